#!/bin/bash
# tools/ab_benchk.sh OUTDIR K "NAME ENV=.." ...: bench.py at K images per generation, one generation at a time, per variant
out=$1; k=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/$out
cd $root
for spec in "$@"; do
  set -- $spec; name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 300 python3 bench.py --images-per-gpu $k --steps $(( 16 / k + 2 )) --warmup 1 --in-flight ${F:-1} --no-cpu-baseline --no-batched-roofline --no-coalesced > gpurun_out/$out/bk${k}_$name.json 2> gpurun_out/$out/bk${k}_$name.err ) || { echo "$name failed"; tail -3 gpurun_out/$out/bk${k}_$name.err; exit 1; }
  python3 -c "
import json; r=json.load(open('gpurun_out/$out/bk${k}_$name.json')); print('k=$k f=${F:-1} $name', r['value'], 'images/s', r['ms_per_step'], 'ms/gen')"
done
