#!/bin/bash
# in-step breakdown at K images per generation: tools/ab_stepk.sh OUTDIR K "NAME ENV=.." ...
out=$1; k=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace_$name -- python3 $root/bench.py --images-per-gpu $k --steps 3 --warmup 1 --no-cpu-baseline --no-batched-roofline --no-coalesced --in-flight 1 > $root/$out/bench_$name.log 2>&1 ) || exit 1
  python3 $root/tools/step_breakdown.py -vv $root/$out/trace_$name/*/*_kernel_trace.csv > $root/$out/breakdown${k}_$name.txt 2>&1
  rm -rf $root/$out/trace_$name
  head -9 $root/$out/breakdown${k}_$name.txt
done
