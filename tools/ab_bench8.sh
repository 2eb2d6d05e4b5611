#!/bin/bash
# tools/ab_bench8.sh OUTDIR "NAME ENV=.." ...: bench.py at 8 images per generation (BASELINE configs[2] per GPU), one generation at a time, per variant
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/$out
cd $root
for spec in "$@"; do
  set -- $spec; name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 300 python3 bench.py --images-per-gpu 8 --regions 4 --steps 4 --warmup 1 --in-flight 1 --no-cpu-baseline --no-batched-roofline --no-coalesced > gpurun_out/$out/b8_$name.json 2> gpurun_out/$out/b8_$name.err ) || { echo "$name failed"; tail -3 gpurun_out/$out/b8_$name.err; exit 1; }
  python3 -c "
import json; r=json.load(open('gpurun_out/$out/b8_$name.json')); print('$name', r['value'], 'images/s', r['ms_per_step'], 'ms/gen', 'xattn', r['roofline']['avg_launch_us'], r['roofline']['forward_only']['avg_launch_us'], 'sa', r['roofline_self_attn']['avg_launch_us'], 'conv', r['roofline_conv3x3']['avg_launch_us'])"
done
