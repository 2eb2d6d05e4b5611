#!/bin/bash
# in-step A/B on the GPU box: tools/ab_step.sh OUTDIR "NAME ENV=.. ENV=.." "NAME ENV=.." ...
# one kernel trace of bench.py per variant, reduced to tools/step_breakdown.py's per-kernel table (the traces are deleted)
out=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; name=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $root/$out/trace_$name -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batched-roofline --no-coalesced --in-flight 1 > $root/$out/bench_$name.log 2>&1 ) || exit 1
  python3 $root/tools/step_breakdown.py -vv $root/$out/trace_$name/*/*_kernel_trace.csv > $root/$out/breakdown_$name.txt 2>&1
  rm -rf $root/$out/trace_$name
  head -12 $root/$out/breakdown_$name.txt
done
