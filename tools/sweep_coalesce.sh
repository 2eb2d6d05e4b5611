#!/bin/bash
# tools/sweep_coalesce.sh OUTDIR: bench.py --images-per-gpu k --in-flight f for k in {1,2,4,8}, f in {1,2} (configs[1]'s masks), one JSON
# line per point under gpurun_out/OUTDIR (committed as profiles/r04_sweep_k{k}_f{f}.json)
out=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/$out
cd $root
for k in 1 2 4 8; do
  for f in 1 2; do
    n=$(( 24 / k )); [ $n -lt 4 ] && n=4
    timeout -k 10 300 python3 bench.py --images-per-gpu $k --in-flight $f --steps $n --warmup 1 --no-cpu-baseline --no-batched-roofline --no-coalesced \
      > gpurun_out/$out/sweep_k${k}_f${f}.json 2> gpurun_out/$out/sweep_k${k}_f${f}.err || { echo "k=$k f=$f failed"; tail -3 gpurun_out/$out/sweep_k${k}_f${f}.err; exit 1; }
    python3 - <<PY
import json
r = json.load(open("gpurun_out/$out/sweep_k${k}_f${f}.json"))
o = r.get("one_generation_at_a_time", {})
print("k=$k f=$f  value", r["value"], "images/s  ms_per_step", r["ms_per_step"], " one-at-a-time", o.get("value"), o.get("ms_per_generation"))
PY
  done
done
