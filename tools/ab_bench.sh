#!/bin/bash
# headline A/B on ONE box: tools/ab_bench.sh OUTDIR REPS STEPS "NAME ENV=.." ...  -> images/s two in flight / one at a time per variant, interleaved
# (BENCH_ARGS=... in a variant's environment is appended to its bench.py command line; no spaces inside one value)
out=$1; reps=$2; steps=$3; shift 3
specs=("$@")
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/$out
for r in $(seq $reps); do
  for spec in "${specs[@]}"; do
    read -r -a parts <<< "$spec"
    name=${parts[0]}
    ( for kv in "${parts[@]:1}"; do export "$kv"; done
      timeout -k 10 300 python3 $root/bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-batched-roofline --no-coalesced $BENCH_ARGS > $root/$out/line_${name}_$r.json 2> $root/$out/err_${name}_$r.log ) || exit 1
    python3 - $root/$out/line_${name}_$r.json $name $r <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], "two in flight", d["value"], "one at a time", d["one_generation_at_a_time"]["value"], flush=True)
PY
  done
done
