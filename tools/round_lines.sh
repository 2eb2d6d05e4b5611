#!/bin/bash
# second call of a round's evidence: the bench LINES on the tree whose stamped files (profiles/in_step_kernels.json,
# pmc_mfma_busy.json, pmc_traffic.json) were committed after tools/round_evidence.sh: the default line, the driver's command, the
# other BASELINE configs as lines (configs[2]: 8 images x 4 masks per GPU; configs[3]: 768x768, 8 images), the k x f sweep
out=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
o=$root/gpurun_out/$out
mkdir -p $o
cd $root
timeout -k 10 500 python3 bench.py > $o/bench_line_default.json 2> $o/bench_default.err || exit 1
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_line_driver_command.json 2> $o/bench_driver.err || exit 1
timeout -k 10 300 python3 bench.py --images-per-gpu 8 --regions 4 --steps 6 --warmup 1 --no-cpu-baseline --no-batched-roofline --no-coalesced > $o/config3_8_images_4_masks.json 2> $o/config3.err || exit 1
timeout -k 10 400 python3 bench.py --size 768 --images-per-gpu 8 --steps 4 --warmup 1 --no-cpu-baseline --no-batched-roofline --no-coalesced > $o/config4_768_8_images.json 2> $o/config4.err || exit 1
bash tools/sweep_coalesce.sh $out || exit 1
python3 - <<PY
import json
for n in ("bench_line_default", "bench_line_driver_command", "config3_8_images_4_masks", "config4_768_8_images"):
    r = json.load(open("$o/" + n + ".json"))
    print(n, r["value"], r.get("one_generation_at_a_time", {}).get("value"), r["roofline"]["frac"], r["roofline"].get("in_step_us"))
PY
