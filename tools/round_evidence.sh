#!/bin/bash
# one gpurun call that produces a round's evidence files under gpurun_out/$1 (copied into profiles/ by hand afterwards):
#   full -m gpu suite, the default bench line, the driver's command line, a kernel trace of the step (stats + per-step breakdown),
#   the PMC passes (separate rocprofv3 runs: FETCH_SIZE, WRITE_SIZE, SQ counters), the 2-rank gloo rehearsal
out=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
o=$root/gpurun_out/$out
mkdir -p $o
cd $root
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $o/t.log 2>&1; echo "pytest rc=$?" >> $o/t.log; tail -3 $o/t.log
timeout -k 10 300 python3 bench.py > $o/bench_line_default.json 2> $o/bench_default.err || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_line_driver_command.json 2> $o/bench_driver.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batched-roofline --in-flight 1 > $o/bench_trace.log 2>&1 || exit 1
python3 $root/tools/step_breakdown.py -vv $o/trace/*/*_kernel_trace.csv > $o/step_breakdown.txt 2>&1
cp $o/trace/*/*_kernel_stats.csv $o/bench_kernel_stats.csv
rm -rf $o/trace
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_f -- python3 $root/tools/pmc_xattn.py > $o/pmc_f.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_w -- python3 $root/tools/pmc_xattn.py > $o/pmc_w.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $o/pmc_sq -- python3 $root/tools/pmc_sa.py > $o/pmc_sq.log 2>&1 || exit 1
cp $o/pmc_f/*/*counter_collection.csv $o/pmc_fetch.csv; cp $o/pmc_w/*/*counter_collection.csv $o/pmc_write.csv
python3 $root/tools/summarize_pmc.py $o/pmc_sq/*/*counter_collection.csv > $o/pmc_sq_counters.csv
rm -rf $o/pmc_f $o/pmc_w $o/pmc_sq
cd $root
DSC_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --no-batched-roofline > $o/bench_line_2_ranks_gloo_rehearsal.json 2> $o/bench_2rank.err
echo "2-rank rc=$?"
tail -c 400 $o/bench_line_default.json
