#!/bin/bash
# one gpurun call that produces a round's evidence files under gpurun_out/$1 (copied into profiles/ by hand afterwards):
#   [full -m gpu suite unless $2 = notests], a kernel trace of the step (stats + per-step breakdown + in_step_kernels.json), the PMC
#   passes (separate rocprofv3 runs: FETCH_SIZE, WRITE_SIZE -> pmc_traffic.json; SQ counters over every family -> pmc_mfma_busy.json),
#   the 2-rank gloo rehearsal.  The bench LINES that quote those stamped files are taken by tools/round_lines.sh in a later call,
#   after the files have been committed under profiles/.
out=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
o=$root/gpurun_out/$out
mkdir -p $o/stamped
export DSC_PROFILES_DIR=$o/stamped
cd $root
if [ "$2" != "notests" ]; then
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $o/t.log 2>&1; echo "pytest rc=$?" >> $o/t.log; tail -3 $o/t.log
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batched-roofline --no-coalesced --in-flight 1 > $o/bench_trace.log 2>&1 || exit 1
python3 $root/tools/step_breakdown.py -vv $o/trace/*/*_kernel_trace.csv > $o/step_breakdown.txt 2>&1
python3 $root/tools/make_in_step.py $o/trace/*/*_kernel_trace.csv r04 > $o/make_in_step.log 2>&1 || { tail -3 $o/make_in_step.log; exit 1; }
cp $o/trace/*/*_kernel_stats.csv $o/bench_kernel_stats.csv
rm -rf $o/trace
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_f -- python3 $root/tools/pmc_xattn.py > $o/pmc_f.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_w -- python3 $root/tools/pmc_xattn.py > $o/pmc_w.log 2>&1 || exit 1
python3 $root/tools/make_pmc_traffic.py $o/pmc_f/*/*counter_collection.csv $o/pmc_w/*/*counter_collection.csv r04 > $o/make_pmc_traffic.log 2>&1 || { tail -3 $o/make_pmc_traffic.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $o/pmc_sq -- python3 $root/tools/pmc_kernels.py > $o/pmc_sq.log 2>&1 || exit 1
python3 $root/tools/make_mfma_busy.py $o/pmc_sq/*/*counter_collection.csv r04 > $o/make_mfma_busy.log 2>&1 || { tail -3 $o/make_mfma_busy.log; exit 1; }
python3 $root/tools/summarize_pmc.py $o/pmc_sq/*/*counter_collection.csv > $o/pmc_sq_counters.csv
rm -rf $o/pmc_f $o/pmc_w $o/pmc_sq
cd $root
DSC_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --no-batched-roofline --no-coalesced > $o/bench_line_2_ranks_gloo_rehearsal.json 2> $o/bench_2rank.err
echo "2-rank rc=$?"
ls $o/stamped
head -9 $o/step_breakdown.txt
