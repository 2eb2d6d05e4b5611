#!/bin/bash
# round 4, first call: host probe, the new / touched tests, a step breakdown of this tree, the SQ-counter pass over every family
root=${GRAFT_REPO_ROOT:-/root/repo}
o=$root/gpurun_out/r4a
mkdir -p $o
cd $root
python3 - > $o/host.txt 2>&1 <<'PY'
import os
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "ERR", e)
PY
cat $o/host.txt
timeout -k 10 900 python -m pytest tests/test_full_size_parity_gpu.py::test_config1_single_step_one_mask tests/test_multi_rank_gpu.py tests/test_cabi.py "tests/test_unet_pipeline_gpu.py::test_linear_library_bias_residual" tests/test_unet_pipeline_gpu.py::test_library_gemms_on_two_streams_finish_and_need_no_workspace tests/test_full_size_parity_gpu.py::test_vae_decode_full_size_vs_oracle -x -q -s > $o/t.log 2>&1; echo "pytest rc=$?" >> $o/t.log; tail -5 $o/t.log
bash tools/ab_step.sh gpurun_out/r4a "base" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $o/pmc_sq -- python3 $root/tools/pmc_kernels.py > $o/pmc_sq.log 2>&1 || { tail -5 $o/pmc_sq.log; exit 1; }
cp $o/pmc_sq/*/*counter_collection.csv $o/pmc_sq_raw.csv
python3 $root/tools/summarize_pmc.py $o/pmc_sq_raw.csv > $o/pmc_sq_counters.csv
rm -rf $o/pmc_sq
grep -c . $o/pmc_sq_counters.csv
