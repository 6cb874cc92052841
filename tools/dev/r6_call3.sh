#!/bin/bash
# round 6, third GPU call: benched-mode parity with L1 sign ties, the densifying 2 000-iteration slice, host profile of the installed loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_mode.py::test_config_C_bench_mode_vs_oracle -q -s -m gpu > $O/bench_mode.txt 2>&1
echo "bench_mode rc=$?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_densify.py::test_training_psnr_parity_2000_steps_with_densification -q -s -m gpu > $O/densify_slice.txt 2>&1
echo "densify_slice rc=$?" >> $O/summary.txt
timeout 600 python tools/dev/host_profile_installed.py > $O/host_profile_installed.txt 2>&1
echo "host_profile rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -5 $O/bench_mode.txt; tail -8 $O/densify_slice.txt
