#!/bin/bash
# round 6, ninth GPU call: the randomised parity sweep on the final kernels under the round-6 rule (flips proven per pixel, no row excused), two seeds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c9; mkdir -p $O
timeout 500 python tests/fuzz_parity.py 330 20261001 > $O/fuzz_seed20261001.txt 2>&1; echo "fuzz1 rc=$?" >> $O/summary.txt
timeout 500 python tests/fuzz_parity.py 330 777123 > $O/fuzz_seed777123.txt 2>&1; echo "fuzz2 rc=$?" >> $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_densify.py::test_training_psnr_parity_2000_steps_with_densification tests/test_gpu_bench_mode.py::test_training_psnr_parity_300_steps_float_and_8bit -q -s -m gpu --durations=3 > $O/psnr_tests.txt 2>&1; echo "psnr tests rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -3 $O/fuzz_seed20261001.txt | cut -c1-600; tail -3 $O/fuzz_seed777123.txt | cut -c1-600; tail -8 $O/psnr_tests.txt
