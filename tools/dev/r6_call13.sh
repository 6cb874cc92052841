#!/bin/bash
# round 6, call 13: flat-image loss tests; the first batch's loads hoisted in front of the loss-gradient prologue of the backward blend (A/B)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c13; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fused.py -q -m gpu > $O/fused_tree.txt 2>&1; echo "fused (tree) rc=$?" >> $O/summary.txt; tail -2 $O/fused_tree.txt >> $O/summary.txt
EGS_RASTER_LIB=$PWD/build_ab/libegs_hoist.so timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_bench_mode.py::test_config_C_bench_mode_vs_oracle tests/test_gpu_parity.py -q -m gpu -x > $O/hoist_tests.txt 2>&1; echo "hoist tests rc=$?" >> $O/summary.txt; tail -2 $O/hoist_tests.txt >> $O/summary.txt
LIBS="egogaussian_amd/libegs_raster.so build_ab/libegs_hoist.so" REPS="1 2 3" bash tools/dev/ab_bench.sh > /dev/null
cp gpurun_out/ab_bench.txt $O/ab_bench.txt
cat $O/summary.txt; grep -A1 "== lib" $O/ab_bench.txt | grep -v "^--"
