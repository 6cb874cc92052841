#!/bin/bash
# round 6, eighth GPU call: the LDS-merged backward blend (one accumulator line per (tile, splat)) -- parity first, then a same-box A/B of six builds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_label.py tests/test_gpu_fused_adam.py tests/test_gpu_bench_mode.py::test_config_C_bench_mode_vs_oracle tests/test_gpu_trained_scene.py tests/test_gpu_offscreen.py -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/summary.txt; tail -3 $O/tests.txt >> $O/summary.txt
LIBS="build_ab/libegs_d03dbe6.so build_ab/libegs_nomerge.so build_ab/libegs_nomerge_fold9.so build_ab/libegs_nomerge_priosel.so egogaussian_amd/libegs_raster.so build_ab/libegs_merge_fold9.so" REPS="1 2" bash tools/dev/ab_bench.sh > /dev/null
cp gpurun_out/ab_bench.txt $O/ab_bench.txt
cat $O/summary.txt $O/ab_bench.txt
