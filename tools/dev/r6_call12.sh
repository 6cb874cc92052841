#!/bin/bash
# round 6, twelfth GPU call: k_preprocess_backward at 5 and 6 waves per SIMD (96 / 80 VGPRs, 136 / 232 B of scratch per lane) against the tree's 4 (120 VGPRs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
LIBS="egogaussian_amd/libegs_raster.so build_ab/libegs_ppb5.so build_ab/libegs_ppb6.so" REPS="1 2" bash tools/dev/ab_bench.sh
mkdir -p gpurun_out/r6c12; cp gpurun_out/ab_bench.txt gpurun_out/r6c12/ab_bench.txt
