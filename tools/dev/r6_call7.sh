#!/bin/bash
# round 6, seventh GPU call: same-box A/B of the backward visit loop -- f772283 (FAR path, per-visit priority select), d03dbe6 (FAR removed), tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
LIBS="build_ab/libegs_f772283.so build_ab/libegs_d03dbe6.so egogaussian_amd/libegs_raster.so" REPS="1 2 3" bash tools/dev/ab_bench.sh
mkdir -p gpurun_out/r6c7; cp gpurun_out/ab_bench.txt gpurun_out/r6c7/ab_bench.txt
