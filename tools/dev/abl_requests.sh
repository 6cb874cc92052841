#!/bin/bash
# Request-count ablations of the backward blend on main's kernels (EGS_ABL 5 / 6 / 8 / 9 builds under build_ab/, made by
#   make OBJDIR=build_ab/ablN LIB=build_ab/libegs_ablN.so EXTRA=-DEGS_ABL=N): config C and the trained scene.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/abl_requests.txt; : > $out
for lib in "" build_ab/libegs_abl5.so build_ab/libegs_abl6.so build_ab/libegs_abl8.so build_ab/libegs_abl9.so $EXTRA_LIBS; do
  for scene in "" bench_data/trained_scene.npz; do
    echo "== lib=${lib:-product} scene=${scene:-configC}" >> $out
    EGS_RASTER_LIB=$PWD/${lib:-egogaussian_amd/libegs_raster.so} SCENE=$scene timeout 300 python tools/time_stages.py 500000 540 960 40 2>&1 | grep -E "render_|total|tile_|preprocess" >> $out
  done
done
cat $out
