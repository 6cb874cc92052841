#!/bin/bash
# A/B of library builds on the rasterizer alone (tools/time_stages.py): config C (depth + alpha gradients -> k_render_backward<2>) and the
# trained scene, then MODE 1 through the colour-only loss of tools/dev/time_bwd_modes.py.   LIBS="a.so b.so" bash tools/dev/ab_stages.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/ab_stages.txt; : > $out
for lib in ${LIBS:-build_ab/libegs_base.so egogaussian_amd/libegs_raster.so}; do
  for scene in "" bench_data/trained_scene.npz; do
    echo "== lib=$lib scene=${scene:-configC}" >> $out
    EGS_RASTER_LIB=$PWD/$lib SCENE=$scene timeout 300 python tools/time_stages.py 500000 540 960 40 2>&1 | grep -E "render_|total" >> $out
    EGS_RASTER_LIB=$PWD/$lib SCENE=$scene timeout 300 python tools/dev/time_bwd_modes.py 2>&1 | grep -E "mode" >> $out
  done
done
cat $out
