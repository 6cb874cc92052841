#!/bin/bash
# request-count ablations of the backward blend (EGS_ABL 8 / 9 builds) against the product build, config C and the trained scene
cd "$(dirname "$0")/../.."
for v in raster ab_8 ab_9; do
  lib=$PWD/egogaussian_amd/libegs_$v.so
  [ -f "$lib" ] || continue
  echo "== $v, config C"; EGS_RASTER_LIB=$lib timeout 300 bash tools/prof_stages.sh abl_$v 500000 540 960 40 | grep "render_backward\|render_forward"
  echo "== $v, trained scene"; SCENE=$PWD/bench_data/trained_scene.npz EGS_RASTER_LIB=$lib timeout 300 bash tools/prof_stages.sh ablt_$v 253202 540 960 40 | grep "render_backward\|render_forward"
done
