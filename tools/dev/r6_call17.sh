#!/bin/bash
# round 6, call 17: the per-tile sort inside the forward blend up to 2816 (instead of 2048) instances of CAPACITY per tile -- configs D and the trained scene then take it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c17; mkdir -p $O
EGS_RASTER_LIB=$PWD/build_ab/libegs_solo.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trained_scene.py tests/test_gpu_capacity.py -q -m gpu -x > $O/solo_tests.txt 2>&1; echo "solo tests rc=$?" >> $O/summary.txt; tail -2 $O/solo_tests.txt >> $O/summary.txt
for rep in 1 2; do for lib in egogaussian_amd/libegs_raster.so build_ab/libegs_solo.so; do
  EGS_RASTER_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --no-sh3-leg --no-fine-all-leg --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
f=lambda k: (j.get(k,{}).get('op_ms'), {a: round(b['ms_per_launch']*1e3,1) for a,b in j.get(k,{}).get('stages',{}).items()})
print('$lib', 'headline', j['value'], 'B', j.get('config_B_forward_only',{}).get('frames_per_s'), 'D', f('config_D_op_only'), 'trained', f('trained_scene_op_only'))" >> $O/summary.txt
done; done
cat $O/summary.txt
