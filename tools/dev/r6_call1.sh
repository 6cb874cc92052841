#!/bin/bash
# round 6, first GPU call: the off-screen family against three builds of the FAR path, the whole GPU suite under the stricter flip rule,
# the VALU issue-rate ubench per occupancy, a baseline bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r6c1
O=gpurun_out/r6c1
./tools/ubench/valu_rate.bin > $O/valu_rate.txt 2>&1
for lib in default far_all far_none; do
  if [ $lib = default ]; then unset EGS_RASTER_LIB; else export EGS_RASTER_LIB=$PWD/build_ab/libegs_$lib.so; fi
  timeout 900 python -m pytest tests/test_gpu_offscreen.py -q -s -m gpu > $O/offscreen_$lib.txt 2>&1
  echo "offscreen $lib rc=$?" >> $O/summary.txt
done
unset EGS_RASTER_LIB
timeout 1500 python -m pytest tests -q -m gpu -s > $O/gpu_suite.txt 2>&1
echo "suite rc=$?" >> $O/summary.txt
tail -5 $O/gpu_suite.txt >> $O/summary.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt
