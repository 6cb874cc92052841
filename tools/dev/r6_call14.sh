#!/bin/bash
# round 6, call 14: the loss forward as a three-stage pipeline across the waves of a workgroup -- bit comparison with the one-wave kernel, tests, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c14; mkdir -p $O
python tools/dev/loss_bits.py /tmp/a.pt > $O/bits.txt 2>&1
EGS_RASTER_LIB=$PWD/build_ab/libegs_pipe.so python tools/dev/loss_bits.py /tmp/b.pt >> $O/bits.txt 2>&1
python tools/dev/loss_bits.py /tmp/a.pt /tmp/b.pt >> $O/bits.txt 2>&1; echo "bits rc=$?" >> $O/summary.txt
EGS_RASTER_LIB=$PWD/build_ab/libegs_pipe.so timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_adam.py tests/test_gpu_bench_mode.py::test_config_C_bench_mode_vs_oracle tests/test_gpu_capacity.py -q -m gpu -x > $O/pipe_tests.txt 2>&1; echo "pipe tests rc=$?" >> $O/summary.txt; tail -2 $O/pipe_tests.txt >> $O/summary.txt
LIBS="egogaussian_amd/libegs_raster.so build_ab/libegs_pipe.so" REPS="1 2 3" bash tools/dev/ab_bench.sh > /dev/null
cp gpurun_out/ab_bench.txt $O/ab_bench.txt
cat $O/bits.txt $O/summary.txt; grep -A1 "== lib" $O/ab_bench.txt | grep -v "^--"
