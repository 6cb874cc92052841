#!/bin/bash
# round 6, second GPU call: conic -> cov2D in float64 (preprocess backward), the flip rule with per-pixel float32 reach, L1 sign ties
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_offscreen.py -q -s -m gpu > $O/offscreen.txt 2>&1
echo "offscreen rc=$?" >> $O/summary.txt
timeout 1800 python -m pytest tests -q -m gpu -s --deselect tests/test_gpu_offscreen.py > $O/gpu_suite.txt 2>&1
echo "suite rc=$?" >> $O/summary.txt
tail -12 $O/gpu_suite.txt >> $O/summary.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt
