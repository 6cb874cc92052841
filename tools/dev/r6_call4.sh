#!/bin/bash
# round 6, fourth GPU call: ABI 6 (per-call flags, no registry), provenance route, strict gradient bar, densifying slice with run-to-run spread
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_reentrant.py tests/test_gpu_provenance.py -q -s -m gpu > $O/new_tests.txt 2>&1
echo "new tests rc=$?" >> $O/summary.txt
timeout 600 python tools/dev/host_profile_installed.py > $O/host_profile_installed.txt 2>&1
echo "host_profile rc=$?" >> $O/summary.txt
timeout 1800 python -m pytest tests -q -m gpu -s --deselect tests/test_gpu_reentrant.py --deselect tests/test_gpu_provenance.py --deselect tests/test_gpu_offscreen.py > $O/gpu_suite.txt 2>&1
echo "suite rc=$?" >> $O/summary.txt
tail -12 $O/gpu_suite.txt >> $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -30 $O/new_tests.txt
