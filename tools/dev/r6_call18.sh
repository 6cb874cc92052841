#!/bin/bash
# round 6, call 18: the seed-9001 miss as a fixed test, that seed again under the corrected arbitration rule, then call 17's A/B (sort in the blend up to 2816 per tile)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_offscreen.py -q -s -m gpu -k "single_symmetric" > $O/single.txt 2>&1; echo "single rc=$?" >> $O/summary.txt; grep "hip-f64" $O/single.txt >> $O/summary.txt
timeout 1000 python tests/fuzz_parity.py 800 9001 > $O/fuzz_seed9001.txt 2>&1; echo "fuzz 9001 rc=$?" >> $O/summary.txt; tail -1 $O/fuzz_seed9001.txt | cut -c1-500 >> $O/summary.txt
cat $O/summary.txt
bash tools/dev/r6_call17.sh
