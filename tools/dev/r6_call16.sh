#!/bin/bash
# round 6, call 16: a longer randomised parity sweep on the final kernels, two more seeds (900 s each)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c16; mkdir -p $O
for seed in 424242 9001; do timeout 1100 python tests/fuzz_parity.py 900 $seed > $O/fuzz_seed$seed.txt 2>&1; echo "fuzz $seed rc=$?" >> $O/summary.txt; tail -1 $O/fuzz_seed$seed.txt | cut -c1-700 >> $O/summary.txt; done
cat $O/summary.txt
