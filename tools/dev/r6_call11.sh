#!/bin/bash
# round 6, eleventh GPU call: flakiness of the chaotic-chain tests (three repeats), then what the driver runs at round end (suite, smoke, bench)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c11; mkdir -p $O
for rep in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_densify.py::test_training_psnr_parity_2000_steps_with_densification tests/test_gpu_bench_mode.py::test_training_psnr_parity_300_steps_float_and_8bit tests/test_gpu_parity.py::test_training_psnr_matches_oracle_training -q -s -m gpu > $O/chaotic_$rep.txt 2>&1
  echo "chaotic $rep rc=$?" >> $O/summary.txt; grep -h "iterations with densification\|PSNR on 4 held-out" $O/chaotic_$rep.txt | cut -c1-330 >> $O/summary.txt
done
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > $O/gpu_suite.txt 2>&1; echo "suite rc=$?" >> $O/summary.txt; tail -14 $O/gpu_suite.txt >> $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/summary.txt; tail -1 $O/smoke.txt >> $O/summary.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>> $O/summary.txt; echo "bench rc=$?" >> $O/summary.txt
python -c "
import json; j = json.loads(open('$O/bench_driver_cmd.json').read().strip().splitlines()[-1]); print('driver cmd', j['value'], j['ms_per_step'], j.get('valid', True), [k for k in j if k.endswith('_step') or k.endswith('_shape')])" >> $O/summary.txt
cat $O/summary.txt
