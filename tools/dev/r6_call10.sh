#!/bin/bash
# round 6, tenth GPU call: deferred overflow check (opt-in), memoised rotation activation, the unchanged-loop legs with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_provenance.py tests/test_gpu_label.py tests/test_gpu_capacity.py tests/test_adapter.py tests/test_gpu_densify.py -q -x -m gpu --durations=5 > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/summary.txt; tail -12 $O/tests.txt >> $O/summary.txt
for rep in 1 2; do timeout 600 python bench.py --unchanged-trainer-legs --steps 200 --warmup 30 > $O/legs_$rep.json 2> $O/legs_$rep.err; echo "legs $rep rc=$?" >> $O/summary.txt; done
python - <<'PY' >> $O/summary.txt
import json
for rep in (1, 2):
    u = json.loads(open(f"gpurun_out/r6c10/legs_{rep}.json").read().strip().splitlines()[-1])
    for k, v in u.items():
        print(rep, k, v.get("value"), v.get("ms_per_step"), v.get("error", ""))
PY
cat $O/summary.txt
