#!/bin/bash
# round 6, fifth GPU call: leaner backward visit loop (no FAR path, priority countdown, ninth value unfolded), fixed tests, the unchanged-loop legs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_adam.py tests/test_gpu_parity.py tests/test_gpu_label.py tests/test_gpu_bench_mode.py::test_config_C_bench_mode_vs_oracle tests/test_gpu_offscreen.py -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/summary.txt; tail -4 $O/tests.txt >> $O/summary.txt
timeout 600 python bench.py --unchanged-trainer-legs --steps 200 --warmup 30 > $O/unchanged_legs.json 2> $O/unchanged_legs.err
echo "unchanged legs rc=$?" >> $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
python - <<'PY' >> $O/summary.txt
import json
j = json.loads(open("gpurun_out/r6c5/bench.json").read().strip().splitlines()[-1])
print("headline", j["value"], j["ms_per_step"], j.get("value_median"), {k: v["ms_per_launch"] for k, v in j["stages"].items()})
for k in ("reference_shaped_step", "label_phase_shape", "eager_fused_step", "trained_scene_op_only", "config_D_op_only"):
    v = j.get(k, {})
    print(k, v.get("value"), v.get("ms_per_step"), v.get("op_ms"), v.get("rasterizer_stage_ms"), v.get("raw_parameter_route", "")[:30])
u = json.loads(open("gpurun_out/r6c5/unchanged_legs.json").read().strip().splitlines()[-1])
for k, v in u.items():
    print("legs-only run:", k, v.get("value"), v.get("ms_per_step"), v.get("raw_parameter_route", "")[:40])
PY
cat $O/summary.txt
