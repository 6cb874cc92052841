#!/bin/bash
# round 6, sixth GPU call: the unfolded ninth value combined by a masked DPP step (one publishing lane per slot again); tool-vs-bench host timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6c6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_adam.py tests/test_gpu_parity.py tests/test_gpu_label.py tests/test_gpu_bench_mode.py::test_config_C_bench_mode_vs_oracle -q -x -m gpu > $O/tests.txt 2>&1
echo "tests rc=$?" >> $O/summary.txt; tail -3 $O/tests.txt >> $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
python - <<'PY' >> $O/summary.txt
import json
j = json.loads(open("gpurun_out/r6c6/bench.json").read().strip().splitlines()[-1])
print("headline", j["value"], j["ms_per_step"], j.get("value_median"), {k: v["ms_per_launch"] for k, v in j["stages"].items()})
for k in ("reference_shaped_step", "label_phase_shape", "eager_fused_step", "trained_scene_op_only", "config_D_op_only"):
    v = j.get(k, {})
    print(k, v.get("value"), v.get("ms_per_step"), v.get("op_ms"), v.get("rasterizer_stage_ms") or {k2: v2["ms_per_launch"] for k2, v2 in v.get("stages", {}).items()})
PY
FRAMES=8 timeout 300 python tools/dev/host_profile_installed.py 2>&1 | grep "wall per step" >> $O/summary.txt
FRAMES=32 timeout 300 python tools/dev/host_profile_installed.py 2>&1 | grep "wall per step" >> $O/summary.txt
cat $O/summary.txt
