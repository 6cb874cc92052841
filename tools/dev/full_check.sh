#!/bin/bash
# everything the driver runs at round end, in one gpurun call: GPU tests, smoke, the default bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/full_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full_smoke.txt 2>&1
timeout 1500 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err
tail -3 gpurun_out/full_tests.txt; tail -2 gpurun_out/full_smoke.txt; tail -c 600 gpurun_out/full_bench.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/full_bench.json").read().strip().splitlines()[-1])
print("value", j["value"], j["unit"], "ms/step", j["ms_per_step"])
for k in ("reference_shaped_step", "import_swap_only_step", "label_phase_shape", "eager_fused_step", "fine_all_shape", "sh_degree_3"):
    v = j.get(k, {})
    print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "error", "label_backward_ms", "label_backward_full_path_ms", "label_backward_ratio") if kk in v})
print({k: v["ms_per_launch"] for k, v in j["stages"].items()})
PY
