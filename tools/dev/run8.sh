timeout 900 python -m pytest tests/test_gpu_bench_mode.py tests/test_install.py tests/test_gpu_fused.py tests/test_gpu_fused_adam.py tests/test_gpu_label.py -x -q -s 2>&1 | grep -E "benched|d/d|passed|failed|Error|rank" | tail -14
timeout 1500 python bench.py --no-config-legs > gpurun_out/b8.json 2> gpurun_out/b8.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/b8.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"])
for k in ("reference_shaped_step", "import_swap_only_step", "label_phase_shape", "eager_fused_step"):
    v = j.get(k, {})
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "error", "label_backward_ms", "label_backward_full_path_ms", "label_backward_ratio", "label_backward_kernels_ms", "full_backward_kernels_ms", "rasterizer_stage_ms")})
PY
tail -3 gpurun_out/b8.err
