timeout 1200 python -m pytest tests/test_gpu_bench_mode.py -x -q -s 2>&1 | grep -E "benched|AssertionError|passed|failed|    xyz|f_dc" | tail -8
timeout 600 python bench.py --unchanged-trainer-legs > gpurun_out/b9.json 2> gpurun_out/b9.err; python - <<'PY'
import json
j = json.loads(open("gpurun_out/b9.json").read().strip().splitlines()[-1])
for k in ("reference_shaped_step", "import_swap_only_step", "label_phase_shape"):
    v = j.get(k, {})
    print(k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "error", "label_backward_ms", "label_backward_full_path_ms", "label_backward_ratio", "label_backward_kernels_ms", "full_backward_kernels_ms")})
PY
timeout 600 python tools/dev/host_profile_installed.py 2>&1 | grep "wall per step"
