timeout 900 python -m pytest tests/test_gpu_bench_mode.py -x -q -s -k "config_C_bench_mode" 2>&1 | grep -vE "Warning|warn|^$" | tail -25
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trained_scene.py -x -q -s 2>&1 | grep -E "grads:|D:|hot|trained|passed|failed|Error|assert" | tail -60
timeout 600 python -m pytest tests/test_gpu_sweep.py -x -q -s 2>&1 | grep -E "seed|passed|failed|Error|assert" | tail -12
for c in "C 500000 540 960 20" "D 1000000 1080 1920 8"; do
    set -- $c
    echo "=== $1: $(timeout 100 tools/prof_stages.sh $1_n $2 $3 $4 $5 2>&1 | grep -E "tile_sort|tile_" | tr '\n' ' ')"
done
