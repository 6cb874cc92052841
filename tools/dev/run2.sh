set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8
for c in "C 500000 540 960 20" "B 100000 540 960 20" "D 1000000 1080 1920 12"; do
  set -- $c
  echo "=== new $1"; timeout 300 tools/prof_stages.sh $1_new $2 $3 $4 $5 2>&1 | grep -E "bin_|tile_sort|tile_|total"
done
timeout 900 python -m pytest tests/test_gpu_label.py tests/test_gpu_fused_adam.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_golden_host.py -x -q 2>&1 | tail -8
