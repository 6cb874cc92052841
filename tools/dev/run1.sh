set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15
for m in 0 1; do
  export EGS_BIN_LEGACY=$m
  echo "=== legacy=$m C"; timeout 300 tools/prof_stages.sh C_$m 500000 540 960 20 2>&1 | grep -E "bin_|tile_sort|table_scan|tile_|total"
  echo "=== legacy=$m B"; timeout 300 tools/prof_stages.sh B_$m 100000 540 960 20 2>&1 | grep -E "bin_|tile_sort|table_scan|tile_|total"
  echo "=== legacy=$m D"; timeout 300 tools/prof_stages.sh D_$m 1000000 1080 1920 12 2>&1 | grep -E "bin_|tile_sort|table_scan|tile_|total"
done
