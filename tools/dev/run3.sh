set -x
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "parity or ties or slab or large or capacity or overflow" 2>&1 | tail -5
for c in "C 500000 540 960 20" "B 100000 540 960 20" "D 1000000 1080 1920 12"; do
  set -- $c
  echo "=== new $1"; timeout 300 tools/prof_stages.sh $1_new $2 $3 $4 $5 2>&1 | grep -E "bin_|tile_sort|tile_|total"
done
EGS_RASTER_LIB=egogaussian_amd/libegs_timing.so timeout 300 python tools/dev/emit_phases.py 500000 540 960
EGS_RASTER_LIB=egogaussian_amd/libegs_timing.so timeout 300 python tools/dev/emit_phases.py 1000000 1080 1920
timeout 900 python -m pytest tests/test_gpu_label.py -x -q -s 2>&1 | tail -25
timeout 300 python -m pytest tests/test_golden_host.py -x -q -k test_covariance_matches_reference 2>&1 | grep -E "^E|assert|passed|failed" | head -20
