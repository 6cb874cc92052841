#!/bin/bash
# rocprofv3 --kernel-trace --stats around tools/time_stages.py: per-kernel averages of the rasterizer alone.
#   usage: tools/prof_stages.sh <tag> [N H W iters]     (run through gpurun from the repo root)
tag=$1; shift
cd "$(dirname "$0")/.."
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$out" -o run -- python "$root/tools/time_stages.py" "$@" > "$out/stages.txt" 2> "$out/err.txt") || { tail -5 "$out/err.txt"; exit 1; }
db=$(find "$out" -name "*.db" | head -1)
python "$root/tools/kstats.py" "$db" k_ | sort -k4 -n -r | head -24
grep "tile_\|total" "$out/stages.txt"
rm -f "$db"
