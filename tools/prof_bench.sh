#!/bin/bash
# rocprofv3 --kernel-trace --stats around one bench.py run on the GPU box; prints the per-kernel averages and writes the
# markdown summary that goes under profiles/.   usage: tools/prof_bench.sh <tag> [bench.py args...]
# (run from the repo root, i.e. through gpurun; output lands in gpurun_out/prof_<tag>/ and gpurun_out/<tag>_kernel_stats.md)
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$out" -o run -- python "$root/bench.py" "$@" > "$out/bench.json" 2> "$out/bench.err") || { tail -5 "$out/bench.err"; exit 1; }
db=$(find "$out" -name "*.db" | head -1)
python "$root/tools/rocprof_summary.py" "$db" "$root/gpurun_out/${tag}_kernel_stats.md" "rocprofv3 --kernel-trace --stats -- python bench.py $*" > /dev/null
python "$root/tools/gpu_gaps.py" "$db" "$root/gpurun_out/${tag}_step_budget.json" > "$root/gpurun_out/${tag}_step_budget.txt" 2>&1 || true
grep "k_\|copyBuffer" "$root/gpurun_out/${tag}_kernel_stats.md" | awk -F'|' '{printf "%-60s x%6s  %8s us\n", substr($2,1,60), $3, $5}' | head -24
tail -1 "$out/bench.json" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under rocprof:', d['value'], 'it/s', d['ms_per_step'], 'ms/step')"
rm -f "$db"    # the database is tens of MB; the summary is what is kept
