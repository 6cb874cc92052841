#!/bin/bash
# Hardware counters of the benched workload, collected on the GPU box in runs of their own (never together with tracing other than
# --kernel-trace), folded into profiles/pmc_traffic.json and profiles/sq_counters.json stamped with the kernel-source hash.
#   usage (through gpurun, from the repo root):  tools/collect_counters.sh [N H W]        default 500000 540 960
# Passes (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 -> separate passes; 8 SQ slots):
#   1 FETCH_SIZE   2 WRITE_SIZE   3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
#   4 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
# Workload: bench.py launched eagerly (--no-graph) so that every kernel of the step is a dispatch of its own.
set -e
cd "$(dirname "$0")/.."
root=$PWD
N=${1:-500000}; H=${2:-540}; W=${3:-960}
export TMPDIR=/tmp
out=$root/gpurun_out/counters
rm -rf "$out"; mkdir -p "$out"
cmd="python $root/bench.py --no-graph --steps 12 --warmup 4 --no-cpu-baseline --no-sh3-leg --no-fine-all-leg --gaussians $N --height $H --width $W"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    i=$((i + 1))
    (cd /tmp && timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/pass$i" -o run -- $cmd > "$out/pass$i.log" 2>&1) || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
    echo "pass $i ($ctrs): $(find "$out/pass$i" -name '*counter_collection.csv' | head -1)"
done
python "$root/tools/counters_to_json.py" "$out" "${N}@${W}x${H}" "$root/gpurun_out"
