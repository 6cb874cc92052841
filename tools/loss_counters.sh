#!/bin/bash
# Instruction-fetch and wait counters of the two loss kernels alone (tools/loss_time.py), one rocprofv3 --pmc pass per group.
#   usage (through gpurun, from the repo root): tools/loss_counters.sh
cd "$(dirname "$0")/.."
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/loss_counters
rm -rf "$out"; mkdir -p "$out"
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -io "SQ[C]*_[A-Z_]*\(ICACHE\|IFETCH\|INST_ANY\|WAIT_INST\|INSTS_\)[A-Z_0-9]*" | sort -u | tr '\n' ' ') > "$out/avail.txt"
i=0
for ctrs in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i + 1))
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/pass$i" -o run -- python "$root/tools/loss_time.py" > "$out/pass$i.log" 2>&1) || { echo "pass $i failed"; tail -3 "$out/pass$i.log"; }
    f=$(find "$out/pass$i" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "ssim" not in k: continue
    acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
done
cat "$out/avail.txt"
