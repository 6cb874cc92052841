#!/bin/bash
# HBM bytes per launch of the two loss kernels ALONE (tools/loss_kernels.py: no side jobs carried), FETCH_SIZE / WRITE_SIZE in passes of
# their own with the gfx950 corrections of tools/pmc_to_json.py -- the measured half of profiles/r4_loss_traffic.md.
#   usage (through gpurun, from the repo root):  tools/loss_traffic.sh
cd "$(dirname "$0")/.."
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/pmc_loss
rm -rf "$out"; mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/$c" -o run -- python "$root/tools/loss_kernels.py" > "$out/$c.log" 2>&1) || { tail -5 "$out/$c.log"; exit 1; }
done
python - "$out" <<'PY'
import sys, glob, pandas as pd
out = sys.argv[1]
res = {}
for c, mul in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    df = pd.read_csv(glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0])
    df = df[df["Counter_Name"] == c]
    g = df.groupby("Kernel_Name")["Counter_Value"].agg(["sum", "count"])
    for k, r in g.iterrows():
        res.setdefault(k, {})[c] = r["sum"] * mul / r["count"]
for k, v in sorted(res.items(), key=lambda kv: -sum(kv[1].values())):
    if "ssim" in k:
        print(f"{k[:70]:70s} fetch {v.get('FETCH_SIZE', 0) / 1e6:8.2f} MB  write {v.get('WRITE_SIZE', 0) / 1e6:8.2f} MB")
PY
rm -rf "$out/FETCH_SIZE" "$out/WRITE_SIZE"
