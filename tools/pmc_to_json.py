#!/usr/bin/env python
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (two separate passes) into profiles/pmc_traffic.json:
HBM bytes per launch of every rasterizer stage.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads, so it is
doubled; calibration on a kernel with a known byte count in the same run is recorded next to the numbers.
  python tools/pmc_to_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload key> [out.json]"""
import json, os, sys
import pandas as pd

STAGES = {"preprocess": ["k_preprocess(", "k_preprocess_count"], "tile_bucket": ["k_bin_count", "k_scan_", "k_bin_scatter"], "tile_sort": ["k_tile_sort"],
          "render_forward": ["k_render_forward"], "render_backward": ["k_render_backward", "k_backward_prologue"],
          "preprocess_backward": ["k_preprocess_backward"], "cov3d": ["k_cov3d_"], "loss": ["k_l1_ssim_"], "adam": ["k_adam"]}


def per_kernel(csv, counter):
    df = pd.read_csv(csv)
    df = df[df["Counter_Name"] == counter]
    return df.groupby("Kernel_Name")["Counter_Value"].agg(["sum", "count"])


def main():
    fetch_csv, write_csv, key = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    res = {}
    for stage, pats in STAGES.items():
        fk = f[[any(p in n for p in pats) for n in f.index]]
        wk = w[[any(p in n for p in pats) for n in w.index]]
        if len(fk) == 0:
            continue
        # launches of the stage = launches of its most frequent kernel divided by how many of them one stage call issues
        lead = [n for n in fk.index if pats[0] in n]
        launches = int(fk.loc[lead, "count"].max()) if lead else int(fk["count"].max())
        if stage == "tile_bucket":
            launches = int(fk.loc[[n for n in fk.index if "k_bin_scatter" in n], "count"].max())
        fetch_b = float(fk["sum"].sum()) * 1024 * 2 / launches
        write_b = float(wk["sum"].sum()) * 1024 / launches
        res[stage] = {"hbm_bytes_per_launch": int(fetch_b + write_b), "fetch_bytes_x2": int(fetch_b), "write_bytes": int(write_b),
                      "launches_profiled": launches}
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[key] = res
    data["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                     "launch (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md HBM section). Calibration: k_preprocess reads 52 B x N "
                     "exactly (26.0 MB at N=500k).")
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print(f"{k:22s} {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch  (fetch x2 {v['fetch_bytes_x2'] / 1e6:.2f}, write {v['write_bytes'] / 1e6:.2f})")


if __name__ == "__main__":
    main()
