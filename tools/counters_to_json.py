#!/usr/bin/env python
"""Fold the rocprofv3 --pmc CSVs of tools/collect_counters.sh into pmc_traffic.json (HBM bytes per launch per stage) and
sq_counters.json (instruction counts, VALU / LDS busy, waiting share per stage), both stamped with the kernel-source hash
(egogaussian_amd.lib.kernel_source_hash) -- bench.py reports their numbers only while that hash is the library's.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request of a wide coalesced read, so it is doubled (calibration: k_preprocess reads 56 B x N in raw-parameter mode);
SQ cycle counters are quad-cycles summed over the 8 XCDs; VALUBusy = SQ_ACTIVE_INST_VALU * 4 / SIMDs / (GRBM_GUI_ACTIVE / 8), the
gfx94x formula rocprof falls back to.
  python tools/counters_to_json.py <dir with pass1..pass4> <workload key> <output dir>"""
import glob
import json
import os
import sys

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egogaussian_amd.lib import kernel_source_hash          # noqa: E402

STAGES = {"preprocess": ["k_preprocess<", "k_preprocess(", "k_preprocess_count"], "tile_bucket": ["k_bin_count", "k_table_scan", "k_bin_scatter"], "tile_sort": ["k_tile_sort"],
          "render_forward": ["k_render_forward"], "render_backward": ["k_render_backward", "k_backward_prologue"],
          "preprocess_backward": ["k_preprocess_backward"], "loss": ["k_l1_ssim_"], "adam": ["k_adam"]}
N_SIMD, XCDS, CLOCK = 1024, 8, 2.4e9


def table(d, i):
    f = glob.glob(os.path.join(d, f"pass{i}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}/pass{i}")
    df = pd.read_csv(f[0])
    return df.groupby(["Kernel_Name", "Counter_Name"])["Counter_Value"].agg(["sum", "count"]).reset_index()


def pick(df, pats, counter):
    m = df[(df["Counter_Name"] == counter) & df["Kernel_Name"].apply(lambda n: any(p in n for p in pats))]
    return float(m["sum"].sum()), (int(m["count"].max()) if len(m) else 0)


def main():
    d, key, outdir = sys.argv[1:4]
    t = [None] + [table(d, i) for i in (1, 2, 3, 4)]
    h = kernel_source_hash()
    traffic, sq = {}, {}
    for stage, pats in STAGES.items():
        # launches of the stage = those of its most frequent kernel (a stage's first frame may run other kernels than the rest: the fused
        # k_preprocess_count needs a capacity, so frame 0 is k_preprocess + k_bin_count)
        launches = max(pick(t[1], [pt], "FETCH_SIZE")[1] for pt in pats)
        if not launches:
            continue
        fetch, _ = pick(t[1], pats, "FETCH_SIZE")
        write, _ = pick(t[2], pats, "WRITE_SIZE")
        fb, wb = fetch * 1024 * 2 / launches, write * 1024 / launches
        traffic[stage] = {"hbm_bytes_per_launch": int(fb + wb), "fetch_bytes_x2": int(fb), "write_bytes": int(wb), "launches_profiled": launches}
        g = lambda c, i: pick(t[i], pats, c)[0] / launches
        gui = g("GRBM_GUI_ACTIVE", 4) / XCDS
        waves = g("SQ_WAVES", 3)
        ent = {"kernels": pats, "gpu_cycles": int(gui), "us_at_2.4GHz": round(gui / CLOCK * 1e6, 1),
               "valu_wave_instructions": int(g("SQ_INSTS_VALU", 3)), "lds_wave_instructions": int(g("SQ_INSTS_LDS", 3)),
               "salu_wave_instructions": int(g("SQ_INSTS_SALU", 3)), "waves": int(waves),
               "valu_per_wave": round(g("SQ_INSTS_VALU", 3) / max(waves, 1), 1),
               "valu_busy": round(g("SQ_ACTIVE_INST_VALU", 3) * 4 / N_SIMD / max(gui, 1), 3),
               "lds_busy": round(g("SQ_ACTIVE_INST_LDS", 3) * 4 / N_SIMD / max(gui, 1), 3),
               "wave_cycles_waiting_frac": round(g("SQ_WAIT_ANY", 4) / max(g("SQ_WAVE_CYCLES", 4), 1), 3),
               "launches_profiled": launches}
        sq[stage] = ent
    note_t = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-graph` (tools/collect_counters.sh); bytes = "
              "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch of the stage's kernels (gfx950 FETCH_SIZE halving, MI355X_MICROARCH.md HBM section).")
    note_s = ("rocprofv3 --pmc, two SQ passes over `bench.py --no-graph` (tools/collect_counters.sh), averages per launch of the stage's kernels. "
              "Counters are summed over the 8 XCDs; valu_busy = SQ_ACTIVE_INST_VALU*4/1024 SIMDs/(GRBM_GUI_ACTIVE/8) (gfx94x VALUBusy formula); "
              "wave_cycles_waiting_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES.")
    for name, data, note in (("pmc_traffic.json", traffic, note_t), ("sq_counters.json", sq, note_s)):
        json.dump({"_source_hash": h, "_note": note, key: data}, open(os.path.join(outdir, name), "w"), indent=1, sort_keys=True)
    for k, v in traffic.items():
        s = sq[k]
        print(f"{k:22s} {v['hbm_bytes_per_launch'] / 1e6:8.2f} MB/launch  {s['us_at_2.4GHz']:7.1f} us  VALU/wave {s['valu_per_wave']:8.1f}  "
              f"VALU busy {s['valu_busy']:.2f}  LDS busy {s['lds_busy']:.2f}  waiting {s['wave_cycles_waiting_frac']:.2f}")
    print("source hash", h)


if __name__ == "__main__":
    main()
