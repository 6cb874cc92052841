#!/usr/bin/env python
"""profiles/r3_footprint_sweep.md from the JSON lines of `bench.py --footprints` and of the default `bench.py` run (config B / D legs)."""
import json, sys
foot = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["footprints"]
base = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]) if len(sys.argv) > 2 else {}
legs = {}
if "config_B_forward_only" in base:
    legs["config B: S(100k) forward only"] = base["config_B_forward_only"]
if "config_D_op_only" in base:
    legs["config D: S(1M) @ 1920x1080"] = base["config_D_op_only"]
legs.update({"S(500k), splats x2": foot.get("S500k_scale_x2", {}), "S(500k), splats x3": foot.get("S500k_scale_x3", {}),
             "densified model": foot.get("densified_model", {})})
print("# Round 3: the rasterizer at heavier footprints than the headline scene\n")
print("`python bench.py --footprints` (+ the config B / D legs of the default run), one MI355X; rasterizer forward + backward with colour, depth and")
print("alpha gradients (config B: forward only); per-stage times from HIP events the library records around each stage (us per launch).")
print("S(500k) itself: 67 pixel-splat pairs per pixel, 918 instances per tile after culling, 37 of 64 lanes kept per (wave, splat) visit.\n")
cols = ["preprocess", "tile_bucket", "tile_sort", "render_forward", "render_backward", "preprocess_backward"]
print("| scene | Gaussians | R (rect.) | after culling | list mean / max | pairs Q | lanes kept | " + " | ".join(cols) + " | op ms |")
print("|---|---:|---:|---:|---:|---:|---:|" + "---:|" * (len(cols) + 1))
for name, v in legs.items():
    if not v or "error" in v:
        print(f"| {name} | error: {v.get('error') if v else 'absent'} |"); continue
    st = v["stages"]
    print(f"| {name} | {v['gaussians']:,} | {v['instances_R']:,} | {v['instances_after_tile_culling']:,} | {v['tile_list_len_mean']:.0f} / {v['tile_list_len_max']} | "
          f"{v['pairs_Q'] / 1e6:.1f} M | {v['lanes_kept_per_visit']:.1f} | " + " | ".join(f"{1e3 * st[c]['ms_per_launch']:.1f}" if c in st else "-" for c in cols) + f" | {v['op_ms']:.3f} |")
print()
for name, v in legs.items():
    if v and "workload" in v:
        print(f"* **{name}**: {v['workload']}.")
