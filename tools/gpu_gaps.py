#!/usr/bin/env python
"""Idle gaps of the GPU between consecutive kernels, from a rocprofv3 --kernel-trace database: which kernel the GPU waited for,
how long, per training step (a step = one k_render_backward launch).   python tools/gpu_gaps.py results.db [budget.json]
budget.json (optional): {"_source_hash", "wall_us_per_step", "kernels_us": {kernel: us per launch}} -- bench.py reads it (profiles/
graph_step_budget.json) for the roofline of the REPLAYED step while the hash is the library's."""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
adam = [k for k, r in enumerate(rows) if "k_render_backward" in r[0]]
rows = rows[adam[len(adam) // 4]:adam[3 * len(adam) // 4] + 1]       # the steady middle half of the training steps
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
gap_by, busy, steps = collections.defaultdict(float), 0.0, 0
big = collections.defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    busy += (e0 - s0)
    gap_by[short(n1)] += max(0, s1 - e0)
    if s1 - e0 > 3000:
        big[short(n1)].append((s1 - e0) / 1e3)
    steps += "k_render_backward" in n1
wall = rows[-1][2] - rows[0][1]
print(f"{steps} steps; per step: wall {wall / steps / 1e3:.1f} us, kernels busy {busy / steps / 1e3:.1f} us, idle {(wall - busy) / steps / 1e3:.1f} us")
for n, g in sorted(gap_by.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  GPU idle before {n:46s} {g / steps / 1e3:7.1f} us / step")
per = collections.defaultdict(lambda: [0.0, 0])
for n, s0, e0 in rows:
    per[short(n)][0] += e0 - s0; per[short(n)][1] += 1
print("kernel time per step (us):")
for n, (t, k) in sorted(per.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {n:46s} {t / steps / 1e3:7.1f}   ({k / steps:.1f} launches / step, {t / k / 1e3:.1f} us each)")

for n, g in big.items():
    print(f"  gaps > 3 us before {n}: {len(g)} (mean {sum(g) / len(g):.1f} us, max {max(g):.1f} us)")

if len(sys.argv) > 2:
    import json, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from egogaussian_amd import lib as _lib
    json.dump({"_source_hash": _lib.kernel_source_hash(), "steps": steps, "wall_us_per_step": round(wall / steps / 1e3, 2),
               "idle_us_per_step": round((wall - busy) / steps / 1e3, 2),
               "kernels_us": {n: round(t / k / 1e3, 3) for n, (t, k) in per.items()},
               "launches_per_step": {n: round(k / steps, 3) for n, (t, k) in per.items()}}, open(sys.argv[2], "w"), indent=1)
