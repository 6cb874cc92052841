#!/usr/bin/env python
"""Per-wave analysis of a blend-kernel timeline saved by tools/lane_use.py (TIMELINE=<file.npy>): which waves finish last and how fast
they ran, duration against visits / batches, waves still running over the span, SIMD ends per XCD.  python tools/timeline_analysis.py f.npy [fwd|bwd]"""
import sys
import numpy as np

a = np.load(sys.argv[1])
fwd = len(sys.argv) > 2 and sys.argv[2] == "fwd"
t0, t1 = a[:, 0] - a[:, 0].min(), a[:, 1] - a[:, 0].min()
dur, vis, n, depth = (t1 - t0).astype(float) * 10, a[:, 6].astype(float), a[:, 3].astype(float), a[:, 7].astype(float)      # ns
batches = np.ceil(np.minimum(depth + 64, n) / 64.0) if fwd else np.ceil(depth / 64.0)
xcc, cu, simd = a[:, 2] >> 16, (a[:, 2] >> 8) & 0xff, (a[:, 2] >> 4) & 3
print(f"span {t1.max() / 100:.1f} us; visits mean {vis.mean():.0f} max {vis.max():.0f}; batches mean {batches.mean():.1f} max {batches.max():.0f}")
print(f"corr(visits, duration) {np.corrcoef(vis, dur)[0, 1]:.2f}  corr(batches, duration) {np.corrcoef(batches, dur)[0, 1]:.2f}")
for lo, hi in ((1, 3), (3, 5), (5, 8), (8, 40)):
    m = (batches >= lo) & (batches < hi)
    if m.sum():
        print(f"  batches [{lo},{hi}): {m.sum():5d} waves, mean duration {dur[m].mean() / 1e3:6.1f} us, mean visits {vis[m].mean():6.1f}")
key = (xcc * 1000 + cu) * 4 + simd
print("ten last-finishing waves: end (us), visits, batches, ns per visit; waves of the same SIMD still running at 70 / 85 / 95 % of the span")
for i in np.argsort(-t1)[:10]:
    same = key == key[i]
    print(f"  {t1[i] / 100:6.1f}  {vis[i]:4.0f}  {batches[i]:2.0f}  {dur[i] / max(vis[i], 1):7.1f}   {[int((t1[same] > f * t1.max()).sum()) for f in (0.7, 0.85, 0.95)]}")
late, early = t1 > 0.9 * t1.max(), t1 < np.percentile(t1, 50)
print(f"mean ns per visit: waves ending in the last 10 % of the span {(dur[late] / np.maximum(vis[late], 1)).mean():.0f}, in the first half {(dur[early] / np.maximum(vis[early], 1)).mean():.0f}")
print("waves still running at 25 / 50 / 75 / 90 / 97 % of the span:", [int((t1 > f * t1.max()).sum()) for f in (0.25, 0.5, 0.75, 0.9, 0.97)], "of", len(a))
for x in np.unique(xcc):
    m = xcc == x
    ids, inv = np.unique(key[m], return_inverse=True)
    e, v = np.zeros(len(ids)), np.zeros(len(ids))
    np.maximum.at(e, inv, t1[m]); np.add.at(v, inv, vis[m])
    print(f"  XCD {x}: SIMD end min/mean/max {e.min() / 100:6.1f} {e.mean() / 100:6.1f} {e.max() / 100:6.1f} us; visits per SIMD mean {v.mean():6.1f} max {v.max():5.0f}; corr {np.corrcoef(v, e)[0, 1]:.2f}")
