#!/usr/bin/env python3
"""project16_kernel's blocks by kind from the phase stamps of an ablation build (DAGL_TIMES_FILE): 4-tile / 3-tile blocks of the
full round (block id -> group as in project16_kernel), the single-tile remainder blocks, the riding thr / bias blocks.
   python tools/p16_timeline.py <times file> <n_full> <n_proj> [skip launches]"""
import sys
import numpy as np

path, n_full, n_proj = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 4
runs = []
for line in open(path):
    f = line.split()
    if f[0] != "project16_kernel":
        continue
    n = int(f[1])
    runs.append(np.array(f[2:2 + 4 * n], dtype=np.float64).reshape(n, 4) / 100.0)
runs = runs[skip:]
def q(a): return "min %5.1f  p10 %5.1f  p50 %5.1f  p90 %5.1f  max %5.1f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
kinds = {}
for t in runs:
    t0 = t[t[:, 0] > 0, 0].min()
    for bid in range(t.shape[0]):
        if t[bid, 3] <= 0: continue
        if bid < (n_full & ~15):
            qq = bid >> 3; grp = (qq ^ (qq >> 5)) & 1; kind = "4-tile" if grp == 0 else "3-tile"
        elif bid < n_full: kind = "4-tile" if (bid & 1) == 0 else "3-tile"
        elif bid < n_proj: kind = "single-tile"
        else: kind = "thr/bias"
        e = t[bid] - t0
        kinds.setdefault(kind, []).append((e[0], (e[1] if t[bid, 1] > 0 else e[0]) - e[0], (e[2] - e[1]) if t[bid, 2] > 0 and t[bid, 1] > 0 else 0.0, e[3] - (e[2] if t[bid, 2] > 0 else e[0]), e[3]))
print(f"{len(runs)} launches; span {np.mean([ (t[t[:,3]>0,3].max() - t[t[:,0]>0,0].min()) for t in runs]):.1f} us")
for kind, rows in kinds.items():
    a = np.array(rows)
    print(f"{kind:12s} n={len(rows) // len(runs)}")
    for j, lab in enumerate(("entry", "prologue", "loop", "epilogue", "exit")):
        print(f"    {lab:9s} {q(a[:, j])}")
