#!/usr/bin/env python3
"""Which blocks of project16_kernel shared a CU (ablation build with -DDAGL_P16_HWID): loop time by the kinds of a CU's blocks.
   python tools/p16_cu_pairs.py <times file> <n_full> <n_proj> [skip]"""
import sys, collections
import numpy as np
path, n_full, n_proj = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 4
runs = []
for line in open(path):
    f = line.split()
    if f[0] != "project16_kernel": continue
    n = int(f[1]); runs.append(np.array(f[2:2 + 4 * n], dtype=np.uint64).reshape(n, 4))
runs = runs[skip:]
agg = collections.defaultdict(list)
for t in runs:
    t0 = min(int(v) for v in t[:, 0] if v > 0)
    cus = collections.defaultdict(list)
    for bid in range(min(n_full, t.shape[0])):
        qq = bid >> 3
        grp = ((qq ^ (qq >> 5)) & 1) if bid < (n_full & ~15) else (bid & 1)
        hw = int(t[bid, 1]); cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 16) & 0xf
        cus[(xcc, se, sh, cu)].append((bid, 4 - grp, (int(t[bid, 0]) - t0) / 100.0, (int(t[bid, 2]) - int(t[bid, 0])) / 100.0, (int(t[bid, 3]) - t0) / 100.0))
    for key, bl in cus.items():
        kinds = tuple(sorted(b[1] for b in bl))
        for b in bl:
            agg[(kinds, b[1])].append((b[3], b[4]))
    if t is runs[0]:
        print("CUs seen:", len(cus), " blocks per CU:", collections.Counter(len(v) for v in cus.values()))
        for key in sorted(cus)[:6]: print("  ", key, [(b[0], b[1]) for b in cus[key]])
for (kinds, k), v in sorted(agg.items()):
    a = np.array(v)
    print(f"CU holds {kinds}: {k}-tile block  n={len(v)//len(runs):4d}  prologue+loop p50 {np.median(a[:,0]):5.1f} (min {a[:,0].min():5.1f} max {a[:,0].max():5.1f})  exit p50 {np.median(a[:,1]):5.1f} max {a[:,1].max():5.1f}")
# loop time by XCC / SE / position in the dispatch order
by = collections.defaultdict(list)
for t in runs:
    for bid in range(min(n_full, t.shape[0])):
        hw = int(t[bid, 1]); cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 16) & 0xf
        lp = (int(t[bid, 2]) - int(t[bid, 0])) / 100.0
        by[("xcc", xcc)].append(lp); by[("se", se)].append(lp); by[("cu", cu)].append(lp); by[("q>>3", bid >> 6)].append(lp)
for k in sorted(by): print(k, "n=%d" % (len(by[k]) // len(runs)), "p50 %.1f  min %.1f  max %.1f" % (np.median(by[k]), min(by[k]), max(by[k])))
# per launch: is it the same CUs that are slow every time?
per_cu = collections.defaultdict(list)
for t in runs:
    for bid in range(min(n_full, t.shape[0])):
        hw = int(t[bid, 1]) & 0xfffff
        per_cu[hw].append((int(t[bid, 2]) - int(t[bid, 0])) / 100.0)
m = np.array([np.mean(v) for v in per_cu.values()]); s = np.array([np.std(v) for v in per_cu.values()])
print("per-CU mean loop: min %.1f p50 %.1f max %.1f; mean within-CU std over launches %.1f" % (m.min(), np.median(m), m.max(), s.mean()))
