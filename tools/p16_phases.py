#!/usr/bin/env python3
"""project16_body2's loop by phase (ablation build with -DDAGL_P16_PHASES): wave 0's shader-clock sums per block.
   python tools/p16_phases.py <times file> <n_full> [skip]"""
import sys
import numpy as np
path, n_full = sys.argv[1], int(sys.argv[2])
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 4
runs = []
for line in open(path):
    f = line.split()
    if f[0] != "project16_kernel": continue
    n = int(f[1]); runs.append(np.array(f[2:2 + 4 * n], dtype=np.float64).reshape(n, 4))
runs = runs[skip:]
lab = ["requests (LDS-DMA issue)", "reads + multiplies issued", "counted wait for tap t+1", "barrier"]
for name, sel in (("first on its CU (ids 0..255)", slice(0, 256)), ("second (ids 256..511)", slice(256, n_full))):
    a = np.mean([r[sel] for r in runs], axis=0)
    tot = a.sum(axis=1).mean()
    print(f"{name}: {tot:.0f} clocks in the loop's 46 steady taps = {tot / 46:.0f} per tap")
    for k in range(4):
        print(f"   {lab[k]:28s} {a[:, k].mean() / 46:7.0f} per tap  {100 * a[:, k].mean() / tot:5.1f} %")
