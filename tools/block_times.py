#!/usr/bin/env python3
"""Timeline of a kernel's blocks from the phase stamps an ablation build writes (dagl_amd/csrc/debug.hip, DAGL_TIMES_FILE).
   python tools/block_times.py <file> [skip_first_launches]
Per kernel, averaged over the recorded launches (100 MHz stamps -> microseconds):
   span            last exit - first entry
   entry           distribution of block entry times relative to the first entry (dispatch ramp / later rounds)
   prologue, loop, epilogue   per-block phase lengths
"""
import sys
import numpy as np

def q(a):
    return "min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (a.min(), np.percentile(a, 50), np.percentile(a, 90), a.max())

def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    by = {}
    for line in open(path):
        f = line.split()
        name, n = f[0], int(f[1])
        if name.endswith("_phases"):                  # per-wave sums of shader-clock deltas, 5 phases per wave
            w = np.array(f[2:2 + 4 * n], dtype=np.float64)
            w = w[:len(w) - len(w) % 5].reshape(-1, 5)
            by.setdefault(name, []).append(w)
            continue
        t = np.array(f[2:2 + 4 * n], dtype=np.float64).reshape(n, 4) / 100.0      # us
        by.setdefault(name, []).append(t)
    for name, runs in by.items():
        runs = runs[skip:] if len(runs) > skip else runs
        if name.endswith("_phases"):
            m = np.mean([r.mean(axis=0) for r in runs], axis=0)
            mx = np.mean([r.max(axis=0) for r in runs], axis=0)
            tot = m.sum()
            print(f"{name}: per-wave clocks inside the loop, mean over waves (max): total {tot:.0f}")
            for k, lab in enumerate(["tile request", "reads + multiplies", "test + extraction", "wait for tile", "barrier / publish"]):
                print(f"  {lab:20s} {m[k]:9.0f} ({mx[k]:9.0f})  {100 * m[k] / tot:5.1f} %")
            continue
        spans, ent, pro, loop, epi, ends = [], [], [], [], [], []
        for t in runs:
            ok = t[:, 3] > 0
            t = t[ok]
            t[:, 1] = np.where(t[:, 1] > 0, t[:, 1], t[:, 0]); t[:, 2] = np.where(t[:, 2] > 0, t[:, 2], t[:, 1])
            t0 = t[:, 0].min()
            spans.append(t[:, 3].max() - t0)
            ent.append(t[:, 0] - t0); pro.append(t[:, 1] - t[:, 0]); loop.append(t[:, 2] - t[:, 1]); epi.append(t[:, 3] - t[:, 2])
            ends.append(t[:, 3] - t0)
        print(f"{name}: {len(runs)} launches, {len(ent[0])} blocks; span {np.mean(spans):.1f} us")
        print("  entry     ", q(np.concatenate(ent)))
        print("  prologue  ", q(np.concatenate(pro)))
        print("  loop      ", q(np.concatenate(loop)))
        print("  epilogue  ", q(np.concatenate(epi)))
        print("  exit      ", q(np.concatenate(ends)))

if __name__ == "__main__":
    main()
