"""Run ON THE GPU BOX: adaptive mode at small sizes over a range of mask densities, block against the fp64 oracle.
   python tools/sweep_adaptive.py   (prints one line per case; exit code 1 on a mismatch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
from oracle.ce_oracle import ce_forward_oracle

def normwise(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

dev = torch.device("cuda:0")
bad = 0
cases = [(B, H, W, g, s) for (B, H, W) in [(1, 64, 64), (2, 72, 56), (1, 96, 96), (1, 128, 128)]
         for g in (1.2, 1.5, 1.7, 1.8, 1.9) for s in (3, 17)]
for B, H, W, gain, seed in cases:
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant="sparse", sparse_gain=gain).items()}
    x = torch.from_numpy(make_features(seed + 1, B, 64, H, W))
    with torch.no_grad():
        want, st = ce_forward_oracle(x, prm, mode="adaptive", k=None, stages=True, dtype=torch.float64)
    deg = st["deg"].numpy().reshape(-1)
    m = CE(in_channels=64); m.load_state_dict(prm, strict=True); m.select_mode = "adaptive"; m = m.to(dev).eval()
    with torch.no_grad():
        out = m(x.to(dev)).cpu()
    e = normwise(out.numpy(), want.float().numpy())
    info = m.last_info
    ok = e <= 1e-4 and abs(info["total_edges"] - int(deg.sum())) <= max(2, int(1e-4 * deg.size))
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} B={B} {H}x{W} gain={gain} seed={seed}: err {e:.2e} mean deg {deg.mean():.1f} max {deg.max()} "
          f"(64..256: {int(((deg >= 64) & (deg <= 256)).sum())}, >256: {int((deg > 256).sum())}) path {info['path']} "
          f"redone {info['redone_queries']} edges {info['total_edges']} vs {int(deg.sum())}")
sys.exit(1 if bad else 0)
