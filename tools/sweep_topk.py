"""Run ON THE GPU BOX: the top-k modes at odd sizes (rows that are no multiple of 32 or 4, batches), block against the oracle.
   python tools/sweep_topk.py   (one line per case; exit code 1 on a mismatch)"""
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
shapes = [(1, 40, 36), (2, 72, 72), (1, 100, 88), (1, 33, 47), (3, 64, 96), (1, 129, 65), (1, 128, 128)]
for (B, H, W) in shapes:
    for mode, k in (("topk", 4), ("topk", 8), ("topk", 16), ("adaptive_topk", 8)):
        seed = 7 + H + k
        prm = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant="sparse" if mode != "topk" else "default",
                                                                  sparse_gain=1.7).items()}
        x = torch.from_numpy(make_features(seed + 1, B, 64, H, W))
        with torch.no_grad():
            want = ce_forward_oracle(x, prm, mode=mode, k=k, dtype=torch.float64).float()
        m = CE(in_channels=64); m.load_state_dict(prm, strict=True); m.select_mode = mode; m.select_k = k; m = m.to(dev).eval()
        with torch.no_grad():
            out = m(x.to(dev)).cpu()
            ms = CE(in_channels=64); ms.load_state_dict(prm, strict=True); ms.select_mode = mode; ms.select_k = k; ms.scan = "exact"
            ms = ms.to(dev).eval()
            out_exact = ms(x.to(dev)).cpu()
        e = normwise(out.numpy(), want.numpy()); e2 = normwise(out_exact.numpy(), want.numpy())
        ok = e <= 1e-4 and e2 <= 1e-4
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} B={B} {H}x{W} {mode} k={k}: screened {e:.2e} (path {m.last_info['path']}) exact scan {e2:.2e}")
sys.exit(1 if bad else 0)
