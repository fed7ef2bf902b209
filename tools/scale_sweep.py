"""Run ON THE GPU BOX: every mode on inputs scaled from 1e-3 to 30 (features from the fp16 denormals of the split operands up to the
range guard): error against the fp64 oracle, degree mismatches, and which path served the call.
   python tools/scale_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
from oracle.ce_oracle import ce_forward_oracle

dev = torch.device("cuda:0")
worst = 0.0
for mode, k, variant in (("adaptive", 0, "sparse"), ("adaptive", 0, "default"), ("topk", 8, "default"), ("adaptive_topk", 16, "sparse"), ("topk", 200, "default")):
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(59, variant=variant, sparse_gain=1.7).items()}
    x = torch.from_numpy(make_features(59, 1, 64, 56, 60))
    for scale in (1e-3, 0.03, 0.3, 1.0, 3.0, 10.0, 30.0):
        xs = x * scale
        want = ce_forward_oracle(xs, params, mode=mode, k=k or None, dtype=torch.float64).float().numpy()
        ce = CE(in_channels=64); ce.load_state_dict(params, strict=True); ce.select_mode = mode
        if k: ce.select_k = k
        ce = ce.to(dev).eval()
        with torch.no_grad():
            out = ce(xs.to(dev)).cpu().numpy()
        den = np.abs(want).max()
        err = float(np.abs(out - want).max() / (den if den > 0 else 1.0))
        info = ce.last_info or {}
        flag = "" if err <= 1e-4 else "   <-- above 1e-4"
        worst = max(worst, err)
        print(f"{mode:14s} {variant:8s} k={k:3d} input x {scale:<6g}: normwise {err:.2e}  path {info.get('path')}  range_fallback {info.get('range_fallback')}  scan {ce.scan}{flag}")
print(f"worst {worst:.2e}")
