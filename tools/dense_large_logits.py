"""Run ON THE GPU BOX: the streamed dense formulation on inputs scaled up until its softmax shift (an upper bound of the row maximum
from the bf16 scan) leaves the fp16 range of the weights: error against the fp64 oracle, and whether the range guard took the call.
   python tools/dense_large_logits.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
from oracle.ce_oracle import ce_forward_oracle

dev = torch.device("cuda:0")
params = {n: torch.from_numpy(a) for n, a in make_ce_params(57, variant="default").items()}
x = torch.from_numpy(make_features(57, 1, 64, 48, 52))
for scale in (1.0, 1.2, 1.4, 1.6, 1.8, 2.0, 2.5, 3.0):
    xs = x * scale
    want, st = ce_forward_oracle(xs, params, mode="adaptive", dtype=torch.float64, stages=True)
    top = float((10.0 * st["S"] * torch.relu(st["S"] - st["T"].unsqueeze(-1))).max())
    ce = CE(in_channels=64); ce.load_state_dict(params, strict=True); ce = ce.to(dev).eval()
    with torch.no_grad():
        out = ce(xs.to(dev)).cpu().numpy()
    w = want.float().numpy()
    err = float(np.abs(out - w).max() / np.abs(w).max())
    print(f"input x {scale}: largest logit {top:8.0f}  path {ce.last_info['path']}  range_fallback {ce.last_info['range_fallback']}  normwise error {err:.2e}  zeros {float((out == 0).mean()):.3f}")
