#!/usr/bin/env python3
"""Run ON THE GPU BOX: what the device-resident top-k threshold policy costs on the headline workload (synthetic 256x256, k = 8, where
the sampled threshold suffices): CE.topk_threshold = "auto" (the kernels read the workspace's policy word) against "sparse"
(DAGL_FLAG_SAMPLED_TOPK: no policy pointer), interleaved in one process."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
dev = torch.device("cuda:0")
mods = {}
for pol in ("sparse", "auto"):
    m = CE(in_channels=64)
    m.load_state_dict({n: torch.from_numpy(a) for n, a in make_ce_params(2024, variant="default").items()}, strict=True)
    m.select_mode, m.select_k, m.topk_threshold = "topk", 8, pol
    mods[pol] = m.to(dev).eval()
x = torch.from_numpy(make_features(100, 1, 64, 256, 256)).to(dev)
with torch.no_grad():
    for m in mods.values():
        for _ in range(300):
            m(x)
    torch.cuda.synchronize()
    for r in range(4):
        for pol, m in mods.items():
            for _ in range(50):
                m(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(400):
                m(x)
            torch.cuda.synchronize()
            print(f"{pol:7s} {(time.perf_counter() - t0) / 400 * 1e3:.4f} ms per forward")
print("auto: policy word tight?", mods["auto"].topk_policy_is_tight())
