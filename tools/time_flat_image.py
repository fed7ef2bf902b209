#!/usr/bin/env python3
"""Run ON THE GPU BOX: the redo path of the top-k modes on a near-constant 256x256 map (every key inside the screen's band, the
candidate slots of every query group overflow, all of them are redone by the fp32 scan + merge behind the screen)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params

dev = torch.device("cuda:0")
params = {n: torch.from_numpy(a) for n, a in make_ce_params(52, variant="default").items()}
ce = CE(in_channels=64)
ce.load_state_dict(params, strict=True)
ce.select_mode = "topk"; ce.select_k = 8
ce = ce.to(dev).eval()
g = torch.Generator().manual_seed(3)
for name, x in (("flat", 0.25 + 1e-2 * torch.randn(1, 64, 256, 256, generator=g)), ("noise", torch.randn(1, 64, 256, 256, generator=g))):
    x = x.to(dev)
    with torch.no_grad():
        for _ in range(3):
            ce(x)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(10):
            ce(x)
        torch.cuda.synchronize()
    print(f"{name} 256x256: {(time.time() - t) * 100:.3f} ms per forward", ce.last_info)
