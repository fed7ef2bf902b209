#!/usr/bin/env python3
"""Run ON THE GPU BOX: N forwards of ONE head in the dense regime (shipped adaptive semantics), for timing / rocprofv3 / ablations.
   python tools/dense_case.py real|synth|leaf [calls]
   real   Set12 img_01 (sigma 50, [0,1]) through the trained checkpoint's head conv + 8 ResBlocks, whole 256x256 map, head c1_1
          (mask density 1.0, logits below ~70: no weight is exactly zero, every multiply runs)
   leaf   the same image's 64 leaf tiles of 72x72 as one batch [64,64,72,72] (what forward_chop feeds a head)
   synth  N(0,1) features, default-initialised head (logits of hundreds: ~3/4 of the A V granules are exactly zero and skipped)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kind = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
if kind == "synth":
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    ce = CE(in_channels=64)
    ce.load_state_dict({n: torch.from_numpy(a) for n, a in make_ce_params(1, variant="default").items()}, strict=True)
    ce = ce.to(dev).eval()
    x = torch.from_numpy(make_features(100, 1, 64, 256, 256)).to(dev)
else:
    from dagl_amd.net import RR, set12_protocol_noise, chop_leaf_boxes
    G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
    net = RR().eval()
    net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True)
    net = net.to(dev)
    imgs = np.load(os.path.join(G, "set12.npz"))
    clean = torch.from_numpy(imgs["img_01"].astype(np.float32) / 255.0)[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
    if kind == "leaf":
        noisy = torch.stack([noisy[0, :, y0:y1, x0:x1] for (y0, y1, x0, x1) in chop_leaf_boxes(256, 256)])
    with torch.no_grad():
        x = net.head(noisy)
        for blk in net.body[:8]:
            x = blk(x)
    x = x.contiguous()
    ce = net.body[8].c1_1
ce.select_mode = "adaptive"
with torch.no_grad():
    for _ in range(6):
        ce(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        ce(x)
    e1.record(); e1.synchronize()
print(f"dense_case {kind} {tuple(x.shape)}: {e0.elapsed_time(e1) / calls:.4f} ms per call (path {(ce.last_info or {}).get('path')})")
