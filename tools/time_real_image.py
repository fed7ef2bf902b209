#!/usr/bin/env python3
"""Run ON THE GPU BOX: one CE head forward on REAL features -- a Set12 image (sigma 50) through the trained checkpoint's head conv
and first eight ResBlocks, whole 256x256 map, top-k k=8 -- next to the benchmark's synthetic N(0,1) features: stage times of
both.  Natural-image scores are not spread like the synthetic map's: the threshold sampled from every 8th key tile lets hundreds to
thousands of keys through and the call lands on the fp32 redo pass; CE.topk_threshold = "auto" notices after the first call and
takes the threshold from every second key tile (DAGL_FLAG_TIGHT_TOPK)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagl_amd import ops
from dagl_amd._lib import STAGE_NAMES
from dagl_amd.net import RR, set12_protocol_noise
from dagl_amd.synth import make_features

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval()
net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True)
net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
ces = net.body[8]
ce = ces.c1_1
ce.select_mode = "topk"; ce.select_k = 8


def time_head(x, label, threshold="auto"):
    ce.topk_threshold = threshold
    ce.reset_topk_policy()
    prof = ops.StageProfile(20)
    with torch.no_grad():
        for _ in range(30):
            ce(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            ce(x)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 100
        prof.select_stage(-1)
        ce.profile = prof
        for _ in range(20):
            ce(x)
        torch.cuda.synchronize()
        ce.profile = None
    st = np.asarray(prof.read()).mean(axis=0)
    print(f"{label:28s} {ms:.4f} ms  " + "  ".join(f"{STAGE_NAMES[i]} {st[i] * 1e3:.1f}" for i in range(8)))


for name in sorted(imgs.files):
    clean = torch.from_numpy(imgs[name].astype(np.float32) / 255.0)      # uint8 images; the network works on [0, 1] (rgb_range 1)
    if clean.ndim == 2:
        clean = clean[None, None]
    if clean.shape[-1] != 256 or clean.shape[-2] != 256:
        continue
    noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
    with torch.no_grad():
        x = net.head(noisy)
        for blk in net.body[:8]:
            x = blk(x)
    time_head(x.contiguous(), f"Set12 {name}, sampled thr.", "sparse")
    time_head(x.contiguous(), f"Set12 {name}, auto")
time_head(torch.from_numpy(make_features(100, 1, 64, 256, 256)).to(dev), "synthetic N(0,1), auto")
