"""Run ON THE GPU BOX: top-k modes on a 1024^2 mosaic of two natural images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd import ops
from dagl_amd.net import RR, set12_protocol_noise
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
a = imgs["img_11"].astype(np.float32) / 255.0; b = imgs["img_12"].astype(np.float32) / 255.0
big = np.block([[a, b[:, ::-1]], [b[::-1], a[::-1, ::-1]]])                       # 1024 x 1024 mosaic of two natural images
clean = torch.from_numpy(big.copy())[None, None]
noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
with torch.no_grad():
    x = net.head(noisy)
    for blk in net.body[:8]: x = blk(x)
x = x.contiguous()
ce = net.body[8].c1_1
for mode, k in (("topk", 8), ("adaptive_topk", 16), ("topk", 50)):
    ce.select_mode, ce.select_k = mode, k
    ce.topk_threshold = "auto"; ce.reset_topk_policy()
    with torch.no_grad():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ce(x); e1.record(); e1.synchronize(); first = e0.elapsed_time(e1)
        for _ in range(3): ce(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5): ce(x)
        e1.record(); e1.synchronize()
        shape, d = ce._last_call
        bad = ops.ce_range_check(shape, mode, k, ce._ws, d)
    print(f"1024x1024 natural mosaic, trained features, {mode} k={k}: first {first:8.2f} ms steady {e0.elapsed_time(e1)/5:8.2f} ms flags {bad}", flush=True)
