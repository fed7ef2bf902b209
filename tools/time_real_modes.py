"""Run ON THE GPU BOX: every select mode / k on trained features of a 256^2 and a 512^2 Set12 image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd.net import RR, set12_protocol_noise
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
ce = net.body[8].c1_1
def run(x, label, mode, k):
    ce.select_mode, ce.select_k = mode, k
    ce.topk_threshold = "auto"; ce.reset_topk_policy(); ce._dense_hint = False
    with torch.no_grad():
        for _ in range(6): ce(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ce(x)
        e1.record(); e1.synchronize()
    print(f"{label:26s} {mode:14s} k={k:3d}: {e0.elapsed_time(e1)/20:8.4f} ms   info {ce.last_info}", flush=True)
for name in ("img_02", "img_11"):
    clean = torch.from_numpy(imgs[name].astype(np.float32) / 255.0)[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
    with torch.no_grad():
        x = net.head(noisy)
        for blk in net.body[:8]: x = blk(x)
    x = x.contiguous(); H = x.shape[-1]
    for mode, k in (("topk", 8), ("topk", 50), ("topk", 64), ("adaptive_topk", 16), ("adaptive_topk", 64), ("adaptive", 0), ("topk", 100)):
        if H == 512 and k == 100: continue
        run(x, f"Set12 {name} {H}x{H}", mode, k)
