"""Run ON THE GPU BOX: a BATCH of 512 x 512 natural-image feature maps (trained checkpoint), top-k 8 / 50: the candidate records of a
large batch of large maps are capped at 2 GiB -- do the slots still hold?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd import ops
from dagl_amd.net import RR, set12_protocol_noise
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
names = [n for n in sorted(imgs.files) if imgs[n].shape[-1] == 512]
clean = torch.stack([torch.from_numpy(imgs[n].astype(np.float32) / 255.0)[None] for n in names[:4]])
noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
with torch.no_grad():
    x = net.head(noisy)
    for blk in net.body[:8]: x = blk(x)
x = x.contiguous()
ce = net.body[8].c1_1
for nb in (1, 2, 4):
    for k in (8, 50):
        ce.select_mode, ce.select_k = "topk", k
        ce.topk_threshold = "auto"; ce.reset_topk_policy()
        xb = x[:nb].contiguous()
        with torch.no_grad():
            for _ in range(4): ce(xb)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ce(xb)
            e1.record(); e1.synchronize()
            shape, d = ce._last_call
            bad = ops.ce_range_check(shape, "topk", k, ce._ws, d)
        ms = e0.elapsed_time(e1) / 5
        print(f"[{nb},64,512,512] natural features top-k {k:2d}: {ms:8.3f} ms = {ms/nb:7.3f} per image  flags {bad}", flush=True)
