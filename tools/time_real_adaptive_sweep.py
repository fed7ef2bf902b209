"""Run ON THE GPU BOX: the shipped adaptive semantics on natural-image features with the mask made sparser and sparser (the trained
head's bias head shifted down): lists + per-query redo, the hand-over to the dense formulation, and what each costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd.net import RR, set12_protocol_noise
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({n: torch.from_numpy(z[n].astype(np.float32)) for n in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
clean = torch.from_numpy(imgs["img_02"].astype(np.float32) / 255.0)[None, None]
noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
with torch.no_grad():
    x = net.head(noisy)
    for blk in net.body[:8]: x = blk(x)
x = x.contiguous()
ce = net.body[8].c1_1; ce.select_mode = "adaptive"
b0 = ce.bias_conv.bias.detach().clone()
with torch.no_grad():
    for shift in [float(v) for v in (sys.argv[1:] or (0.0, 0.3, 0.6, 0.9, 1.1, 1.2, 1.3, 1.4, 1.5, 1.6, 1.7, 1.8, 1.9, 2.0, 2.5, 5.0))]:
        ce.bias_conv.bias.copy_(b0 - shift); ce.invalidate_packed() if hasattr(ce, "invalidate_packed") else None
        ce._dense_hint = False
        for _ in range(4): ce(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): ce(x)
        e1.record(); e1.synchronize()
        i = ce.last_info or {}
        L = 4096
        print(f"bias shift -{shift:4.1f}: {e0.elapsed_time(e1)/8:8.4f} ms  path {i.get('path')}  mean degree {i.get('total_edges', 0)/L:9.1f}  max {i.get('max_degree')}  redone {i.get('redone_queries')}", flush=True)
