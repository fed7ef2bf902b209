"""Run ON THE GPU BOX: whole RR on odd-sized natural images, batched vs tile by tile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd.ce import CE
from dagl_amd.net import RR, chop_forward_batched, chop_forward, set12_protocol_noise
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
def t(fn, n=3):
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter(); y = fn(); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return first, (time.perf_counter() - t0) / n * 1e3, y
clean = torch.from_numpy(imgs["img_11"].astype(np.float32) / 255.0)[None, None]
for (h, w) in ((321, 481), (180, 180), (481, 321)):
    noisy = set12_protocol_noise(clean[..., :h, :w].contiguous(), 50.0, 1.0).to(dev)
    for mode, k in (("adaptive", 8), ("topk", 8)):
        for m in net.modules():
            if isinstance(m, CE):
                m.select_mode, m.select_k = mode, k; m.reset_topk_policy()
        f1, ms1, y1 = t(lambda: chop_forward_batched(net, noisy))
        f2, ms2, y2 = t(lambda: chop_forward(net, noisy), n=1)
        print(f"whole RR {h}x{w} {mode} k={k}: batched first {f1:8.1f} steady {ms1:7.1f} ms | tile by tile first {f2:8.1f} steady {ms2:7.1f} ms | max diff {(y1 - y2).abs().max().item():.2e}", flush=True)
