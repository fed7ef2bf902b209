"""Run ON THE GPU BOX: one Set12 image through the whole RR (trained weights, tiled, all leaves as one batch): adaptive, top-k 8, top-k 50."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd.ce import CE
from dagl_amd.net import RR, chop_forward_batched, set12_protocol_noise
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
def t(fn, n=3):
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return first, (time.perf_counter() - t0) / n * 1e3
for name in ("img_02", "img_11"):
    clean = torch.from_numpy(imgs[name].astype(np.float32) / 255.0)[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
    for mode, k in (("adaptive", 8), ("topk", 8), ("topk", 50)):
        for h in net.modules():
            if isinstance(h, CE):
                h.select_mode, h.select_k = mode, k
                h.reset_topk_policy()
        first, ms = t(lambda: chop_forward_batched(net, noisy))
        print(f"whole RR, Set12 {name} {tuple(noisy.shape[-2:])}, trained weights, {mode} k={k}: first call {first:8.1f} ms, steady {ms:8.1f} ms", flush=True)
