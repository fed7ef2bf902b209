"""Run ON THE GPU BOX: RR training step on natural crops with the trained checkpoint."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd.ce import CE
from dagl_amd.net import RR
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
imgs = np.load(os.path.join(G, "set12.npz"))
crops = []
for i, n in enumerate(sorted(imgs.files)[:8]):
    a = imgs[n].astype(np.float32) / 255.0
    crops.append(torch.from_numpy(a[40:168, 60:188])[None])
clean = torch.stack(crops).to(dev)                                  # [8,1,128,128]
gen = torch.Generator(device="cpu").manual_seed(0)
noisy = (clean.cpu() + torch.randn(clean.shape, generator=gen) * 50 / 255).to(dev)
for mode, k in (("topk", 8), ("topk", 50), ("adaptive", 0)):
    net = RR().train(); net.load_state_dict({kk: torch.from_numpy(z[kk].astype(np.float32)) for kk in z.files}, strict=True); net = net.to(dev)
    for m in net.modules():
        if isinstance(m, CE):
            m.select_mode = mode
            if k: m.select_k = k
    opt = torch.optim.Adam(net.parameters(), lr=1e-5)
    for i in range(7):
        if i == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = (net(noisy) - clean).abs().mean()
        loss.backward(); opt.step()
    torch.cuda.synchronize()
    print(f"RR train step on natural crops [8,1,128,128], trained weights, {mode} k={k}: {(time.perf_counter()-t0)/4*1e3:.1f} ms  loss {loss.item():.4f}", flush=True)
