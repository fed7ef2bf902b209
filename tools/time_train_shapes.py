"""Run ON THE GPU BOX: RR training step at the reference trainers' default shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagl_amd.ce import CE
from dagl_amd.net import RR, seeded_state_dict
dev = torch.device("cuda:0")
for colors, B, crop in ((1, 32, 64), (3, 64, 64), (3, 8, 128)):
    for mode in ("topk", "topk50", "adaptive"):
        net = RR(n_colors=colors) if colors != 1 else RR()
        net.load_state_dict(seeded_state_dict(net.state_dict(), 7), strict=True)
        net = net.to(dev).train()
        for m in net.modules():
            if isinstance(m, CE):
                m.select_mode, m.select_k = ("topk", 50) if mode == "topk50" else (mode, 8)
        opt = torch.optim.Adam(net.parameters(), lr=1e-5)
        g = torch.Generator().manual_seed(1)
        x = torch.rand(B, colors, crop, crop, generator=g).to(dev); y = torch.rand(B, colors, crop, crop, generator=g).to(dev)
        for i in range(7):
            if i == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            loss = (net(x) - y).abs().mean(); loss.backward(); opt.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        print(f"RR(n_colors={colors}) train step [{B},{colors},{crop},{crop}] {mode}: {ms:7.1f} ms = {B / ms * 1e3:7.1f} img/s", flush=True)
        del net, opt
        torch.cuda.empty_cache()
