"""Run ON THE GPU BOX: one 256x256 image through the whole RR (12 CE heads, reference tiling) -- tile by tile as the reference
does it (chop_forward) and with all leaf tiles as one batch (chop_forward_batched), dense masks (stand-in weights) and top-k 8."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagl_amd.ce import CE
from dagl_amd.net import RR, chop_forward, chop_forward_batched, seeded_state_dict

dev = torch.device("cuda:0")
m = RR().eval()
m.load_state_dict(seeded_state_dict(m.state_dict(), 7), strict=True)
m = m.to(dev)
x = torch.rand(1, 1, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)

def t(fn, n=3):
    with torch.no_grad():
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): y = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, y

modes = [a for a in sys.argv[1:] if a in ("adaptive", "topk")] or ["adaptive", "topk"]
only_batched = "batched" in sys.argv[1:]
for mode in modes:
    for h in m.modules():
        if isinstance(h, CE):
            h.select_mode, h.select_k = mode, 8
    b, yb = t(lambda: chop_forward_batched(m, x))
    if only_batched:
        print(f"{mode:9s} all leaves as one batch {b:7.1f} ms"); continue
    a, ya = t(lambda: chop_forward(m, x))
    print(f"{mode:9s} tile by tile {a:7.1f} ms   all leaves as one batch {b:7.1f} ms   max |diff| {(ya - yb).abs().max().item():.2e}")
