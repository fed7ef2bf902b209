"""Run ON THE GPU BOX: one head, top-k 8, trained features of every Set12 image (whole map + the 64 leaf tiles of 72x72 for the 256^2 ones): first call, steady state, policy / redo flags.  THR=full|sparse forces a threshold."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd import ops
from dagl_amd.net import RR, set12_protocol_noise, chop_leaf_boxes
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
ce = net.body[8].c1_1; ce.select_mode = "topk"; ce.select_k = 8
def run(x, label):
    ce.topk_threshold = os.environ.get("THR", "auto"); ce.reset_topk_policy()
    with torch.no_grad():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ce(x); e1.record(); e1.synchronize(); first = e0.elapsed_time(e1)
        for _ in range(10): ce(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(30): ce(x)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 30
        shape, d = ce._last_call
        bad = ops.ce_range_check(shape, "topk", 8, ce._ws, d)
    print(f"{label:34s} first call {first:7.3f} ms   steady {ms:7.4f} ms   flags {bad} (4 = last call's redo pass had work, 8 = tight policy)", flush=True)
for name in sorted(imgs.files):
    clean = torch.from_numpy(imgs[name].astype(np.float32) / 255.0)
    if clean.ndim == 2: clean = clean[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
    with torch.no_grad():
        x = net.head(noisy)
        for blk in net.body[:8]: x = blk(x)
    H = x.shape[-1]
    run(x.contiguous(), f"Set12 {name} whole {H}x{H}")
    if H == 256:
        # the 64 leaf tiles of 72x72 the tiled driver feeds a head
        tiles = torch.stack([x[0, :, y0:y0+72, x0:x0+72] for y0 in range(0, 256-71, 26)[:8] for x0 in range(0, 256-71, 26)[:8]])
        run(tiles.contiguous(), f"Set12 {name} 64 tiles of 72x72")
