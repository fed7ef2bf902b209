"""Run ON THE GPU BOX: the same on leaf-tile batches [64|256, 64, 72, 72]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd import ops
from dagl_amd._lib import STAGE_NAMES
from dagl_amd.net import RR, set12_protocol_noise, chop_leaf_boxes
from dagl_amd.synth import make_features
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
ce = net.body[8].c1_1
def run(x, label, k, thr="auto"):
    ce.select_mode, ce.select_k = "topk", k
    ce.topk_threshold = thr; ce.reset_topk_policy()
    prof = ops.StageProfile(10)
    with torch.no_grad():
        for _ in range(5): ce(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ce(x)
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        shape, d = ce._last_call
        bad = ops.ce_range_check(shape, "topk", k, ce._ws, d)
        prof.select_stage(-1); ce.profile = prof
        for _ in range(10): ce(x)
        torch.cuda.synchronize(); ce.profile = None
    st = np.asarray(prof.read()).mean(axis=0)
    print(f"{label:30s} k={k:3d} {thr:6s}: {ms:8.4f} ms flags {bad}  " + "  ".join(f"{STAGE_NAMES[i]} {st[i]*1e3:.0f}" for i in range(8)), flush=True)
clean = torch.from_numpy(imgs["img_02"].astype(np.float32) / 255.0)[None, None]
noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
tiles = torch.stack([noisy[0, :, y0:y1, x0:x1] for (y0, y1, x0, x1) in chop_leaf_boxes(256, 256)])
with torch.no_grad():
    x = net.head(tiles)
    for blk in net.body[:8]: x = blk(x)
x = x.contiguous()
xs = torch.from_numpy(make_features(100, 64, 64, 72, 72)).to(dev)
for k in (8, 16, 32, 50):
    run(x, "img_02 leaf tiles [64,64,72,72]", k)
    run(xs, "synthetic [64,64,72,72]", k)
run(torch.cat([x, x, x, x]).contiguous(), "img_02 leaf tiles x4 [256,..]", 50); run(torch.cat([x, x, x, x]).contiguous(), "img_02 leaf tiles x4 [256,..]", 8)
run(x[:1].contiguous(), "img_02 1 leaf tile", 50)
