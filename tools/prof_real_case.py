"""Run ON THE GPU BOX under rocprofv3: N forwards of one head on the TRAINED checkpoint's features of a Set12 image.
   python tools/prof_real_case.py <img> <mode> <k> [leaf] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagl_amd.net import RR, set12_protocol_noise, chop_leaf_boxes
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
img, mode, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
leaf = len(sys.argv) > 4 and sys.argv[4] == "leaf"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 100
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({n: torch.from_numpy(z[n].astype(np.float32)) for n in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
clean = torch.from_numpy(imgs[img].astype(np.float32) / 255.0)[None, None]
noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
if leaf:
    noisy = torch.stack([noisy[0, :, y0:y1, x0:x1] for (y0, y1, x0, x1) in chop_leaf_boxes(noisy.shape[-2], noisy.shape[-1])])
with torch.no_grad():
    x = net.head(noisy)
    for blk in net.body[:8]: x = blk(x)
    x = x.contiguous()
    ce = net.body[8].c1_1
    ce.select_mode = mode
    if k: ce.select_k = k
    for _ in range(steps): ce(x)
torch.cuda.synchronize()
print(ce.last_info)
