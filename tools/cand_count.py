"""Run ON THE GPU BOX: how many keys lie within the screen band of the TRUE k-th best score (real vs synthetic features)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from dagl_amd.net import RR, set12_protocol_noise
from dagl_amd.synth import make_features, make_ce_params
from dagl_amd.ce import CE
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
def rows(ce, x):
    with torch.no_grad():
        b1 = ce.g(x)
        kp = F.unfold(F.pad(b1, (3, 3, 3, 3)), 7).transpose(1, 2)[0]               # [N,784] (c,kh,kw)
        X = F.relu(ce.fc2(kp))
        H = x.shape[-1]
        qp = F.unfold(F.pad(b1, (1, 2, 1, 2)), 7, stride=4).transpose(1, 2)[0]
        Q = F.relu(ce.fc1(qp))
    return Q, X
def report(label, Q, X):
    idx = torch.linspace(0, Q.shape[0] - 1, 256).long().to(dev)
    S = Q[idx] @ X.t()
    srt = S.sort(dim=1, descending=True).values
    for k in (8, 50):
        kth = srt[:, k - 1:k]
        out = []
        for band in (0.0, 0.008, 0.016, 0.032):
            c = (S >= kth * (1 - band)).sum(dim=1).float()
            out.append(f"band {band*100:.1f}%: median {int(c.median())} p90 {int(c.quantile(0.9))} max {int(c.max())}")
        print(f"{label} k={k}: keys with S >= kth*(1-band): " + " | ".join(out), flush=True)
for name in ("img_02", "img_05"):
    clean = torch.from_numpy(imgs[name].astype(np.float32) / 255.0)[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
    with torch.no_grad():
        x = net.head(noisy)
        for blk in net.body[:8]: x = blk(x)
    Q, X = rows(net.body[8].c1_1, x)
    report(f"Set12 {name} trained", Q, X)
prm = {n: torch.from_numpy(a) for n, a in make_ce_params(2024, variant="default").items()}
ce = CE(in_channels=64); ce.load_state_dict(prm, strict=True); ce = ce.to(dev)
Q, X = rows(ce, torch.from_numpy(make_features(100, 1, 64, 256, 256)).to(dev))
report("synthetic N(0,1)", Q, X)
