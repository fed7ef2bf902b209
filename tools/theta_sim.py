"""Run ON THE GPU BOX: simulate the top-k threshold estimate (group maxima per (chunk, half, slot), k-th largest) on real scores: candidates per query and per segment under several grouping schemes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from dagl_amd.net import RR, set12_protocol_noise
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"); dev = torch.device("cuda:0")
z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
net = RR().eval(); net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True); net = net.to(dev)
imgs = np.load(os.path.join(G, "set12.npz"))
DELTA = 0.0079
def rows(ce, x):
    with torch.no_grad():
        b1 = ce.g(x)
        X = F.relu(ce.fc2(F.unfold(F.pad(b1, (3, 3, 3, 3)), 7).transpose(1, 2)[0]))
        Q = F.relu(ce.fc1(F.unfold(F.pad(b1, (1, 2, 1, 2)), 7, stride=4).transpose(1, 2)[0]))
    return Q, X
def sim(S, k, stride, hashed, gkeep, steps_per_split=32):
    nq, N = S.shape
    n = torch.arange(N, device=dev)
    step = n // 64; p = n % 64; row = p % 32
    c = (step * 13 + (step >> 2) * 7) & 31 if hashed else torch.zeros_like(step)
    rowp = row ^ c
    h = (rowp >> 2) & 1; r = (rowp & 3) + 4 * (rowp >> 3)
    split = step // steps_per_split
    sampled = ((step % steps_per_split) % stride) == 0
    nsplit = int(split.max()) + 1
    gid = (split * 2 + h) * 16 + r                                  # group id
    Sb = S.to(torch.bfloat16).float()
    gm = torch.full((nq, nsplit * 2 * 16), -1.0, device=dev)
    gm.scatter_reduce_(1, gid[sampled].expand(nq, -1), Sb[:, sampled], reduce="amax")
    kept = gm.view(nq, nsplit * 2, 16).topk(gkeep, dim=2).values.reshape(nq, -1)
    kth = kept.topk(min(k, kept.shape[1]), dim=1).values[:, -1:]
    theta = kth * ((1 - DELTA) / (1 + DELTA))
    cand = Sb >= theta
    tot = cand.sum(1).float()
    seg = torch.zeros(nq, nsplit * 2, device=dev)
    seg.scatter_add_(1, ((split * 2 + ((row >> 2) & 1))).expand(nq, -1), cand.float())
    return tot, seg.max(1).values
for name in ("img_02", "img_05", "img_07"):
    clean = torch.from_numpy(imgs[name].astype(np.float32) / 255.0)[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
    with torch.no_grad():
        x = net.head(noisy)
        for blk in net.body[:8]: x = blk(x)
    Q, X = rows(net.body[8].c1_1, x)
    idx = torch.linspace(0, Q.shape[0] - 1, 512).long().to(dev)
    S = Q[idx] @ X.t()
    for k in (8, 32, 50):
        for label, stride, hashed, gk in (("now(tight)", 2 if k <= 32 else 1, False, 4), ("s1 alias g16", 1, False, 16), ("s1 hash g4", 1, True, 4), ("s1 hash g16", 1, True, 16), ("s2 hash g16", 2, True, 16)):
            tot, segmax = sim(S, k, stride, hashed, gk)
            print(f"{name} k={k:2d} {label:13s}: candidates median {int(tot.median()):6d} p90 {int(tot.quantile(0.9)):6d} max {int(tot.max()):6d} | worst segment median {int(segmax.median()):5d} max {int(segmax.max()):5d} | queries with a segment > 127: {int((segmax > 127).sum())}/512, with total > 1024: {int((tot > 1024).sum())}", flush=True)
