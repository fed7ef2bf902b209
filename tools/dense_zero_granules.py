"""CPU measurement (runs anywhere; python tools/dense_zero_granules.py [all|synth|ckpt|big]): fraction of (32-query x 16-key) A.V granules of the dense formulation whose weights are all exactly zero
after the 2^14 fp16 split (l - M' < -27.1), for default-init synthetic features and for the trained checkpoint's features."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ce_oracle import ce_forward_oracle, params_to_torch
from dagl_amd.synth import make_ce_params, make_features
torch.set_num_threads(8)

def analyse(S, T_thr, H, W, label, slack=0.0):
    # S [L,N]; m = relu(S - mean*thr + bias) -> here T = mean*thr - bias
    L, N = S.shape
    m = torch.relu(S - T_thr[:, None])
    l = 10.0 * S * m
    M = l.max(dim=1, keepdim=True).values * (1.0 + slack)
    d = l - M
    zero = d < -27.1                       # p * 2^14 rounds to 0 in fp16 (hi = lo = 0)
    dens = (m > 0).float().mean().item()
    # key tile = 8x4 pixel block; kblock = 2 rows x 8 px; granule = 32 consecutive queries x one kblock
    Hk, Wk = H // 2 * 2, W // 8 * 8
    z = zero.view(L, H, W)[:, :Hk, :Wk]
    z = z.reshape(L, Hk // 2, 2, Wk // 8, 8).permute(0, 1, 3, 2, 4).reshape(L, Hk // 2, Wk // 8, 16)
    zk = z.all(dim=-1)                                      # [L, Hk/2, Wk/8] all 16 keys zero for this query
    Lq = L // 32 * 32
    g32 = zk[:Lq].reshape(Lq // 32, 32, -1).all(dim=1)      # 32 queries x 16 keys
    g64 = zk[:L // 64 * 64].reshape(L // 64, 64, -1).all(dim=1)     # 64 queries x 16 keys: the granule dense.hip skips
    Lq64 = L // 64 * 64
    zt = zk[:Lq64].reshape(Lq64 // 64, 64, Hk // 4, 2, Wk // 8).all(dim=3).all(dim=1)   # 64 q x 32 keys
    g16 = zk[:Lq // 16 * 16].reshape(-1, 16, zk.shape[1] * zk.shape[2]).all(dim=1)
    print(f"{label}: mask density {dens:.3f}, max logit median {M.median().item():.1f} (min {M.min().item():.1f}, max {M.max().item():.1f}); "
          f"pairs exactly zero {zero.float().mean().item():.4f}; per-query 16-key groups zero {zk.float().mean().item():.4f}; "
          f"16q x 16k granules {g16.float().mean().item():.4f}; 32q x 16k granules {g32.float().mean().item():.4f}; 64q x 16k granules (the kernel's) {g64.float().mean().item():.4f}; 64q x 32k tiles {zt.float().mean().item():.4f}", flush=True)

def run_synth(H, W, seed_p=1, seed_x=100):
    p = params_to_torch(make_ce_params(seed_p, variant="default"))
    x = torch.from_numpy(make_features(seed_x, 1, 64, H, W))
    with torch.no_grad():
        out, st = ce_forward_oracle(x, p, stages=True)
    analyse(st["S"][0], st["T"][0], H, W, f"synthetic default init {H}x{W}")
    analyse(st["S"][0], st["T"][0], H, W, f"  ... with M' 3% high", slack=0.03)

def run_ckpt(size):
    from dagl_amd.net import RR, set12_protocol_noise
    G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    z = np.load(os.path.join(G, "quality_ckpt_fp16.npz"))
    net = RR().eval()
    net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True)
    imgs = np.load(os.path.join(G, "set12.npz"))
    name = sorted(imgs.files)[0]
    clean = torch.from_numpy(imgs[name].astype(np.float32) / 255.0)
    if clean.ndim == 2: clean = clean[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0)[..., :size, :size]
    with torch.no_grad():
        x = net.head(noisy)
        for blk in net.body[:8]:
            x = blk(x)
    ces = net.body[8]
    for hn in ("c1_1", "c1_3"):
        ce = getattr(ces, hn)
        p = {k: v.detach().float() for k, v in ce.state_dict().items()}
        with torch.no_grad():
            out, st = ce_forward_oracle(x, p, stages=True)
        analyse(st["S"][0], st["T"][0], size, size, f"trained ckpt head {hn} on {name} {size}x{size}")

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    t = time.time()
    if which in ("all", "synth"):
        run_synth(64, 64); run_synth(128, 128)
    if which in ("all", "ckpt"):
        run_ckpt(72); run_ckpt(128)
    if which == "big":
        run_synth(256, 256); run_ckpt(256)
    print(f"{time.time() - t:.1f} s")
