"""Run ON THE GPU BOX: the same call repeated -- outputs must be bit-identical from run to run (fixed summation orders, integer-only
atomics where order must not matter) in every regime.
   python tools/check_determinism.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features

dev = torch.device("cuda:0")
bad = 0
for name, variant, gain, mode, k, sync in (("adaptive sparse (mean degree 8)", "sparse", 1.95, "adaptive", 0, "always"),
                                           ("adaptive sparse (mean degree 8), no wait", "sparse", 1.95, "adaptive", 0, "auto"),
                                           ("adaptive, mean degree 55", "sparse", 1.8, "adaptive", 0, "always"),
                                           ("adaptive dense", "default", 2.0, "adaptive", 0, "always"),
                                           ("top-k 8", "default", 2.0, "topk", 8, "always"),
                                           ("adaptive AND top-16", "sparse", 1.7, "adaptive_topk", 16, "always"),
                                           ("top-k 300", "default", 2.0, "topk", 300, "always")):
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(41, variant=variant, sparse_gain=gain).items()}
    ce = CE(in_channels=64); ce.load_state_dict(prm, strict=True); ce.select_mode = mode; ce.adaptive_sync = sync
    if k: ce.select_k = k
    ce = ce.to(dev).eval()
    x = torch.from_numpy(make_features(41, 1, 64, 256, 256)).to(dev)
    with torch.no_grad():
        ref = ce(x).clone()
        same = all(torch.equal(ce(x), ref) for _ in range(12))
        fresh = CE(in_channels=64); fresh.load_state_dict(prm, strict=True); fresh.select_mode = mode
        if k: fresh.select_k = k
        same2 = torch.equal(fresh.to(dev).eval()(x), ref)
    print(f"{name:44s} 12 repeats identical: {same}; a fresh module identical: {same2}; path {(ce.last_info or {}).get('path')}")
    bad += (not same) + (not same2)
print("bad", bad)
