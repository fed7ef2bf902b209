#!/usr/bin/env python3
"""Golden vectors for NON-DEFAULT patch geometries, minted by running the REFERENCE block itself (CPU, build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_geometry.py

``ksize``, ``stride_1``, ``stride_2`` and ``inter_channels`` are constructor arguments of the reference's ``CE``
(/root/reference/DN_Gray/model/dagl.py:175-176; the fixed-k variant DN_Gray/model/.ipynb_checkpoints/
GReccR2b_3mh_1-checkpoint.py:153-155) that its own builders never pass (dagl.py:94-109).  Each case builds the reference module
with those arguments, loads weights from ``dagl_amd.synth.make_ce_params`` (numpy PCG64, regenerable anywhere), runs ``forward``
under ``torch.no_grad`` and stores OUTPUT data only: ``tests/golden/geom_<case>.npz`` with ``out`` [B,c,H,W], ``deg`` [B,L]
(re-derived with the module's own layers) and ``meta`` (json).  One more entry records a geometry on which the reference itself
raises (F.fold's block grid does not hold the query patches).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _load_module, make_ce_params, make_features  # noqa: E402

# (name, task, seed, variant, sparse_gain, mode, k, B, H, W, Cin, ksize, stride_1, stride_2, inter_channels, softmax_scale)
CASES = [
    ("k5s3_sparse_30x27",       "DN_Gray", 51, "sparse",  1.5, "adaptive", 0, 1, 30, 27, 64, 5, 3, 1, 16, 10),
    ("k3s2_c8_default_b2_21x20", "DN_Gray", 52, "default", 2.0, "adaptive", 0, 2, 21, 20, 32, 3, 2, 1, 8, 10),
    ("k7s4_kv2_sparse_32x36",   "DN_Gray", 53, "sparse",  1.3, "adaptive", 0, 1, 32, 36, 64, 7, 4, 2, 16, 10),
    ("k7s4_c32_default_24x28",  "DN_Gray", 54, "default", 2.0, "adaptive", 0, 1, 24, 28, 64, 7, 4, 1, 32, 10),
    ("k9s4_c4_sparse_28x24",    "DN_Gray", 55, "sparse",  1.4, "adaptive", 0, 1, 28, 24, 64, 9, 4, 1, 4, 10),
    ("k5s3_scale4_sparse_b2_24x30", "CAR", 56, "sparse",  1.5, "adaptive", 0, 2, 24, 30, 64, 5, 3, 1, 16, 4),
    ("k5s3_topk6_30x27",        "TOPK",    57, "default", 2.0, "topk",     6, 1, 30, 27, 64, 5, 3, 1, 16, 10),
    ("k3s2_c8_topk100_k_gt_n_9x8", "TOPK", 58, "default", 2.0, "topk",   100, 1, 9, 8, 32, 3, 2, 1, 8, 10),
    ("k7s2_c12_cin20_sparse_18x22", "DN_Gray", 59, "sparse", 1.5, "adaptive", 0, 1, 18, 22, 20, 7, 2, 1, 12, 10),
]
# the reference raises here (fold's block grid 7x7, 64 query patches): recorded, not computed
RAISES = [("k7s4_kv4_32x32", "DN_Gray", 60, 1, 32, 32, 64, 7, 4, 4, 16)]


def run_case(case):
    name, task, seed, variant, gain, mode, k, B, H, W, Cin, ks, s1, s2, c, scale = case
    mod = _load_module(task)
    np_params = make_ce_params(seed, in_channels=Cin, inter_channels=c, ksize=ks, variant=variant, sparse_gain=gain)
    x = torch.from_numpy(make_features(seed, B, Cin, H, W))
    kw = dict(ksize=ks, stride_1=s1, stride_2=s2, in_channels=Cin, inter_channels=c, softmax_scale=scale)
    if task == "TOPK":
        ce = mod.CE(num_edge=k, **kw)
        sd = {n: torch.from_numpy(a) for n, a in np_params.items() if not n.startswith(("thr_conv", "bias_conv"))}
        missing = ce.load_state_dict(sd, strict=False)
        assert set(missing.missing_keys) <= {"conv33.weight", "conv33.bias"}, missing
    else:
        ce = mod.CE(**kw)
        ce.load_state_dict({n: torch.from_numpy(a) for n, a in np_params.items()}, strict=True)
    ce.eval()
    grabbed = {}
    if task == "TOPK":
        ce.W.register_forward_pre_hook(lambda m, inp: grabbed.__setitem__("y", inp[0].detach().clone()))
    with torch.no_grad():
        y = ce(x)
    out = grabbed["y"] if task == "TOPK" else y
    degs = []
    with torch.no_grad():
        b1 = ce.g(x)
        q, _ = mod.extract_image_patches(b1, [ks, ks], [s1, s1], [1, 1], padding="same")
        kx, _ = mod.extract_image_patches(b1, [ks, ks], [s2, s2], [1, 1], padding="same")
        if task != "TOPK":
            b4, _ = mod.same_padding(x, [ks, ks], [s1, s1], [1, 1])
            thr, bia = ce.thr_conv(b4).view(B, -1), ce.bias_conv(b4).view(B, -1)
        for n in range(B):
            S = ce.fc1(q[n].t()) @ ce.fc2(kx[n].t()).t()
            if task == "TOPK":
                d = torch.full((S.shape[0],), min(k, S.shape[1]), dtype=torch.int32)
            else:
                m = torch.relu(S - S.mean(dim=1, keepdim=True) * thr[n].unsqueeze(1) + bia[n].unsqueeze(1))
                d = (m != 0).sum(1).to(torch.int32)
            degs.append(d)
    assert torch.isfinite(out).all(), name
    meta = dict(name=name, task=task, seed=seed, variant=variant, sparse_gain=gain, mode=mode, k=k, B=B, C=Cin, H=H, W=W,
                ksize=ks, stride_1=s1, stride_2=s2, inter_channels=c, softmax_scale=scale, torch=torch.__version__)
    np.savez_compressed(os.path.join(HERE, "geom_" + name + ".npz"), out=out.numpy().astype(np.float32),
                        deg=torch.stack(degs).numpy(), meta=json.dumps(meta))
    d = torch.stack(degs).float()
    print(f"{name:32s} out{tuple(out.shape)} |out|max={out.abs().max():.4f} deg mean={d.mean():.1f} min={d.min():.0f} max={d.max():.0f} "
          f"N={kx.shape[-1]}")


def run_raises(case):
    name, task, seed, B, H, W, Cin, ks, s1, s2, c = case
    mod = _load_module(task)
    ce = mod.CE(ksize=ks, stride_1=s1, stride_2=s2, in_channels=Cin, inter_channels=c).eval()
    try:
        with torch.no_grad():
            ce(torch.from_numpy(make_features(seed, B, Cin, H, W)))
    except RuntimeError as e:
        return dict(name=name, B=B, C=Cin, H=H, W=W, ksize=ks, stride_1=s1, stride_2=s2, inter_channels=c, error=type(e).__name__)
    raise AssertionError(f"{name}: the reference did not raise")


if __name__ == "__main__":
    torch.manual_seed(0)
    only = set(sys.argv[1:])
    for cs in CASES:
        if not only or cs[0] in only:
            run_case(cs)
    if not only:
        with open(os.path.join(HERE, "geom_raises.json"), "w") as f:
            json.dump([run_raises(cs) for cs in RAISES], f, indent=1)
        print("geom_raises.json written")
