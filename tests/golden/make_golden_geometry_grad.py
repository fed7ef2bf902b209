#!/usr/bin/env python3
"""Gradient goldens for NON-DEFAULT patch geometries from the REFERENCE block's own autograd (CPU, build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_geometry_grad.py

As make_golden_grad.py (loss = sum(out * G), G seeded), with the reference ``CE`` built with ``ksize / stride_1 / stride_2 /
inter_channels`` other than its defaults (DN_Gray/model/dagl.py:175-176; fixed-k variant GReccR2b_3mh_1-checkpoint.py:153-155).
Stores DATA only: ``tests/golden/geomgrad_<case>.npz`` = out, d_x, d_<param> (fc weights: every ``fc_step``-th element), meta.
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
from make_golden_grad import FC_STEP, loss_weights  # noqa: E402

# (name, task, seed, variant, sparse_gain, mode, k, B, H, W, Cin, ksize, stride_1, stride_2, inter_channels, softmax_scale)
CASES = [
    ("k5s3_sparse_b2_24x27",   "DN_Gray", 71, "sparse",  1.5, "adaptive", 0, 2, 24, 27, 64, 5, 3, 1, 16, 10),
    ("k3s2_c8_default_21x20",  "DN_Gray", 72, "default", 2.0, "adaptive", 0, 1, 21, 20, 32, 3, 2, 1, 8, 10),
    ("k7s4_kv2_sparse_32x36",  "DN_Gray", 73, "sparse",  1.3, "adaptive", 0, 1, 32, 36, 64, 7, 4, 2, 16, 10),
    ("k5s3_topk6_24x27",       "TOPK",    74, "default", 2.0, "topk",     6, 1, 24, 27, 64, 5, 3, 1, 16, 10),
    ("k5s2_scale4_sparse_20x22", "DN_Gray", 75, "sparse", 1.5, "adaptive", 0, 1, 20, 22, 32, 5, 2, 1, 8, 4),
]


def run_case(case):
    name, task, seed, variant, gain, mode, k, B, H, W, Cin, ks, s1, s2, c, scale = case
    mod = _load_module(task)
    np_params = make_ce_params(seed, in_channels=Cin, inter_channels=c, ksize=ks, variant=variant, sparse_gain=gain)
    x = torch.from_numpy(make_features(seed, B, Cin, H, W)).requires_grad_(True)
    kw = dict(ksize=ks, stride_1=s1, stride_2=s2, in_channels=Cin, inter_channels=c, softmax_scale=scale)
    if task == "TOPK":
        ce = mod.CE(num_edge=k, **kw)
        ce.load_state_dict({n: torch.from_numpy(a) for n, a in np_params.items() if not n.startswith(("thr_conv", "bias_conv"))}, strict=False)
    else:
        ce = mod.CE(**kw)
        ce.load_state_dict({n: torch.from_numpy(a) for n, a in np_params.items()}, strict=True)
    ce.train()
    grabbed = {}
    if task == "TOPK":
        ce.W.register_forward_pre_hook(lambda m, inp: grabbed.__setitem__("y", inp[0]))
    y = ce(x)
    out = grabbed["y"] if task == "TOPK" else y
    G = torch.from_numpy(loss_weights(seed, tuple(out.shape)))
    (out * G).sum().backward()
    arrays = dict(out=out.detach().numpy().astype(np.float32), d_x=x.grad.numpy().astype(np.float32))
    for n, p in ce.named_parameters():
        if n.startswith(("W.", "conv33")) or p.grad is None:
            continue
        g = p.grad.numpy().astype(np.float32)
        arrays["d_" + n] = g.reshape(-1)[::FC_STEP].copy() if n in ("fc1.0.weight", "fc2.0.weight") else g
    meta = dict(name=name, task=task, seed=seed, variant=variant, sparse_gain=gain, mode=mode, k=k, B=B, C=Cin, H=H, W=W, ksize=ks,
                stride_1=s1, stride_2=s2, inter_channels=c, softmax_scale=scale, fc_step=FC_STEP, torch=torch.__version__)
    np.savez_compressed(os.path.join(HERE, "geomgrad_" + name + ".npz"), meta=json.dumps(meta), **arrays)
    print(name, {n: (a.shape, float(np.abs(a).max())) for n, a in arrays.items()}, flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    only = set(sys.argv[1:])
    for cs in CASES:
        if not only or cs[0] in only:
            run_case(cs)
