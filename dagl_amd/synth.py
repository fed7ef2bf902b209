"""Seeded synthetic parameters / inputs for the patch-graph attention block.

Everything is drawn from numpy's PCG64 (``np.random.default_rng``) in a fixed,
documented order so that the GPU box, this container and the golden-vector
script regenerate bit-identical tensors without shipping them.

Parameter names and shapes follow the reference block
(``DN_Gray/model/dagl.py:175-205``): ``g`` (3x3 conv C->c), ``W`` (1x1 conv
c->C, registered but never applied, dagl.py:192), ``theta`` (1x1 conv C->c),
``fc1.0`` / ``fc2.0`` (Linear P->P/4), ``thr_conv`` / ``bias_conv`` (7x7
stride-4 conv C->1).

Draw order (one rng, seed given by the caller):
    g.weight, g.bias, W.weight, W.bias, theta.weight, theta.bias,
    fc1.0.weight, fc1.0.bias, fc2.0.weight, fc2.0.bias,
    thr_conv.weight, thr_conv.bias, bias_conv.weight, bias_conv.bias
each ``uniform(-1/sqrt(fan_in), +1/sqrt(fan_in))`` in float64 then cast to
float32 (the bound torch's default Conv2d/Linear init uses).  Variants then
overwrite the thr/bias heads; they never consume extra draws, so all variants
of one seed share g/theta/fc weights.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

KSIZE = 7
STRIDE_Q = 4

VARIANTS = ("default", "sparse", "allpass", "nonepass")


def _uniform(rng, shape, fan_in):
    bound = 1.0 / math.sqrt(fan_in)
    return rng.uniform(-bound, bound, size=shape).astype(np.float32)


def make_ce_params(seed: int, in_channels: int = 64, inter_channels: int = 16,
                   ksize: int = KSIZE, variant: str = "default",
                   sparse_gain: float = 2.0) -> "OrderedDict[str, np.ndarray]":
    """Return the block's state_dict as numpy float32 arrays.

    variant:
      default   torch-default-like init everywhere (dense regime: ~95 % of the
                keys pass the adaptive mask)
      sparse    thr head = small random weights + bias ``sparse_gain`` and a
                zero bias head, so a query keeps only keys whose score exceeds
                ~``sparse_gain`` x its row mean (a handful of neighbours)
      allpass   thr head outputs -1 (threshold below every score): every key
                is a neighbour of every query
      nonepass  bias head outputs -1e4: no key passes, output is exactly zero
    """
    if variant not in VARIANTS:
        raise ValueError(f"unknown variant {variant!r}")
    C, c, k = in_channels, inter_channels, ksize
    P = c * k * k
    D = P // 4
    rng = np.random.default_rng(seed)
    p = OrderedDict()
    p["g.weight"] = _uniform(rng, (c, C, 3, 3), C * 9)
    p["g.bias"] = _uniform(rng, (c,), C * 9)
    p["W.weight"] = _uniform(rng, (C, c, 1, 1), c)
    p["W.bias"] = _uniform(rng, (C,), c)
    p["theta.weight"] = _uniform(rng, (c, C, 1, 1), C)
    p["theta.bias"] = _uniform(rng, (c,), C)
    p["fc1.0.weight"] = _uniform(rng, (D, P), P)
    p["fc1.0.bias"] = _uniform(rng, (D,), P)
    p["fc2.0.weight"] = _uniform(rng, (D, P), P)
    p["fc2.0.bias"] = _uniform(rng, (D,), P)
    p["thr_conv.weight"] = _uniform(rng, (1, C, k, k), C * k * k)
    p["thr_conv.bias"] = _uniform(rng, (1,), C * k * k)
    p["bias_conv.weight"] = _uniform(rng, (1, C, k, k), C * k * k)
    p["bias_conv.bias"] = _uniform(rng, (1,), C * k * k)
    if variant == "sparse":
        p["thr_conv.weight"] = (p["thr_conv.weight"] * np.float32(0.25)).astype(np.float32)
        p["thr_conv.bias"] = np.full((1,), sparse_gain, np.float32)
        p["bias_conv.weight"] = np.zeros_like(p["bias_conv.weight"])
        p["bias_conv.bias"] = np.zeros((1,), np.float32)
    elif variant == "allpass":
        p["thr_conv.weight"] = np.zeros_like(p["thr_conv.weight"])
        p["thr_conv.bias"] = np.full((1,), -1.0, np.float32)
        p["bias_conv.weight"] = np.zeros_like(p["bias_conv.weight"])
        p["bias_conv.bias"] = np.zeros((1,), np.float32)
    elif variant == "nonepass":
        p["bias_conv.weight"] = np.zeros_like(p["bias_conv.weight"])
        p["bias_conv.bias"] = np.full((1,), -1.0e4, np.float32)
    return p


def make_features(seed: int, B: int, C: int, H: int, W: int) -> np.ndarray:
    """Synthetic feature map ``[B,C,H,W]`` fp32, N(0,1), own rng stream."""
    rng = np.random.default_rng([seed, 0xFEA7])
    return rng.standard_normal((B, C, H, W)).astype(np.float32)


def same_pad_amounts(size: int, ksize: int, stride: int):
    """TF-"SAME" zero padding (lo, hi) along one axis.

    Mirrors the arithmetic of ``same_padding`` (DN_Gray/model/dagl.py:123-139):
    total = max(0, (ceil(size/stride)-1)*stride + ksize - size), lo = total//2
    (``int(total/2.)`` there), hi = the rest -- asymmetric when total is odd.
    """
    out = (size + stride - 1) // stride
    total = max(0, (out - 1) * stride + ksize - size)
    lo = total // 2
    return lo, total - lo


def query_grid(H: int, W: int, ksize: int = KSIZE, stride: int = STRIDE_Q):
    """(Lh, Lw, pad_top, pad_left) of the stride-4 query patch grid."""
    pt, _ = same_pad_amounts(H, ksize, stride)
    pl, _ = same_pad_amounts(W, ksize, stride)
    return (H + stride - 1) // stride, (W + stride - 1) // stride, pt, pl
