#!/usr/bin/env python3
"""Golden values of the reference's metric functions (build container only):
Demosaic/pytorch_ssim ``ssim`` and DN_Gray/utility ``calc_psnr``/``quantize`` on seeded inputs -> tests/golden/metrics.json"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def inputs():
    rng = np.random.default_rng(123)
    a = torch.from_numpy(rng.random((2, 3, 40, 36)).astype(np.float32))
    b = (a + torch.from_numpy(rng.normal(0, 0.05, (2, 3, 40, 36)).astype(np.float32))).clamp(0, 1)
    return a, b


if __name__ == "__main__":
    ps = load("/root/reference/Demosaic/pytorch_ssim/__init__.py", "ref_ssim")
    a, b = inputs()
    out = {"ssim_mean": float(ps.ssim(a, b)), "ssim_per_image": [float(v) for v in ps.ssim(a, b, size_average=False)]}
    # utility.py imports matplotlib/imageio at module level: restate nothing, just exec the two pure functions
    src = open("/root/reference/DN_Gray/utility.py").read()
    ns = {"math": __import__("math"), "torch": torch}
    for fn in ("quantize", "calc_psnr"):
        start = src.index(f"def {fn}(")
        end = src.index("\ndef ", start + 1)
        exec(src[start:end], ns)                       # runs the reference's own code, in memory only
    out["calc_psnr_train"] = ns["calc_psnr"](b.clone(), a.clone(), 1, 1.0, benchmark=False)
    out["calc_psnr_bench_rgb"] = ns["calc_psnr"](b.clone(), a.clone(), 2, 1.0, benchmark=True)
    out["quantize_sum"] = float(ns["quantize"](b * 0.7, 1.0).double().sum())
    json.dump(out, open(os.path.join(HERE, "metrics.json"), "w"), indent=1)
    print(out)
