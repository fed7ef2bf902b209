#!/usr/bin/env python3
"""Run ON THE GPU BOX: the two gradient products of the fc2 projection at BASELINE config 5's size (B = 8, 128 x 128: 131 072 key
patches) -- split-fp16 GEMM (dagl_fc_grad16) against the fp32 matrix-core path (unfold + two dagl_gemm_f32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagl_amd import _lib, ops  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    B, H, W = 8, 128, 128
    n = B * H * W
    g = torch.Generator(device=dev).manual_seed(0)
    pmap = torch.zeros(B, H + 6, W + 6, 16, device=dev)
    pmap[:, 3:3 + H, 3:3 + W] = torch.randn(B, H, W, 16, device=dev, generator=g)
    w = (torch.rand(196, 784, device=dev, generator=g) - 0.5) * 0.07
    dz = torch.randn(n, 196, device=dev, generator=g) * 1e-4
    need = lib.dagl_fc_grad16_scratch_bytes(B, H, W)
    scratch = torch.empty(need + 256, device=dev, dtype=torch.uint8)
    base = (scratch.data_ptr() + 255) // 256 * 256
    d_w = torch.empty(196, 784, device=dev); d_rows = torch.empty(n, 784, device=dev)

    def fast(both=True, dw=True):
        _lib.check(lib.dagl_fc_grad16(ops._stream(), B, H + 6, W + 6, 1, 0, 0, H, W, pmap.data_ptr(), w.data_ptr(), None, dz.data_ptr(),
                                      d_w.data_ptr() if (both or dw) else None, None, d_rows.data_ptr() if (both or not dw) else None,
                                      base, need), "dagl_fc_grad16")

    rows = torch.empty(n, 784, device=dev)

    def slow():
        _lib.check(lib.dagl_unfold_patches(ops._stream(), B, H + 6, W + 6, 16, 7, 1, 0, 0, H, W, pmap.data_ptr(), rows.data_ptr()), "unfold")
        ops.gemm_f32(dz, rows, a_k_contiguous=False, b_k_contiguous=False)
        ops.gemm_f32(dz, w, a_k_contiguous=True, b_k_contiguous=False, out=rows)

    flop = 2.0 * 196 * 784 * n
    t_all, t_w, t_r, t_old = timed(fast), timed(lambda: fast(False, True)), timed(lambda: fast(False, False)), timed(slow)
    print(f"fc2 gradients at n = {n}: split-fp16 {t_all:.3f} ms (d W alone {t_w:.3f} = {flop / t_w / 1e9:.0f} TF-equivalent, "
          f"d rows alone {t_r:.3f} = {flop / t_r / 1e9:.0f}); fp32 matrix cores {t_old:.3f} ms")


if __name__ == "__main__":
    main()
