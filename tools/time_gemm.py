"""Run ON THE GPU BOX: time dagl_gemm_f32 on the shapes of the training path (weight-gradient GEMMs in particular)."""
import time, torch
from dagl_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

dev = torch.device("cuda:0")
for name, R, O, K in [("fc2 dW", 131072, 196, 784), ("fc1 dW", 8192, 196, 784), ("g dW", 131072, 16, 576), ("theta dW", 131072, 16, 64)]:
    dz = torch.randn(R, O, device=dev); rows = torch.randn(R, K, device=dev)
    for ct in (8, 0):
        us = t(lambda: ops.gemm_f32(dz, rows, a_k_contiguous=False, b_k_contiguous=False, chunk_tiles=ct))
        print(f"{name:9s} [{O}x{R}]x[{R}x{K}] chunk_tiles={ct}: {us:8.1f} us  {2.0 * R * O * K / us / 1e6:6.1f} TFLOP/s")
    w = torch.randn(O, K, device=dev)
    us = t(lambda: ops.gemm_f32(rows, w, a_k_contiguous=True, b_k_contiguous=True, chunk_tiles=7))
    print(f"{name[:-3]:9s} fwd rows x W^T: {us:8.1f} us  {2.0 * R * O * K / us / 1e6:6.1f} TFLOP/s")
    us = t(lambda: ops.gemm_f32(dz, w, a_k_contiguous=True, b_k_contiguous=False))
    print(f"{name[:-3]:9s} d rows = dZ W : {us:8.1f} us  {2.0 * R * O * K / us / 1e6:6.1f} TFLOP/s")
