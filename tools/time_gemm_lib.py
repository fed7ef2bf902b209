"""Run ON THE GPU BOX: dagl_gemm_f32 against the library GEMM (torch.matmul -> hipBLASLt / rocBLAS, fp32) on the training
path's backward shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagl_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
for name, R, O, K in [("fc2", 131072, 196, 784), ("fc1", 8192, 196, 784)]:
    dz = torch.randn(R, O, device=dev); rows = torch.randn(R, K, device=dev); w = torch.randn(O, K, device=dev)
    fl = 2.0 * R * O * K
    a = t(lambda: ops.gemm_f32(dz, rows, a_k_contiguous=False, b_k_contiguous=False, chunk_tiles=8)); b = t(lambda: torch.matmul(dz.t(), rows))
    print(f"{name} dW     [{O}x{R}]x[{R}x{K}]: dagl {a:7.1f} us {fl / a / 1e6:6.1f} TF | library {b:7.1f} us {fl / b / 1e6:6.1f} TF")
    a = t(lambda: ops.gemm_f32(dz, w, a_k_contiguous=True, b_k_contiguous=False)); b = t(lambda: torch.matmul(dz, w))
    print(f"{name} d rows [{R}x{O}]x[{O}x{K}]: dagl {a:7.1f} us {fl / a / 1e6:6.1f} TF | library {b:7.1f} us {fl / b / 1e6:6.1f} TF")
    a = t(lambda: ops.gemm_f32(rows, w, a_k_contiguous=True, b_k_contiguous=True, chunk_tiles=7)); b = t(lambda: torch.matmul(rows, w.t()))
    print(f"{name} fwd    [{R}x{K}]x[{K}x{O}]: dagl {a:7.1f} us {fl / a / 1e6:6.1f} TF | library {b:7.1f} us {fl / b / 1e6:6.1f} TF")
