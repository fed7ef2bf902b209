"""Run ON THE GPU BOX: one CE head, adaptive mask in the sparse regime (the benchmark's mean-degree-8 / -55 cases), ms per call.
   python tools/sparse_case.py <gain: 1.95 | 1.8> [calls] [always | auto]  (adaptive_sync; default auto)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features

gain = float(sys.argv[1])
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
prm = {n: torch.from_numpy(a) for n, a in make_ce_params(41, variant="sparse", sparse_gain=gain).items()}
ce = CE(in_channels=64)
ce.load_state_dict(prm, strict=True)
ce.select_mode = "adaptive"
ce.adaptive_sync = sys.argv[3] if len(sys.argv) > 3 else "auto"
ce = ce.to(dev).eval()
x = torch.from_numpy(make_features(41, 1, 64, 256, 256)).to(dev)
with torch.no_grad():
    for _ in range(400):
        ce(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        ce(x)
    e1.record()
    torch.cuda.synchronize()
print(f"sparse_case gain {gain} sync={ce.adaptive_sync}: {e0.elapsed_time(e1) / calls:.4f} ms per call (path {(ce.last_info or {}).get('path')})")
