"""Run ON THE GPU BOX: ms per call of one CE head in a given regime.  python tools/time_case.py <mode> <variant> <gain> <wseed> <fseed> <size> <k> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
mode, variant, gain, ws, fs, size, k = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
steps = int(sys.argv[8]) if len(sys.argv) > 8 else 100
dev = torch.device("cuda:0")
prm = {n: torch.from_numpy(a) for n, a in make_ce_params(ws, variant=variant, sparse_gain=gain).items()}
m = CE(in_channels=64); m.load_state_dict(prm, strict=True); m.select_mode = mode
if k: m.select_k = k
m = m.to(dev).eval()
x = torch.from_numpy(make_features(fs, 1, 64, size, size)).to(dev)
with torch.no_grad():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): m(x)
    torch.cuda.synchronize()
print(f"{mode} k={k} {size}^2: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per call (path {m.last_info and m.last_info.get('path')})")
