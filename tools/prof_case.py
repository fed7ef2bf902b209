"""Run ON THE GPU BOX under rocprofv3: N forwards of one CE head in a given regime (for per-kernel averages).
   python tools/prof_case.py <mode> <variant> <gain> <wseed> <fseed> [size] [k] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features

mode, variant, gain, ws, fs = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
size = int(sys.argv[6]) if len(sys.argv) > 6 else 256
k = int(sys.argv[7]) if len(sys.argv) > 7 else 0
steps = int(sys.argv[8]) if len(sys.argv) > 8 else 200
dev = torch.device("cuda:0")
prm = {n: torch.from_numpy(a) for n, a in make_ce_params(ws, variant=variant, sparse_gain=gain).items()}
m = CE(in_channels=64)
m.load_state_dict(prm, strict=True)
m.select_mode = mode
if k:
    m.select_k = k
m = m.to(dev).eval()
x = torch.from_numpy(make_features(fs, 1, 64, size, size)).to(dev)
with torch.no_grad():
    for _ in range(steps):
        m(x)
torch.cuda.synchronize()
print(m.last_info)
