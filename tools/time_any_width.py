"""Run ON THE GPU BOX: CE(in_channels = n) for n != 64 (the unfused any-width prologue: unfold + fp32 matrix-core GEMM, then
dagl_ce_forward) next to the fused 64-channel head, 256 x 256, top-k 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
dev = torch.device("cuda:0")
for C in (64, 32, 96, 128, 48):
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(2024, in_channels=C, variant="default").items()}
    ce = CE(in_channels=C); ce.load_state_dict(prm, strict=True); ce.select_mode, ce.select_k = "topk", 8
    ce = ce.to(dev).eval()
    x = torch.from_numpy(make_features(100, 1, C, 256, 256)).to(dev)
    with torch.no_grad():
        for _ in range(20): ce(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ce(x)
        e1.record(); e1.synchronize()
    print(f"CE(in_channels={C:3d}) [1,{C},256,256] top-k 8: {e0.elapsed_time(e1)/50:.4f} ms", flush=True)
