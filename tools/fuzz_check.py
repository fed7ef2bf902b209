"""Run ON THE GPU BOX: the mismatching cases of tools/fuzz_shapes.py against the fp64 oracle, with the smallest relative gap between the k-th and
the (k+1)-th score of any query (near-ties below fp32 resolution: either key is a legitimate k-th neighbour)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagl_amd import ops
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
from oracle.ce_oracle import ce_forward_oracle
from tests.helpers import normwise
dev = torch.device("cuda:0")
torch.set_num_threads(64)
for (B, H, W, k, mode) in [(1, 83, 125, 16, "topk"), (7, 83, 126, 33, "topk"), (1, 105, 126, 33, "adaptive_topk"), (2, 145, 129, 50, "adaptive_topk")]:
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(90 + H, variant="default" if mode == "topk" else "allpass").items()}
    x = torch.from_numpy(make_features(91 + W, B, 64, H, W))
    with torch.no_grad():
        want, st = ce_forward_oracle(x, prm, mode=mode, k=k, dtype=torch.float64, stages=True)
    outs = {}
    for scan in ("screened", "exact"):
        ce = CE(in_channels=64); ce.load_state_dict(prm, strict=True); ce.select_mode, ce.select_k, ce.scan = mode, k, scan
        ce = ce.to(dev).eval()
        with torch.no_grad():
            outs[scan] = ce(x.to(dev)).cpu()
    es, ee = normwise(outs["screened"].numpy(), want.float().numpy()), normwise(outs["exact"].numpy(), want.float().numpy())
    # near-ties at the k-th place in the oracle's fp64 scores
    gaps = []
    for s in (st if isinstance(st, list) else [st]):
        S = s["S"]; kk = min(k, S.shape[1] - 1)
        if mode != "topk":
            S = torch.where(s["mask_b"] > 0, S, S) 
        top = S.topk(kk + 1, dim=1).values
        gaps.append(((top[:, kk - 1] - top[:, kk]) / top[:, kk - 1]).min().item())
    print(f"[{B},64,{H},{W}] {mode} k={k}: screened vs fp64 oracle {es:.2e}, exact scan vs fp64 oracle {ee:.2e}, smallest relative gap k-th/(k+1)-th score {min(gaps):.2e}", flush=True)
