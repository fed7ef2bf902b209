"""Adaptive mask over its whole range of densities at small sizes, block against the fp64 oracle: lists only, lists + heavy
queries (64+ candidates: refine_heavy_kernel), lists + per-query redo (overflow.hip: up to half the queries at these sizes;
the CSR redo beyond its capacity is tests/test_gpu_configs.py's 512^2 case) and the dense formulation -- the selection path follows the mask, the result must not.
(tools/sweep_adaptive.py is the same check over 40 cases.)  Reference: dagl.py:256-265."""
import numpy as np
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,gain,seed,path", [
    (1, 64, 64, 1.9, 17, 3),       # sparse: nothing beyond the lists
    (2, 72, 56, 1.7, 3, 3),        # a few heavy queries, 3 redone one by one
    (1, 64, 64, 1.5, 3, 3),        # mean degree 90: 74 heavy queries, 20 redone
    (1, 96, 96, 1.5, 17, 3),       # 187 heavy, 140 redone
    (1, 128, 128, 1.5, 3, 3),      # mean degree 384: a third of the queries redone one by one
    (1, 64, 64, 1.2, 3, 4),        # most queries overflow: dense formulation
])
def test_adaptive_mask_density_sweep(B, H, W, gain, seed, path):
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle
    dev = torch.device("cuda:0")
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant="sparse", sparse_gain=gain).items()}
    x = torch.from_numpy(make_features(seed + 1, B, 64, H, W))
    with torch.no_grad():
        want, st = ce_forward_oracle(x, prm, mode="adaptive", k=None, stages=True, dtype=torch.float64)
    deg = st["deg"].numpy().reshape(-1)
    m = CE(in_channels=64)
    m.load_state_dict(prm, strict=True)
    m.select_mode = "adaptive"
    m = m.to(dev).eval()
    with torch.no_grad():
        out = m(x.to(dev)).cpu()
        again = m(x.to(dev)).cpu()
    info = m.last_info
    assert normwise(out.numpy(), want.float().numpy()) <= 1e-4
    assert torch.equal(out, again)                                   # same call, same bits (list order is not)
    assert info["path"] == path, info
    # (a key within rounding of its threshold may flip between the fp32 block and the fp64 oracle)
    assert abs(info["total_edges"] - int(deg.sum())) <= max(2, int(1e-4 * deg.size)), (info, int(deg.sum()))
    assert abs(info["max_degree"] - int(deg.max())) <= 1, (info, int(deg.max()))
