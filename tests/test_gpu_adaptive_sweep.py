"""Adaptive mask over its whole range of densities at small sizes, block against the fp64 oracle: lists only, lists + heavy
queries (64+ candidates: refine_heavy_kernel), lists + per-query redo (overflow.hip: while the flagged rows hold less than
1/96 of all pairs; the CSR redo beyond its capacity is tests/test_gpu_configs.py's 512^2 case) and the dense formulation -- the selection path follows the mask, the result must not.
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
    (1, 128, 128, 1.7, 3, 3),      # mean degree 52: 140 heavy queries, 42 redone
    (1, 96, 96, 1.5, 17, 4),       # 140 flagged queries holding 2.6 % of all pairs: too heavy to redo one by one
    (1, 128, 128, 1.5, 3, 4),      # mean degree 384, a third of the queries flagged: likewise
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


def test_a_few_dense_rows_among_sparse_ones_go_to_the_dense_formulation():
    """Every fifth query keeps ALL keys, the others ~1.5 of 16 384: 205 flagged queries are fewer than half (the lists would
    serve the call) but their rows hold 20 % of all pairs -- redone one by one that is a value patch per edge (~1 ns each),
    so the call must end in the dense formulation (path 4), whose cost does not depend on the mask.  thr / bias go in
    through the C ABI's own arguments (dagl_ce_forward takes them per query); oracle: the same rows, fp64."""
    from dagl_amd import ops
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_core_oracle, ce_forward_oracle
    dev = torch.device("cuda:0")
    H = W = 128
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(17, variant="sparse", sparse_gain=1.9).items()}
    x = torch.from_numpy(make_features(18, 1, 64, H, W))
    with torch.no_grad():
        _, st = ce_forward_oracle(x, prm, mode="adaptive", k=None, stages=True, dtype=torch.float64)
    thr, bias = st["thr"].clone(), st["bias"].clone()                  # [1, L]
    thr[:, ::5] = 0.0
    bias[:, ::5] = 0.01                                                # m = S + 0.01 > 0 for every key (S >= 0)
    with torch.no_grad():
        want, ref = ce_core_oracle(st["Wq"], st["X"], st["b2"], thr, bias, mode="adaptive", stages=True)
    deg = ref["deg"][0].numpy()
    assert int((deg == H * W).sum()) == 205 and np.median(deg) <= 4

    m = CE(in_channels=64)
    m.load_state_dict(prm, strict=True)
    m = m.to(dev).eval()
    with torch.no_grad():
        b1, b2, _, _ = m._prologue(x.to(dev))
        out, info = ops.ce_forward(b1.contiguous(), b2.contiguous(), thr.float().to(dev).contiguous(),
                                   bias.float().to(dev).contiguous(), m.fc1[0].weight, m.fc1[0].bias, m.fc2[0].weight,
                                   m.fc2[0].bias, mode="adaptive", k=0, debug=True)
    assert info["path"] == 4, {k: v for k, v in info.items() if not torch.is_tensor(v)}
    assert normwise(out.cpu().numpy(), want.float().numpy()) <= 1e-4
    got = info["deg"][0].cpu().numpy()
    assert int((got != deg).sum()) <= max(1, int(1e-4 * deg.size))
    assert normwise(info["rowsum"][0].cpu().numpy(), ref["rowsum"][0].float().numpy()) <= 1e-4
