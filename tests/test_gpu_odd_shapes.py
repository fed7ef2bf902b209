"""Rows that are no multiple of 32 (the projections' work items wrap around row ends), of 4 (the thr / bias heads' scalar
staging), of anything (129 x 65), batches: the block, screened and exact scan, against the fp64 oracle.
(tools/sweep_topk.py is the same check over 28 cases.)  Reference: dagl.py:207-275, GReccR2b_3mh_1-checkpoint.py:242-250."""
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,mode,k", [(2, 72, 72, "topk", 8), (1, 100, 88, "topk", 16), (1, 129, 65, "adaptive_topk", 8),
                                          (3, 64, 96, "topk", 4), (1, 100, 88, "adaptive", 0)])
def test_odd_shapes_match_the_fp64_oracle(B, H, W, mode, k):
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle
    dev = torch.device("cuda:0")
    seed = 7 + H + k
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant="default" if mode == "topk" else "sparse",
                                                              sparse_gain=1.7).items()}
    x = torch.from_numpy(make_features(seed + 1, B, 64, H, W))
    with torch.no_grad():
        want = ce_forward_oracle(x, prm, mode=mode, k=k or None, dtype=torch.float64).float()
    for scan in ("screened", "exact"):
        m = CE(in_channels=64)
        m.load_state_dict(prm, strict=True)
        m.select_mode, m.scan = mode, scan
        if k:
            m.select_k = k
        m = m.to(dev).eval()
        with torch.no_grad():
            out = m(x.to(dev)).cpu()
        assert normwise(out.numpy(), want.numpy()) <= 1e-4, (scan, m.last_info)
