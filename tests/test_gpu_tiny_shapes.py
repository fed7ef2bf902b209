"""Maps smaller than a patch, single rows / columns, odd batches: every select mode, inference and training route, k below and
beyond the lists' width, against the fp64 oracle (dagl.py:207-275: SAME padding makes any H, W >= 1 legal; the fixed-k variant takes
top_k = min(num_edge, N), GReccR2b_3mh_1-checkpoint.py:243)."""
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1, 1), (1, 1, 5), (1, 2, 3), (1, 3, 3), (1, 4, 4), (2, 5, 9), (1, 7, 7), (1, 8, 8), (3, 9, 13), (1, 13, 4), (1, 4, 31),
          (5, 6, 6), (1, 1, 64), (1, 64, 1), (1, 33, 2), (2, 8, 10), (1, 16, 5)]
CASES = [("adaptive", "default", 0), ("adaptive", "sparse", 0), ("adaptive", "allpass", 0), ("adaptive", "nonepass", 0),
         ("topk", "default", 1), ("topk", "default", 8), ("topk", "default", 50), ("topk", "default", 100),
         ("adaptive_topk", "sparse", 3), ("adaptive_topk", "default", 70)]


@pytest.mark.parametrize("B,H,W", SHAPES, ids=[f"{b}x{h}x{w}" for b, h, w in SHAPES])
def test_tiny_and_degenerate_shapes(B, H, W):
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle
    dev = torch.device("cuda:0")
    worst = 0.0
    for mode, variant, k in CASES:
        seed = 11 + H * 7 + W + k
        prm = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant=variant, sparse_gain=1.5).items()}
        x = torch.from_numpy(make_features(seed + 1, B, 64, H, W))
        with torch.no_grad():
            want = ce_forward_oracle(x, prm, mode=mode, k=k or None, dtype=torch.float64).float().numpy()
        for route in ("infer", "infer_exact", "train"):
            m = CE(in_channels=64)
            m.load_state_dict(prm, strict=True)
            m.select_mode, m.scan = mode, ("exact" if route == "infer_exact" else "screened")
            if k:
                m.select_k = k
            m = m.to(dev)
            if route == "train":
                m.train()
                xx = x.to(dev).requires_grad_(True)
                out = m(xx)
                out.sum().backward()
                assert torch.isfinite(xx.grad).all(), (mode, variant, k, route)
                out = out.detach()
            else:
                m.eval()
                with torch.no_grad():
                    out = m(x.to(dev))
            e = normwise(out.cpu().numpy(), want)
            worst = max(worst, e)
            assert e <= 1e-4, (mode, variant, k, route, e, m.last_info)
    print(f"[{B},64,{H},{W}]: worst of {3 * len(CASES)} mode / route cases {worst:.2e} from the fp64 oracle")
