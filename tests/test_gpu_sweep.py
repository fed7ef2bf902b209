"""Seeded random sweep over shapes, batch sizes, selection modes, k, weight regimes and both scans against the fp64
oracle: ragged widths (the dense path's 32-key row tiles, the projection's 32-patch items), maps smaller than the 7x7
window, every mask regime, the hinted second call of a dense module."""
import numpy as np
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu

_rng = np.random.default_rng(20260928)
CASES = []
for _ in range(36):
    mode = ["adaptive", "topk", "adaptive_topk"][int(_rng.integers(0, 3))]
    CASES.append((int(_rng.integers(1, 4)), int(_rng.integers(5, 90)), int(_rng.integers(5, 90)), mode,
                  int(_rng.integers(1, 33)),
                  ["default", "sparse", "allpass", "nonepass"][int(_rng.integers(0, 4))] if mode != "topk" else "default",
                  float(_rng.uniform(1.5, 3.0)), ["screened", "exact"][int(_rng.integers(0, 2))]))
CASES += [(1, 7, 7, "adaptive", 1, "default", 2.0, "screened"), (2, 4, 4, "topk", 3, "default", 2.0, "screened"),
          (1, 1, 1, "adaptive", 1, "default", 2.0, "screened"), (1, 46, 47, "adaptive", 1, "default", 2.0, "screened"),
          (1, 64, 33, "adaptive", 1, "default", 2.0, "screened"), (3, 50, 41, "adaptive", 1, "default", 2.0, "screened")]


@pytest.mark.parametrize("case", list(enumerate(CASES)), ids=lambda c: f"{c[0]}-B{c[1][0]}-{c[1][1]}x{c[1][2]}-{c[1][3]}-{c[1][5]}-{c[1][7]}")
def test_random_case_matches_fp64_oracle(case):
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle
    idx, (B, H, W, mode, k, variant, gain, scan) = case
    seed = 5000 + idx
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant=variant, sparse_gain=gain).items()}
    x = torch.from_numpy(make_features(seed, B, 64, H, W))
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode, ce.select_k, ce.scan = mode, k, scan
    ce = ce.cuda().eval()
    with torch.no_grad():
        first = ce(x.cuda())
        second = ce(x.cuda())                               # prepared workspace / dense hint
    ref = ce_forward_oracle(x, params, mode=mode, k=k if mode != "adaptive" else None, dtype=torch.float64).numpy()
    assert normwise(first.cpu().numpy(), ref) <= 1e-4
    assert normwise(second.cpu().numpy(), ref) <= 1e-4
