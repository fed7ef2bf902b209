"""The module's host-side policy state TOGETHER (review of round 5, item 12): packed-weight key, dense hint and its 16-call cadence,
served streaks per shape, the workspace's top-k threshold policy word, the range poll cadence -- each has its own test; here ONE module
of each mode goes through a seeded random sequence of shapes, inputs (ordinary, x 3e3, x 1e-3, nearly flat, non-finite) and weight
edits, and every call must give what a FRESH module gives on the same input (same kernels, cold state): no call of the default
configuration returns NaN for finite input, whatever came before it."""
import warnings

import numpy as np
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu

SHAPES = [(1, 48, 52), (2, 64, 64), (1, 96, 80), (3, 40, 40)]


def _module(params, mode, k):
    from dagl_amd.ce import CE
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode = mode
    if k:
        ce.select_k = k
    return ce.to("cuda:0").eval()


def _input(rng, kind, shape):
    from dagl_amd.synth import make_features
    B, H, W = shape
    x = torch.from_numpy(make_features(int(rng.integers(1, 10000)), B, 64, H, W))
    if kind == "big":
        x = x * 3.0e3
    elif kind == "tiny":
        x = x * 1.0e-3
    elif kind == "flat":
        g = torch.Generator().manual_seed(int(rng.integers(1, 10000)))
        x = 0.25 + 2e-4 * torch.randn(B, 64, H, W, generator=g)
    elif kind == "inf":
        x[0, 5, 3, 4] = float("inf")
    return x


@pytest.mark.parametrize("mode,k,variant,gain", [("topk", 8, "default", 2.0), ("adaptive", 0, "sparse", 1.9), ("adaptive", 0, "default", 2.0),
                                                 ("adaptive_topk", 16, "sparse", 1.7)])
def test_a_module_with_history_gives_what_a_fresh_module_gives(mode, k, variant, gain):
    from dagl_amd.synth import make_ce_params
    rng = np.random.default_rng(20260929 + k)
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(31, variant=variant, sparse_gain=gain).items()}
    ce = _module(params, mode, k)
    kinds = ["normal"] * 6 + ["big", "big", "tiny", "flat", "inf"]
    worst = 0.0
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(70):
            shape = SHAPES[int(rng.integers(0, len(SHAPES)))]
            kind = kinds[int(rng.integers(0, len(kinds)))]
            if step % 23 == 22:                                   # an in-place weight edit (the optimizer's way): the packed copies must follow
                with torch.no_grad():
                    ce.fc2[0].weight.mul_(1.01)
                params = {n: p.detach().cpu().clone() for n, p in ce.state_dict().items()}
            x = _input(rng, kind, shape)
            got = ce(x.to("cuda:0")).cpu()
            fresh = _module(params, mode, k)
            if ce.scan == "exact":                                # (a non-finite input has moved the module to the fp32 path: so be the reference)
                fresh.scan = "exact"
            want = fresh(x.to("cuda:0")).cpu()
            if kind == "inf":
                assert not torch.isfinite(got).all()              # NaN / inf in, never finite garbage out
                continue
            assert torch.isfinite(got).all(), (step, kind, shape, ce.scan, ce.last_info)
            assert torch.isfinite(want).all()
            # same kernels on a cold workspace: equal up to the routes the history may choose differently (threshold policy tight / sampled,
            # dense hint / optimistic lists, exact scan after an inf) -- each of them within 1e-4 of the fp64 oracle, so 2e-4 of each other;
            # nearly flat maps: thousands of near-ties, which route keeps which is a matter of last bits (tests/test_gpu_round5.py)
            e = normwise(got.numpy(), want.numpy())
            worst = max(worst, e if kind != "flat" else 0.0)
            assert e <= (2e-4 if kind != "flat" else 2e-3), (step, kind, shape, e, ce.scan, ce.last_info)
    print(f"[policy product] {mode} k={k}: 70 calls with history vs fresh modules, worst {worst:.2e}, module ended on scan={ce.scan}")
