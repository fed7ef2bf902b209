"""Activations beyond the fine tier of the split-fp16 kernels (round 6: include/dagl_ce.h "range").  Up to round 5 a top-k call
whose input left |x| < 3750 came back NaN-filled and stayed so until the module's next poll; now the input is split with a scale of
the block's own and the key / query map exists in two tiers (csrc/dagl_common.h B1Tiers): the SAME launches serve the call -- finite
in, finite out, as dagl.py:207-275 -- on the first call, with no host poll, under HIP-graph replay, for the fused CES stage too."""
import warnings

import numpy as np
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu


def _setup(mode, k, variant, gain, scale=3.0e3, hw=(64, 64)):
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(77, variant=variant, sparse_gain=gain).items()}
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode = mode
    if k:
        ce.select_k = k
    x = torch.from_numpy(make_features(77, 1, 64, *hw)) * scale          # |x| up to ~1.3e4: 16 x overflows fp16
    return ce.to("cuda:0").eval(), params, x


def test_adaptive_call_beyond_the_fine_tier_is_served():
    from oracle.ce_oracle import ce_forward_oracle
    ce, params, x = _setup("adaptive", 0, "sparse", 1.9)
    with torch.no_grad(), warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        out = ce(x.to("cuda:0")).cpu()
    want = ce_forward_oracle(x, params, mode="adaptive", dtype=torch.float64).float()
    assert torch.isfinite(out).all()
    assert normwise(out.numpy(), want.numpy()) <= 1e-4
    with torch.no_grad():                                                   # same route, same bits on a second call
        again = ce(x.to("cuda:0")).cpu()
    assert torch.equal(again, out)


@pytest.mark.parametrize("scale", [3.0e3, 30.0, 3.0e5])
def test_topk_call_beyond_the_fine_tier_is_finite_and_right_on_the_first_call(scale):
    """x 30: only the INPUT leaves the fine range of some blocks (|16 x| >= 60000 needs |x| >= 3750: no -- the map b1 does at x 3e3);
    x 3e3: input and map; x 3e5: both deep in the coarse tier (|b1| ~ 1e6)."""
    from oracle.ce_oracle import ce_forward_oracle
    ce, params, x = _setup("topk", 8, "default", 2.0, scale)
    xd = x.to("cuda:0")
    with torch.no_grad():
        out = ce(xd)                                                        # the FIRST call of the module: cold workspace, no poll yet
        assert ce._calls_since_range_check == 1 and ce.scan == "screened"
        out = out.cpu()
    assert torch.isfinite(out).all()
    want = ce_forward_oracle(x, params, mode="topk", k=8, dtype=torch.float64).float()
    w32 = ce_forward_oracle(x, params, mode="topk", k=8).float()
    e_ref32 = normwise(w32.numpy(), want.numpy())
    e = normwise(out.numpy(), want.numpy())
    print(f"[range] top-k x{scale:g}: e_hip {e:.2e} e_ref32 {e_ref32:.2e}")
    assert e <= max(1e-4, 2.0 * e_ref32 + 1e-5), (e, e_ref32)
    with torch.no_grad():
        assert ce.range_ok() and ce.scan == "screened"                      # nothing to report
        small = ce(xd * (1.0 / scale)).cpu()                                # back in the fine tier on the same (prepared) workspace
    want_s = ce_forward_oracle(x * (1.0 / scale), params, mode="topk", k=8, dtype=torch.float64).float()
    assert normwise(small.numpy(), want_s.numpy()) <= 1e-4


def test_topk_call_beyond_the_fine_tier_under_graph_replay():
    from oracle.ce_oracle import ce_forward_oracle
    ce, params, x = _setup("topk", 8, "default", 2.0, 1.0)
    xs = torch.empty_like(x, device="cuda:0")
    with torch.no_grad():
        xs.copy_(x)
        for _ in range(3):
            ce(xs)                                                          # warm: packed weights, prepared workspace
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = ce(xs)
        for scale in (1.0, 3.0e3, 1.0):                                     # in range, far out, in range again: one captured launch set
            xs.copy_(x * scale)
            g.replay()
            torch.cuda.synchronize()
            got = out.cpu()
            assert torch.isfinite(got).all(), scale
            want = ce_forward_oracle(x * scale, params, mode="topk", k=8, dtype=torch.float64).float()
            assert normwise(got.numpy(), want.numpy()) <= 1e-4, scale


def test_fused_stage_beyond_the_fine_tier_first_call():
    """One CES stage (four top-k heads + 1x1 mix + residual, dagl.py:112-119) on the fused launch set at input x 3e3: finite and equal to
    the per-head modules' exact-scan (fp32) results."""
    from dagl_amd.ce import CE
    from dagl_amd.net import CES
    from dagl_amd.synth import make_features
    torch.manual_seed(5)
    ces = CES(64).to("cuda:0").eval()
    for m in ces.modules():
        if isinstance(m, CE):
            m.select_mode, m.select_k = "topk", 8
    x = (torch.from_numpy(make_features(91, 1, 64, 48, 52)) * 3.0e3).to("cuda:0")
    with torch.no_grad():
        got = ces._stage(1, x)
        assert ces.last_info is not None                                    # the fused path ran
        assert torch.isfinite(got).all()
        heads = [getattr(ces, f"c1_{h}") for h in (1, 2, 3, 4)]
        for hd in heads:
            hd.scan = "exact"
        want = ces.c1_c(torch.cat([hd(x) for hd in heads], dim=1)) + x
    assert normwise(got.cpu().numpy(), want.cpu().numpy()) <= 1e-4


def test_nonfinite_input_is_nan_filled_and_reported():
    ce, params, x = _setup("topk", 8, "default", 2.0, 1.0)
    x[0, 3, 10, 10] = float("inf")
    with torch.no_grad():
        out = ce(x.to("cuda:0"))
        assert torch.isnan(out).all()
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            assert not ce.range_ok()


@pytest.mark.parametrize("scale", [1.0, 3.0e3])
def test_topk_beyond_64_neighbours_has_no_fixed_feature_range(scale):
    """k = 100 (row-wise dense form, csrc/topk_wide.hip): the split-fp16 scores take their power of two from the image's largest
    feature since round 6 -- an input x 3e3 (features ~1e3 x the fixed scale's limit of 937) is served on the first call."""
    from oracle.ce_oracle import ce_forward_oracle
    ce, params, x = _setup("topk", 100, "default", 2.0, scale, hw=(48, 56))
    with torch.no_grad():
        out = ce(x.to("cuda:0")).cpu()
        assert ce.scan == "screened"
    assert torch.isfinite(out).all()
    want = ce_forward_oracle(x, params, mode="topk", k=100, dtype=torch.float64).float()
    e_ref32 = normwise(ce_forward_oracle(x, params, mode="topk", k=100).numpy(), want.numpy())
    e = normwise(out.numpy(), want.numpy())
    print(f"[range] top-k 100 x{scale:g}: e_hip {e:.2e} e_ref32 {e_ref32:.2e} path {ce.last_info and ce.last_info.get('path')}")
    assert e <= max(1e-4, 2.0 * e_ref32 + 1e-5), (e, e_ref32)


def test_random_shapes_and_input_scales_first_call():
    """Twelve seeded (batch, H, W, k, scale) draws -- ragged widths (partial 64-pixel strips of the prologue, rows that are no multiple
    of 32 in the projection), batches, scales from 1e-3 to 1e5 -- each on a cold module's first call against the fp64 oracle."""
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle
    rng = np.random.default_rng(606)
    worst = 0.0
    for case in range(12):
        B = int(rng.integers(1, 4)); H = int(rng.integers(24, 90)); W = int(rng.integers(24, 90))
        k = int(rng.choice([4, 8, 16])); scale = float(rng.choice([1e-3, 1.0, 1e2, 3e3, 1e5]))
        params = {n: torch.from_numpy(a) for n, a in make_ce_params(100 + case, variant="default").items()}
        x = torch.from_numpy(make_features(200 + case, B, 64, H, W)) * scale
        ce = CE(in_channels=64)
        ce.load_state_dict(params, strict=True)
        ce.select_mode, ce.select_k = "topk", k
        ce = ce.to("cuda:0").eval()
        with torch.no_grad():
            out = ce(x.to("cuda:0")).cpu()
        assert torch.isfinite(out).all(), (case, B, H, W, k, scale)
        want = ce_forward_oracle(x, params, mode="topk", k=k, dtype=torch.float64).float()
        e_ref32 = normwise(ce_forward_oracle(x, params, mode="topk", k=k).numpy(), want.numpy())
        e = normwise(out.numpy(), want.numpy())
        worst = max(worst, e)
        assert e <= max(1e-4, 2.0 * e_ref32 + 1e-5), (case, B, H, W, k, scale, e, e_ref32)
    print(f"[range] 12 random (batch, H, W, k, scale) draws, first calls: worst {worst:.2e} of the fp64 oracle")
