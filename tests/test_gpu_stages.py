"""GPU parity, stage by stage: every HIP stage (through the C ABI) against the CPU oracle's stage output."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import normwise

pytestmark = pytest.mark.gpu

SHAPES = [(1, 16, 16), (2, 23, 30), (1, 63, 50), (1, 64, 64), (1, 40, 100)]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def _case(seed, B, H, W, variant="sparse"):
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle, params_to_torch
    p = params_to_torch(make_ce_params(seed, variant=variant, sparse_gain=1.6))
    x = torch.from_numpy(make_features(seed, B, 64, H, W))
    out, st = ce_forward_oracle(x, p, stages=True)
    return x, p, out, st


def test_device_is_gfx950():
    from dagl_amd import _lib
    _dev()
    _lib.check(_lib.load().dagl_device_check(), "dagl_device_check")


@pytest.mark.parametrize("B,H,W", SHAPES)
def test_pad_nhwc_bit_exact(B, H, W):
    from dagl_amd import ops
    d = _dev()
    x = torch.randn(B, 16, H, W, generator=torch.Generator().manual_seed(1))
    got = ops.pad_nhwc(x.to(d)).cpu()
    want = F.pad(x, (3, 3, 3, 3)).permute(0, 2, 3, 1).contiguous()
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,H,W", SHAPES)
def test_projection_matches_oracle(B, H, W):
    """fc1/fc2 + ReLU on every patch (dagl.py:248-249) as implicit GEMM vs the unfold+Linear oracle."""
    from dagl_amd import ops
    d = _dev()
    x, p, _, st = _case(31, B, H, W)
    b1p = ops.pad_nhwc(st["b1"].to(d).contiguous())
    L, N = st["Wq"].shape[1], st["X"].shape[1]
    wp2 = ops.pack_fc_weight(p["fc2.0.weight"].to(d))
    wp1 = ops.pack_fc_weight(p["fc1.0.weight"].to(d))
    X, colsum = ops.project_patches(b1p, wp2, p["fc2.0.bias"].to(d), H, W, queries=False, want_colsum=True)
    Wq, _ = ops.project_patches(b1p, wp1, p["fc1.0.bias"].to(d), H, W, queries=True)
    X, Wq, colsum = X.cpu(), Wq.cpu(), colsum.cpu()
    # fp32 dot products of 784 terms in a different order than MKL: ~1e-6 normwise
    assert normwise(X[:, :N, :196].numpy(), st["X"].numpy()) <= 5e-6
    assert normwise(Wq[:, :L, :196].numpy(), st["Wq"].numpy()) <= 5e-6
    assert float(X[:, :, 196:].abs().max()) == 0.0 and float(X[:, N:, :].abs().max()) == 0.0
    assert float(Wq[:, :, 196:].abs().max()) == 0.0 and float(Wq[:, L:, :].abs().max()) == 0.0
    want_cs = st["X"].double().sum(dim=1)
    assert normwise(colsum[:, :196].numpy(), want_cs.numpy()) <= 5e-6


@pytest.mark.parametrize("B,H,W", SHAPES[:4])
def test_threshold_and_dense_scores(B, H, W):
    from dagl_amd import ops
    d = _dev()
    x, p, _, st = _case(32, B, H, W)
    b1p = ops.pad_nhwc(st["b1"].to(d).contiguous())
    L, N = st["Wq"].shape[1], st["X"].shape[1]
    X, colsum = ops.project_patches(b1p, ops.pack_fc_weight(p["fc2.0.weight"].to(d)), p["fc2.0.bias"].to(d), H, W,
                                    queries=False, want_colsum=True)
    Wq, _ = ops.project_patches(b1p, ops.pack_fc_weight(p["fc1.0.weight"].to(d)), p["fc1.0.bias"].to(d), H, W,
                                queries=True)
    S = ops.scores_dense(Wq, X, L, N).cpu()
    assert normwise(S.numpy(), st["S"].numpy()) <= 5e-6
    mt = ops.query_thresholds(Wq, colsum, st["thr"].to(d).contiguous(), L, N).cpu()
    want = st["S"].mean(dim=2) * st["thr"]
    assert normwise(mt.numpy(), want.numpy()) <= 5e-6


@pytest.mark.parametrize("L,k,N,P", [(1, 1, 1, 4), (37, 3, 50, 784), (300, 8, 4096, 784), (64, 16, 1000, 128),
                                     (100, 5, 333, 20)])
def test_gather_aggregate_matches_oracle(L, k, N, P):
    from dagl_amd import ops
    from oracle.ce_oracle import gather_aggregate_oracle
    d = _dev()
    g = torch.Generator().manual_seed(L * 131 + k)
    values = torch.randn(N, P, generator=g)
    idx = torch.randint(0, N, (L, k), generator=g, dtype=torch.int32)
    idx[torch.rand(L, k, generator=g) < 0.15] = -1          # empty slots
    wgt = torch.rand(L, k, generator=g)
    got = ops.gather_aggregate(idx.to(d), wgt.to(d), values.to(d)).cpu()
    want = gather_aggregate_oracle(idx, wgt, values)
    assert normwise(got.numpy(), want.numpy()) <= 2e-6


def test_gather_aggregate_empty_and_errors():
    from dagl_amd import ops
    from dagl_amd._lib import DaglError
    d = _dev()
    v = torch.randn(10, 8, device=d)
    out = ops.gather_aggregate(torch.zeros(0, 4, dtype=torch.int32, device=d), torch.zeros(0, 4, device=d), v)
    assert out.shape == (0, 8)
    with pytest.raises(DaglError):
        ops.gather_aggregate(torch.zeros(2, 4, dtype=torch.int32, device=d), torch.zeros(2, 4, device=d),
                             torch.randn(10, 7, device=d))          # P not a multiple of 4
    with pytest.raises(DaglError):
        ops.gather_aggregate(torch.zeros(2, 4, dtype=torch.int32), torch.zeros(2, 4), torch.randn(10, 8))  # CPU tensors


@pytest.mark.parametrize("B,H,W", SHAPES)
def test_unfold_values_matches_unfold(B, H, W):
    from dagl_amd import ops
    from oracle.ce_oracle import patch_rows
    d = _dev()
    b2 = torch.randn(B, 16, H, W, generator=torch.Generator().manual_seed(4))
    rows = ops.unfold_values(ops.pad_nhwc(b2.to(d)), H, W).cpu()              # (kh,kw,c) order
    want = patch_rows(b2, 7, 1).view(B, H * W, 16, 7, 7).permute(0, 1, 3, 4, 2).reshape(B, H * W, 784)
    assert torch.equal(rows, want)


@pytest.mark.parametrize("B,H,W", SHAPES + [(1, 5, 9), (1, 1, 1)])
def test_fold_normalize_matches_fold(B, H, W):
    """fold(agg)/fold(unfold(1)) of dagl.py:265-272, including the fold-grid offset and the overlap count."""
    from dagl_amd import ops
    from oracle.ce_oracle import overlap_count
    d = _dev()
    Lh, Lw = -(-H // 4), -(-W // 4)
    agg = torch.randn(B, Lh * Lw, 16, 7, 7, generator=torch.Generator().manual_seed(5))      # reference order
    z = F.fold(agg.reshape(B, Lh * Lw, 784).transpose(1, 2), (H, W), (7, 7), padding=3, stride=4)
    want = z / overlap_count(H, W, torch.float32)
    mine = agg.permute(0, 1, 3, 4, 2).reshape(B, Lh * Lw, 784).contiguous()                   # (kh,kw,c)
    got = ops.fold_normalize(mine.to(d), H, W).cpu()
    assert normwise(got.numpy(), want.numpy()) <= 1e-6


@pytest.mark.parametrize("B,H,W", SHAPES + [(1, 5, 9), (1, 72, 200)])
def test_fused_prologue_matches_stock_convs(B, H, W):
    """g / theta / thr_conv / bias_conv (dagl.py:208-215) computed by prologue.hip vs torch's CPU convs."""
    from dagl_amd import ops
    d = _dev()
    x, p, _, st = _case(33, B, H, W, variant="default")
    pd = {n: t.to(d).contiguous() for n, t in p.items()}
    b1p, b2p, thr, bias = ops.ce_prologue(x.to(d), pd["g.weight"], pd["g.bias"], pd["theta.weight"], pd["theta.bias"],
                                          pd["thr_conv.weight"], pd["thr_conv.bias"], pd["bias_conv.weight"],
                                          pd["bias_conv.bias"])
    want1 = F.pad(st["b1"], (3, 3, 3, 3)).permute(0, 2, 3, 1)
    want2 = F.pad(st["b2"], (3, 3, 3, 3)).permute(0, 2, 3, 1)
    assert normwise(b1p.cpu().numpy(), want1.numpy()) <= 2e-6
    assert normwise(b2p.cpu().numpy(), want2.numpy()) <= 2e-6
    # zero border exactly zero
    assert float(b1p[:, :3].abs().max()) == 0.0 and float(b1p[:, :, -3:].abs().max()) == 0.0
    assert normwise(thr.cpu().numpy(), st["thr"].numpy()) <= 2e-6
    assert normwise(bias.cpu().numpy(), st["bias"].numpy()) <= 2e-6


def test_stage_profile_can_bracket_a_single_stage():
    """dagl_profile_select_stage: only the two events around one stage are recorded (the benchmark's timed steps use it:
    nine event records per call cost ~12 % of a 256^2 forward)."""
    from dagl_amd import ops
    from dagl_amd._lib import STAGE_NAMES
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(81, variant="default").items()}
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode, ce.select_k = "topk", 8
    ce = ce.cuda().eval()
    x = torch.from_numpy(make_features(81, 1, 64, 96, 96)).cuda()
    prof = ops.StageProfile(4)
    ce.profile = prof
    with torch.no_grad():
        prof.select_stage("select")
        ce(x); ce(x)
        sel = prof.read()
        assert len(sel) == 2
        k = list(STAGE_NAMES).index("select")
        assert all(row[k] > 0 and sum(row) == row[k] for row in sel)
        prof.select_stage(-1)
        ce(x)
        full = prof.read()
        assert len(full) == 1 and sum(v > 0 for v in full[0]) >= 6


@pytest.mark.parametrize("B,H,W,k", [(1, 64, 64, 8), (2, 45, 38, 4), (1, 130, 97, 16), (1, 7, 9, 3)])
def test_fused_gather_fold_is_bit_identical_to_the_two_kernels(B, H, W, k):
    """Top-k calls gather, weight and fold in one kernel (aggregate_fold_kernel); the debug entry point (which hands out the
    aggregated rows) runs aggregate_direct_kernel + fold_kernel.  Same fma chains, same summation order: same bits."""
    from dagl_amd import ops
    DEV = _dev()
    g = torch.Generator().manual_seed(B * 1000 + H + W + k)
    b1 = torch.randn(B, 16, H, W, generator=g).to(DEV)
    b2 = torch.randn(B, 16, H, W, generator=g).to(DEV)
    fc1_w = (torch.randn(196, 784, generator=g) * 0.05).to(DEV); fc1_b = (torch.randn(196, generator=g) * 0.1).to(DEV)
    fc2_w = (torch.randn(196, 784, generator=g) * 0.05).to(DEV); fc2_b = (torch.randn(196, generator=g) * 0.1).to(DEV)
    out_fused = ops.ce_forward(b1, b2, None, None, fc1_w, fc1_b, fc2_w, fc2_b, mode="topk", k=k)
    out_two, meta = ops.ce_forward(b1, b2, None, None, fc1_w, fc1_b, fc2_w, fc2_b, mode="topk", k=k, debug=True)
    assert torch.equal(out_fused, out_two)
    assert torch.isfinite(out_fused).all()
