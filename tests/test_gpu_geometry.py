"""``dagl_amd.CE`` built with non-default ``ksize / stride_1 / stride_2 / inter_channels`` (ctor arguments of the reference's CE,
DN_Gray/model/dagl.py:175-176) on the MI355X: ``dagl_ce_generic_forward`` (csrc/generic.hip) against goldens minted from the reference
built with the same arguments (tests/golden/make_golden_geometry.py) and against the fp64 oracle on further geometries."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import REPO, geometry_cases
from tests.helpers import case_inputs, load_geometry_golden, normwise

pytestmark = pytest.mark.gpu

CASES = geometry_cases()


def _module(meta_or_kw, params, mode="adaptive", k=0, scale=10):
    from dagl_amd.ce import CE
    kw = dict(ksize=meta_or_kw["ksize"], stride_1=meta_or_kw["stride_1"], stride_2=meta_or_kw["stride_2"],
              in_channels=meta_or_kw["C"], inter_channels=meta_or_kw["inter_channels"], softmax_scale=scale)
    ce = CE(**kw)
    ce.load_state_dict(params, strict=True)
    ce.select_mode = mode
    if k:
        ce.select_k = k
    return ce.cuda().eval()


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-4] for p in CASES])
def test_generic_geometry_matches_reference_goldens(path):
    meta, g = load_geometry_golden(path)
    x, params = case_inputs(meta)
    ce = _module(meta, params, meta["mode"], meta["k"], meta["softmax_scale"])
    with torch.no_grad():
        out = ce(x.cuda())
    assert out.shape == g["out"].shape and ce.last_info["path"] == 7
    assert normwise(out.cpu().numpy(), g["out"]) <= 1e-4
    deg = ce.last_info["degree"].cpu().numpy()
    if meta["mode"] == "topk":
        np.testing.assert_array_equal(deg, g["deg"])
    else:
        # a key whose mask value lies within an ulp of zero may fall on either side (the product's summation order is not MKL's)
        diff = np.abs(deg.astype(np.int64) - g["deg"])
        assert diff.max() <= 2 and (diff != 0).mean() <= 0.02, (diff.max(), (diff != 0).mean())


# (ksize, stride_1, stride_2, inter_channels, Cin, B, H, W, variant, gain, mode, k, scale)
ORACLE_CASES = [
    (5, 2, 1, 16, 64, 1, 64, 64, "sparse", 1.6, "adaptive", 0, 10),
    (3, 1, 1, 4, 4, 2, 96, 96, "sparse", 1.4, "adaptive", 0, 10),         # two images, L = N = 9216 each: two chunks of score rows per image
    (7, 4, 1, 16, 64, 2, 40, 44, "sparse", 1.6, "adaptive_topk", 12, 10),  # (default geometry through the generic entry point: see below)
    (11, 4, 1, 8, 16, 1, 36, 40, "default", 2.0, "topk", 20, 7),
    (4, 4, 1, 16, 64, 1, 32, 48, "sparse", 1.5, "adaptive", 0, 10),        # even window: SAME pad (1, 2)
    (5, 3, 1, 16, 64, 3, 27, 33, "default", 2.0, "adaptive", 0, 2.5),
    (3, 4, 1, 16, 64, 1, 32, 40, "sparse", 1.3, "adaptive", 0, 10),         # stride_1 > ksize: pixels no query window covers (zero guard, dagl.py:271)
    (1, 1, 1, 16, 64, 1, 24, 20, "sparse", 1.2, "adaptive", 0, 10),         # 1 x 1 patches: P = c, D = c / 4
    (7, 4, 2, 16, 64, 1, 64, 72, "default", 2.0, "topk", 40, 10),           # keys on a stride-2 grid
]


@pytest.mark.parametrize("case", ORACLE_CASES, ids=[f"k{c[0]}s{c[1]}kv{c[2]}c{c[3]}_{c[10]}_{c[6]}x{c[7]}" for c in ORACLE_CASES])
def test_generic_geometry_matches_fp64_oracle(case):
    from dagl_amd import ops
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle
    ks, s1, s2, c, Cin, B, H, W, variant, gain, mode, k, scale = case
    seed = 700 + ks * 13 + s1
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, in_channels=Cin, inter_channels=c, ksize=ks, variant=variant,
                                                                sparse_gain=gain).items()}
    x = torch.from_numpy(make_features(seed, B, Cin, H, W))
    want, st = ce_forward_oracle(x, params, mode=mode, k=k or None, dtype=torch.float64, stages=True, softmax_scale=float(scale),
                                 ksize=ks, stride_q=s1, stride_kv=s2)
    dev = {n: t.cuda() for n, t in params.items()}
    out, deg = ops.ce_forward_generic(x.cuda(), dev, ks, s1, s2, c, mode=mode, k=k, softmax_scale=float(scale), want_degree=True)
    assert normwise(out.cpu().numpy(), want.numpy()) <= 1e-4
    diff = (deg.cpu().long() - st["deg"].long()).abs()
    assert int(diff.max()) <= 2 and float((diff != 0).float().mean()) <= 0.02


def test_generic_entry_point_equals_the_tuned_kernels_on_the_default_geometry():
    """(7, 4, 1, 16) through ``dagl_ce_generic_forward`` and through the module's tuned path: two implementations, one result."""
    from dagl_amd import ops
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(91, variant="sparse", sparse_gain=1.6).items()}
    x = torch.from_numpy(make_features(91, 2, 64, 48, 52)).cuda()
    for mode, k in (("adaptive", 0), ("topk", 8)):
        ce = CE()
        ce.load_state_dict(params, strict=True)
        ce.select_mode, ce.select_k = mode, k or ce.select_k
        ce = ce.cuda().eval()
        with torch.no_grad():
            a = ce(x)
        b = ops.ce_forward_generic(x, {n: t.cuda() for n, t in params.items()}, 7, 4, 1, 16, mode=mode, k=k)
        assert normwise(a.cpu().numpy(), b.cpu().numpy()) <= 5e-5, mode


def test_geometry_on_which_the_reference_raises_is_refused():
    from dagl_amd._lib import DaglError
    from dagl_amd.synth import make_ce_params, make_features
    with open(os.path.join(REPO, "tests", "golden", "geom_raises.json")) as f:
        cases = json.load(f)
    for cs in cases:
        params = {n: torch.from_numpy(a) for n, a in make_ce_params(5, in_channels=cs["C"], inter_channels=cs["inter_channels"],
                                                                    ksize=cs["ksize"]).items()}
        ce = _module(cs, params)
        with pytest.raises(DaglError, match="fold"):
            ce(torch.from_numpy(make_features(5, cs["B"], cs["C"], cs["H"], cs["W"])).cuda())


def test_generic_module_contract():
    """Half-precision I/O, an input width that is not a multiple of 4, train() mode, workspace check of the C ABI."""
    from dagl_amd import _lib
    from dagl_amd._lib import DaglError
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle
    meta = dict(ksize=5, stride_1=3, stride_2=1, C=22, inter_channels=8)
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(77, in_channels=22, inter_channels=8, ksize=5, variant="sparse",
                                                                sparse_gain=1.5).items()}
    x = torch.from_numpy(make_features(77, 1, 22, 24, 27))
    want = ce_forward_oracle(x, params, dtype=torch.float64, ksize=5, stride_q=3, stride_kv=1)
    ce = _module(meta, params)
    with torch.no_grad():
        out = ce(x.cuda())
    assert normwise(out.cpu().numpy(), want.numpy()) <= 1e-4
    xh = x.cuda().half()
    with torch.no_grad():
        oh = ce(xh)
    assert oh.dtype == torch.float16
    want_h = ce_forward_oracle(xh.float().cpu(), params, dtype=torch.float64, ksize=5, stride_q=3, stride_kv=1)
    assert normwise(oh.float().cpu().numpy(), want_h.numpy()) <= 2e-3           # (fp16 output rounding)
    ce.train()
    o2 = ce(x.cuda())                                                            # train() under autograd: the differentiable route
    assert o2.requires_grad and normwise(o2.detach().cpu().numpy(), want.numpy()) <= 1e-4
    with torch.no_grad():                                                        # train() without autograd: the inference route
        assert torch.equal(ce(x.cuda()), out)
    lib = _lib.load()
    need = lib.dagl_ce_generic_workspace_bytes(1, 24, 24, 27, 5, 3, 1, 8)
    assert need > 0 and lib.dagl_ce_generic_workspace_bytes(1, 24, 24, 27, 0, 3, 1, 8) == 0
    buf = torch.empty(1024, device="cuda", dtype=torch.uint8)
    p = {n: t.cuda() for n, t in params.items()}
    xx = torch.zeros(1, 24, 24, 27, device="cuda")
    o = torch.empty(1, 8, 24, 27, device="cuda")
    rc = lib.dagl_ce_generic_forward(None, 1, 24, 24, 27, 5, 3, 1, 8, C.c_float(10.0), 0, 0, xx.data_ptr(), *[p[n].data_ptr() for n in (
        "g.weight", "g.bias", "theta.weight", "theta.bias", "thr_conv.weight", "thr_conv.bias", "bias_conv.weight", "bias_conv.bias",
        "fc1.0.weight", "fc1.0.bias", "fc2.0.weight", "fc2.0.bias")], o.data_ptr(), None, buf.data_ptr(), buf.numel())
    assert rc == _lib.ERR_WORKSPACE and b"workspace" in lib.dagl_last_error()


# ---- autograd through a module with a non-default geometry (dagl_ce_generic_core_forward / _backward) -------------------------------
from tests.test_geometry_oracle import GEOM_GRAD_CASES, geom_grad_inputs, geom_oracle_grads  # noqa: E402


def _hip_geom_grads(meta):
    x, params, G = geom_grad_inputs(meta)
    ce = _module(meta, params, meta["mode"], meta["k"], meta["softmax_scale"]).train()
    xg = x.cuda().requires_grad_(True)
    out = ce(xg)
    (out * G.cuda()).sum().backward()
    grads = {"d_x": xg.grad}
    grads.update({"d_" + n: p.grad for n, p in ce.named_parameters() if p.grad is not None})
    return ce, out.detach(), grads


@pytest.mark.parametrize("path", GEOM_GRAD_CASES, ids=[os.path.basename(p)[9:-4] for p in GEOM_GRAD_CASES])
def test_generic_geometry_gradients_match_reference_autograd(path):
    from tests.test_oracle_grad import compare_grads, load_grad_case
    meta, want = load_grad_case(path)
    ce, out, grads = _hip_geom_grads(meta)
    assert normwise(out.cpu().numpy(), want["out"]) <= 1e-4
    assert "d_W.weight" not in grads
    compare_grads(grads, want, meta["fc_step"], 5e-4)
    if meta["mode"] == "topk":
        assert "d_thr_conv.weight" not in grads
    # ... and as close to the fp64 oracle's autograd as the reference's own fp32 gradients are
    _, g64 = geom_oracle_grads(meta, torch.float64)
    for name, w in g64.items():
        w = w.numpy()
        e_hip = normwise(grads[name].cpu().numpy(), w)
        if name in ("d_fc1.0.weight", "d_fc2.0.weight"):
            w = w.reshape(-1)[::meta["fc_step"]]
        e_ref = normwise(want[name], w)
        assert e_hip <= 3 * e_ref + 1e-4, (name, e_hip, e_ref)


def test_generic_geometry_training_forward_equals_inference_and_eval_input_gradients():
    from tests.test_oracle_grad import load_grad_case
    meta, _ = load_grad_case([p for p in GEOM_GRAD_CASES if "k5s3_sparse_b2" in p][0])
    x, params, G = geom_grad_inputs(meta)
    for mode, k in (("adaptive", 0), ("topk", 5), ("adaptive_topk", 9)):
        ce = _module(meta, params, mode, k, meta["softmax_scale"])
        xd = x.cuda()
        with torch.no_grad():
            ref = ce(xd)
        ce.train()
        out = ce(xd.clone().requires_grad_(True))
        assert out.requires_grad and normwise(out.detach().cpu().numpy(), ref.cpu().numpy()) <= 2e-5, mode
        # an eval() module whose INPUT requires a gradient (the reference's test loop runs with autograd on): same input gradient
        ce.eval()
        xa, xb = xd.clone().requires_grad_(True), xd.clone().requires_grad_(True)
        (ce(xa) * G.cuda()).sum().backward()
        ce.train()
        (ce(xb) * G.cuda()).sum().backward()
        assert torch.equal(xa.grad, xb.grad), mode
    m2 = dict(meta, mode="adaptive_topk", k=9)
    _, g64 = geom_oracle_grads(m2, torch.float64)
    ce = _module(meta, params, "adaptive_topk", 9, meta["softmax_scale"]).train()
    xg = x.cuda().requires_grad_(True)
    (ce(xg) * G.cuda()).sum().backward()
    got = {"d_x": xg.grad, **{"d_" + n: p.grad for n, p in ce.named_parameters() if p.grad is not None}}
    for name, w in g64.items():
        assert normwise(got[name].cpu().numpy(), w.numpy()) <= 5e-4, name
