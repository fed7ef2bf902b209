"""Non-default patch geometries (ksize, stride_1, stride_2, inter_channels: ctor arguments of the reference's CE,
DN_Gray/model/dagl.py:175-176): the CPU oracle against goldens minted from the reference built with those arguments
(tests/golden/make_golden_geometry.py), and the host-side contract of ``dagl_amd.CE`` for them."""
import json
import os

import numpy as np
import pytest
import torch

from tests.conftest import REPO, geometry_cases
from tests.helpers import case_inputs, load_geometry_golden, normwise
from oracle.ce_oracle import ce_forward_oracle

CASES = geometry_cases()


def test_geometry_goldens_present():
    assert len(CASES) >= 9


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[5:-4] for p in CASES])
def test_oracle_matches_reference_geometry(path):
    meta, g = load_geometry_golden(path)
    x, params = case_inputs(meta)
    out, st = ce_forward_oracle(x, params, mode=meta["mode"], k=meta["k"] or None, stages=True, softmax_scale=float(meta["softmax_scale"]),
                                ksize=meta["ksize"], stride_q=meta["stride_1"], stride_kv=meta["stride_2"])
    assert out.shape == g["out"].shape
    assert normwise(out.numpy(), g["out"]) <= 1e-4                         # (same bar and reasoning as tests/test_oracle_golden.py)
    np.testing.assert_array_equal(st["deg"].numpy().astype(np.int32), g["deg"])


def test_ce_accepts_any_geometry_and_keeps_the_reference_state_dict():
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params
    ce = CE(ksize=5, stride_1=3, stride_2=1, in_channels=32, inter_channels=8)
    p = make_ce_params(3, in_channels=32, inter_channels=8, ksize=5)
    ce.load_state_dict({n: torch.from_numpy(a) for n, a in p.items()}, strict=True)     # names / shapes of dagl.py:190-205
    assert ce.fc1[0].weight.shape == (50, 200) and ce.thr_conv.weight.shape == (1, 32, 5, 5) and ce.thr_conv.stride == (3, 3)
    assert ce._generic and not CE()._generic


def test_ce_rejects_what_the_generic_route_does_not_hold():
    from dagl_amd.ce import CE
    from dagl_amd._lib import DaglError
    with pytest.raises(DaglError, match="multiple of 4"):
        CE(inter_channels=6)
    with pytest.raises(DaglError, match="ksize"):
        CE(ksize=0)
    with pytest.raises(DaglError, match="GPU"):
        CE(ksize=5, stride_1=3)(torch.zeros(1, 64, 12, 12))


def test_recorded_geometry_on_which_the_reference_raises():
    with open(os.path.join(REPO, "tests", "golden", "geom_raises.json")) as f:
        cases = json.load(f)
    assert cases and all(c["error"] == "RuntimeError" for c in cases)


# ---- gradients: the oracle's autograd against the reference's own, for non-default geometries -----------------------------------------
import glob  # noqa: E402

from tests.helpers import GOLDEN_DIR  # noqa: E402

GEOM_GRAD_CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "geomgrad_*.npz")))


def geom_grad_inputs(meta):
    from dagl_amd.synth import make_ce_params, make_features
    p = make_ce_params(meta["seed"], in_channels=meta["C"], inter_channels=meta["inter_channels"], ksize=meta["ksize"],
                       variant=meta["variant"], sparse_gain=meta["sparse_gain"])
    x = torch.from_numpy(make_features(meta["seed"], meta["B"], meta["C"], meta["H"], meta["W"]))
    G = np.random.Generator(np.random.PCG64(meta["seed"] + 1000)).standard_normal(
        (meta["B"], meta["inter_channels"], meta["H"], meta["W"])).astype(np.float32)
    return x, {n: torch.from_numpy(a) for n, a in p.items()}, torch.from_numpy(G)


def geom_oracle_grads(meta, dtype):
    x, params, G = geom_grad_inputs(meta)
    x = x.to(dtype).requires_grad_(True)
    P = {n: t.to(dtype).requires_grad_(True) for n, t in params.items()}
    out = ce_forward_oracle(x, P, mode=meta["mode"], k=meta["k"] or None, dtype=dtype, softmax_scale=float(meta["softmax_scale"]),
                            ksize=meta["ksize"], stride_q=meta["stride_1"], stride_kv=meta["stride_2"])
    (out * G.to(dtype)).sum().backward()
    grads = {"d_x": x.grad}
    grads.update({"d_" + n: t.grad for n, t in P.items() if t.grad is not None})
    return out.detach(), grads


@pytest.mark.parametrize("path", GEOM_GRAD_CASES, ids=[os.path.basename(p)[9:-4] for p in GEOM_GRAD_CASES])
def test_oracle_autograd_matches_reference_gradients_geometry(path):
    from tests.test_oracle_grad import compare_grads, load_grad_case
    meta, want = load_grad_case(path)
    out, grads = geom_oracle_grads(meta, torch.float32)
    assert normwise(out.numpy(), want["out"]) <= 1e-4
    compare_grads(grads, want, meta["fc_step"], 5e-4)
