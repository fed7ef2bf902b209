"""CPU tests of the host-side mirror: geometry helpers, module surface, state_dict compatibility."""
import numpy as np
import pytest
import torch

from dagl_amd.synth import make_ce_params, make_features, query_grid, same_pad_amounts
from oracle.ce_oracle import same_pad


@pytest.mark.parametrize("size", list(range(1, 41)) + [63, 64, 72, 255, 256, 1024])
def test_same_pad_matches_oracle(size):
    x = torch.zeros(1, 1, size, size)
    for k, s in ((7, 4), (7, 1)):
        _, (l, r, t, b) = same_pad(x, k, s)
        assert same_pad_amounts(size, k, s) == (t, b) == (l, r)
        assert t <= 3 and b <= 3          # every patch fits the 3-pixel zero border of the NHWC maps


def test_query_grid():
    assert query_grid(256, 256) == (64, 64, 1, 1)
    assert query_grid(63, 50) == (16, 13, 2, 2)
    assert query_grid(23, 30)[:2] == (6, 8)


def test_synth_is_deterministic_and_variant_stable():
    a = make_ce_params(5, variant="default")
    b = make_ce_params(5, variant="sparse")
    for n in ("g.weight", "theta.bias", "fc1.0.weight", "fc2.0.bias"):
        np.testing.assert_array_equal(a[n], b[n])
    assert float(b["thr_conv.bias"][0]) == 2.0
    x1, x2 = make_features(3, 1, 4, 5, 6), make_features(3, 1, 4, 5, 6)
    np.testing.assert_array_equal(x1, x2)


def test_module_surface_matches_reference_block():
    from dagl_amd.ce import CE
    ce = CE(in_channels=64)
    sd = ce.state_dict()
    expect = {"g.weight": (16, 64, 3, 3), "g.bias": (16,), "W.weight": (64, 16, 1, 1), "W.bias": (64,),
              "theta.weight": (16, 64, 1, 1), "theta.bias": (16,), "fc1.0.weight": (196, 784), "fc1.0.bias": (196,),
              "fc2.0.weight": (196, 784), "fc2.0.bias": (196,), "thr_conv.weight": (1, 64, 7, 7),
              "thr_conv.bias": (1,), "bias_conv.weight": (1, 64, 7, 7), "bias_conv.bias": (1,)}
    assert list(sd.keys()) == list(expect.keys())          # same names, same registration order (dagl.py:190-205)
    for n, shp in expect.items():
        assert tuple(sd[n].shape) == shp
    # a synthetic "reference checkpoint" loads strictly
    ce.load_state_dict({n: torch.from_numpy(a) for n, a in make_ce_params(1).items()}, strict=True)
    assert sum(p.numel() for p in ce.parameters()) == 325354


def test_module_rejects_cpu_and_foreign_hyperparameters():
    from dagl_amd._lib import DaglError
    from dagl_amd.ce import CE
    assert CE(ksize=5)._generic                              # (any patch geometry since round 6: tests/test_geometry_oracle.py)
    with pytest.raises(DaglError, match="inter_channels"):
        CE(inter_channels=10)
    with pytest.raises(DaglError, match="softmax_scale"):
        CE(softmax_scale=0)
    # softmax_scale is served by scaling fc1 (and the bias head) on the module side: c^2 = scale / 10, c = scale / 10 in top-k mode
    ce = CE(softmax_scale=40)
    assert ce._scale_c() == 2.0
    ce.select_mode = "topk"
    assert ce._scale_c() == 4.0
    assert CE()._scale_c() == 1.0
    prm = ce._params_f32()
    assert torch.equal(prm["fc1.0.weight"], ce.fc1[0].weight.detach() * 4.0) and prm["fc2.0.weight"] is not None
    assert torch.equal(prm["bias_conv.weight"], ce.bias_conv.weight.detach())          # (no bias head in the fixed-k variant)
    ce = CE(in_channels=64)
    with pytest.raises(DaglError, match="GPU"):
        with torch.no_grad():
            ce(torch.zeros(1, 64, 8, 8))
    with pytest.raises(DaglError, match="expected"):
        with torch.no_grad():
            ce(torch.zeros(1, 32, 8, 8))


def test_network_width_other_than_64_fails_like_the_reference():
    """``CES(in_channels=n_feats)`` concatenates four 16-channel heads into its ``Conv2d(n_feats, n_feats, 1)`` mix
    (DN_Gray/model/dagl.py:94-119): the reference's own ``RR(n_feats=32)`` cannot run a forward (a shape error at ``c1_c``),
    although each ``CE(in_channels=32)`` head is a valid module on its own (goldens ``*_c32_*`` / ``*_c96_*`` / ``*_c128_*``).
    Same wiring here: the network builds, strict-loads its own state_dict and raises at the mix -- on the CPU already, before any
    kernel runs, because the heads are never reached with a consistent shape."""
    import torch
    from dagl_amd.net import CES
    ces = CES(32)
    assert ces.fuse_stage is False
    assert ces.c1_1.in_channels == 32 and ces.c1_c.weight.shape == (32, 32, 1, 1)
    with pytest.raises(RuntimeError):
        ces.c1_c(torch.zeros(1, 64, 8, 8))          # what cat(four 16-channel heads) hands the mix
