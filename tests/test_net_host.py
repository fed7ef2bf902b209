"""CPU tests of the caller-side restatement (dagl_amd/net.py): state_dict compatibility with the reference trunk,
tiling geometry, PSNR helper.  The CE heads are replaced by a CPU stand-in: the block itself needs the GPU."""
import json
import os

import numpy as np
import torch
import torch.nn as nn

from tests.helpers import GOLDEN_DIR


class _StubCE(nn.Module):
    def __init__(self, in_channels=64):
        super().__init__()
        self.g = nn.Conv2d(in_channels, 16, 3, padding=1)

    def forward(self, x):
        return self.g(x)


def test_rr_state_dict_matches_reference_names_and_shapes():
    """Key names, order and shapes equal the reference RR's (fixture written by tests/golden/make_set12_psnr.py's
    companion check against /root/reference/DN_Gray/model/dagl.py)."""
    from dagl_amd.net import RR
    want = [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLDEN_DIR, "rr_state_keys.json")))]
    got = [(k, tuple(v.shape)) for k, v in RR().state_dict().items()]
    assert got == want
    assert sum(int(np.prod(s)) for _, s in got) == 5727453


def test_seeded_state_dict_is_reproducible():
    from dagl_amd.net import RR, seeded_state_dict
    m = RR()
    a, b = seeded_state_dict(m.state_dict(), 7), seeded_state_dict(m.state_dict(), 7)
    assert all(torch.equal(a[k], b[k]) for k in a)
    m.load_state_dict(a, strict=True)
    assert float(a["body.0.body.1.weight"]) == 0.25


def test_chop_forward_tiles_like_the_reference():
    """Leaf-tile geometry documented in SURVEY.md section 0.4: 256x256 -> 64 tiles of 72x72, 512x512 -> 256 of 76x76;
    stitching an identity network reproduces the input exactly."""
    from dagl_amd.net import chop_forward
    seen = []

    class Probe(nn.Module):
        def forward(self, x):
            seen.append(tuple(x.shape[-2:]))
            return x

    for size, n, t in ((256, 64, 72), (512, 256, 76)):
        seen.clear()
        x = torch.rand(1, 1, size, size)
        y = chop_forward(Probe(), x)
        assert torch.equal(x, y)
        assert len(seen) == n and set(seen) == {(t, t)}


def test_trunk_runs_with_stub_heads_and_psnr_helper():
    from dagl_amd.net import RR, psnr, set12_protocol_noise
    net = RR(ce_cls=_StubCE).eval()
    x = torch.rand(1, 1, 24, 20)
    with torch.no_grad():
        y = net(x)
    assert y.shape == x.shape
    clean = torch.full((1, 1, 8, 8), 0.5)
    assert abs(psnr(clean + 0.1, clean) - 20.0) < 1e-4          # fp32 image, 0.1 is not exact
    n1, n2 = set12_protocol_noise(clean), set12_protocol_noise(clean)
    assert torch.equal(n1, n2)


def test_batched_chop_equals_sequential_chop():
    from dagl_amd.net import chop_forward, chop_forward_batched
    net = nn.Conv2d(1, 1, 5, padding=2).eval()
    for shape in ((1, 1, 256, 256), (1, 1, 203, 310), (2, 1, 128, 160)):
        x = torch.rand(*shape)
        with torch.no_grad():
            a, b = chop_forward(net, x), chop_forward_batched(net, x, max_batch=16)
        assert torch.allclose(a, b, atol=1e-6, rtol=0)


def test_forward_x8_is_exact_for_an_equivariant_network():
    """A pointwise network commutes with flips / transposes: the ensemble must return its plain output."""
    from dagl_amd.net import forward_x8
    f = lambda t: t * 2.0 + 1.0
    x = torch.rand(2, 1, 12, 20)
    assert torch.allclose(forward_x8(f, x), f(x), atol=1e-6)
    seen = []
    forward_x8(lambda t: (seen.append(tuple(t.shape[-2:])), t)[1], x)
    assert len(seen) == 8 and seen.count((12, 20)) == 4 and seen.count((20, 12)) == 4


def test_rr_load_state_dict_follows_the_reference_rule_for_tail_keys():
    """``RR.load_state_dict`` is the reference's own loader (DN_Gray/model/dagl.py:56-73), not torch's: a gray
    checkpoint loads into an ``n_colors = 3`` network (the fine-tune route: the ``tail.*`` shapes differ and are
    skipped, ``head.*`` is NOT tolerated), unknown ``tail`` keys pass even under ``strict``, other unknown keys raise
    ``KeyError`` only under ``strict``, keys the checkpoint lacks never raise, the return value is ``None``."""
    import pytest
    from dagl_amd.net import RR, seeded_state_dict
    gray = RR(n_resblocks=2, n_colors=1, ce_cls=_StubCE)
    ckpt = seeded_state_dict(gray.state_dict(), 3)
    color = RR(n_resblocks=2, n_colors=3, ce_cls=_StubCE)
    before = {k: v.clone() for k, v in color.state_dict().items()}

    # head.0.weight is [64,1,3,3] in the checkpoint and [64,3,3,3] here: copy_ broadcasts it (as in the reference);
    # tail.0.weight [1,64,3,3] -> [3,64,3,3] broadcasts too; tail.0.bias [1] -> [3] as well.  A shape that cannot be
    # broadcast is what the rule is about:
    bad = dict(ckpt)
    bad["tail.0.weight"] = torch.zeros(5, 64, 3, 3)
    bad["tail.9.extra"] = torch.zeros(2)                     # unknown, but a tail key: tolerated under strict
    assert color.load_state_dict(bad, strict=True) is None
    after = color.state_dict()
    assert torch.equal(after["tail.0.weight"], before["tail.0.weight"])          # skipped, left as it was
    assert torch.equal(after["body.0.body.0.weight"], ckpt["body.0.body.0.weight"])
    assert torch.equal(after["head.0.weight"], ckpt["head.0.weight"].expand(64, 3, 3, 3))

    worse = dict(ckpt)
    worse["head.0.weight"] = torch.zeros(5, 1, 3, 3)          # a mismatch anywhere else raises
    with pytest.raises(RuntimeError, match="head.0.weight"):
        color.load_state_dict(worse)
    extra = dict(ckpt)
    extra["body.77.weight"] = torch.zeros(1)
    with pytest.raises(KeyError, match="body.77.weight"):
        color.load_state_dict(extra, strict=True)
    assert color.load_state_dict(extra, strict=False) is None
    partial = {k: v for k, v in ckpt.items() if not k.startswith("body.1.")}    # missing keys: never an error
    assert color.load_state_dict(partial, strict=True) is None
    assert gray.load_state_dict({k: nn.Parameter(v) for k, v in ckpt.items()}) is None      # Parameters are accepted
