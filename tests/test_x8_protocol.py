"""The reference's test-time protocol around the network, pinned on outputs of the reference itself
(tests/golden/x8_protocol.npz, minted by make_golden_stage_x8.py): self-ensemble ``test_x8`` (transform order and all),
``forward_chop`` with ``--ensemble``, and the Demosaic / DN_Real tilings (shave 12, min_size 70000).  The cheap-conv cases
run on the CPU and must be bit-identical; the RR cases need the HIP block (-m gpu)."""
import json
import os

import numpy as np
import pytest
import torch

from dagl_amd.net import CHOP_PRESETS, chop_forward, chop_forward_batched, forward_x8
from tests.helpers import GOLDEN_DIR, normwise


def _fix():
    z = np.load(os.path.join(GOLDEN_DIR, "x8_protocol.npz"), allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def _cheap_conv(c, seed):
    m = torch.nn.Conv2d(c, c, 5, padding=2)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.1)
        m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return m


def _draws(meta):
    g = torch.Generator().manual_seed(meta["rng_seed"])
    return [torch.rand(*s, generator=g) for s in ((2, 1, 37, 52), (1, 1, 40, 44), (1, 1, 96, 104), (1, 1, 203, 310),
                                                  (1, 3, 530, 610))]


def test_x8_and_tilings_equal_the_reference_bit_for_bit_on_a_cheap_conv():
    z, meta = _fix()
    xs, _, _, xchop, xd = _draws(meta)
    conv1, conv3 = _cheap_conv(1, meta["conv1_seed"]), _cheap_conv(3, meta["conv3_seed"])
    with torch.no_grad():
        assert np.array_equal(forward_x8(conv1, xs).numpy(), z["x8_conv"])
        got = chop_forward(conv1, xchop, ensemble=True)
        assert np.array_equal(got.numpy()[..., ::3, ::5], z["chop_x8_conv"])
        got_b = chop_forward_batched(conv1, xchop, ensemble=True)
        assert normwise(got_b.numpy(), got.numpy()) <= 1e-6          # batched convs may pick another kernel: not bitwise
        for task, key in (("demosaic", "chop_demosaic_conv"), ("dn_real", "chop_real_conv")):
            ms, sh = CHOP_PRESETS[task]
            got = chop_forward(conv3, xd, min_size=ms, shave_size_max=sh)
            assert np.array_equal(got.numpy()[..., ::3, ::5], z[key]), task
    assert CHOP_PRESETS["dn_gray"] == (10000, 24) and CHOP_PRESETS["car"] == (10000, 24)


def _rr(dev, seed):
    from dagl_amd.net import RR, seeded_state_dict
    net = RR().eval()
    net.load_state_dict(seeded_state_dict(net.state_dict(), seed), strict=True)
    return net.to(dev)


@pytest.mark.gpu
def test_self_ensemble_of_the_whole_network_matches_the_reference():
    z, meta = _fix()
    dev = torch.device("cuda:0")
    _, xr, xc, _, _ = _draws(meta)
    net = _rr(dev, meta["rr_seed"])
    with torch.no_grad():
        got = forward_x8(net, xr.to(dev)).cpu().numpy()
        assert normwise(got, z["x8_rr"]) <= 1e-4
        got = chop_forward(net, xc.to(dev), ensemble=True).cpu().numpy()
        assert normwise(got, z["chop_x8_rr"]) <= 1e-4
        got_b = chop_forward_batched(net, xc.to(dev), ensemble=True).cpu().numpy()
        assert normwise(got_b, z["chop_x8_rr"]) <= 1e-4
