"""The quality leg of BASELINE.json's metric: PSNR on Set12 sigma=50 of the full network with the HIP block vs the
same network with the reference block, identical (regenerable) weights, reference test protocol.  The reference
numbers were produced on CPU in the build container by tests/golden/make_set12_psnr.py; bar: |dPSNR| <= 0.02 dB."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN_DIR, normwise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    from dagl_amd.net import RR, seeded_state_dict
    assert torch.cuda.is_available()
    ref = json.load(open(os.path.join(GOLDEN_DIR, "set12_psnr_ref.json")))
    m = RR().eval()
    m.load_state_dict(seeded_state_dict(m.state_dict(), ref["seed"]), strict=True)
    return m.to("cuda:0"), ref


@pytest.mark.parametrize("name", ["01", "05", "07", "09", "12"])
def test_set12_psnr_delta(net, name):
    from dagl_amd.net import chop_forward, chop_forward_batched, psnr, set12_protocol_noise
    model, ref = net
    imgs = np.load(os.path.join(GOLDEN_DIR, "set12.npz"))
    subs = np.load(os.path.join(GOLDEN_DIR, "set12_out_sub.npz"))
    clean = torch.from_numpy(imgs[f"img_{name}"].astype(np.float32) / 255.0)[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0)
    r = ref["images"][name]
    assert abs(psnr(noisy, clean) - r["psnr_noisy"]) < 1e-6          # same noise as the reference run
    with torch.no_grad():
        out = torch.clamp(chop_forward(model, noisy.to("cuda:0")), 0.0, 1.0).cpu()
    d = psnr(out, clean) - r["psnr_out"]
    print(f"Set12/{name}: reference {r['psnr_out']:.4f} dB, HIP {psnr(out, clean):.4f} dB, delta {d:+.5f} dB")
    assert abs(d) <= 0.02
    # and the images themselves agree (every 8th pixel of the reference output is committed)
    assert normwise(out[0, 0, ::8, ::8].numpy(), subs[f"out_{name}"]) <= 2e-3
    # the batched tile driver gives the same image (tiles are independent)
    with torch.no_grad():
        outb = torch.clamp(chop_forward_batched(model, noisy.to("cuda:0")), 0.0, 1.0).cpu()
    assert abs(psnr(outb, clean) - r["psnr_out"]) <= 0.02
    assert normwise(outb.numpy(), out.numpy()) <= 2e-3
