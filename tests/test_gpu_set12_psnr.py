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


@pytest.fixture(scope="module")
def net_sparse():
    """Second regime (tests/golden/make_set12_psnr.py --sparse): the 12 heads' thr / bias heads keep a handful of neighbours
    per query (sparse_heads_state_dict, gain 1.65: long-tailed degrees) -- neighbour lists, the per-query overflow rows and
    the stage-level launch set are what serves the reference tiling here, not the streamed dense formulation."""
    from dagl_amd.net import RR, seeded_state_dict, sparse_heads_state_dict
    ref = json.load(open(os.path.join(GOLDEN_DIR, "set12_psnr_ref_sparse.json")))
    m = RR().eval()
    m.load_state_dict(sparse_heads_state_dict(seeded_state_dict(m.state_dict(), ref["seed"]), ref["seed"] + 100, 1.65), strict=True)
    return m.to("cuda:0"), ref


@pytest.mark.parametrize("name", ["01", "05", "09"])
def test_set12_psnr_delta_sparse_masks(net_sparse, name):
    from dagl_amd.ce import CE
    from dagl_amd.net import chop_forward, chop_forward_batched, psnr, set12_protocol_noise
    model, ref = net_sparse
    imgs = np.load(os.path.join(GOLDEN_DIR, "set12.npz"))
    subs = np.load(os.path.join(GOLDEN_DIR, "set12_out_sub_sparse.npz"))
    clean = torch.from_numpy(imgs[f"img_{name}"].astype(np.float32) / 255.0)[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0)
    r = ref["images"][name]
    with torch.no_grad():
        out = torch.clamp(chop_forward(model, noisy.to("cuda:0")), 0.0, 1.0).cpu()
    d = psnr(out, clean) - r["psnr_out"]
    print(f"Set12/{name} (sparse masks): reference {r['psnr_out']:.4f} dB, HIP {psnr(out, clean):.4f} dB, delta {d:+.5f} dB")
    assert abs(d) <= 0.02
    assert normwise(out[0, 0, ::8, ::8].numpy(), subs[f"out_{name}"]) <= 2e-3
    paths = {m.last_info["path"] for m in model.modules() if isinstance(m, CE) and m.last_info}
    print(f"Set12/{name} (sparse masks): serving paths {sorted(paths)}")
    assert paths & {1, 2, 3}, paths                  # neighbour lists (with or without overflow rows) serve heads here
    with torch.no_grad():
        outb = torch.clamp(chop_forward_batched(model, noisy.to("cuda:0")), 0.0, 1.0).cpu()
    assert abs(psnr(outb, clean) - r["psnr_out"]) <= 0.02


@pytest.fixture(scope="module")
def net_trained():
    """Third regime: weights that mean something.  tests/golden/quality_ckpt_fp16.npz = RR after 400 DN_Gray training steps on
    the HIP path (tools/train_quality_ckpt.py, float16 values); the reference forward with exactly those values ran on the CPU
    in the build container (tests/golden/make_set12_psnr.py --ckpt): 25.3 dB mean against 14.1 dB of the noisy input."""
    from dagl_amd.net import RR
    ref = json.load(open(os.path.join(GOLDEN_DIR, "set12_psnr_ref_trained.json")))
    z = np.load(os.path.join(GOLDEN_DIR, "quality_ckpt_fp16.npz"))
    m = RR().eval()
    m.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True)
    return m.to("cuda:0"), ref


@pytest.mark.parametrize("name", ["%02d" % i for i in range(1, 13)])
def test_set12_psnr_delta_trained_checkpoint(net_trained, name):
    from dagl_amd.ce import CE
    from dagl_amd.net import chop_forward_batched, psnr, set12_protocol_noise
    model, ref = net_trained
    imgs = np.load(os.path.join(GOLDEN_DIR, "set12.npz"))
    subs = np.load(os.path.join(GOLDEN_DIR, "set12_out_sub_trained.npz"))
    clean = torch.from_numpy(imgs[f"img_{name}"].astype(np.float32) / 255.0)[None, None]
    noisy = set12_protocol_noise(clean, 50.0, 1.0)
    r = ref["images"][name]
    with torch.no_grad():
        out = torch.clamp(chop_forward_batched(model, noisy.to("cuda:0")), 0.0, 1.0).cpu()
    p = psnr(out, clean)
    dens = []
    for m in model.modules():
        if isinstance(m, CE) and m.last_info and m.last_info.get("total_edges", -1) >= 0:
            dens.append(m.last_info["total_edges"])
    print(f"Set12/{name} (trained checkpoint): noisy {r['psnr_noisy']:.2f} dB, reference {r['psnr_out']:.4f} dB, HIP {p:.4f} dB, "
          f"delta {p - r['psnr_out']:+.6f} dB; paths {sorted({m.last_info['path'] for m in model.modules() if isinstance(m, CE) and m.last_info})}")
    assert p > r["psnr_noisy"] + 8.0                                     # the network denoises
    assert abs(p - r["psnr_out"]) <= 0.02
    assert normwise(out[0, 0, ::8, ::8].numpy(), subs[f"out_{name}"]) <= 2e-3
