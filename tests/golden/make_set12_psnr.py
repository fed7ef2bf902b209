#!/usr/bin/env python3
"""Set12 sigma=50 reference run (CPU, build container only): the quality leg of BASELINE.json's metric.

    PYTHONDONTWRITEBYTECODE=1 nohup python tests/golden/make_set12_psnr.py > /tmp/set12.log 2>&1 &

No trained checkpoint ships with the reference (.MISSING_LARGE_BLOBS), so the run uses a regenerable stand-in
(``dagl_amd.net.seeded_state_dict``, numpy PCG64 seed 7) loaded into the REFERENCE network
(/root/reference/DN_Gray/model/dagl.py ``RR``, 12 ``CE`` heads) and follows the reference test protocol
(DN_Gray/test.py:49-66): clean image / 255, ``torch.manual_seed(1)`` Gaussian noise sigma = 50/255, tiled inference
(``Model.forward_chop`` geometry, no self-ensemble), clamp to [0,1], PSNR against the clean image.

Outputs (data only):
    tests/golden/set12.npz            the 12 Set12 test images as uint8 arrays (test data of the reference)
    tests/golden/set12_psnr_ref.json  per image: noisy PSNR, output PSNR of the reference forward, output mean/std
    tests/golden/set12_out_sub.npz    every 8th pixel of each reference output (fp32) for a direct comparison
Also checks that ``dagl_amd.net.chop_forward`` reproduces the reference's own ``forward_chop`` tiling.
"""
from __future__ import annotations

import glob
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference/DN_Gray"
SEED = 7


def reference_modules():
    sys.path.insert(0, REF)
    import model as ref_model_pkg          # DN_Gray/model/__init__.py (Model, forward_chop)
    from model import dagl as ref_dagl      # RR / CES / CE
    sys.path.remove(REF)
    return ref_model_pkg, ref_dagl


def main():
    from dagl_amd.net import chop_forward, psnr, seeded_state_dict, set12_protocol_noise, sparse_heads_state_dict
    sparse = "--sparse" in sys.argv       # second regime: sparse adaptive masks (sparse_heads_state_dict, gain 1.65)
    # third: a briefly TRAINED checkpoint (tools/train_quality_ckpt.py on the GPU box: 400 DN_Gray steps from the seeded
    # init on Set12 crops, stored as float16) -- weights that mean something: output PSNR ~10 dB above the noisy input
    trained = "--ckpt" in sys.argv
    suffix = "_sparse" if sparse else ("_trained" if trained else "")
    torch.set_num_threads(os.cpu_count() or 1)
    ref_pkg, ref_dagl = reference_modules()
    args = SimpleNamespace(n_resblocks=16, n_feats=64, n_colors=1, res_scale=1, rgb_range=1.0)
    net = ref_dagl.RR(args).eval()
    sd = seeded_state_dict(net.state_dict(), SEED)
    if sparse:
        sd = sparse_heads_state_dict(sd, SEED + 100, 1.65)
    if trained:
        z = np.load(os.path.join(HERE, "quality_ckpt_fp16.npz"))
        sd = {k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}
    net.load_state_dict(sd, strict=True)

    # 1. tiling check: my chop_forward == the reference's Model.forward_chop (cheap stand-in network)
    m = ref_pkg.Model.__new__(ref_pkg.Model)
    torch.nn.Module.__init__(m)
    m.scale, m.idx_scale, m.n_GPUs, m.ensemble = [1], 0, 1, False
    m.model = torch.nn.Conv2d(1, 1, 5, padding=2)
    xt = torch.rand(1, 1, 203, 310)
    with torch.no_grad():
        a = m.forward_chop(xt)
        b = chop_forward(m.model, xt)
    assert torch.equal(a, b), "chop_forward deviates from the reference forward_chop"
    print("chop_forward == reference forward_chop (203x310 probe)", flush=True)

    files = sorted(glob.glob(os.path.join(REF, "testsets", "Set12", "*.png")))
    assert len(files) == 12
    images = {os.path.basename(f)[:-4]: np.asarray(Image.open(f).convert("L"), dtype=np.uint8) for f in files}
    if trained:
        pass                                                              # all twelve images
    elif not sparse:
        np.savez_compressed(os.path.join(HERE, "set12.npz"), **{f"img_{k}": v for k, v in images.items()})
    else:
        images = {k: images[k] for k in ("01", "05", "09")}             # three images (two 256^2, one 512^2) suffice here

    result, subs = {}, {}
    for name, img in images.items():
        clean = torch.from_numpy(img.astype(np.float32) / 255.0)[None, None]
        noisy = set12_protocol_noise(clean, 50.0, 1.0)
        t0 = time.time()
        with torch.no_grad():
            out = torch.clamp(chop_forward(net, noisy), 0.0, 1.0)
        dt = time.time() - t0
        result[name] = dict(shape=list(img.shape), psnr_noisy=psnr(noisy, clean), psnr_out=psnr(out, clean),
                            out_mean=float(out.mean()), out_std=float(out.std()), seconds=dt)
        subs[f"out_{name}"] = out[0, 0, ::8, ::8].numpy().astype(np.float32)
        print(name, result[name], flush=True)
        json.dump(dict(seed=SEED, sigma=50, protocol="DN_Gray/test.py:49-66, forward_chop without ensemble",
                       heads=("sparse_heads_state_dict(seed + 100, gain 1.65)" if sparse else
                              ("tests/golden/quality_ckpt_fp16.npz (tools/train_quality_ckpt.py, float16 values)" if trained
                               else "seeded_state_dict")),
                       torch=torch.__version__, images=result),
                  open(os.path.join(HERE, f"set12_psnr_ref{suffix}.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(HERE, f"set12_out_sub{suffix}.npz"), **subs)
    print("mean output PSNR", np.mean([r["psnr_out"] for r in result.values()]))


if __name__ == "__main__":
    main()
