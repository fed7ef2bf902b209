#!/usr/bin/env python3
"""Golden vectors for the callers either side of the block, minted by running the REFERENCE itself (CPU, build container):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stage_x8.py

  ces_stage_48x48.npz   one CES stage -- conv1x1(cat(c1_1(x)..c1_4(x))) + x, DN_Gray/model/dagl.py:114 -- and the whole
                        CES.forward (:112-119, three stages + the two ResBlock groups) of the reference modules, with sparse
                        thr/bias heads so that the fused stage launch set (dagl_ces_stage_forward) is what gets compared
  x8_protocol.npz       the reference's self-ensemble: test_x8 (DN_Gray/model/__init__.py:53-62) on a cheap seeded conv
                        (bit-exact pin of the transform order, CPU-checkable) and on the reference RR; forward_chop with
                        ensemble=True (:179-231, test_x8 per leaf) on a 96x104 image; the Demosaic / DN_Real tilings
                        (shave 12; min_size 70000, Demosaic/model/__init__.py:188, DN_Real/model/__init__.py:135,144)
                        on a cheap seeded conv
Only outputs are stored; inputs and weights are regenerated from numpy PCG64 seeds (dagl_amd.synth / dagl_amd.net).
"""
from __future__ import annotations

import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"

from dagl_amd.net import seeded_state_dict  # noqa: E402
from dagl_amd.synth import make_ce_params, make_features  # noqa: E402


def ref_pkg(task):
    for m in [m for m in sys.modules if m == "model" or m.startswith("model.")]:
        del sys.modules[m]
    root = os.path.join(REF, task)
    sys.path.insert(0, root)
    try:
        import model as pkg
        from model import dagl
    finally:
        sys.path.remove(root)
    return pkg, dagl


def ces_state(template, seed=31, gain=1.65):
    """CES weights: heads from make_ce_params(seed + index, 'sparse'), everything else from seeded_state_dict."""
    sd = seeded_state_dict(template, seed)
    idx = 0
    for s in (1, 2, 3):
        for h in (1, 2, 3, 4):
            p = make_ce_params(seed + idx, variant="sparse", sparse_gain=gain)
            for n, a in p.items():
                sd[f"c{s}_{h}.{n}"] = torch.from_numpy(a)
            idx += 1
    return sd


def cheap_conv(c, seed):
    m = torch.nn.Conv2d(c, c, 5, padding=2)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.1)
        m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return m


def bare_model(pkg, net, ensemble):
    m = pkg.Model.__new__(pkg.Model)
    torch.nn.Module.__init__(m)
    m.scale, m.idx_scale, m.n_GPUs, m.ensemble = [1], 0, 1, ensemble
    m.model = net
    return m


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    pkg, dagl = ref_pkg("DN_Gray")

    # ---- CES stage / whole CES -------------------------------------------------------------------------------------
    ces = dagl.CES(in_channels=64).eval()
    ces.load_state_dict(ces_state(ces.state_dict()), strict=True)
    x = torch.from_numpy(make_features(31, 1, 64, 48, 48)) * 0.5
    with torch.no_grad():
        stage1 = ces.c1_c(torch.cat([ces.c1_1(x), ces.c1_2(x), ces.c1_3(x), ces.c1_4(x)], dim=1)) + x      # dagl.py:114
        whole = ces(x)
        degs = []
        for hd in (ces.c1_1, ces.c1_2, ces.c1_3, ces.c1_4):
            b1 = hd.g(x)
            q, _ = dagl.extract_image_patches(b1, [7, 7], [4, 4], [1, 1], padding="same")
            kx, _ = dagl.extract_image_patches(b1, [7, 7], [1, 1], [1, 1], padding="same")
            b4, _ = dagl.same_padding(x, [7, 7], [4, 4], [1, 1])
            S = hd.fc1(q[0].t()) @ hd.fc2(kx[0].t()).t()
            m = torch.relu(S - S.mean(1, keepdim=True) * hd.thr_conv(b4).view(-1, 1) + hd.bias_conv(b4).view(-1, 1))
            degs.append((m != 0).sum(1).to(torch.int32))
    degs = torch.stack(degs)
    print("CES stage: |out|max", float(stage1.abs().max()), "head degrees mean/max", float(degs.float().mean()), int(degs.max()))
    np.savez_compressed(os.path.join(HERE, "ces_stage_48x48.npz"),
                        stage1=stage1[0, :, ::2, ::2].numpy().astype(np.float32),
                        whole=whole[0, :, 1::2, ::2].numpy().astype(np.float32), deg=degs.numpy(),
                        meta=json.dumps(dict(seed=31, gain=1.65, H=48, W=48, scale=0.5, torch=torch.__version__)))

    # ---- x8 protocol ---------------------------------------------------------------------------------------------------
    out = {}
    conv1 = cheap_conv(1, 5)
    g = torch.Generator().manual_seed(9)
    xs = torch.rand(2, 1, 37, 52, generator=g)
    with torch.no_grad():
        out["x8_conv"] = pkg.test_x8(conv1, xs).numpy()
    args = SimpleNamespace(n_resblocks=16, n_feats=64, n_colors=1, res_scale=1, rgb_range=1.0)
    net = dagl.RR(args).eval()
    net.load_state_dict(seeded_state_dict(net.state_dict(), 7), strict=True)
    xr = torch.rand(1, 1, 40, 44, generator=g)
    with torch.no_grad():
        out["x8_rr"] = pkg.test_x8(net, xr).numpy()
        print("x8 on RR done", flush=True)
        xc = torch.rand(1, 1, 96, 104, generator=g)
        out["chop_x8_rr"] = bare_model(pkg, net, True).forward_chop(xc).numpy()
        print("forward_chop(ensemble) on RR done", flush=True)
        out["chop_x8_conv"] = bare_model(pkg, conv1, True).forward_chop(torch.rand(1, 1, 203, 310, generator=g)).numpy()[..., ::3, ::5]
    # the other forks' tilings (two levels of recursion each), cheap conv with 3 channels; every (3rd, 5th) pixel is stored
    conv3 = cheap_conv(3, 6)
    xd = torch.rand(1, 3, 530, 610, generator=g)
    for task, key in (("Demosaic", "chop_demosaic_conv"), ("DN_Real", "chop_real_conv")):
        p2, _ = ref_pkg(task)
        with torch.no_grad():
            out[key] = bare_model(p2, conv3, False).forward_chop(xd).numpy()[..., ::3, ::5]
    np.savez_compressed(os.path.join(HERE, "x8_protocol.npz"), **{k: v.astype(np.float32) for k, v in out.items()},
                        meta=json.dumps(dict(rr_seed=7, conv1_seed=5, conv3_seed=6, rng_seed=9, torch=torch.__version__,
                                             draws="xs[2,1,37,52], xr[1,1,40,44], xc[1,1,96,104], chop conv [1,1,203,310], "
                                                   "xd[1,3,530,610] from one torch.Generator(9), in this order; the three "
                                                   "cheap-conv tilings are stored at [..., ::3, ::5]")))
    print("written")


if __name__ == "__main__":
    main()
