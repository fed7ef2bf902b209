#!/usr/bin/env python3
"""Run ON THE GPU BOX: a briefly trained checkpoint for the quality leg (no trained weights ship with the reference).

    python tools/train_quality_ckpt.py [--steps 400] [--out gpurun_out/quality_ckpt_fp16.npz]

RR (n_colors = 1, 12 CE heads in the shipped ``adaptive`` mode) from the regenerable seeded init (numpy PCG64 seed 7), trained
with the DN_Gray step (dagl_amd.train.TrainStep: sigma = 50 noise drawn on the device, MSE(sum)/(2B), Adam) on random 64 x 64
crops of the twelve Set12 images (tests/golden/set12.npz) -- the test images themselves: the point is weights that mean
something (PSNR above the noisy input, masks that have moved away from their initialisation), not generalisation.  The
state_dict is written as float16 arrays: BOTH sides of the comparison (the reference forward on the CPU in the build
container, tests/golden/make_set12_psnr.py --ckpt, and the HIP path) load exactly those rounded values.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--crop", type=int, default=64)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "quality_ckpt_fp16.npz"))
    args = ap.parse_args()
    from dagl_amd.ce import CE
    from dagl_amd.net import RR, seeded_state_dict
    from dagl_amd.train import TrainOptions, TrainStep, freeze_unused, make_optimizer
    dev = torch.device("cuda:0")
    net = RR(n_colors=1)
    net.load_state_dict(seeded_state_dict(net.state_dict(), 7), strict=True)
    net = net.to(dev)
    heads = [m for m in net.modules() if isinstance(m, CE)]
    freeze_unused(net)
    opt = TrainOptions(task="dn_gray", lr=args.lr, noise_sigma=50.0)
    step = TrainStep(net, make_optimizer(net, opt), opt, generator=torch.Generator(device=dev).manual_seed(11))
    imgs = np.load(os.path.join(REPO, "tests", "golden", "set12.npz"))
    clean = [torch.from_numpy(imgs[k].astype(np.float32) / 255.0) for k in sorted(imgs.files)]
    rng = np.random.default_rng(5)
    log, t0 = [], time.time()
    for it in range(args.steps):
        crops = []
        for _ in range(args.batch):
            im = clean[int(rng.integers(len(clean)))]
            y, x = int(rng.integers(im.shape[0] - args.crop + 1)), int(rng.integers(im.shape[1] - args.crop + 1))
            c = im[y:y + args.crop, x:x + args.crop]
            if rng.integers(2):
                c = c.flip(1)
            crops.append(c)
        hr = torch.stack(crops)[:, None].to(dev)
        loss, ps = step(hr)
        if it % 20 == 0 or it == args.steps - 1:
            dens = []
            for m in heads:
                i = m.last_info or {}
                if i.get("total_edges", -1) >= 0:
                    dens.append(i["total_edges"] / (args.batch * (args.crop // 4) ** 2 * args.crop ** 2))
            log.append(dict(step=it, loss=float(loss), psnr=float(ps), mask_density=float(np.mean(dens)) if dens else None,
                            paths=sorted({(m.last_info or {}).get("path") for m in heads}, key=str)))
            print(log[-1], f"{time.time() - t0:.1f}s", flush=True)
    net.eval()
    sd = {k: v.detach().cpu().numpy().astype(np.float16) for k, v in net.state_dict().items()}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.savez_compressed(args.out, **sd)
    json.dump(dict(steps=args.steps, batch=args.batch, crop=args.crop, lr=args.lr, log=log,
                   seconds=time.time() - t0), open(args.out[:-4] + "_log.json", "w"), indent=1)
    print("wrote", args.out, os.path.getsize(args.out) / 2 ** 20, "MiB")


if __name__ == "__main__":
    main()
