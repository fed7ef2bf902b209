#!/usr/bin/env python3
"""Run ON THE GPU BOX: the training step (RR, 12 heads, top-k 8, [8,3,128,128]) under variants of the TRUNK's setup:
   python tools/time_train_variants.py [channels_last] [find_fast]"""
import os, sys, time
if "find_fast" in sys.argv:
    os.environ["MIOPEN_FIND_MODE"] = "FAST"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagl_amd.ce import CE
from dagl_amd.net import RR, seeded_state_dict
from dagl_amd.train import TrainOptions, TrainStep, freeze_unused, make_optimizer
dev = torch.device("cuda:0")
net = RR(n_colors=3)
net.load_state_dict(seeded_state_dict(net.state_dict(), 7), strict=True)
for m in net.modules():
    if isinstance(m, CE):
        m.select_mode, m.select_k = "topk", 8
net = net.to(dev)
if "channels_last" in sys.argv:
    net = net.to(memory_format=torch.channels_last)
freeze_unused(net)
opt = TrainOptions(task="dn_real", lr=1e-4)
step = TrainStep(net, make_optimizer(net, opt), opt, generator=torch.Generator(device=dev).manual_seed(300))
hr = torch.rand(8, 3, 128, 128, generator=torch.Generator().manual_seed(200)).to(dev)
if "channels_last" in sys.argv:
    hr = hr.contiguous(memory_format=torch.channels_last)
for _ in range(4):
    step(hr)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    l, _ = step(hr)
torch.cuda.synchronize()
print(sys.argv[1:], f"{(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per step, loss {float(l):.2f}")
