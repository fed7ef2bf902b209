"""Training-step glue on CPU: the task losses of the reference trainers, the loss-spike guard, the frozen never-applied
``W`` convs, and a world-size-2 gloo DDP step that must equal the single-process step on the whole batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

from dagl_amd.train import (TrainOptions, TrainStep, batch_psnr_device, freeze_unused, make_optimizer, make_scheduler,
                            task_loss, wrap_ddp)


class _StandInHead(nn.Module):
    """CPU stand-in with the block's parameter surface that matters here (a used conv + the never-applied ``W``)."""

    def __init__(self, in_channels=64):
        super().__init__()
        self.g = nn.Conv2d(in_channels, 16, 3, padding=1)
        self.W = nn.Conv2d(16, in_channels, 1)

    def forward(self, x):
        return self.g(x)


def _net(seed=0):
    from dagl_amd.net import RR
    torch.manual_seed(seed)
    return RR(n_resblocks=2, n_feats=64, n_colors=1, ce_cls=_StandInHead)


def test_task_losses_are_the_trainers_hard_coded_ones():
    torch.manual_seed(1)
    sr, hr = torch.rand(4, 1, 8, 8), torch.rand(4, 1, 8, 8)
    assert torch.allclose(task_loss(sr, hr, "dn_gray"), nn.MSELoss(reduction="sum")(sr, hr) / (4 * 2))
    assert torch.allclose(task_loss(sr, hr, "car"), nn.MSELoss()(sr, hr) * 255 ** 2)
    assert torch.allclose(task_loss(sr, hr, "demosaic"), nn.L1Loss(reduction="sum")(sr, hr) / (4 * 2))
    from dagl_amd.metrics import batch_psnr
    assert abs(float(batch_psnr_device(sr, hr)) - batch_psnr(sr, hr, 1.0)) < 1e-4


def test_unused_w_convs_are_frozen_and_adam_defaults_follow_option_py():
    net = _net()
    n = freeze_unused(net)
    assert n == 12 * (16 * 64 + 64)
    frozen = [k for k, p in net.named_parameters() if not p.requires_grad]
    assert all(".W." in k or k.startswith("add_mean") for k in frozen) and sum(".W." in k for k in frozen) == 24
    opt = TrainOptions()
    adam = make_optimizer(net, opt)
    g = adam.param_groups[0]
    assert (g["lr"], g["betas"], g["eps"], g["weight_decay"]) == (4e-4, (0.9, 0.999), 1e-8, 0.0)
    sch = make_scheduler(adam, opt)
    assert (sch.step_size, sch.gamma) == (200, 0.5)


def test_train_step_reduces_loss_and_spike_guard_skips():
    net = _net()
    freeze_unused(net)
    opt = TrainOptions(lr=1e-3)
    step = TrainStep(net, make_optimizer(net, opt), opt, generator=torch.Generator().manual_seed(3))
    hr = torch.rand(4, 1, 16, 16)
    losses = [float(step(hr)[0]) for _ in range(8)]
    assert losses[-1] < losses[0]
    before = [p.detach().clone() for p in net.parameters()]
    step.error_last = 1e-12                                   # any loss is now a "spike"
    step(hr, check_spike=True)
    assert step.skipped == 1
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))


def test_train_step_does_not_apply_a_non_finite_loss():
    """A forward that left the block's split-fp16 range is NaN-filled (never wrong numbers): the step's default guard -- the
    reference's own ``loss < skip_threshold * error_last`` test, which is false for NaN -- must keep Adam from writing it into
    the weights (round-3 advisory)."""
    net = _net()
    freeze_unused(net)
    opt = TrainOptions(lr=1e-3)
    step = TrainStep(net, make_optimizer(net, opt), opt, generator=torch.Generator().manual_seed(5))
    hr = torch.rand(2, 1, 16, 16)
    step(hr)
    before = [p.detach().clone() for p in net.parameters()]
    bad = hr.clone(); bad[0, 0, 3, 3] = float("nan")
    loss, _ = step(bad)                                        # defaults: the guard is on
    assert not torch.isfinite(loss) and step.skipped == 1
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
    step(hr)                                                   # and training goes on
    assert all(torch.isfinite(p).all() for p in net.parameters())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        net = _net(seed=0)                                    # same initial weights on every rank
        ddp = wrap_ddp(net, torch.device("cpu"))
        opt = TrainOptions(lr=1e-3)
        step = TrainStep(ddp, make_optimizer(ddp, opt), opt)
        g = torch.Generator().manual_seed(11)
        hr = torch.rand(4, 1, 12, 12, generator=g)
        lr = hr + 0.1 * torch.randn(4, 1, 12, 12, generator=g)
        lo, hi = rank * 2, rank * 2 + 2                       # image-batch data parallel: 2 crops per rank
        step(hr[lo:hi], lr[lo:hi])
        flat = torch.cat([p.detach().flatten() for p in net.parameters()])
        q.put((rank, flat.numpy()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_ddp_step_equals_single_process_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, whole batch: sum-loss / (2B) -> the average of the two ranks' gradients is the same quantity
    torch.set_num_threads(2)
    net = _net(seed=0)
    freeze_unused(net)
    opt = TrainOptions(lr=1e-3)
    step = TrainStep(net, make_optimizer(net, opt), opt)
    g = torch.Generator().manual_seed(11)
    hr = torch.rand(4, 1, 12, 12, generator=g)
    lr = hr + 0.1 * torch.randn(4, 1, 12, 12, generator=g)
    step(hr, lr)
    want = torch.cat([p.detach().flatten() for p in net.parameters()]).numpy()
    import numpy as np
    assert np.array_equal(got[0], got[1])                     # replicas stay in lock-step
    assert np.allclose(got[0], want, rtol=1e-4, atol=1e-6)
