"""Oracle parity at BASELINE.json's own sizes and configurations, through the module / C ABI:

  config 2  [1,64,256,256] fp32, top-k k=8 and adaptive at mean degree ~8: the whole dense oracle (10 s of host time)
  config 3  512x512 with bf16 feature maps: a sample of the queries against ALL keys (the full [L,N] matrix is 16 GiB),
            and the bf16-input block against the oracle fed the same bf16-rounded input
  config 4  1024x1024, adaptive AND top-16: a sample of the queries against all 1 048 576 keys
  config 5  one optimisation step of the whole RR (n_colors=3) on [8,3,128,128]; whole-network gradients against the
            fp64 oracle's autograd at a size the oracle can do; the RCCL process group on one GPU
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _params(seed, variant, gain):
    from dagl_amd.synth import make_ce_params
    return {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant=variant, sparse_gain=gain).items()}


def _module(params, mode, k):
    from dagl_amd.ce import CE
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode = mode
    if k:
        ce.select_k = k
    return ce.to(_dev()).eval()


def _debug(ce, x):
    from dagl_amd import ops
    with torch.no_grad():
        b1, b2, thr, bias = ce._prologue(x)
        return ops.ce_forward(b1.contiguous(), b2.contiguous(), thr.contiguous(), bias.contiguous(),
                              ce.fc1[0].weight, ce.fc1[0].bias, ce.fc2[0].weight, ce.fc2[0].bias,
                              mode=ce.select_mode, k=ce.select_k, debug=True)


def _agg_ckk(agg):                       # [.., 784] (kh,kw,c) -> (c,kh,kw), the reference's row order
    s = agg.shape[:-1]
    return agg.reshape(*s, 7, 7, 16).movedim(-1, -3).reshape(*s, 784)


@pytest.mark.parametrize("mode,k,variant,gain", [("topk", 8, "default", 2.0), ("adaptive", 0, "sparse", 1.8),
                                                 ("adaptive", 0, "sparse", 1.95)])
def test_config2_256x256_matches_the_dense_oracle(mode, k, variant, gain):
    """BASELINE configs[1] itself: [1,64,256,256], L = 4096 queries x N = 65536 keys, against the dense fp32 oracle."""
    from dagl_amd.synth import make_features
    from oracle.ce_oracle import ce_forward_oracle
    params = _params(41, variant, gain)
    x = torch.from_numpy(make_features(41, 1, 64, 256, 256))
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        want, st = ce_forward_oracle(x, params, mode=mode, k=k or None, stages=True)
    deg_ref = st["deg"][0].numpy().astype(np.int64)
    rowsum_ref, agg_ref = st["rowsum"][0].numpy(), st["agg"][0]
    del st
    # the oracle in fp32 IS the reference's arithmetic, rounding noise included: logits 10 S m of a few tens turn the last
    # bits of S into ~1e-4 of a softmax weight.  So the yardstick is an fp64 evaluation: the block must be within 1e-4 of
    # it, and within 1e-4 + (the fp32 oracle's own distance to fp64) of the fp32 oracle
    with torch.no_grad():
        want64 = ce_forward_oracle(x, params, mode=mode, k=k or None, dtype=torch.float64).float()
    e_ref = normwise(want.numpy(), want64.numpy())
    ce = _module(params, mode, k)
    with torch.no_grad():
        out = ce(x.to(_dev())).cpu()
    assert normwise(out.numpy(), want64.numpy()) <= TOL
    assert normwise(out.numpy(), want.numpy()) <= TOL + e_ref
    out_d, info = _debug(ce, x.to(_dev()))
    deg = info["deg"][0].cpu().numpy().astype(np.int64)
    ndiff = int((deg != deg_ref).sum())
    # (a key within rounding of its threshold may flip between two fp32 evaluations)
    assert ndiff <= max(1, int(1e-4 * deg.size)), f"{ndiff} of {deg.size} queries differ in degree"
    assert normwise(info["rowsum"][0].cpu().numpy(), rowsum_ref) <= TOL + e_ref
    assert normwise(_agg_ckk(info["agg"][0].cpu()).numpy(), agg_ref.numpy()) <= TOL + e_ref
    assert normwise(out_d.cpu().numpy(), want64.numpy()) <= TOL
    if mode == "adaptive":
        # gain 1.95 is the "mean degree ~8" regime of section 8d (7.7, maximum 890: long-tailed); gain 1.8: mean 55, maximum
        # 4578.  Either way the lists serve all but a few queries and those are redone one by one (overflow.hip): no fp32 rescan
        assert (4.0 <= deg_ref.mean() <= 16.0) if gain > 1.9 else deg_ref.mean() > 16.0, deg_ref.mean()
        assert deg_ref.max() > 256
        assert info["path"] == 3 and 0 < info["redone_queries"] <= 256, info


@pytest.mark.parametrize("H,W,mode,k,in_dtype", [(512, 512, "topk", 8, torch.bfloat16),
                                                 (1024, 1024, "adaptive_topk", 16, torch.float32)])
def test_config3_config4_query_sample_against_all_keys(H, W, mode, k, in_dtype):
    """512^2 (bf16 feature maps) and 1024^2 (adaptive AND top-16): 64 queries spread over the image, each against ALL keys,
    on the oracle; the block's degrees, softmax mass and aggregated patches for those queries must match."""
    from dagl_amd.synth import make_features
    from oracle.ce_oracle import ce_rows_oracle
    params = _params(61, "sparse", 1.7)
    x = torch.from_numpy(make_features(61 + H, 1, 64, H, W))
    if in_dtype != torch.float32:
        x = x.to(in_dtype).float()                                   # what the block sees after its boundary conversion
    L = (H // 4) * (W // 4)
    rows = torch.linspace(0, L - 1, 64).long()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ref = ce_rows_oracle(x, params, rows, mode=mode, k=k)
    ce = _module(params, mode, k)
    out_d, info = _debug(ce, x.to(_dev()))
    deg = info["deg"][0].cpu()[rows].numpy()
    assert np.array_equal(deg, ref["deg"].numpy().astype(deg.dtype))
    assert normwise(info["rowsum"][0].cpu()[rows].numpy(), ref["rowsum"].numpy()) <= TOL
    agg = _agg_ckk(info["agg"][0].cpu()[rows])
    assert normwise(agg.numpy(), ref["agg"].numpy()) <= TOL
    with torch.no_grad():
        out = ce(x.to(_dev()).to(in_dtype))
    assert out.dtype == in_dtype and torch.isfinite(out.float()).all()
    # module output (fused fp16-split prologue) vs the debug entry point (stock-conv prologue): the same fold of the same
    # aggregated patches up to the rounding of b1, plus one bf16 rounding at the boundary in config 3
    assert normwise(out.float().cpu().numpy(), out_d.cpu().numpy()) <= (2.0 ** -8 if in_dtype == torch.bfloat16 else TOL)


def test_topk500_at_256_query_sample_against_all_keys():
    """num_edge = 500 (CA_model-checkpoint.py:134-143) at 256^2: the row-wise form in two batches of 2048 queries; 64 queries
    spread over the image, each against all 65 536 keys, on the oracle."""
    import time
    from dagl_amd.synth import make_features
    from oracle.ce_oracle import ce_rows_oracle
    params = _params(61, "default", 2.0)
    x = torch.from_numpy(make_features(61, 1, 64, 256, 256))
    rows = torch.linspace(0, 4095, 64).long()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ref = ce_rows_oracle(x, params, rows, mode="topk", k=500)
    ce = _module(params, "topk", 500)
    out_d, info = _debug(ce, x.to(_dev()))
    assert info["path"] == 6 and info["max_degree"] == 500 and info["total_edges"] == 500 * 4096
    deg = info["deg"][0].cpu()[rows].numpy()
    assert np.array_equal(deg, ref["deg"].numpy().astype(deg.dtype))
    assert normwise(info["rowsum"][0].cpu()[rows].numpy(), ref["rowsum"].numpy()) <= TOL
    agg = _agg_ckk(info["agg"][0].cpu()[rows])
    err = normwise(agg.numpy(), ref["agg"].numpy())
    with torch.no_grad():
        out = ce(x.to(_dev()))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            ce(x.to(_dev()))
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"[parity] topk k=500 256x256: aggregated patches of 64 sampled queries vs the oracle, normwise {err:.2e}; {ms:.2f} ms per call")
    assert err <= TOL
    assert normwise(out.cpu().numpy(), out_d.cpu().numpy()) <= TOL


def test_adaptive_with_more_flagged_queries_than_the_redo_holds_takes_the_csr_lists():
    """512^2, adaptive at a density where ~10 % of the queries overflow their lists: more than the per-query redo holds at
    this size (512 score rows of 1 MiB), fewer than half -- the call is redone by the fp32 scan with two-pass CSR lists
    (path 1).  64 sampled queries against all 262 144 keys on the oracle."""
    from dagl_amd.synth import make_features
    from oracle.ce_oracle import ce_rows_oracle
    H = W = 512
    params = _params(61, "sparse", 1.8)
    x = torch.from_numpy(make_features(61 + H, 1, 64, H, W))
    L = (H // 4) * (W // 4)
    rows = torch.linspace(0, L - 1, 64).long()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        ref = ce_rows_oracle(x, params, rows, mode="adaptive", k=None)
        ref64 = ce_rows_oracle(x, params, rows, mode="adaptive", k=None, dtype=torch.float64)
    ce = _module(params, "adaptive", 0)
    out_d, info = _debug(ce, x.to(_dev()))
    assert info["path"] == 1 and 512 < info["redone_queries"] <= L // 2, info
    deg = info["deg"][0].cpu()[rows].numpy().astype(np.int64)
    ref_deg = ref["deg"].numpy().astype(np.int64)
    # (a key within rounding of its threshold may flip between two fp32 evaluations)
    assert np.abs(deg - ref_deg).max() <= 1 and int((deg != ref_deg).sum()) <= 2, (deg, ref_deg)
    # logits of several hundred turn the last bits of an fp32 score into ~1e-4 of a softmax weight: the yardstick is the
    # fp64 evaluation, the fp32 oracle gets its own distance to it as slack (as in the 256^2 test above)
    rowsum, agg = info["rowsum"][0].cpu()[rows].numpy(), _agg_ckk(info["agg"][0].cpu()[rows]).numpy()
    rs64, agg64 = ref64["rowsum"].float().numpy(), ref64["agg"].float().numpy()
    e_rs, e_agg = normwise(ref["rowsum"].numpy(), rs64), normwise(ref["agg"].numpy(), agg64)
    print("rowsum: block vs fp64", normwise(rowsum, rs64), "fp32 oracle vs fp64", e_rs,
          "| agg: block vs fp64", normwise(agg, agg64), "fp32 oracle vs fp64", e_agg)
    assert normwise(rowsum, rs64) <= TOL and normwise(agg, agg64) <= TOL
    assert normwise(rowsum, ref["rowsum"].numpy()) <= TOL + e_rs and normwise(agg, ref["agg"].numpy()) <= TOL + e_agg
    with torch.no_grad():
        out = ce(x.to(_dev()))
    assert normwise(out.cpu().numpy(), out_d.cpu().numpy()) <= TOL


def test_config3_bf16_feature_maps_vs_oracle_on_the_rounded_input():
    """bf16 I/O (config 3) against the ORACLE (not against the block itself): the block computes in fp32 on the bf16-rounded
    map, so oracle(x.bf16().float()) is the yardstick; the output carries one bf16 rounding (2^-8 relative)."""
    from dagl_amd.synth import make_features
    from oracle.ce_oracle import ce_forward_oracle
    params = _params(63, "default", 2.0)
    x = torch.from_numpy(make_features(63, 2, 64, 128, 128))
    xr = x.to(torch.bfloat16)
    with torch.no_grad():
        want = ce_forward_oracle(xr.float(), params, mode="topk", k=8)
    ce = _module(params, "topk", 8)
    with torch.no_grad():
        y16 = ce(xr.to(_dev()))
        y32 = ce(xr.float().to(_dev()))
    assert y16.dtype == torch.bfloat16
    assert normwise(y32.cpu().numpy(), want.numpy()) <= TOL
    err = (y16.float().cpu() - want).abs()
    assert bool((err <= 2.0 ** -8 * want.abs() + TOL * want.abs().max()).all())


# ---- config 5: training ----------------------------------------------------------------------------------------------

def _oracle_ce_cls():
    from dagl_amd.ce import CE
    from oracle.ce_oracle import ce_forward_oracle

    class OracleCE(CE):
        """The block's parameter surface with the dense differentiable CPU oracle as forward (autograd derives what the
        reference's autograd derives)."""

        def forward(self, b):
            prm = {n: p for n, p in self.named_parameters() if not n.startswith("W.")}
            if self.select_mode == "adaptive":
                return ce_forward_oracle(b, prm, mode="adaptive", k=None, dtype=b.dtype)
            out, st = ce_forward_oracle(b, prm, mode=self.select_mode, k=self.select_k, dtype=b.dtype, stages=True)
            # how close this head comes to a tie between its k-th and (k+1)-th neighbour: a fixed-k selection is discontinuous there,
            # and two evaluations that differ in the last bits of a score may resolve it differently
            with torch.no_grad():
                S = st["S"].detach()
                kk = min(int(self.select_k), S.shape[-1] - 1)
                top = S.topk(kk + 1, dim=-1).values
                self.min_topk_gap = float(((top[..., kk - 1] - top[..., kk]) / top[..., kk - 1].abs().clamp_min(1e-30)).min())
            return out
    return OracleCE


@pytest.mark.parametrize("mode,k", [("topk", 8), ("adaptive", 0)])
def test_config5_whole_network_gradients_match_oracle_autograd(mode, k):
    from dagl_amd.ce import CE
    from dagl_amd.net import RR, seeded_state_dict
    from dagl_amd.train import freeze_unused, task_loss
    # N = 2304 keys per image: the screened scan runs.  "adaptive" = the shipped semantics at default-like init: dense
    # neighbourhoods, forward and backward in the dense formulation (dense_train.hip)
    B, C, H, W = 2, 3, 48, 48
    ref = RR(n_colors=C, ce_cls=_oracle_ce_cls())
    sd = seeded_state_dict(ref.state_dict(), 19)
    ref.load_state_dict(sd, strict=True)
    net = RR(n_colors=C)
    net.load_state_dict(sd, strict=True)
    for m in list(ref.modules()) + list(net.modules()):
        if isinstance(m, CE):
            m.select_mode, m.select_k = mode, k or m.select_k
    freeze_unused(ref); freeze_unused(net)
    g = torch.Generator().manual_seed(23)
    hr = torch.rand(B, C, H, W, generator=g)
    lr = hr + (50.0 / 255.0) * torch.randn(B, C, H, W, generator=g)
    torch.set_num_threads(min(16, os.cpu_count() or 1))                # (hundreds of threads crawl on these small fp64 ops)
    # the reference's OWN arithmetic as the yardstick: the same oracle network in fp32 (what the reference computes on its CPU path) --
    # every tensor's bound below is tied to ITS distance from the fp64 autograd, not to an absolute number
    ref32 = RR(n_colors=C, ce_cls=_oracle_ce_cls())
    ref32.load_state_dict(sd, strict=True)
    for m in ref32.modules():
        if isinstance(m, CE):
            m.select_mode, m.select_k = mode, k or m.select_k
    freeze_unused(ref32)
    ref32 = ref32.train()
    task_loss(ref32(lr), hr, "dn_real").backward()
    g32 = {n: p.grad for n, p in ref32.named_parameters()}
    ref = ref.double().train()
    loss_ref = task_loss(ref(lr.double()), hr.double(), "dn_real")
    loss_ref.backward()
    net = net.to(_dev()).train()
    loss = task_loss(net(lr.to(_dev())), hr.to(_dev()), "dn_real")
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-4 * abs(float(loss_ref.detach()))
    worst, errs = ("", 0.0), []
    gref = dict(ref.named_parameters())
    for name, p in net.named_parameters():
        if not p.requires_grad:
            continue
        if gref[name].grad is None:                                    # fixed-k selection: the thr / bias heads take no part
            assert p.grad is None and mode == "topk" and ("thr_conv" in name or "bias_conv" in name), name
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        e = normwise(p.grad.cpu().numpy(), gref[name].grad.numpy())
        errs.append(e)
        if e > worst[1]:
            worst = (name, e)
    top = sorted(((normwise(p.grad.cpu().numpy(), gref[n].grad.numpy()), n) for n, p in net.named_parameters()
                  if p.requires_grad and gref[n].grad is not None), reverse=True)[:5]
    print(f"[config5 gradients, {mode}] median {float(np.median(errs)):.2e}; worst tensors: " + ", ".join(f"{n} {e:.2e}" for e, n in top))
    # e_hip <= 3 e_ref32 + floor per tensor (both against the fp64 autograd).  The floor covers tensors on which the reference's fp32
    # rounding happens to cancel (its distance is a draw of the same noise, not a bound on it): the median reference distance over the
    # network's tensors, i.e. "no worse than three times what fp32 arithmetic typically costs here"
    e32 = {n: normwise(g32[n].numpy(), gref[n].grad.numpy()) for n, p in net.named_parameters()
           if p.requires_grad and gref[n].grad is not None and g32[n] is not None}
    floor32 = float(np.median(list(e32.values())))
    ratios = sorted(((normwise(p.grad.cpu().numpy(), gref[n].grad.numpy()) / (e32[n] + floor32), n) for n, p in net.named_parameters()
                     if n in e32), reverse=True)
    print(f"[config5 gradients, {mode}] reference-fp32 autograd vs fp64: median {floor32:.2e}, worst {max(e32.values()):.2e}; "
          f"e_hip / (e_ref32 + median): worst " + ", ".join(f"{n} {r:.2f}" for r, n in ratios[:4]))
    if mode == "topk":
        # fixed-k selection is discontinuous: a near-tie between an 8th and a 9th neighbour in one of the 12 heads, resolved
        # differently by fp64 and fp32 scores, moves THAT head's gradients by ~1e-2 while everything else agrees.  The oracle heads
        # report their closest call (relative gap between the k-th and (k+1)-th score over all queries, fp64): only the parameters
        # of heads that come within 1e-6 of a tie get the loose bound, every other tensor of the network the tight one
        near_tie = {n for n, m in ref.named_modules() if getattr(m, "min_topk_gap", 1.0) < 1e-6}
        print(f"[config5 gradients, topk] heads within 1e-6 of a k-th / (k+1)-th tie: {sorted(near_tie)}")
        # a flipped neighbour changes that head's output, hence the gradient that flows BACK through it: the head's own parameters
        # and everything upstream of it in the forward graph (trunk before the CES, earlier stages, the ResBlocks between them) see
        # it; the head's siblings and everything downstream (the stage's mix, the trunk behind, the tail) do not
        loose_prefixes = set()
        for h in near_tie:                                  # e.g. "body.8.c3_2"
            ces, stage = h.rsplit(".", 1)[0], int(h.rsplit(".", 1)[1][1])
            loose_prefixes |= {h + ".", "head."} | {f"body.{i}." for i in range(int(ces.split(".")[1]))}
            for t in range(1, stage):
                loose_prefixes |= {f"{ces}.c{t}_", f"{ces}.RBS{t}."}
        for name, p in net.named_parameters():
            if not p.requires_grad or gref[name].grad is None:
                continue
            e = normwise(p.grad.cpu().numpy(), gref[name].grad.numpy())
            loose = any(name.startswith(pre) for pre in loose_prefixes)
            # (everything else still sees the flip at second order -- the activations behind that head move a little, and with them
            # the gradient that reaches its siblings: 2.1e-3 on one sibling's bias in the committed draw, 3e-4 typical)
            assert e <= (5e-2 if loose else 3e-3), (name, e, "touched by a near-tie head" if loose else "")
            # (per tensor against the reference's own fp32 distance: only meaningful when no head sits on a tie -- a neighbour that the
            # HIP scores and the fp64 oracle resolve differently moves that head's output, and through the activations behind it EVERY
            # gradient of the network at the 1e-3 level, while the fp32 oracle, resolving the tie as fp64 does, stays at 1e-6)
            if not near_tie and name in e32:
                assert e <= 3.0 * (e32[name] + floor32), (name, e, e32[name], floor32)
        assert float(np.median(errs)) <= 1e-3, float(np.median(errs))
    else:
        # dense regime at default-like init: logits of several hundred, so the ~8e-8 relative rounding noise of a score (split-fp16
        # or fp32 alike, tools/mfma_precision.hip) reaches the stage-3 heads' gradients amplified ~1e4 times; which tensor catches it
        # is a property of the weight draw, not of the kernels (profiles/r04_grad_noise_by_seed.log: the round-3 and round-4
        # libraries each reach 2.2-2.4e-3 on one of six seeds and 1e-5..4e-4 on the others; the reference's own fp32 autograd sits
        # 8e-4 from fp64 on this draw).  Typical tensors must agree far better: the median.
        assert worst[1] <= 5e-3 and float(np.median(errs)) <= 1e-4, (worst, float(np.median(errs)))
        for r, n in ratios:
            assert r <= 3.0, (n, r, e32[n], floor32)


@pytest.mark.parametrize("mode,k", [("topk", 8), ("adaptive", 0)])
def test_config5_train_step_128x128_batch8(mode, k):
    """BASELINE configs[4] per GPU: RR(n_colors=3), [8,3,128,128] crops, MSE(sum)/(2B), Adam: the loss goes down."""
    from dagl_amd.ce import CE
    from dagl_amd.net import RR, seeded_state_dict
    from dagl_amd.train import TrainOptions, TrainStep, freeze_unused, make_optimizer
    net = RR(n_colors=3)
    net.load_state_dict(seeded_state_dict(net.state_dict(), 7), strict=True)
    for m in net.modules():
        if isinstance(m, CE):
            m.select_mode, m.select_k = mode, k or m.select_k
    net = net.to(_dev())
    freeze_unused(net)
    opt = TrainOptions(task="dn_real", lr=1e-4)
    step = TrainStep(net, make_optimizer(net, opt), opt, generator=torch.Generator(device=_dev()).manual_seed(5))
    hr = torch.rand(8, 3, 128, 128, generator=torch.Generator().manual_seed(6)).to(_dev())
    losses = [float(step(hr)[0]) for _ in range(6)]
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    for name, p in net.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
        elif mode == "topk":
            assert ".W." in name or "thr_conv" in name or "bias_conv" in name or name.startswith("add_mean"), name


def test_config5_rccl_process_group_on_one_gpu():
    """The 8-GPU run must not be the first execution of init_process_group("nccl") / DDP over RCCL: world size 1 here."""
    env = dict(os.environ, DAGL_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--train", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--crop", "64"], env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["parallelism"] == "ddp1" and line["allreduce_ms"] is not None and line["allreduce_ms"] > 0
    assert np.isfinite(line["loss_first_last"]).all()
    # forward bench through the same process-group path
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--size", "64",
                        "--no-quality", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.parametrize("mode,k,variant,gain", [("topk", 8, "default", 2.0), ("adaptive_topk", 16, "sparse", 1.7),
                                                 ("adaptive", 0, "sparse", 1.95)])
def test_config2_batches_equal_their_images_at_256x256(mode, k, variant, gain):
    """Samples are independent (dagl.py:245): a batch of three 256x256 images through one call -- the batch is a grid dimension
    of every kernel, the screen then runs 512-query blocks with fewer key chunks per image -- must give, image by image, what
    three calls give.  Not bit for bit: the projection cuts the overhang of its grid into single-tile blocks whose three
    split products meet in a different order, and WHICH patches those are depends on the batch size (and the CU count), so the
    features' last bits do; the softmax turns that into a few 1e-6 of the output.  The same call repeated is bit-identical."""
    from dagl_amd.synth import make_features
    ce = _module(_params(41, variant, gain), mode, k)
    x = torch.from_numpy(np.concatenate([make_features(41 + i, 1, 64, 256, 256) for i in range(3)], axis=0)).to(_dev())
    with torch.no_grad():
        whole = ce(x).clone()
        again = ce(x).clone()
        singles = [ce(x[i:i + 1]).clone() for i in range(3)]
    assert torch.isfinite(whole).all()
    assert torch.equal(whole, again)
    for i in range(3):
        e = normwise(whole[i:i + 1].cpu().numpy(), singles[i].cpu().numpy())
        assert e <= 2e-5, f"image {i} differs inside the batch: {e:.2e}"
