#!/usr/bin/env python3
"""Benchmark of the patch-graph attention head on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode topk|adaptive] [--k 8] [--size 256] [--batch 1]

A "step" is one forward of one ``CE`` head (prologue convolutions + the whole HIP block) over one batch
of synthetic feature maps ``[batch,64,size,size]`` that is already resident in HBM.  Default workload =
BASELINE.json configs[1]: 256x256x64 features, k=8, fp32, one GPU.  Multi-GPU (launched by
``python -m torch.distributed.run``) is plain image-batch data parallel: every rank runs the same step
on its own images (weak scaling), no collective on the data path; timing = barrier + sync on both
sides, MAX over ranks.

Rank 0 prints ONE JSON line: whole-job query-patches/s, plus
  roofline      the dominant kernel (streamed similarity + select, fp32 matrix cores): algorithmic FLOP
                per launch / its mean duration, measured live with hipEvents at the stage boundaries of
                the very steps that are timed (dagl_profile_*, recorded on the launch stream)
  roofline_gather  the stand-alone gather/weighted-sum kernel over materialised value rows (HBM bound,
                algorithmic bytes (k+1)*4P+8k per query) and the fused in-block gather
  cpu_baseline  the dense CPU oracle (a port of the reference's algorithm) timed on this box's host cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
D_FEAT, P_ROW = 196, 784


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="topk", choices=["topk", "adaptive", "adaptive_topk"])
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--sparse-gain", type=float, default=2.4, help="adaptive modes: threshold gain of the synthetic thr head")
    ap.add_argument("--scan", default="screened", choices=["screened", "exact"],
                    help="screened: bf16 matrix-core screen + exact refine (default); exact: all scores on the fp32 matrix cores")
    ap.add_argument("--stage", action="store_true",
                    help="time one CES stage (4 heads sharing the input + 1x1 mix + residual, dagl_ces_stage_forward) instead of one head")
    ap.add_argument("--train", action="store_true",
                    help="secondary workload (BASELINE configs[4]): one optimisation step of the whole RR network on "
                         "--crop x --crop crops, --batch crops per GPU, DDP gradient all-reduce over RCCL")
    ap.add_argument("--crop", type=int, default=128)
    ap.add_argument("--colors", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quality", action="store_true", help="skip the Set12 sigma=50 PSNR-delta leg")
    ap.add_argument("--cpu-size", type=int, default=0, help="feature-map size of the CPU-baseline sample (default: --size)")
    return ap.parse_args()


def cpu_baseline(args, params, mode, k):
    """Dense CPU oracle on the host cores, bounded sample (one forward of the benchmark workload after a small
    warm-up).  The reference algorithm is dense: its cost does not depend on k."""
    from dagl_amd.synth import make_features
    from oracle.ce_oracle import ce_forward_oracle
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    size = args.cpu_size or args.size
    x_small = torch.from_numpy(make_features(1, 1, 64, 64, 64))
    x = torch.from_numpy(make_features(2, 1, 64, size, size))
    with torch.no_grad():
        ce_forward_oracle(x_small, params, mode=mode, k=k or None)          # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        ce_forward_oracle(x, params, mode=mode, k=k or None)
        dt = time.perf_counter() - t0
    L = ((size + 3) // 4) ** 2
    return {"value": L / dt, "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 forward of oracle/ce_oracle.py (dense torch-CPU restatement of CE.forward) on "
                      f"[1,64,{size},{size}] fp32, L={L} query patches, {dt:.2f} s, after one 64x64 warm-up"}


def committed_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc pass (profiles/r01_traffic.json): PMC
    counters cannot be read from inside the timed process, so the number comes from the profile of this same command."""
    try:
        t = json.load(open(os.path.join(REPO, "profiles", "r01_traffic.json")))
        return t.get(kernel_key)
    except Exception:
        return None


def quality_leg(dev):
    """PSNR on Set12 sigma=50 of the full network (12 HIP heads) vs the reference forward with the same regenerable
    weights (numbers committed by tests/golden/make_set12_psnr.py, which ran the reference on CPU)."""
    import numpy as np
    from dagl_amd.net import RR, chop_forward_batched, psnr, seeded_state_dict, set12_protocol_noise
    gdir = os.path.join(REPO, "tests", "golden")
    ref = json.load(open(os.path.join(gdir, "set12_psnr_ref.json")))
    imgs = np.load(os.path.join(gdir, "set12.npz"))
    net = RR().eval()
    net.load_state_dict(seeded_state_dict(net.state_dict(), ref["seed"]), strict=True)
    net = net.to(dev)
    deltas, t0 = {}, time.perf_counter()
    for name in sorted(ref["images"]):
        clean = torch.from_numpy(imgs[f"img_{name}"].astype(np.float32) / 255.0)[None, None]
        noisy = set12_protocol_noise(clean, 50.0, 1.0)
        with torch.no_grad():
            out = torch.clamp(chop_forward_batched(net, noisy.to(dev)), 0.0, 1.0).cpu()
        deltas[name] = psnr(out, clean) - ref["images"][name]["psnr_out"]
    return {"dataset": "Set12 (12 images), sigma=50, reference test protocol (tiled inference, no self-ensemble)",
            "weights": f"regenerable stand-in checkpoint (numpy PCG64 seed {ref['seed']}); no trained weights ship with the reference",
            "psnr_delta_db_max_abs": max(abs(v) for v in deltas.values()),
            "psnr_delta_db_mean": sum(deltas.values()) / len(deltas),
            "psnr_ref_mean_db": sum(r["psnr_out"] for r in ref["images"].values()) / len(ref["images"]),
            "bar_db": 0.02, "seconds": time.perf_counter() - t0}


def train_bench(args, dev, dist, world, rank):
    """BASELINE configs[4] in the sparse regime the backward serves: RR (12 CE heads, fixed-k selection) trained on
    synthetic crops, fwd + bwd + Adam per step, one process per GPU under DDP (RCCL all-reduce of 5.7 M gradients)."""
    from dagl_amd.net import RR, seeded_state_dict
    from dagl_amd.shard import rank_seed, reduce_max_seconds
    from dagl_amd.train import TrainOptions, TrainStep, freeze_unused, make_optimizer, wrap_ddp
    from dagl_amd.ce import CE
    B = args.batch if args.batch > 1 else 8
    net = RR(n_colors=args.colors)
    net.load_state_dict(seeded_state_dict(net.state_dict(), 7), strict=True)       # same weights on every rank
    for m in net.modules():
        if isinstance(m, CE):
            m.select_mode, m.select_k, m.scan = args.mode, args.k, args.scan
    net = net.to(dev)
    freeze_unused(net)
    model = wrap_ddp(net, dev) if dist is not None else net
    opt = TrainOptions(task="dn_real", lr=1e-4)
    step = TrainStep(model, make_optimizer(model, opt), opt,
                     generator=torch.Generator(device=dev).manual_seed(rank_seed(300, rank)))
    g = torch.Generator().manual_seed(rank_seed(200, rank))
    hr = torch.rand(B, args.colors, args.crop, args.crop, generator=g).to(dev)     # resident in HBM
    losses = []
    for _ in range(args.warmup):
        step(hr)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step(hr)[0])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = reduce_max_seconds(time.perf_counter() - t0, dist, dev)
    n_par = sum(p.numel() for p in net.parameters() if p.requires_grad)
    allreduce_ms = None
    if dist is not None:
        # the collective on its own (outside the timed steps): one all-reduce of a gradient-sized fp32 buffer over RCCL
        flat = torch.zeros(n_par, device=dev)
        for _ in range(3):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(flat)
        torch.cuda.synchronize()
        allreduce_ms = reduce_max_seconds((time.perf_counter() - t1) / 10, dist, dev) * 1e3
    if rank == 0:
        line = {"metric": f"RR train crops/s (fwd+bwd+Adam) @{args.crop}x{args.crop} {args.mode} k={args.k}",
                "value": world * B * args.steps / elapsed, "unit": "crops/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[4] (sparse regime): RR with 12 CE heads, crops [{B},{args.colors},"
                                       f"{args.crop},{args.crop}] per GPU, MSE(sum)/(2B) loss, Adam, gradients of "
                                       f"{n_par} parameters all-reduced over RCCL",
                           "parallelism": f"ddp{world}", "select_mode": args.mode, "k": args.k, "batch_per_gpu": B},
                "roofline": None, "cpu_baseline": None,
                "allreduce_ms": allreduce_ms, "allreduce_bytes": 4 * n_par if dist is not None else None,
                "loss_first_last": [float(losses[0]), float(losses[-1])] if losses else None}
        print(json.dumps(line))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DAGL_BENCH_FORCE_DIST"):          # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI: used for the barriers/max only

    if args.train:
        train_bench(args, dev, dist, world, rank)
        if dist is not None:
            dist.destroy_process_group()
        return

    from dagl_amd import ops
    from dagl_amd.ce import CE
    from dagl_amd.shard import rank_seed, reduce_max_seconds
    from dagl_amd.synth import make_ce_params, make_features

    mode, k = args.mode, (args.k if args.mode != "adaptive" else 0)
    variant = "default" if mode == "topk" else "sparse"
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(2024, variant=variant, sparse_gain=args.sparse_gain).items()}
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode = mode
    ce.scan = args.scan
    if k:
        ce.select_k = k
    ce = ce.to(dev).eval()

    B, H, W = args.batch, args.size, args.size
    x = torch.from_numpy(make_features(rank_seed(100, rank), B, 64, H, W)).to(dev)     # resident in HBM
    L, N = ((H + 3) // 4) * ((W + 3) // 4), H * W

    prof = ops.StageProfile(max(args.steps, 1))
    heads_per_step = 1
    if args.stage:
        heads_per_step = 4
        heads = [ce]
        for hseed in (2025, 2026, 2027):
            hp = {n: torch.from_numpy(a) for n, a in make_ce_params(hseed, variant=variant, sparse_gain=args.sparse_gain).items()}
            hm = CE(in_channels=64)
            hm.load_state_dict(hp, strict=True)
            hm.select_mode, hm.scan, hm.select_k = ce.select_mode, ce.scan, ce.select_k
            heads.append(hm.to(dev).eval())
        prm = [{n: q.detach().contiguous() for n, q in hd.named_parameters() if not n.startswith("W.")} for hd in heads]
        gmix = torch.Generator().manual_seed(3)
        mix_w = ((torch.rand(64, 64, 1, 1, generator=gmix) - 0.5) * 0.25).to(dev)
        mix_b = ((torch.rand(64, generator=gmix) - 0.5) * 0.1).to(dev)
        ws_stage = ops.Workspace()
        info_box = {}

        def step(profile=None):
            out, inf = ops.ces_stage_forward(x, prm, mix_w, mix_b, mode=mode, k=ce.select_k, workspace=ws_stage, profile=profile)
            assert out is not None, "dense neighbourhoods: the stage entry point handed the call back"
            info_box["info"] = inf
    else:
        def step(profile=None):
            ce.profile = profile
            ce(x)
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        # timed region: only the dominant kernel is bracketed by hipEvents (on the launch stream); an event record costs
        # ~4 us of stream time, so the full nine-boundary stage profile is taken in a separate pass below
        prof.select_stage("select")
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(prof)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        dominant_ms = [c[4] for c in prof.read()]
        prof.select_stage(-1)
        for _ in range(min(args.steps, 20)):
            step(prof)
        torch.cuda.synchronize()
        ce.profile = None
    elapsed = reduce_max_seconds(elapsed, dist, dev)
    stage_ms = prof.read()
    info = info_box["info"] if args.stage else ce.last_info

    # stand-alone gather kernel over materialised value rows (rank 0 only; outside the timed region)
    gather = None
    if rank == 0:
        with torch.no_grad():
            b1, b2, thr, bias = ce._prologue(x[:1])
            rows = ops.unfold_values(ops.pad_nhwc(b2.contiguous()), H, W)[0].contiguous()       # [N,784]
            kk = k or 8
            g = torch.Generator(device="cpu").manual_seed(5)
            idx = torch.randint(0, N, (L, kk), generator=g, dtype=torch.int32).to(dev)
            wgt = torch.rand(L, kk, generator=g).to(dev)
            for _ in range(5):
                ops.gather_aggregate(idx, wgt, rows)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record()
            for _ in range(reps):
                ops.gather_aggregate(idx, wgt, rows)
            e1.record()
            e1.synchronize()
            g_ms = e0.elapsed_time(e1) / reps
            g_bytes = L * ((kk + 1) * 4 * P_ROW + 8 * kk)
            gather = {"bound": "hbm", "kernel": "gather_rows_kernel (dagl_gather_aggregate)", "k": kk,
                      "achieved": g_bytes / (g_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": g_bytes / (g_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                      "traffic": committed_traffic("gather_rows_kernel") if (H, kk) == (256, 8) else None,
                      "bytes_per_launch": g_bytes, "ms_per_launch": g_ms,
                      "timing": "mean of 50 back-to-back launches (launch gaps included), torch events on the launch stream"}
            del rows

    if rank == 0:
        import numpy as np
        sm = np.asarray(stage_ms, dtype=np.float64)                      # [steps, 8]
        mean_ms = sm.mean(axis=0) if len(sm) else np.zeros(8)
        from dagl_amd._lib import STAGE_NAMES
        sel_ms = float(np.mean(dominant_ms)) if len(dominant_ms) else float(mean_ms[4])     # from the timed steps
        flops = 2.0 * heads_per_step * B * L * N * D_FEAT                # algorithmic: 2*L*N*D per image and head
        ach = flops / (sel_ms * 1e-3) / 1e12 if sel_ms > 0 else 0.0
        screened = (info or {}).get("path") == 3
        peak = PEAK_BF16_MATRIX_TFLOPS if screened else PEAK_F32_MATRIX_TFLOPS
        roofline = {"bound": "mfma",
                    "kernel": "screen_kernel<1> (bf16 v_mfma_f32_32x32x16_bf16, full L*N candidate filter)" if screened
                              else "score_select_kernel (fp32 v_mfma_f32_32x32x2_f32)",
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "traffic": committed_traffic("screen_kernel<1>") if (screened and (H, mode, k, B) == (256, "topk", 8, 1)) else None,
                    "flop_per_launch": flops, "ms_per_launch": sel_ms}
        if gather is not None and mean_ms[6] > 0:
            kk = k or max(1, int(round((info or {}).get("total_edges", 0) / max(1, B * L))))
            fb = B * L * ((kk + 1) * 4 * P_ROW + 8 * kk)
            gather["fused_in_block"] = {"kernel": "aggregate_direct_kernel", "ms_per_launch": float(mean_ms[6]),
                                        "achieved": fb / (mean_ms[6] * 1e-3) / 1e9, "unit": "GB/s",
                                        "frac": fb / (mean_ms[6] * 1e-3) / 1e9 / PEAK_HBM_GBS}
        total_patches = world * heads_per_step * B * L * args.steps
        line = {
            "metric": "graph-attn fwd query-patches/s @256x256x64 k=8" if (H, mode, k, args.stage) == (256, "topk", 8, False)
                      else f"graph-attn fwd query-patches/s @{H}x{W}x64 {mode} k={k}",
            "value": total_patches / elapsed, "unit": "patches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("one CES stage (4 heads + 1x1 mix + residual)" if args.stage else "BASELINE configs[1]: one CE head forward")
                                   + f", features [{B},64,{H},{W}] fp32 per GPU, "
                                   f"select mode {mode} k={k}, L={L} queries x N={N} keys per image",
                       "parallelism": f"dp{world} (independent images per rank, no data-path collective)",
                       "select_mode": mode, "k": k, "batch_per_gpu": B, "scan": args.scan,
                       "selection_path": (info or {}).get("path"), "max_degree": (info or {}).get("max_degree")},
            "roofline": roofline,
            "roofline_gather": gather,
            "stage_ms": {STAGE_NAMES[i]: float(mean_ms[i]) for i in range(8)},
            "stage_ms_note": "separate instrumented pass after the timed steps (nine event records per call add ~35 us)",
            "hip_block_ms": float(mean_ms.sum()),
        }
        if world == 1 and not args.no_quality and not args.stage:
            line["quality"] = quality_leg(dev)
        if world == 1 and not args.no_cpu_baseline and not args.stage:
            line["cpu_baseline"] = cpu_baseline(args, params, mode, k)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
