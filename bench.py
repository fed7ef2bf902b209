#!/usr/bin/env python3
"""Benchmark of the patch-graph attention head on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode topk|adaptive] [--k 8] [--size 256] [--batch 1]

A "step" is one forward of one ``CE`` head (prologue convolutions + the whole HIP block) over one batch
of synthetic feature maps ``[batch,64,size,size]`` that is already resident in HBM.  Default workload =
BASELINE.json configs[1]: 256x256x64 features, k=8, fp32, one GPU.  Multi-GPU is plain image-batch data
parallel, one process per GPU: every rank runs the same step on its own images (weak scaling), no collective
on the data path; timing = barrier + sync on both sides, MAX over ranks.  ``python bench.py --gpus N`` starts
its own N ranks (dagl_amd/launch.py: re-executes itself with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
rank 0 prints); launched under ``python -m torch.distributed.run --nproc-per-node N`` it uses the ranks it is
given.  The line carries ``n_ranks_seen`` (the process group's own world size) and every rank's device name.

Rank 0 prints ONE JSON line: whole-job query-patches/s, plus
  roofline      the dominant kernel (screen_kernel<1>: the full L x N similarity scan + candidate filter on the
                bf16 matrix cores, v_mfma_f32_32x32x16_bf16; the fp32-MFMA score_select_kernel with --scan exact):
                algorithmic FLOP per launch / its mean duration, measured live with hipEvents at the stage
                boundaries of the very steps that are timed (dagl_profile_*, recorded on the launch stream)
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
    ap.add_argument("--variant", default="auto", choices=["auto", "default", "sparse"],
                    help="synthetic thr/bias heads: default = torch-like init (adaptive: ~95 %% of the keys pass, dense regime); "
                         "sparse = threshold gain --sparse-gain; auto = default for topk, sparse otherwise")
    ap.add_argument("--scan", default="screened", choices=["screened", "exact"],
                    help="screened: bf16 matrix-core screen + exact refine (default); exact: all scores on the fp32 matrix cores")
    ap.add_argument("--stage", action="store_true",
                    help="time one CES stage (4 heads sharing the input + 1x1 mix + residual, dagl_ces_stage_forward) instead of one head")
    ap.add_argument("--train", action="store_true",
                    help="secondary workload (BASELINE configs[4]): one optimisation step of the whole RR network on "
                         "--crop x --crop crops, --batch crops per GPU, DDP gradient all-reduce over RCCL")
    ap.add_argument("--crop", type=int, default=128)
    ap.add_argument("--topk-redo", choices=["auto", "always"], default="always",
                    help="CE.topk_redo: 'always' (module default, what the benchmark times) queues the fp32 redo launch in every call; "
                         "'auto' (opt-in, A/B) drops it once a poll found the workspace without redo work")
    ap.add_argument("--dense-backward", choices=["f16", "fp32"], default="f16",
                    help="--train A/B: the dense graph core's backward products on the fp16 (split operands, default) or fp32 matrix cores")
    ap.add_argument("--prologue-backward", choices=["direct", "unfold"], default="direct",
                    help="--train A/B: gradients of g / theta as tap-wise products on the maps (default) or through unfold + GEMM + fold")
    ap.add_argument("--colors", type=int, default=3)
    ap.add_argument("--wseed", type=int, default=2024, help="seed of the synthetic head weights")
    ap.add_argument("--fseed", type=int, default=None, help="seed of the synthetic feature maps (+ rank); default: the per-rank stream of seed 100")
    ap.add_argument("--prewarm", type=float, default=0.5,
                    help="seconds of untimed passes before the W warm-up steps (clock ramp of an idle GPU); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quality", action="store_true", help="skip the Set12 sigma=50 PSNR-delta leg")
    ap.add_argument("--miopen-probe-limit", type=float, default=75.0,
                    help="seconds a child process may take to start and run its first stock convolutions (usually ~12) before the legs that run "
                         "the trunk (quality, Set12 features, training steps) are skipped with a note (a cold MIOpen on slow storage: minutes)")
    ap.add_argument("--cpu-size", type=int, default=0, help="feature-map size of the CPU-baseline sample (default: --size)")
    ap.add_argument("--cpu-runs", type=int, default=5, help="timed forwards of the CPU baseline (median reported)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs block (other BASELINE configs, short runs)")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test: the same spawn / rendezvous / barrier / MAX-over-ranks / one-JSON-line protocol with "
                         "a trivial CPU step on the gloo backend (tests/test_bench_spawn.py); not a measurement")
    return ap.parse_args()


def rank_devices(dist, name):
    """Every rank's device name, gathered on the job's own process group (shows that RCCL / gloo saw N ranks)."""
    if dist is None or not dist.is_initialized():
        return [name]
    names = [None] * dist.get_world_size()
    dist.all_gather_object(names, name)
    return names


def stub_bench(args, world, rank):
    """``--stub``: the launch protocol end to end without a GPU -- gloo process group, per-rank step on the CPU, barrier +
    MAX-over-ranks timing, rank 0 prints the one JSON line."""
    import torch.distributed as dist
    from dagl_amd.shard import reduce_max_seconds
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = torch.randn(64, 64, generator=torch.Generator().manual_seed(rank))
    for _ in range(args.warmup):
        (a @ a).sum()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        (a @ a).sum()
    dist.barrier()
    elapsed = reduce_max_seconds(time.perf_counter() - t0, dist)
    devices = rank_devices(dist, f"cpu:{rank}")
    if rank == 0:
        print(json.dumps({"metric": "stub steps/s (launcher self-test, not a measurement)", "value": world * args.steps / elapsed,
                          "unit": "steps/s", "n_gpus": world, "n_ranks_seen": dist.get_world_size(), "devices": devices,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "stub": True}), flush=True)
    dist.destroy_process_group()


def cpu_baseline(args, params, mode, k, x_cpu=None, hip_out=None):
    """Dense CPU oracle on the host cores: 3 warm-ups (64^2, 128^2, full size), median of 5 timed
    forwards of the benchmark workload on all cores, plus a 1-thread figure on a 128^2 sample (the dense algorithm is
    O(L*N): one thread at 256^2 would take minutes).  The reference algorithm is dense: its cost does not depend on k.
    With the HIP output of the same input at hand, the normwise parity error against this oracle run is reported too."""
    import statistics
    from dagl_amd.synth import make_features
    from oracle.ce_oracle import ce_forward_oracle
    cores = os.cpu_count() or 1
    size = args.cpu_size or args.size
    x = x_cpu if (x_cpu is not None and x_cpu.shape[-1] == size) else torch.from_numpy(make_features(2, 1, 64, size, size))
    runs = max(1, args.cpu_runs)
    out = None
    with torch.no_grad():
        torch.set_num_threads(cores)
        for ws in (64, 128):                                               # warm-ups (thread pool, allocator)
            ce_forward_oracle(torch.from_numpy(make_features(1, 1, 64, ws, ws)), params, mode=mode, k=k or None)
        ce_forward_oracle(x, params, mode=mode, k=k or None)
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            out = ce_forward_oracle(x, params, mode=mode, k=k or None)
            ts.append(time.perf_counter() - t0)
        dt = statistics.median(ts)
        threads = torch.get_num_threads()
        # one thread, 128^2 sample
        torch.set_num_threads(1)
        x1 = torch.from_numpy(make_features(3, 1, 64, 128, 128))
        ce_forward_oracle(x1, params, mode=mode, k=k or None)
        t1 = []
        for _ in range(3):
            t0 = time.perf_counter()
            ce_forward_oracle(x1, params, mode=mode, k=k or None)
            t1.append(time.perf_counter() - t0)
        torch.set_num_threads(cores)
    L = ((size + 3) // 4) ** 2
    res = {"value": L / dt, "unit": "patches/s", "cores": threads, "kind": "port",
           "sample": f"median of {runs} forwards of oracle/ce_oracle.py (dense torch-CPU restatement of CE.forward) on "
                     f"[1,64,{size},{size}] fp32, L={L} query patches, {dt:.2f} s each (min {min(ts):.2f}, max {max(ts):.2f}), "
                     f"after 3 warm-ups (64^2, 128^2, {size}^2)",
           "one_thread": {"value": 1024 / statistics.median(t1), "unit": "patches/s", "cores": 1,
                          "sample": f"median of 3 forwards on [1,64,128,128] (L=1024), {statistics.median(t1):.2f} s each"}}
    if hip_out is not None and out is not None and tuple(hip_out.shape) == tuple(out.shape):
        res["parity_err"] = float((hip_out - out).abs().max() / out.abs().max())
        res["parity_note"] = "max|hip - oracle| / max|oracle| of the block output on the benchmark input (bar 1e-4)"
    return res


def _prewarm(step, seconds):
    """Untimed passes of the same step until `seconds` of wall clock have gone by: a GPU that sat idle while Python started
    runs its first tens of milliseconds below the sustained clock (measured: 0.300 ms/step over the first 15 ms, 0.266 ms
    from ~50 ms on), and the contract's W warm-up steps last 1.5 ms at this step size.  Returns the passes made."""
    n, t0 = 0, time.perf_counter()
    while seconds > 0:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        n += 20
        if time.perf_counter() - t0 >= seconds:
            break
    return n


def _time_steps(step, steps, warmup, prewarm_s=0.0):
    if prewarm_s > 0:                      # building a case leaves the GPU idle for a second: same clock pre-warm as the main run
        _prewarm(step, prewarm_s)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


EXTRA_PREWARM_S = 0.15


def dense_roofline(stage_ms, B, L, N, info, source):
    """`roofline` block of a call that ran the streamed dense formulation (info.path 4): dense_attend_kernel (dense.hip).
    Algorithmic work = S (2 L N 196) + A V (2 L N 784) per image; executed on the fp16 matrix cores = three split products of
    each, S on 224 padded columns.  `stage_ms` = hipEvent-bracketed time of the call's stage "gather" (value-map split 6 us +
    dense_attend_kernel + combine 9 us + the two gated launches of the second pass) on the launch stream."""
    flops = 2.0 * B * L * N * (D_FEAT + P_ROW)
    executed = 3.0 * 2.0 * B * L * N * (224 + P_ROW)
    ach = flops / (stage_ms * 1e-3) / 1e12 if stage_ms > 0 else 0.0
    rerun = int((info or {}).get("dense_rerun_blocks", 0) or 0)
    return {"bound": "mfma", "kernel": "dense_attend_kernel (fp16 v_mfma_f32_16x16x32_f16 scores + v_mfma_f32_32x32x16_f16 A V, split "
                                       "operands hi + lo: three products each)",
            "achieved": ach, "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MATRIX_TFLOPS,
            "executed_frac": executed / (stage_ms * 1e-3) / 1e12 / PEAK_BF16_MATRIX_TFLOPS if stage_ms > 0 else 0.0,
            "executed_note": "upper figure: exact-zero weight granules (64 queries x 16 keys) are skipped, so fewer multiplies than "
                             "this run on maps whose logits reach hundreds (synthetic N(0,1) features); none are on trained features",
            "traffic": committed_traffic("dense:dense_attend_kernel") if (B, L, N) == (1, 4096, 65536) else None,
            "traffic_note": "HBM-side bytes per dense_attend_kernel launch from the committed rocprofv3 --pmc passes of `bench.py --mode "
                            "adaptive --variant default` (profiles/rNN_traffic.json, key dense:dense_attend_kernel; mean over the call's "
                            "two launches, of which the gated second one moves nothing): not measured in this run",
            "flop_per_launch": flops, "ms_per_launch": stage_ms,
            "launches_per_call": 1 if rerun == 0 else 2, "dense_rerun_blocks": rerun,
            "launches_note": "dense_attend_kernel launches of a call that do work (the second, gated pass exits at once unless the "
                             "first flagged blocks of 64 queries: rows the top-1 screen could not give an exact shift)",
            "timing": source}


def miopen_probe_seconds(dev, limit):
    """Wall-clock seconds a CHILD process needs for its first stock convolutions (forward + backward of a 64-channel 3x3 layer on a
    leaf tile), inf when it is not done after ``limit`` seconds (the child is killed; this process has not touched MIOpen then).
    A child, because the start-up cannot be interrupted from inside; its run also leaves MIOpen's files in the page cache."""
    import subprocess
    code = ("import time, torch\n"
            "t0 = time.perf_counter()\n"
            f"dev = torch.device('cuda:{dev.index or 0}')\n"
            "conv = torch.nn.Conv2d(64, 64, 3, padding=1).to(dev)\n"
            "x = torch.randn(2, 64, 72, 72, device=dev, requires_grad=True)\n"
            "conv(x).sum().backward()\n"
            "torch.cuda.synchronize()\n"
            "print('PROBE', time.perf_counter() - t0)\n")
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=limit)
        for ln in r.stdout.splitlines():
            if ln.startswith("PROBE"):
                return float(ln.split()[1])
        return float("inf")
    except subprocess.TimeoutExpired:
        return float("inf")
    finally:
        _ = time.perf_counter() - t0


def extra_configs(dev, trunk_ok=True):
    """The other BASELINE configurations and regimes, each a short run (0.15 s of untimed passes for the clock -- see
    --prewarm --, 3 warm-ups + 10 timed steps) inside the one driver-timed command, so that their numbers are measured by
    the driver's run as well."""
    from dagl_amd import ops
    from dagl_amd._lib import STAGE_NAMES
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features

    def head(seed, variant, gain, mode, k):
        prm = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant=variant, sparse_gain=gain).items()}
        m = CE(in_channels=64)
        m.load_state_dict(prm, strict=True)
        m.select_mode = mode
        if k:
            m.select_k = k
        return m.to(dev).eval()

    out = {}
    cases = [
        ("512x512_topk8_bf16_io", 512, "default", 2.0, "topk", 8, torch.bfloat16, "BASELINE configs[2]: CAR 512x512, k=8, bf16 feature maps"),
        ("1024x1024_adaptive_topk16", 1024, "sparse", 1.7, "adaptive_topk", 16, torch.float32, "BASELINE configs[3]: 1024x1024, adaptive AND k_max=16, whole-image search window"),
        ("256x256_adaptive_dense_default_init", 256, "default", 2.0, "adaptive", 0, torch.float32, "shipped semantics at default init (~95 % of the keys pass): streamed dense formulation; on this synthetic N(0,1) map the logits reach hundreds and ~2/3 of the (64 query x 16 key) weight granules are exactly zero and skipped -- see 256x256_set12_features.adaptive_dense for the regime where none are"),
        ("256x256_adaptive_dense_default_init_bench_map", 256, "default", 2.0, "adaptive", 0, torch.float32, "the same configuration on the map `bench.py --mode adaptive --variant default` itself times (features seed rank_seed(100, 0) = 100000): in round 4 every block of THIS map took the dense kernel's second pass (1.52 ms a call) while the entry above (features seed 100) did not"),
        ("256x256_adaptive_mean_degree_8", 256, "sparse", 1.95, "adaptive", 0, torch.float32, "adaptive mask tuned to a mean degree of ~8 (7.7; long-tailed: maximum 890) (SURVEY 8d config 2)"),
        ("256x256_adaptive_mean_degree_55", 256, "sparse", 1.8, "adaptive", 0, torch.float32, "adaptive mask at mean degree 55, maximum 4578"),
        ("256x256_topk8_batch8", 256, "default", 2.0, "topk", 8, torch.float32, "the headline configuration with EIGHT images per call ([8,64,256,256]; batch = grid dimension): what the fixed per-launch costs of the one-image step are worth", 8),
        ("256x256_topk500", 256, "default", 2.0, "topk", 500, torch.float32, "num_edge = 500 (CA_model-checkpoint.py:134-143): beyond the 64-entry lists, every query's score row in the dense form (csrc/topk_wide.hip)"),
    ]
    seeds = {"256x256_adaptive_mean_degree_8": (41, 41), "256x256_adaptive_mean_degree_55": (41, 41),       # (weights, features): the pair tests/test_gpu_configs.py checks against the oracle
             "256x256_adaptive_dense_default_init_bench_map": (2024, 100000)}
    with torch.no_grad():
        for name, size, variant, gain, mode, k, dt, what, *rest in cases:
            nb = rest[0] if rest else 1
            ws, fs = seeds.get(name, (2024, 100))
            ce = head(ws, variant, gain, mode, k)
            if name == "256x256_adaptive_mean_degree_8":
                ce.adaptive_sync = "auto"        # steady sparse workload: stops waiting for the verdict after four served calls (ce.py)
            x = torch.from_numpy(make_features(fs, nb, 64, size, size)).to(dev).to(dt)
            ms = _time_steps(lambda: ce(x), 10, 3, EXTRA_PREWARM_S)
            L = nb * (size // 4) ** 2
            info = ce.last_info or {}
            out[name] = {"what": what, "ms_per_step": ms, "patches_per_s": L / (ms * 1e-3), "L": L, "N": size * size, "batch": nb,
                         "selection_path": info.get("path"), "max_degree": info.get("max_degree"),
                         "mean_degree": (info.get("total_edges", -1) / L) if info.get("total_edges", -1) >= 0 else None}
            if info.get("path") == 3 and size >= 512:
                # configs[2] / configs[3]: the same accounting as the headline -- the dominant kernel (the bf16 filter pass over all L N
                # scores) event-bracketed on the launch stream in 10 more calls, then the nine-boundary stage profile of 5 calls
                pr = ops.StageProfile(10)
                pr.select_stage("select")
                ce.profile = pr
                for _ in range(10):
                    ce(x)
                torch.cuda.synchronize()
                s_ms = [c[4] for c in pr.read()]
                pr.select_stage(-1)
                pr.reset()
                for _ in range(5):
                    ce(x)
                torch.cuda.synchronize()
                ce.profile = None
                st = pr.read()
                sel = sum(s_ms) / max(len(s_ms), 1)
                flop = 2.0 * L * (size * size) * D_FEAT
                ach = flop / (sel * 1e-3) / 1e12 if sel > 0 else 0.0
                out[name]["roofline"] = {"bound": "mfma", "kernel": "screen_ring_kernel<1> (bf16 v_mfma_f32_32x32x16_bf16, full L*N candidate filter)",
                                         "achieved": ach, "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MATRIX_TFLOPS,
                                         "traffic": None, "flop_per_launch": flop, "ms_per_launch": sel,
                                         "timing": "mean of 10 calls, two hipEvents around the stage 'select' on the launch stream"}
                if st:
                    mean = [sum(c[i] for c in st) / len(st) for i in range(8)]
                    out[name]["stage_ms"] = {STAGE_NAMES[i]: float(mean[i]) for i in range(8)}
                    out[name]["stage_ms_note"] = "separate instrumented pass (nine event records per call); stage 'select' there includes the records' stream time"
                del pr
            if info.get("path") == 4:              # dense regime: its dominant kernel's roofline, event-bracketed in 10 more calls
                pr = ops.StageProfile(10)
                pr.select_stage("gather")
                ce.profile = pr
                for _ in range(10):
                    ce(x)
                torch.cuda.synchronize()
                ce.profile = None
                g_ms = [c[6] for c in pr.read()]
                out[name]["roofline"] = dense_roofline(sum(g_ms) / max(len(g_ms), 1), nb, L // nb, size * size, ce.last_info,
                                                       "mean of 10 calls, two hipEvents around the stage on the launch stream (these "
                                                       "calls also read their statistics back: not the calls of ms_per_step)")
                del pr
            del ce, x
            torch.cuda.empty_cache()
        # one CES stage: 4 heads sharing x + 1x1 mix + residual
        heads = [head(2024 + h, "default", 2.0, "topk", 8) for h in range(4)]
        prm = [{n: q.detach().contiguous() for n, q in hd.named_parameters() if not n.startswith("W.")} for hd in heads]
        gm = torch.Generator().manual_seed(3)
        mix_w = ((torch.rand(64, 64, 1, 1, generator=gm) - 0.5) * 0.25).to(dev)
        mix_b = ((torch.rand(64, generator=gm) - 0.5) * 0.1).to(dev)
        x = torch.from_numpy(make_features(100, 1, 64, 256, 256)).to(dev)
        ws = ops.Workspace()
        ops.ces_stage_forward(x, prm, mix_w, mix_b, mode="topk", k=8, workspace=ws)      # packs the weights into the workspace
        ms = _time_steps(lambda: ops.ces_stage_forward(x, prm, mix_w, mix_b, mode="topk", k=8, workspace=ws,
                                                       weights_packed=True), 10, 3, EXTRA_PREWARM_S)     # (CES._stage's protocol)
        out["256x256_ces_stage_topk8"] = {"what": "one CES stage (4 heads + 1x1 mix + residual, dagl_ces_stage_forward)",
                                          "ms_per_step": ms, "patches_per_s": 4 * 4096 / (ms * 1e-3), "L": 4096, "N": 65536}
        del heads, prm, x, ws
        torch.cuda.empty_cache()
    out["128x128_geometry_k5_s2_generic"] = generic_geometry_extra(dev)
    if not trunk_ok:
        out["skipped"] = ("256x256_set12_features and the three train_rr_* entries run the trunk's stock convolutions: skipped, MIOpen is cold on "
                          "this box (miopen_probe_s)")
        return out
    out["256x256_set12_features"] = real_features_extra(dev)
    out["train_rr_topk8_128x128_b8"] = train_extra(dev)
    out["train_rr_adaptive_128x128_b8"] = train_extra(dev, steps=4, warmup=2, mode="adaptive")
    # the reference trainer's own default shape: DN_Gray --patch_size 64 --batch_size 32 (option.py:42,88), shipped adaptive semantics
    out["train_rr_adaptive_64x64_b32_gray"] = train_extra(dev, B=32, crop=64, colors=1, steps=4, warmup=2, mode="adaptive")
    return out


def generic_geometry_extra(dev):
    """A head built with a NON-default patch geometry -- CE(ksize=5, stride_1=2, stride_2=1): ctor arguments of the reference block
    (dagl.py:175-176) that none of the tuned kernels holds -- through ``dagl_ce_generic_forward`` (csrc/generic.hip: the reference's dense
    formulation, unfold + fp32 matrix cores, score rows a chunk at a time).  [1,64,128,128]: L = 4096 queries, N = 16384 keys, P = 400."""
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(2024, ksize=5, variant="sparse", sparse_gain=1.6).items()}
    ce = CE(ksize=5, stride_1=2, stride_2=1)
    ce.load_state_dict(prm, strict=True)
    ce = ce.to(dev).eval()
    x = torch.from_numpy(make_features(100, 1, 64, 128, 128)).to(dev)
    with torch.no_grad():
        ms = _time_steps(lambda: ce(x), 10, 3, EXTRA_PREWARM_S)
        deg = ce.last_info["degree"].float()
    L, N, P_, D_ = 4096, 16384, 400, 100
    flop = 2.0 * L * N * (D_ + P_) + 2.0 * (L + N) * P_ * D_
    return {"what": "CE(ksize=5, stride_1=2, stride_2=1) on [1,64,128,128], shipped adaptive semantics: the run-time-geometry route "
                    "(dagl_ce_generic_forward; fp32 matrix cores, not tuned)", "ms_per_step": ms, "patches_per_s": L / (ms * 1e-3), "L": L, "N": N,
            "mean_degree": float(deg.mean()), "max_degree": int(deg.max()),
            "fp32_tflops": flop / (ms * 1e-3) / 1e12, "flop_note": "scores + A V + the two projections, 2 L N (D + P) + 2 (L + N) P D"}


def real_features_extra(dev):
    """REAL features: the seven 256 x 256 Set12 images (sigma 50, the reference's test protocol, scaled to [0, 1]) through the
    trained checkpoint's head conv and first eight ResBlocks, whole 256x256 map, one head.
    * top-k k = 8: natural-image scores are not spread like the synthetic map's -- the threshold sampled from every 8th key tile
      lets hundreds of keys per query through and the call lands on the fp32 redo pass; CE.topk_threshold = "auto" notices
      after the first call and takes the threshold from every second key tile.  All seven maps: worst / median, and the error of
      64 sampled queries' aggregated patches against the oracle (all 65 536 keys) on the first and the slowest map.
    * the shipped adaptive semantics on the same features (mask density 0.3-1.0, logits below ~70: no weight underflows to an exact
      zero, the dense kernel's zero-granule skip never fires): whole map, and the 64 leaf tiles of 72 x 72 that the reference's
      forward_chop feeds a head (DN_Gray/model/__init__.py:179-231) as one batch."""
    import numpy as np
    from dagl_amd import ops
    from dagl_amd.net import RR, chop_leaf_boxes, set12_protocol_noise
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    try:
        z = np.load(os.path.join(gdir, "quality_ckpt_fp16.npz"))
        imgs = np.load(os.path.join(gdir, "set12.npz"))
    except OSError as e:
        return {"what": "skipped", "error": str(e)}
    net = RR().eval()
    net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True)
    net = net.to(dev)
    names = sorted(n for n in imgs.files if imgs[n].shape[-1] == 256 and imgs[n].shape[-2] == 256)

    def features(name, leaf=False):
        clean = torch.from_numpy(imgs[name].astype(np.float32) / 255.0)[None, None]     # uint8 images; the network works on [0, 1]
        noisy = set12_protocol_noise(clean, 50.0, 1.0).to(dev)
        if leaf:
            noisy = torch.stack([noisy[0, :, y0:y1, x0:x1] for (y0, y1, x0, x1) in chop_leaf_boxes(256, 256)])
        x = net.head(noisy)
        for blk in net.body[:8]:
            x = blk(x)
        return x.contiguous()

    def parity(ce, x):
        from oracle.ce_oracle import ce_rows_oracle
        rows = torch.linspace(0, 4095, 64).long()
        b1, b2, thr, bias = ce._prologue(x)
        _, info = ops.ce_forward(b1.contiguous(), b2.contiguous(), thr.contiguous(), bias.contiguous(), ce.fc1[0].weight,
                                 ce.fc1[0].bias, ce.fc2[0].weight, ce.fc2[0].bias, mode="topk", k=8, debug=True, tight_topk=True)
        ref = ce_rows_oracle(x.cpu(), {n: q.detach().cpu() for n, q in ce.named_parameters()}, rows, mode="topk", k=8)
        agg = info["agg"][0].cpu()[rows].reshape(64, 7, 7, 16).movedim(-1, -3).reshape(64, 784)
        return float((agg - ref["agg"]).abs().max() / ref["agg"].abs().max())

    res = {"what": "one head on the features of the seven 256x256 Set12 images (sigma 50, [0,1]) after the trained RR's head conv + 8 "
                   "ResBlocks, whole 256x256 map", "L": 4096, "N": 65536}
    with torch.no_grad():
        ce = net.body[8].c1_1
        ce.select_mode, ce.select_k = "topk", 8
        per, feats = {}, {}
        for n in names:
            x = feats[n] = features(n)
            ce.topk_threshold = "auto"
            ce.reset_topk_policy()
            per[n] = _time_steps(lambda: ce(x), 10, 3, EXTRA_PREWARM_S)
        ms = sorted(per.values())
        worst = max(per, key=per.get)
        # the first call on a cold workspace under "auto": packs the weights, overflows under the sampled threshold, flips the policy
        # word and re-runs tight in-stream (round 3: the fp32 redo pass of every query group + a host poll)
        x0 = feats[names[0]]
        cold = []
        for _ in range(3):
            ce.topk_threshold = "auto"; ce.reset_topk_policy()
            torch.cuda.synchronize(); tcold = time.perf_counter()
            ce(x0)
            torch.cuda.synchronize(); cold.append((time.perf_counter() - tcold) * 1e3)
        ce.topk_threshold = "sparse"; ce.reset_topk_policy()
        res["topk8"] = {"ms_per_step_by_image": {n: round(v, 4) for n, v in per.items()}, "ms_per_step_worst": ms[-1],
                        "ms_per_step_median": ms[len(ms) // 2], "worst_image": worst, "patches_per_s_worst": 4096 / (ms[-1] * 1e-3),
                        "ms_per_step_sampled_threshold": _time_steps(lambda: ce(x0), 5, 2, 0.0),
                        "ms_first_call_cold_workspace_auto": min(cold),
                        "threshold": "auto: the workspace's policy word, flipped on the device by the first call (no host poll)",
                        "parity_err": {names[0]: parity(ce, feats[names[0]]), worst: parity(ce, feats[worst])},
                        "parity_note": "max|hip - oracle| / max|oracle| of 64 sampled queries' aggregated patches against all keys (bar 1e-4)"}
        res["ms_per_step"] = ms[-1]                          # (kept: the entry's headline = the WORST of the seven maps)
        res["patches_per_s"] = 4096 / (ms[-1] * 1e-3)
        # the fixed-k variant's OWN default, num_edge = 50 (GReccR2b_3mh_1-checkpoint.py:176): a few queries of a natural-image map
        # have one hot candidate segment; until the spill area (ScreenArgs::spill) their 128-query groups took the fp32 redo pass
        # every call (7.2 ms)
        ce.select_k = 50
        ce.topk_threshold = "auto"
        per50 = {}
        for n in names[:3]:
            x = feats[n]
            ce.reset_topk_policy()
            per50[n] = _time_steps(lambda: ce(x), 8, 3, EXTRA_PREWARM_S)
        res["topk50"] = {"what": "k = 50 (the variant's default num_edge) on the same features", "ms_per_step_by_image": {n: round(v, 4) for n, v in per50.items()},
                         "ms_per_step": max(per50.values())}
        ce.select_k = 8
        xl8 = features(names[1], leaf=True)
        ce.reset_topk_policy()
        msl8 = _time_steps(lambda: ce(xl8), 8, 3, EXTRA_PREWARM_S)
        res["topk8_leaf_tiles"] = {"what": f"the 64 leaf tiles of 72x72 of {names[1]} as one batch {list(xl8.shape)}, top-k 8 (maps of <= 16 384 keys start on the tight threshold)",
                                   "ms_per_step": msl8, "patches_per_s": xl8.shape[0] * 324 / (msl8 * 1e-3)}
        del xl8
        # the shipped adaptive semantics on the same features: dense formulation, nothing skipped
        ce.select_mode = "adaptive"
        dper = {}
        for n in names[:3]:
            x = feats[n]
            dper[n] = _time_steps(lambda: ce(x), 8, 3, EXTRA_PREWARM_S)
        info = ce.last_info or {}
        res["adaptive_dense"] = {"what": "shipped adaptive semantics, head c1_1, whole map: streamed dense formulation, every multiply runs",
                                 "ms_per_step_by_image": {n: round(v, 4) for n, v in dper.items()}, "ms_per_step": max(dper.values()),
                                 "selection_path": info.get("path")}
        del feats
        xl = features(names[0], leaf=True)
        msl = _time_steps(lambda: ce(xl), 8, 3, EXTRA_PREWARM_S)
        res["adaptive_dense_leaf_tiles"] = {"what": f"the 64 leaf tiles of 72x72 of {names[0]} (forward_chop) as one batch {list(xl.shape)}, head c1_1, adaptive",
                                            "ms_per_step": msl, "patches_per_s": xl.shape[0] * 324 / (msl * 1e-3), "L": 324, "N": 5184,
                                            "selection_path": (ce.last_info or {}).get("path")}
    del net, xl
    torch.cuda.empty_cache()
    return res


def train_extra(dev, B=8, crop=128, colors=3, steps=5, warmup=2, mode="topk"):
    """BASELINE configs[4] on one GPU inside the driver-timed command: RR (12 heads) fwd + bwd + Adam on synthetic crops
    [8,3,128,128], plus the roofline of the step's dominant matrix products -- the fc2 projection's two gradient products on the
    split-fp16 GEMM (gemm_roofline).  mode "topk": fixed-k 8 (neighbour lists); "adaptive": the shipped semantics from the seeded
    initialisation -- dense neighbourhoods, i.e. the streamed dense forward and the dense backward (dense_train.hip)."""
    from dagl_amd import ops
    from dagl_amd.ce import CE
    from dagl_amd.net import RR, seeded_state_dict
    from dagl_amd.train import TrainOptions, TrainStep, freeze_unused, make_optimizer
    net = RR(n_colors=colors)
    net.load_state_dict(seeded_state_dict(net.state_dict(), 7), strict=True)
    for m in net.modules():
        if isinstance(m, CE):
            if mode == "topk":
                m.select_mode, m.select_k = "topk", 8
            else:
                m.select_mode = mode
    net = net.to(dev)
    freeze_unused(net)
    opt = TrainOptions(task="dn_real", lr=1e-4)
    step = TrainStep(net, make_optimizer(net, opt), opt, generator=torch.Generator(device=dev).manual_seed(300))
    hr = torch.rand(B, colors, crop, crop, generator=torch.Generator().manual_seed(200)).to(dev)
    ms = _time_steps(lambda: step(hr), steps, warmup)
    infos = [m.last_info or {} for m in net.modules() if isinstance(m, CE)]
    del net, step
    torch.cuda.empty_cache()
    if mode != "topk":
        return {"what": "the same step with the shipped adaptive semantics from the seeded initialisation (every head in the dense "
                        "regime: streamed split-fp16 forward, dense backward with its five matrix products on the fp16 matrix cores, "
                        "split operands)",
                "ms_per_step": ms, "crops_per_s": B / (ms * 1e-3), "steps": steps, "warmup": warmup,
                "selection_paths": sorted({i.get("path") for i in infos if i.get("path") is not None})}
    return {"what": "BASELINE configs[4] (sparse regime) on one GPU: RR with 12 CE heads, crops [8,3,128,128], fwd + bwd + Adam, "
                    "fixed-k 8; no RCCL leg here (world size 1) -- `bench.py --train --gpus N` runs it under DDP",
            "ms_per_step": ms, "crops_per_s": B / (ms * 1e-3), "steps": steps, "warmup": warmup,
            "roofline": gemm_roofline(dev, B, crop)}


def mfma_sustained(dev, nominal_tflops):
    """What the matrix pipes hold on THIS box under the screen's multiply stream alone (dagl_probe_mfma_bf16: 16 waves per CU, 26
    v_mfma_f32_32x32x16_bf16 per step, operands in registers): TFLOP/s of a launch about as long as the screen kernel, timed with
    events, and the shader clock of its loop (s_memtime against the 100 MHz counter).  The nominal peak assumes 2.4 GHz."""
    import ctypes
    from dagl_amd import _lib, ops
    lib = _lib.load()
    blocks, steps = 256, 40                       # 256 x 16 waves x 40 x 26 multiplies ~ the screen's 3.4 M
    clocks = torch.zeros(2 * blocks, dtype=torch.int64, device=dev)
    sink = torch.zeros(1, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        def run():
            _lib.check(lib.dagl_probe_mfma_bf16(ops._stream(), blocks, steps, clocks.data_ptr(), sink.data_ptr()), "dagl_probe_mfma_bf16")
        for _ in range(20):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    c = clocks.cpu().numpy().reshape(blocks, 2).astype("float64")
    ghz = float((c[:, 0] / (c[:, 1] * 10.0)).mean())
    flop = blocks * 16 * steps * 26 * 32768.0
    tf = flop / (ms * 1e-3) / 1e12
    return {"mfma_only_tflops": tf, "frac_of_nominal": tf / nominal_tflops, "shader_clock_ghz": ghz, "ms_per_launch": ms,
            "what": "dagl_probe_mfma_bf16: the screen's multiply stream alone (256 blocks x 16 waves x 40 steps x 26 v_mfma_f32_32x32x16_bf16, "
                    "register operands), back-to-back launches timed with events; clock = s_memtime / s_memrealtime over wave 0's loop"}


def gemm_roofline(dev, B, crop):
    """The training step's dominant matrix products on their own, hipEvent-bracketed on the launch stream: the two gradient
    products of the fc2 projection -- d W = d Z^T rows ([196 x n] x [n x 784]) and d rows = d Z W ([n x 196] x [196 x 784]), n =
    B * crop^2 key patches -- as the step runs them since round 3: split-fp16 operands on the fp16 matrix cores
    (dagl_fc_grad16 -> fcg_* producers + gemm16s_kernel, train_ops.FAST_FC_BACKWARD).  `achieved` counts the ALGORITHMIC products
    (2 x 2 x 196 x 784 x n FLOP); the kernel executes three split products for each (`executed_frac`)."""
    from dagl_amd import _lib, ops
    lib = _lib.load()
    H = W = crop
    n = B * H * W
    O, K = 196, 784
    g = torch.Generator(device=dev).manual_seed(1)
    pmap = torch.zeros(B, H + 6, W + 6, 16, device=dev)
    pmap[:, 3:3 + H, 3:3 + W] = torch.randn(B, H, W, 16, device=dev, generator=g)
    w = (torch.rand(O, K, device=dev, generator=g) - 0.5) * 0.07
    dz = torch.randn(n, O, device=dev, generator=g) * 1e-4
    # (round 6) the call the training step makes: d rows folded inside the product (dagl_fc_grad16_dmap: d map out, no [n, 784] rows) where
    # the geometry allows, dagl_fc_grad16 (rows out; the fold is a separate kernel, not timed here) elsewhere
    folded = bool(lib.dagl_fc_grad16_dmap_ok(1, W))
    need = (lib.dagl_fc_grad16_dmap_scratch_bytes if folded else lib.dagl_fc_grad16_scratch_bytes)(B, H, W)
    scratch = torch.empty(need + 256, device=dev, dtype=torch.uint8)
    base = (scratch.data_ptr() + 255) // 256 * 256
    d_w = torch.empty(O, K, device=dev)
    d_out = torch.empty(B, H + 6, W + 6, 16, device=dev) if folded else torch.empty(n, K, device=dev)

    def call():
        _lib.check((lib.dagl_fc_grad16_dmap if folded else lib.dagl_fc_grad16)(
            ops._stream(), B, H + 6, W + 6, 1, 0, 0, H, W, pmap.data_ptr(), w.data_ptr(), None, dz.data_ptr(), d_w.data_ptr(), None,
            d_out.data_ptr(), base, need), "dagl_fc_grad16")
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 8
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    e1.synchronize()
    g_ms = e0.elapsed_time(e1) / reps
    flop = 2.0 * 2.0 * O * K * n
    ach = flop / (g_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": ("dagl_fc_grad16_dmap" if folded else "dagl_fc_grad16") +
                                      " (fc2 backward: d W = d Z^T rows [196 x %d] x [%d x 784] and d rows = d Z W%s, split-fp16 "
                                      "operands, gemm16s_kernel v_mfma_f32_32x32x16_f16 + its operand producers)"
                                      % (n, n, " folded to d map inside the product" if folded else ""),
            "achieved": ach, "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MATRIX_TFLOPS,
            "executed_frac": 3.0 * ach / PEAK_BF16_MATRIX_TFLOPS, "fp32_matrix_peak_equivalent": ach / PEAK_F32_MATRIX_TFLOPS,
            "flop_per_launch": flop, "ms_per_launch": g_ms, "traffic": committed_traffic("train:gemm16s_kernel"),
            "traffic_note": "HBM-side bytes of ONE gemm16s_kernel launch (the larger of the call's two products) from the committed "
                            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --train` (profiles/rNN_traffic.json, key "
                            "train:gemm16s_kernel); not measured in this run",
            "note": "algorithmic FLOP of both products / time of the whole call (producers included); x3 = executed fp16 products "
                    "against the 2.5 PF fp16 peak; the same FLOP against the 157.3 TF fp32 matrix peak in fp32_matrix_peak_equivalent",
            "calls_per_step": "12 heads x (fc2 on the key rows; fc1 on the query rows is 1/16 of it)"}


TRAFFIC_SOURCE = ("HBM bytes per launch read from the newest committed profiles/rNN_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / "
                  "WRITE_SIZE passes of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes): PMC counters cannot be "
                  "read inside the timed process, so this field is NOT measured in this run")


def committed_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc pass (the newest profiles/rNN_traffic.json): PMC
    counters cannot be read from inside the timed process, so the number comes from the profile of this same command."""
    import glob
    try:
        files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_traffic.json")))
        t = json.load(open(files[-1]))
        return t.get(kernel_key)
    except Exception:
        return None


def quality_leg(dev):
    """PSNR on Set12 sigma=50 of the full network (12 HIP heads) vs the reference forward with the same regenerable
    weights (numbers committed by tests/golden/make_set12_psnr.py, which ran the reference on CPU)."""
    import numpy as np
    from dagl_amd.net import RR, chop_forward_batched, psnr, seeded_state_dict, set12_protocol_noise
    gdir = os.path.join(REPO, "tests", "golden")
    trained = os.path.exists(os.path.join(gdir, "quality_ckpt_fp16.npz")) and os.path.exists(os.path.join(gdir, "set12_psnr_ref_trained.json"))
    ref = json.load(open(os.path.join(gdir, "set12_psnr_ref_trained.json" if trained else "set12_psnr_ref.json")))
    imgs = np.load(os.path.join(gdir, "set12.npz"))
    net = RR().eval()
    if trained:      # 400 DN_Gray steps from the seeded init (tools/train_quality_ckpt.py), float16 values on both sides
        z = np.load(os.path.join(gdir, "quality_ckpt_fp16.npz"))
        net.load_state_dict({k: torch.from_numpy(z[k].astype(np.float32)) for k in z.files}, strict=True)
    else:
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
            "weights": ("tests/golden/quality_ckpt_fp16.npz: RR after 400 DN_Gray training steps on the HIP path from the seeded init "
                        "(Set12 crops, sigma 50; float16 values loaded by both sides) -- no trained weights ship with the reference"
                        if trained else f"regenerable stand-in checkpoint (numpy PCG64 seed {ref['seed']})"),
            "psnr_noisy_mean_db": sum(r["psnr_noisy"] for r in ref["images"].values()) / len(ref["images"]),
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
    devices = rank_devices(dist, torch.cuda.get_device_name(dev))
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
                "value": world * B * args.steps / elapsed, "unit": "crops/s", "n_gpus": world,
                "n_ranks_seen": dist.get_world_size() if dist is not None else 1, "devices": devices, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[4] (sparse regime): RR with 12 CE heads, crops [{B},{args.colors},"
                                       f"{args.crop},{args.crop}] per GPU, MSE(sum)/(2B) loss, Adam, gradients of "
                                       f"{n_par} parameters all-reduced over RCCL",
                           "parallelism": f"ddp{world}", "select_mode": args.mode, "k": args.k, "batch_per_gpu": B},
                "roofline": gemm_roofline(dev, B, args.crop), "cpu_baseline": None,
                "allreduce_ms": allreduce_ms, "allreduce_bytes": 4 * n_par if dist is not None else None,
                "ddp": {"what": "dagl_amd.train.wrap_ddp: torch DistributedDataParallel over RCCL, one process per GPU",
                        "bucket_cap_mb": 12, "gradient_as_bucket_view": True, "broadcast_buffers": False,
                        "find_unused_parameters": False,
                        "why": "every trainable parameter gets a gradient in every step: freeze_unused() takes the heads' never-applied "
                               "`W` convolution (dagl.py:199-201; registered for checkpoint compatibility) and, in the fixed-k mode, "
                               "the unused thr / bias heads out of the reducer (requires_grad = False), so no unused-parameter scan; "
                               "the gradients travel in ~2 buckets of <= 12 MB (xGMI rings are per-link bound: few large messages), "
                               "the first while the backward of the first half of the trunk is still running",
                        "gradient_bytes": 4 * n_par,
                        "expected_allreduce_ms_at_8_gpus": round(2.0 * (7.0 / 8.0) * 4 * n_par / 153e9 * 1e3, 3),
                        "expected_note": "ring all-reduce over xGMI: 2 (N-1)/N x bytes over one ~153 GB/s link direction per hop, N = 8, "
                                         "latency terms ignored -- an expectation, NOT a measurement: no multi-GPU node was available to "
                                         "this repository's rounds; `allreduce_ms` above is the live figure of THIS run's world size"},
                "loss_first_last": [float(losses[0]), float(losses[-1])] if losses else None}
        print(json.dumps(line))


def main():
    args = parse()
    from dagl_amd.launch import launched_by_torchrun, spawn_ranks
    if not launched_by_torchrun() and (args.gpus > 1 or os.environ.get("DAGL_BENCH_FORCE_SPAWN")):
        # plain ``python bench.py --gpus N``: start the N ranks ourselves (one process per GPU), rank 0 prints the line
        if not args.stub:
            visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if visible < args.gpus:
                raise SystemExit(f"bench.py: {args.gpus} GPUs requested, {visible} visible on this node")
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks "
                         f"(use --nproc-per-node {args.gpus}, or drop the launcher: bench.py spawns its own ranks)")
    if args.stub:
        return stub_bench(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank}, {torch.cuda.device_count()} GPUs visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("DAGL_BENCH_FORCE_DIST") or os.environ.get("DAGL_SPAWNED"):   # (the switches exercise the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)       # RCCL over xGMI: used for the barriers/max only

    if args.train:
        from dagl_amd import ops as _ops, train_ops as _train_ops
        _ops.DENSE_BACKWARD_FP32 = args.dense_backward == "fp32"                    # A/B switches of the training routes
        _train_ops._FORCE_UNFOLD_BACKWARD = args.prologue_backward == "unfold"
        train_bench(args, dev, dist, world, rank)
        if dist is not None:
            dist.destroy_process_group()
        return

    from dagl_amd import ops
    from dagl_amd.ce import CE
    from dagl_amd.shard import rank_seed, reduce_max_seconds
    from dagl_amd.synth import make_ce_params, make_features

    mode, k = args.mode, (args.k if args.mode != "adaptive" else 0)
    variant = ("default" if mode == "topk" else "sparse") if args.variant == "auto" else args.variant
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(args.wseed, variant=variant, sparse_gain=args.sparse_gain).items()}
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode = mode
    ce.scan = args.scan
    if k:
        ce.select_k = k
    ce.topk_redo = args.topk_redo
    ce = ce.to(dev).eval()

    B, H, W = args.batch, args.size, args.size
    x = torch.from_numpy(make_features(rank_seed(100, rank) if args.fseed is None else args.fseed + rank, B, 64, H, W)).to(dev)     # resident in HBM
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
            # the module's protocol (dagl_amd/net.py CES._stage): the first call on a workspace packs the heads' weights and
            # zeroes the map borders there, later calls with the same weights / shape / workspace say so
            wsb = ws_stage.peek(x.device)
            packed = info_box.get("ws_ptr") is not None and wsb is not None and wsb.data_ptr() == info_box["ws_ptr"]
            out, inf = ops.ces_stage_forward(x, prm, mix_w, mix_b, mode=mode, k=ce.select_k, workspace=ws_stage, profile=profile,
                                             weights_packed=packed)
            assert out is not None, "dense neighbourhoods: the stage entry point handed the call back"
            info_box["info"] = inf
            info_box["ws_ptr"] = ws_stage.peek(x.device).data_ptr()
    else:
        def step(profile=None):
            ce.profile = profile
            ce(x)
    with torch.no_grad():
        # the COLD figure first (what --prewarm 0 reports): W warm-up steps on a GPU that sat idle while the case was built, then K
        # timed steps -- kept beside the sustained figure in the line ("ms_per_step_cold")
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - tc) / max(args.steps, 1) * 1e3
        prewarm_steps = _prewarm(step, args.prewarm)
        for _ in range(args.warmup):
            step()
        # timed region: only the dominant kernel is bracketed by hipEvents (on the launch stream); an event record costs
        # ~4 us of stream time, so the full nine-boundary stage profile is taken in a separate pass below
        dense_regime = (not args.stage) and mode == "adaptive" and (ce.last_info or {}).get("path") == 4
        prof.select_stage("gather" if dense_regime else "select")
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
        dominant_ms = [c[6 if dense_regime else 4] for c in prof.read()]
        prof.select_stage(-1)
        for _ in range(min(args.steps, 20)):
            step(prof)
        torch.cuda.synchronize()
        ce.profile = None
    elapsed = reduce_max_seconds(elapsed, dist, dev)
    devices = rank_devices(dist, torch.cuda.get_device_name(dev))
    stage_ms = prof.read()
    info = info_box["info"] if args.stage else ce.last_info

    # stand-alone gather kernel over materialised value rows (rank 0 only; outside the timed region).  The launches rotate
    # over GATHER_SETS distinct (idx, wgt, rows, out) sets: 4 x (205 MB of rows + 12.8 MB out) = 0.87 GB at 256^2, more than
    # three times the 256 MiB Infinity Cache, so no launch finds its rows on the die from the previous one.
    gather = None
    if rank == 0:
        GATHER_SETS = 4
        with torch.no_grad():
            pf = ce._params_f32()                                   # (the library's own prologue: no MIOpen kernel in the profile of this command)
            _b1p, b2p, _thr, _bias = ops.ce_prologue(x[:1].contiguous(), pf["g.weight"], pf["g.bias"], pf["theta.weight"], pf["theta.bias"])
            kk = k or 8
            g = torch.Generator(device="cpu").manual_seed(5)
            sets = []
            for si in range(GATHER_SETS):
                rows = ops.unfold_values((b2p * (1.0 + 0.25 * si)).contiguous(), H, W)[0].contiguous()   # [N,784] (zero border stays zero)
                idx = torch.randint(0, N, (L, kk), generator=g, dtype=torch.int32).to(dev)
                wgt = torch.rand(L, kk, generator=g).to(dev)
                sets.append((idx, wgt, rows))
            for i in range(2 * GATHER_SETS):
                ops.gather_aggregate(*sets[i % GATHER_SETS])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 48
            e0.record()
            for i in range(reps):
                ops.gather_aggregate(*sets[i % GATHER_SETS])
            e1.record()
            e1.synchronize()
            g_ms = e0.elapsed_time(e1) / reps
            # the same launches on ONE set (cache-resident rows), for comparison only
            e0.record()
            for i in range(reps):
                ops.gather_aggregate(*sets[0])
            e1.record()
            e1.synchronize()
            g_ms_hot = e0.elapsed_time(e1) / reps
            g_bytes = L * ((kk + 1) * 4 * P_ROW + 8 * kk)
            set_bytes = sets[0][2].numel() * 4 + L * P_ROW * 4
            gather = {"bound": "hbm", "kernel": "gather_rows_kernel (dagl_gather_aggregate)", "k": kk,
                      "achieved": g_bytes / (g_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": g_bytes / (g_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                      "traffic": committed_traffic("gather_rows_kernel") if (H, kk) == (256, 8) else None,
                      "bytes_per_launch": g_bytes, "ms_per_launch": g_ms,
                      "working_set_bytes": GATHER_SETS * set_bytes,
                      "same_set_ms_per_launch": g_ms_hot,
                      "timing": f"mean of {reps} back-to-back launches (launch gaps included) rotating over {GATHER_SETS} distinct "
                                f"(idx, rows) sets = {GATHER_SETS * set_bytes / 2**20:.0f} MiB > Infinity Cache (256 MiB); torch events "
                                "on the launch stream; same_set_ms_per_launch = the same loop on one cache-resident set"}
            del sets, rows

    if rank == 0:
        import numpy as np
        sm = np.asarray(stage_ms, dtype=np.float64)                      # [steps, 8]
        mean_ms = sm.mean(axis=0) if len(sm) else np.zeros(8)
        from dagl_amd._lib import STAGE_NAMES
        sel_ms = float(np.mean(dominant_ms)) if len(dominant_ms) else float(mean_ms[4])     # from the timed steps
        flops = 2.0 * heads_per_step * B * L * N * D_FEAT                # algorithmic: 2*L*N*D per image and head
        ach = flops / (sel_ms * 1e-3) / 1e12 if sel_ms > 0 else 0.0
        screened = (info or {}).get("path") == 3 and not dense_regime
        peak = PEAK_BF16_MATRIX_TFLOPS if screened else PEAK_F32_MATRIX_TFLOPS
        roofline = {"bound": "mfma",
                    "kernel": "screen_ring_kernel<1> (bf16 v_mfma_f32_32x32x16_bf16, full L*N candidate filter; screen_kernel<1> for "
                              "256-query blocks)" if screened
                              else "score_select_kernel (fp32 v_mfma_f32_32x32x2_f32)",
                    "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "traffic": (committed_traffic("screen_ring_kernel<1>") or committed_traffic("screen_kernel<1>")) if (screened and (H, mode, k, B) == (256, "topk", 8, 1)) else None,
                    "flop_per_launch": flops, "ms_per_launch": sel_ms}
        if dense_regime:
            roofline = dense_roofline(sel_ms, B, L, N, info, "mean over the timed steps, two hipEvents around the stage on the launch stream")
        if roofline["traffic"] is not None:
            roofline["traffic_source"] = TRAFFIC_SOURCE
        if gather is not None and gather.get("traffic") is not None:
            gather["traffic_source"] = TRAFFIC_SOURCE
        if screened:
            roofline["sustained"] = mfma_sustained(dev, peak)
            if roofline["sustained"]:
                roofline["frac_of_sustained"] = ach / roofline["sustained"]["mfma_only_tflops"]
        if gather is not None and mean_ms[6] > 0 and not dense_regime:
            kk = k or max(1, int(round((info or {}).get("total_edges", 0) / max(1, B * L))))
            fb = B * L * ((kk + 1) * 4 * P_ROW + 8 * kk)
            gather["fused_in_block"] = {"kernel": "aggregate_fold_kernel (top-k: gather + weighted sum + fold in one kernel; value patches "
                                                  "read from the 4 MB map, so this is a cache number)" if mode != "adaptive"
                                                  else "aggregate_direct_kernel", "ms_per_launch": float(mean_ms[6]),
                                        "achieved": fb / (mean_ms[6] * 1e-3) / 1e9, "unit": "GB/s",
                                        "frac": fb / (mean_ms[6] * 1e-3) / 1e9 / PEAK_HBM_GBS}
        total_patches = world * heads_per_step * B * L * args.steps
        line = {
            "metric": "graph-attn fwd query-patches/s @256x256x64 k=8" if (H, mode, k, args.stage) == (256, "topk", 8, False)
                      else f"graph-attn fwd query-patches/s @{H}x{W}x64 {mode} k={k}",
            "value": total_patches / elapsed, "unit": "patches/s", "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if dist is not None else 1, "devices": devices, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ms_per_step_cold": cold_ms,
            "prewarm": {"seconds": args.prewarm, "untimed_steps": prewarm_steps,
                        "note": "untimed passes of the same step before the W warm-up steps: an idle GPU runs its first tens "
                                "of ms below the sustained clock (--prewarm 0 shows the cold figure)"},
            "config": {"workload": ("one CES stage (4 heads + 1x1 mix + residual)" if args.stage else
                                    ("BASELINE configs[1]: one CE head forward" if (H, W, mode, k, B) == (256, 256, "topk", 8, 1)
                                     else "one CE head forward"))
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
        # The optional legs below run the trunk's stock convolutions (MIOpen).  On a box whose MIOpen start-up is slow (cold kernel
        # database on slow storage: one pool box in round 6 spent 313 s in the quality leg that takes 4.6 s elsewhere) they would
        # turn a two-minute command into ten: one probe convolution decides.
        trunk_ok, probe_s = True, 0.0
        if world == 1 and not args.stage and not (args.no_quality and args.no_extra):
            probe_s = miopen_probe_seconds(dev, args.miopen_probe_limit)
            trunk_ok = probe_s != float("inf")
            line["miopen_probe_s"] = probe_s if trunk_ok else None
        if world == 1 and not args.no_quality and not args.stage:
            line["quality"] = quality_leg(dev) if trunk_ok else {
                "skipped": f"a child process was not through its first stock convolutions after {args.miopen_probe_limit:.0f} s: MIOpen is cold on "
                           "this box; tests/test_gpu_set12_psnr.py asserts the same figure (max |delta PSNR| <= 0.02 dB)"}
        if world == 1 and not args.no_extra and not args.stage and (H, mode, k, B) == (256, "topk", 8, 1):
            line["extra_configs"] = extra_configs(dev, trunk_ok=trunk_ok)
        if world == 1 and not args.no_cpu_baseline and not args.stage:
            with torch.no_grad():
                hip_out = ce(x[:1]).cpu()
            line["cpu_baseline"] = cpu_baseline(args, params, mode, k, x_cpu=x[:1].cpu(), hip_out=hip_out)
            if "parity_err" in line["cpu_baseline"]:
                line["parity_err"] = line["cpu_baseline"]["parity_err"]
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
