#!/usr/bin/env python3
"""Run ON THE GPU BOX: the headline forward ([1,64,256,256], top-k 8) eagerly against replayed from a HIP graph (torch.cuda.CUDAGraph),
interleaved on one box; also the 64 leaf tiles [64,64,72,72] and the dense regime (hinted adaptive calls without statistics)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
dev = torch.device("cuda:0")


def timed(fn, n):
    for _ in range(10):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, shape, mode, k, variant, fseed in (("headline", (1, 64, 256, 256), "topk", 8, "default", 100000),
                                            ("leaf tiles", (64, 64, 72, 72), "topk", 8, "default", 100000)):
    ce = CE(in_channels=64)
    ce.load_state_dict({n: torch.from_numpy(a) for n, a in make_ce_params(2024, variant=variant, sparse_gain=2.0).items()}, strict=True)
    ce.select_mode, ce.select_k = mode, k
    ce = ce.to(dev).eval()
    x = torch.from_numpy(make_features(fseed, *shape)).to(dev)
    with torch.no_grad():
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(80):
                ref = ce(x).clone()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = ce(x)
        g.replay(); torch.cuda.synchronize()
        assert torch.equal(out, ref)
        t = time.perf_counter()
        while time.perf_counter() - t < 0.5:
            for _ in range(20): ce(x)
            torch.cuda.synchronize()
        for r in range(3):
            print(f"{name}: eager {timed(lambda: ce(x), 300):.4f} ms, graph replay {timed(g.replay, 300):.4f} ms", flush=True)
