import sys, time, torch
sys.path.insert(0, "/root/repo")
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
dev = torch.device("cuda:0")
params = {n: torch.from_numpy(a) for n, a in make_ce_params(71, variant="default").items()}
for (B, S) in ((1, 72), (4, 72), (16, 72), (1, 128), (1, 256)):
    ce = CE(in_channels=64); ce.load_state_dict(params, strict=True); ce.select_mode, ce.select_k = "topk", 8
    ce = ce.to(dev).eval()
    x = torch.from_numpy(make_features(71, B, 64, S, S)).to(dev)
    with torch.no_grad():
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): ce(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): out = ce(x)
        def t(fn, n=200):
            for _ in range(20): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
        te = t(lambda: ce(x)); tg = t(g.replay)
    L = ((S + 3) // 4) ** 2 * B
    print(f"B={B} {S}x{S}: eager {te:7.1f} us  graph {tg:7.1f} us  ({L/tg:.2f} M patches/s graphed)")
