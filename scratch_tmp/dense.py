import sys, time, torch
sys.path.insert(0, "/root/repo")
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
for size in (64, 128, 256):
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(2024, variant="default").items()}
    ce = CE(in_channels=64); ce.load_state_dict(params, strict=True)
    ce = ce.cuda().eval()
    x = torch.from_numpy(make_features(100, 1, 64, size, size)).cuda()
    with torch.no_grad():
        for _ in range(2): ce(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5
        for _ in range(n): ce(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    L = ((size + 3) // 4) ** 2
    print(size, "ms %.2f" % (dt * 1e3), "patches/s %.0f" % (L / dt), ce.last_info)
