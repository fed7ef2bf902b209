import sys, time, torch
sys.path.insert(0, "/root/repo")
from dagl_amd import ops
from dagl_amd.ce import CE
from dagl_amd.synth import make_ce_params, make_features
params = {n: torch.from_numpy(a) for n, a in make_ce_params(2024, variant="default").items()}
ce = CE(in_channels=64); ce.load_state_dict(params, strict=True); ce.select_mode = "topk"; ce.select_k = 8
ce = ce.cuda().eval()
x = torch.from_numpy(make_features(100, 1, 64, 256, 256)).cuda()
prof = ops.StageProfile(200)
def run(n, p):
    ce.profile = p
    with torch.no_grad():
        for _ in range(10): ce(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): ce(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    prof.reset(); a = run(100, prof); b = run(100, None)
    print("ms/step with stage events %.4f, without %.4f" % (a, b))
