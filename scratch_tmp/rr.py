import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from dagl_amd.net import RR, seeded_state_dict, chop_forward
net = RR(n_colors=1)
net.load_state_dict(seeded_state_dict(net.state_dict(), 7), strict=True)
net = net.cuda().eval()
x = torch.rand(1, 1, 256, 256, device="cuda")
with torch.no_grad():
    for _ in range(2): chop_forward(net, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): chop_forward(net, x)
    torch.cuda.synchronize()
print("chop_forward 256x256: %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3))
