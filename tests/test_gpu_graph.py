"""The top-k forward makes no host round trip, so it can be captured into a HIP graph and replayed (launch-bound small
tiles: one graph launch instead of ~15 kernel launches)."""
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu


def test_topk_forward_replays_from_a_hip_graph():
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    dev = torch.device("cuda:0")
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(71, variant="default").items()}
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode, ce.select_k = "topk", 8
    ce = ce.to(dev).eval()
    x_static = torch.from_numpy(make_features(71, 4, 64, 72, 72)).to(dev)
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up on the capture stream: workspace, packed weights
            for _ in range(3):
                eager = ce(x_static).clone()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_static = ce(x_static)
        for seed in (72, 73):
            x_new = torch.from_numpy(make_features(seed, 4, 64, 72, 72)).to(dev)
            x_static.copy_(x_new)
            graph.replay()
            torch.cuda.synchronize()
            got = out_static.clone()
            want = ce(x_new)
            assert normwise(got.cpu().numpy(), want.cpu().numpy()) <= 1e-6
        x_static.copy_(torch.from_numpy(make_features(71, 4, 64, 72, 72)).to(dev))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_static, eager)
