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


def _sparse_adaptive_module(seed=41, gain=1.95):
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant="sparse", sparse_gain=gain).items()}
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode = "adaptive"
    return ce.to("cuda:0").eval(), params


def test_adaptive_forward_stops_waiting_for_its_verdict_and_replays_from_a_hip_graph():
    """Shipped (adaptive) semantics at a mean degree of ~8: after four calls served in-stream the module no longer reads the
    verdict back (DAGL_FLAG_NO_WAIT) -- same numbers, no host synchronisation -- and the call can be captured and replayed."""
    from dagl_amd.synth import make_features
    dev = torch.device("cuda:0")
    ce, _ = _sparse_adaptive_module()
    x_static = torch.from_numpy(make_features(41, 1, 64, 128, 128)).to(dev)
    with torch.no_grad():
        want = ce(x_static).clone()                                           # (default: the verdict is read every call)
        assert ce.last_info["path"] == 3 and 2 < ce.last_info["total_edges"] / 1024 < 64      # sparse, served by lists (+ redo)
        ce.adaptive_sync = "auto"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(6):
                eager = ce(x_static).clone()
        torch.cuda.current_stream().wait_stream(side)
        assert ce._served_streak >= 4 and ce._nowait_calls >= 1              # the last calls did not wait
        assert torch.equal(eager, want)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_static = ce(x_static)
        for seed in (42, 43):
            x_new = torch.from_numpy(make_features(seed, 1, 64, 128, 128)).to(dev)
            x_static.copy_(x_new)
            graph.replay()
            torch.cuda.synchronize()
            got = out_static.clone()
            ce.adaptive_sync = "always"
            ref = ce(x_new)
            ce.adaptive_sync = "auto"
            assert torch.isfinite(got).all() and normwise(got.cpu().numpy(), ref.cpu().numpy()) <= 1e-6


def test_unserved_no_wait_call_is_nan_filled_and_reported():
    """The device-side verdict: a call that did not wait and met dense neighbourhoods (the host would have sent it to the dense
    formulation) returns NaN -- never the clipped lists' numbers --, the sticky word reports it, the module waits again."""
    import warnings
    from dagl_amd.synth import make_features
    from oracle.ce_oracle import ce_forward_oracle
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params
    dev = torch.device("cuda:0")
    # default-initialised heads: ~95 % of the keys pass on a plain N(0,1) input (dense regime); the same input shifted along the
    # threshold head's weights raises thr to ~3 and leaves a mean degree of ~10 (served by the lists + the per-query redo)
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(41, variant="default").items()}
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce = ce.to(dev).eval()
    ce.adaptive_sync = "auto"
    x_dense = torch.from_numpy(make_features(41, 1, 64, 64, 64))
    w = params["thr_conv.weight"][0].sum(dim=(1, 2))                                  # [64]
    x = (x_dense + 0.9 * torch.sign(w)[None, :, None, None]).to(dev)
    with torch.no_grad():
        for _ in range(5):
            ce(x)
            assert ce.last_info["path"] == 3
        assert ce._served_streak >= 4
        out = ce(x_dense.to(dev))                                                     # no wait: unserved
        assert torch.isnan(out).all()
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            assert not ce.range_ok()
        assert any("did not wait" in str(m.message) for m in wlist) and ce._served_streak == 0 and ce.scan == "screened"
        out = ce(x_dense.to(dev)).cpu()                                               # waits again: dense formulation
    assert ce.last_info["path"] == 4
    want = ce_forward_oracle(x_dense, params, mode="adaptive", dtype=torch.float64).float()
    assert normwise(out.numpy(), want.numpy()) <= 1e-4


def test_one_out_of_range_replay_does_not_poison_the_later_ones():
    """A captured call carries its range-guard tag as a kernel argument: every replay has the same one.  A replay that leaves the
    split-fp16 range (since round 6: a non-finite input; |activation| >= 3750 up to round 5) is NaN-filled and leaves the tag in the sticky word; before round 4's last fix every LATER
    replay found its own tag there and was NaN-filled too.  Now the first launch of a captured call re-labels the word ("an earlier
    call", still non-zero): the next replay is served, the poll still reports the violation."""
    import warnings
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    dev = torch.device("cuda:0")
    params = {n: torch.from_numpy(a) for n, a in make_ce_params(75, variant="default").items()}
    ce = CE(in_channels=64)
    ce.load_state_dict(params, strict=True)
    ce.select_mode, ce.select_k = "topk", 8
    ce = ce.to(dev).eval()
    good = torch.from_numpy(make_features(75, 2, 64, 48, 48)).to(dev)
    x_static = good.clone()
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                want = ce(x_static).clone()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out_static = ce(x_static)
        graph.replay(); torch.cuda.synchronize()
        assert torch.equal(out_static, want)
        x_static.copy_(good * 3.0e4)                         # far outside the fine tier: served by the same captured launches (round 6)
        graph.replay(); torch.cuda.synchronize()
        assert torch.isfinite(out_static).all()
        x_static.copy_(good)
        x_static[1, 7, 20, 21] = float("inf")                # what the guard is left with: a non-finite input
        graph.replay(); torch.cuda.synchronize()
        assert torch.isnan(out_static).all()                 # never numbers computed from inf halves
        x_static.copy_(good)
        graph.replay(); torch.cuda.synchronize()
        assert torch.equal(out_static, want)                 # (was: NaN from here on)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            assert not ce.range_ok()                         # the sticky report survives the re-labelling
