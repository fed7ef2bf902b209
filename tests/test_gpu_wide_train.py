"""Top-k modes with min(k, N) beyond the lists' width (DAGL_MAX_TOPK = 64) UNDER AUTOGRAD: the dense formulation with the row-wise
selection of the k best scores as its mask (dagl_ce_core_wide_forward / _backward; dense_train.hip, wide_select.h), against the
fp64 oracle and its autograd.  Reference: top_k = min(num_edge, N), GReccR2b_3mh_1-checkpoint.py:242-250; a stray sibling takes 500
(CA_model-checkpoint.py:134-143)."""
import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle(x, prm, G, mode, k):
    from oracle.ce_oracle import ce_forward_oracle
    xx = x.double().requires_grad_(True)
    P = {n: t.double().requires_grad_(True) for n, t in prm.items()}
    out = ce_forward_oracle(xx, P, mode=mode, k=k, dtype=torch.float64)
    (out * G.double()).sum().backward()
    grads = {"d_x": xx.grad}
    grads.update({"d_" + n: t.grad for n, t in P.items() if t.grad is not None})
    return out.detach(), grads


@pytest.mark.parametrize("B,H,W,mode,variant,k", [(2, 24, 28, "topk", "default", 100), (1, 36, 40, "topk", "default", 500),
                                                  (1, 20, 24, "topk", "default", 5000),          # k > N: every key
                                                  (2, 24, 28, "adaptive_topk", "default", 100),
                                                  (1, 33, 30, "adaptive_topk", "sparse", 80)])
def test_wide_topk_trains_and_matches_the_fp64_oracle_autograd(B, H, W, mode, variant, k):
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    seed = 300 + H + k
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(seed, variant=variant, sparse_gain=1.2).items()}
    x = torch.from_numpy(make_features(seed + 1, B, 64, H, W))
    G = torch.randn(B, 16, H, W, generator=torch.Generator().manual_seed(seed))
    want, g64 = _oracle(x, prm, G, mode, k)
    dev = _dev()
    ce = CE(in_channels=64)
    ce.load_state_dict(prm, strict=True)
    ce.select_mode, ce.select_k = mode, k
    ce = ce.to(dev).train()
    xg = x.to(dev).requires_grad_(True)
    out = ce(xg)
    (out * G.to(dev)).sum().backward()
    e = normwise(out.detach().cpu().numpy(), want.numpy())
    print(f"wide {mode} k={k} [{B},{H},{W}]: forward {e:.2e} from the fp64 oracle")
    assert e <= 1e-4
    got = {"d_x": xg.grad}
    got.update({"d_" + n: p.grad for n, p in ce.named_parameters() if p.grad is not None})
    assert "d_W.weight" not in got
    if mode == "topk":                                   # a 0/1 mask: the threshold heads get no gradient, as in the oracle
        assert "d_thr_conv.weight" not in got and "d_thr_conv.weight" not in g64
    for name, w in g64.items():
        err = normwise(got[name].cpu().numpy(), w.numpy())
        print(f"   {name}: {err:.2e}")
        assert err <= 1e-3, name
    # the same module under no_grad (row-wise inference form, topk_wide.hip) agrees with its training forward
    ce.eval()
    with torch.no_grad():
        inf = ce(x.to(dev))
    assert normwise(inf.cpu().numpy(), out.detach().cpu().numpy()) <= 2e-5


def test_wide_topk_ties_at_the_kth_place_go_to_the_lower_key():
    """A map of identical pixels: every score of a row is the same number, the k best are the k LOWEST key indices (torch.topk on
    the oracle's CPU rows and the list path do the same); forward and gradients against the fp64 oracle."""
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(77, variant="default").items()}
    B, H, W, k = 1, 18, 20, 90
    x = torch.full((B, 64, H, W), 0.37)
    x[:, :, 5:9, 7:12] += 0.05                            # (a patch of other pixels: rows with two distinct score values and ties in both)
    G = torch.randn(B, 16, H, W, generator=torch.Generator().manual_seed(5))
    want, g64 = _oracle(x, prm, G, "topk", k)
    dev = _dev()
    ce = CE(in_channels=64)
    ce.load_state_dict(prm, strict=True)
    ce.select_mode, ce.select_k = "topk", k
    ce = ce.to(dev).train()
    xg = x.to(dev).requires_grad_(True)
    out = ce(xg)
    (out * G.to(dev)).sum().backward()
    # (fp32 scores of identical patches are identical numbers on both sides only where the SAME products are summed in the same order:
    # the oracle's fp64 ties are the HIP path's fp32 ties here because the patches are bit-identical)
    assert normwise(out.detach().cpu().numpy(), want.numpy()) <= 1e-4
    assert normwise(xg.grad.cpu().numpy(), g64["d_x"].numpy()) <= 1e-3


def test_wide_core_is_bit_reproducible_and_chunk_invariant():
    from dagl_amd import ops
    g = torch.Generator().manual_seed(3)
    B, H, W, k = 2, 30, 26, 200
    L, N = 8 * 7, H * W
    dev = _dev()
    wq = (torch.rand(B, L, 196, generator=g) * 0.1).to(dev); xr = (torch.rand(B, N, 196, generator=g) * 0.1).to(dev)
    b2 = torch.randn(B, 16, H, W, generator=g).to(dev); G = torch.randn(B, 16, H, W, generator=g).to(dev)
    o1, _ = ops.ce_core_wide_forward(wq, xr, b2, None, None, "topk", k)
    g1 = ops.ce_core_wide_backward(G, wq, xr, b2, None, None, "topk", k)
    o2, _ = ops.ce_core_wide_forward(wq, xr, b2, None, None, "topk", k)
    g2 = ops.ce_core_wide_backward(G, wq, xr, b2, None, None, "topk", k)
    assert torch.equal(o1, o2) and all(torch.equal(a, b) for a, b in zip(g1[:3], g2[:3]))
    o3, info = ops.ce_core_wide_forward(wq[1:], xr[1:], b2[1:], None, None, "topk", k, want_info=True)
    assert torch.equal(o3, o1[1:]) and info["max_degree"] == k and info["total_edges"] == L * k
