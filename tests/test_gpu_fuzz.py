"""Random (batch, H, W, k, mode) cases through the MODULE (bf16 screen, device-side threshold policy) against the fp64 oracle: the
36 cases of tools/fuzz_shapes.py -- the sweep that found two of round 4's three last-day cliffs from outside the suite -- now inside
it.  Batches: up to four images of the batch are checked on the oracle (images are independent, dagl.py:245).  What must not
appear: an output further than 1e-4 from the fp64 oracle (1e-3 where the oracle's own k-th and (k+1)-th scores of some query lie
within 1e-6 relative of each other: either key is then a legitimate k-th neighbour), or redo work (the fp32 pass behind the screen)."""
import os
import random

import pytest
import torch

from tests.helpers import normwise

pytestmark = pytest.mark.gpu


def _cases():
    rnd = random.Random(4)                   # (the stream of tools/fuzz_shapes.py)
    out = []
    for _ in range(36):
        B = rnd.choice([1, 2, 7, 33, 80]); H = rnd.randint(46, 150); W = rnd.randint(46, 150)
        if B * H * W > 600000:
            B = max(1, 600000 // (H * W))
        out.append((B, H, W, rnd.choice([8, 16, 33, 50, 64]), rnd.choice(["topk", "topk", "adaptive_topk"])))
    return out


@pytest.mark.parametrize("B,H,W,k,mode", _cases(), ids=lambda v: str(v))
def test_random_shape_against_the_fp64_oracle(B, H, W, k, mode):
    from dagl_amd import ops
    from dagl_amd.ce import CE
    from dagl_amd.synth import make_ce_params, make_features
    from oracle.ce_oracle import ce_forward_oracle
    dev = torch.device("cuda:0")
    prm = {n: torch.from_numpy(a) for n, a in make_ce_params(90 + H, variant="default" if mode == "topk" else "allpass").items()}
    x = torch.from_numpy(make_features(91 + W, B, 64, H, W))
    ce = CE(in_channels=64)
    ce.load_state_dict(prm, strict=True)
    ce.select_mode, ce.select_k = mode, k
    ce = ce.to(dev).eval()
    with torch.no_grad():
        out = ce(x.to(dev))
        again = ce(x.to(dev))
    assert torch.equal(out, again)
    shape, d = ce._last_call
    verdict = ops.ce_range_check(shape, mode, min(k, H * W), ce._ws, d)
    assert not verdict & 1, "left the split-fp16 range"
    assert not verdict & 4, "the last call's redo pass had flagged query groups"
    idx = sorted({0, B // 3, (2 * B) // 3, B - 1})
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    with torch.no_grad():
        want, st = ce_forward_oracle(x[idx], prm, mode=mode, k=k, dtype=torch.float64, stages=True)
    kk = min(k, H * W - 1)
    top = st["S"].topk(kk + 1, dim=2).values                       # [images, L, kk + 1]
    gap = float(((top[..., kk - 1] - top[..., kk]) / top[..., kk - 1].clamp(min=1e-30)).min())
    err = normwise(out.cpu()[idx].numpy(), want.float().numpy())
    print(f"[fuzz] [{B},64,{H},{W}] {mode} k={k}: {len(idx)} image(s) vs fp64 oracle {err:.2e}, smallest relative gap k-th/(k+1)-th score {gap:.1e}")
    assert err <= (1e-4 if gap >= 1e-6 else 1e-3)
