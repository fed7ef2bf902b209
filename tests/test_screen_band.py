"""The reduced-precision screen's relative band, proven on the CPU with the adversarial rows of tests/adversarial.py:
the band the kernels use (SCREEN_DELTA, dagl_amd/csrc/dagl_common.h) keeps every true neighbour, the round-1 value
(0.004, half the true worst case of bf16 round-to-nearest) would have dropped one."""
import re
import os

import numpy as np
import pytest
import torch

from tests import adversarial as adv

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _screen_delta():
    src = open(os.path.join(REPO, "dagl_amd", "csrc", "dagl_common.h")).read()
    return float(re.search(r"SCREEN_DELTA\s*=\s*([0-9.eE+-]+)f", src).group(1))


def _screened(case, dtype):
    wq = torch.from_numpy(case["wq"][0]).to(dtype).float()
    x = torch.from_numpy(case["x"][0]).to(dtype).float()
    return (wq.double() @ x.double().t()).numpy()          # products of bf16 values are exact in fp32; sum in fp64


def test_construction_true_neighbours():
    case = adv.build()
    S = case["wq"][0].astype(np.float64) @ case["x"][0].astype(np.float64).T
    want = adv.expected_neighbours(case)
    top8 = np.argsort(-S, axis=1)[:, :8]
    assert all(set(int(v) for v in row) == want for row in top8)
    thr, bias = adv.adaptive_heads(case)
    T = S.mean(axis=1) * thr[0] - bias[0]
    passing = S > T[:, None]
    assert (passing.sum(axis=1) == 8).all()
    assert all(set(np.nonzero(r)[0].tolist()) == want for r in passing)


@pytest.mark.parametrize("dtype,half_ulp,nudge", [(torch.bfloat16, 2.0 ** -8, 2.0 ** -12)])
def test_band_keeps_the_adversarial_neighbour(dtype, half_ulp, nudge):
    delta = _screen_delta()
    case = adv.build(half_ulp=half_ulp, nudge=nudge)
    Ss = _screened(case, dtype)
    a = case["a_key"]
    # top-k: theta = k-th largest screened score (the tightest threshold any partition into groups can produce)
    theta = -np.sort(-Ss, axis=1)[:, 7]
    keep_new = Ss[:, a] >= theta * (1 - delta) / (1 + delta)
    keep_old = Ss[:, a] >= theta * (1 - 0.004) / (1 + 0.004)
    assert keep_new.all()
    assert not keep_old.any(), "the case no longer exercises the band: it must beat the round-1 value"
    # adaptive: candidate iff S~ (1 + delta) - T > 0
    thr, bias = adv.adaptive_heads(case)
    S = case["wq"][0].astype(np.float64) @ case["x"][0].astype(np.float64).T
    T = S.mean(axis=1) * thr[0] - bias[0]
    assert (Ss[:, a] * (1 + delta) > T).all()
    assert not (Ss[:, a] * (1 + 0.004) > T).any()


def test_band_covers_bf16_round_to_nearest_worst_case():
    """|bf16(v) - v| <= 2^-8 v, attained just above a power of two; two operands: (1 + 2^-8)^2 - 1 < SCREEN_DELTA."""
    delta = _screen_delta()
    v = torch.tensor([1.0 + 2.0 ** -8 - 2.0 ** -20, 1.0 + 2.0 ** -8 + 2.0 ** -20], dtype=torch.float32)
    r = v.to(torch.bfloat16).float()
    rel = ((r - v) / v).abs().max().item()
    assert 0.0038 < rel <= 2.0 ** -8
    assert (1 + 2.0 ** -8) ** 2 - 1 + 196 * 2.0 ** -23 < delta < 0.0082
