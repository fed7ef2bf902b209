"""Adversarial feature rows for the reduced-precision screen of the similarity search.

The screen (dagl_amd/csrc/screen.hip) decides from S~ = sum bf16(q) bf16(x) which keys MIGHT be neighbours; the
claim is that it never drops a key the reference keeps (dagl.py:256-257; GReccR2b_3mh_1-checkpoint.py:242-246).
Round-to-nearest moves a value by up to half a unit in the last place: 2^-8 relative for bf16 (8 significant bits)
right above a power of two, so a product moves by up to ~2^-7 -- and in OPPOSITE directions for two keys whose mass
sits on different features.  This builder makes exactly that happen:

  query q     u_lo on features I1 (rounds DOWN to 1), u_hi on I2 (rounds UP to 1 + 2 half_ulp), 1 on a tuning feature
  key  A      u_lo on I1                      -> S~_A / S_A = 1 / u_lo^2        (about 1 - 2^-7)
  keys C_c    (2 + c/64) u_lo on I1, c < 7    -> seven keys with about twice A's score (round down as well)
  keys B_m    u_hi on I2 + t_m on the tuning feature -> S~_B / S_B about 1 + 2^-7, true score 0.2 % BELOW A's
  the rest    small uniform features, scores far below

so the true 8 best keys of every query are {C_0..C_6, A}, an adaptive threshold half way between S_A and S_B keeps
exactly those 8 -- while in the screened scores the B keys overtake A by ~1.2 %.  Everything is scaled by 2^-5
(exact in any binary format) so that the logits 10 S m stay O(1) and A carries visible softmax weight.
"""
from __future__ import annotations

import numpy as np

D = 196
N_I1, N_I2, TUNE = 97, 96, 194          # |I1|, |I2|, index of the tuning feature (feature 195 stays zero)
GSCALE = 2.0 ** -5


def build(H=64, W=64, half_ulp=2.0 ** -8, nudge=2.0 ** -12, seed=0):
    """-> dict(wq [1,L,196], x [1,N,196] float32, a_key, c_keys, b_keys, S_A, S_B0 (float64, unscaled by query))"""
    rng = np.random.default_rng(seed)
    Lh, Lw = -(-H // 4), -(-W // 4)
    L, N = Lh * Lw, H * W
    assert N >= 4096 and N % 128 == 0
    u_lo = np.float32(1.0 + half_ulp - nudge)
    u_hi = np.float32(1.0 + half_ulp + nudge)
    I1 = np.arange(0, N_I1)
    I2 = np.arange(N_I1, N_I1 + N_I2)

    q = np.zeros(D, np.float32)
    q[I1] = u_lo; q[I2] = u_hi; q[TUNE] = 1.0
    qscale = np.float32(2.0) ** ((np.arange(L) % 3) - 1)                 # per-query power of two: 0.5, 1, 2
    wq = (q[None, :] * qscale[:, None] * np.float32(GSCALE)).astype(np.float32)

    x = rng.uniform(0.2, 0.45, size=(N, D)).astype(np.float32)           # far keys
    x[:, TUNE:] = 0.0
    a_key = 1777
    c_keys = np.array([5, 700, 1300, 2100, 2900, 3500, 4000])
    S_A = float(N_I1) * float(u_lo) ** 2
    t0 = np.floor((0.998 * S_A - N_I2 * float(u_hi) ** 2) * 256.0) / 256.0     # multiple of 2^-8 in [0.5, 1): exact in bf16 / fp16
    assert 0.5 <= t0 < 1.0
    n_b = N // 128                                                       # one B key in every second 64-key step
    b_keys = np.array([128 * m + (37 * m) % 64 for m in range(n_b)])
    b_keys = np.array([kk + 1 if kk in (a_key, *c_keys) else kk for kk in b_keys])
    x[a_key] = 0.0; x[a_key, I1] = u_lo
    for c, kk in enumerate(c_keys):
        x[kk] = 0.0; x[kk, I1] = np.float32(2.0 + c / 64.0) * u_lo
    for m, kk in enumerate(b_keys):
        x[kk] = 0.0; x[kk, I2] = u_hi; x[kk, TUNE] = np.float32(t0 - (m % 4) * 2.0 ** -8)
    x *= np.float32(GSCALE)
    S_B0 = N_I2 * float(u_hi) ** 2 + t0
    return dict(wq=wq[None], x=x[None], a_key=a_key, c_keys=c_keys, b_keys=b_keys, S_A=S_A, S_B0=S_B0,
                qscale=qscale.astype(np.float64), L=L, N=N, H=H, W=W)


def adaptive_heads(case):
    """thr = 1 and a bias that puts every query's threshold T = mean*thr - bias half way between S_A and S_B."""
    wq = case["wq"][0].astype(np.float64); x = case["x"][0].astype(np.float64)
    mu = wq @ x.mean(axis=0)
    T = 0.5 * (case["S_A"] + case["S_B0"]) * GSCALE * GSCALE * case["qscale"]
    thr = np.ones((1, case["L"]), np.float32)
    bias = (mu - T).astype(np.float32)[None]
    return thr, bias


def expected_neighbours(case):
    return set(int(v) for v in case["c_keys"]) | {int(case["a_key"])}
