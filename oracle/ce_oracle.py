"""CPU oracle for DAGL's patch-graph attention head -- TEST INFRASTRUCTURE ONLY.

This file restates, in plain dense torch-CPU arithmetic, what the reference
method ``CE.forward`` (/root/reference/DN_Gray/model/dagl.py:207-275; forks in
CAR/, Demosaic/, DN_Real/) computes.  It exists so that the HIP path can be
checked on a box where the reference itself is absent.  Nothing under
``dagl_amd/`` may import it; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg do.

Pinning: the reference ships no tests or golden vectors,
so this oracle is pinned against the reference *itself*, imported in the build
container by ``tests/golden/make_golden.py``; the resulting arrays are committed
under ``tests/golden/`` and ``tests/test_oracle_golden.py`` re-checks the oracle
against them everywhere.  The fixed-k ("top-k") mode is pinned the same way
against the reference's autosaved variant
(DN_Gray/model/.ipynb_checkpoints/GReccR2b_3mh_1-checkpoint.py:242-250).

Every stage is kept as a separate, dense, un-clever function on purpose:
the product path is streaming / sparse / fused, the oracle must not share its
algebra.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

KSIZE = 7          # dagl.py:175  ksize
STRIDE_Q = 4       # dagl.py:175  stride_1 (query patch stride)
STRIDE_KV = 1      # dagl.py:175  stride_2 (key / value patch stride)
SOFTMAX_SCALE = 10.0  # dagl.py:175


def same_pad(x: torch.Tensor, ksize: int, stride: int):
    """Zero-pad NCHW ``x`` the TF-"SAME" way; returns (padded, (l, r, t, b)).

    Follows same_padding, dagl.py:123-139: the odd unit of padding goes to the
    bottom / right.
    """
    H, W = x.shape[-2:]
    oh, ow = -(-H // stride), -(-W // stride)
    ph = max(0, (oh - 1) * stride + ksize - H)
    pw = max(0, (ow - 1) * stride + ksize - W)
    t, l = ph // 2, pw // 2
    pads = (l, pw - l, t, ph - t)
    return F.pad(x, pads), pads


def patch_rows(x: torch.Tensor, ksize: int, stride: int) -> torch.Tensor:
    """``[B,c,H,W]`` -> ``[B, n_patches, c*k*k]`` rows, element order (c, kh, kw).

    extract_image_patches (dagl.py:142-169) followed by the view/permute of
    dagl.py:220-221: SAME-pad, Unfold, one row per patch.
    """
    xp, _ = same_pad(x, ksize, stride)
    cols = F.unfold(xp, kernel_size=ksize, padding=0, stride=stride)  # [B, c*k*k, n]
    return cols.transpose(1, 2).contiguous()


def overlap_count(H: int, W: int, dtype, ksize=KSIZE, stride=STRIDE_Q, pad=3):
    """fold(unfold(ones)) of dagl.py:268-270: how many query windows cover a pixel."""
    ones = torch.ones(1, 1, H, W, dtype=dtype)
    u = F.unfold(ones, (ksize, ksize), padding=pad, stride=stride)
    return F.fold(u, (H, W), (ksize, ksize), padding=pad, stride=stride)[0, 0]


def ce_forward_oracle(x: torch.Tensor, params: Dict[str, torch.Tensor], *,
                      mode: str = "adaptive", k: Optional[int] = None,
                      dtype: torch.dtype = torch.float32,
                      zero_guard: bool = True,
                      stages: bool = False, softmax_scale: float = 10.0,
                      ksize: int = KSIZE, stride_q: int = STRIDE_Q, stride_kv: int = STRIDE_KV):
    """Dense restatement of ``CE.forward``.

    x       [B, C, H, W]
    params  the block's state_dict (torch tensors), names as in dagl.py:190-205
    mode    "adaptive"  shipped behaviour, dagl.py:256-261
            "topk"      fixed-k variant (GReccR2b_3mh_1-checkpoint.py:242-250):
                        0/1 mask over the k best scores, logits 10*S on them,
                        exp(0) for every other key, no renormalisation
            "adaptive_topk"  adaptive mask intersected with the k best scores
                        (no reference equivalent; equals "adaptive" whenever
                        k >= every row's degree)
    dtype   arithmetic type (float64 gives a rounding-free yardstick)
    zero_guard  DN_Gray's ``out_mask += (out_mask==0)`` (dagl.py:271); a
            numerical no-op because the count is never 0
    stages  also return the per-sample intermediates
    ksize, stride_q, stride_kv   the ctor's ksize / stride_1 / stride_2 (dagl.py:175;
            inter_channels is read off the weights); defaults = what ships

    Returns out ``[B, c, H, W]`` (and a dict of stage tensors when asked).
    """
    if mode not in ("adaptive", "topk", "adaptive_topk"):
        raise ValueError(mode)
    if mode != "adaptive" and (k is None or k < 1):
        raise ValueError("k >= 1 required for top-k modes")
    P = {n: t.to(dtype) for n, t in params.items()}
    x = x.to(dtype)
    B, C, H, W = x.shape

    # prologue, dagl.py:208-215
    b1 = F.conv2d(x, P["g.weight"], P["g.bias"], padding=1)           # keys + queries
    b2 = F.conv2d(x, P["theta.weight"], P["theta.bias"])              # values
    xq, _ = same_pad(x, ksize, stride_q)
    if mode == "topk" and "thr_conv.weight" not in P:                 # (the fixed-k variant has no threshold heads)
        thr = bias = torch.zeros(B, (-(-H // stride_q)) * (-(-W // stride_q)), dtype=dtype)
    else:
        thr = F.conv2d(xq, P["thr_conv.weight"], P["thr_conv.bias"], stride=stride_q).reshape(B, -1)
        bias = F.conv2d(xq, P["bias_conv.weight"], P["bias_conv.bias"], stride=stride_q).reshape(B, -1)

    # patch extraction, dagl.py:216-240
    q_rows = patch_rows(b1, ksize, stride_q)    # [B, L, 784]
    v_rows = patch_rows(b2, ksize, stride_kv)   # [B, N, 784]
    k_rows = patch_rows(b1, ksize, stride_kv)   # [B, N, 784]
    fold_pad = same_pad(b1[:1, :1], ksize, stride_kv)[1][0]           # dagl.py:243 -> 3 (paddings[0]: the LEFT pad, for both axes)

    cnt = overlap_count(H, W, dtype, ksize=ksize, stride=stride_q, pad=fold_pad)
    if zero_guard:
        cnt = cnt + (cnt == 0).to(dtype)                              # dagl.py:271

    outs = []
    st = {k_: [] for k_ in ("Wq", "X", "S", "T", "deg", "rowsum", "agg", "mask_b")} if stages else None
    for n in range(B):                                                # dagl.py:245
        Wq = F.relu(F.linear(q_rows[n], P["fc1.0.weight"], P["fc1.0.bias"]))  # [L,196] :248
        X = F.relu(F.linear(k_rows[n], P["fc2.0.weight"], P["fc2.0.bias"]))   # [N,196] :249
        z, c = _graph_core(Wq, X, thr[n], bias[n], v_rows[n], mode, k, H, W, fold_pad, softmax_scale, ksize, stride_q)
        outs.append(z / cnt)                                                   # :272
        if stages:
            st["Wq"].append(Wq); st["X"].append(X)
            for k_ in ("S", "T", "deg", "rowsum", "agg", "mask_b"):
                st[k_].append(c[k_])
    out = torch.cat(outs, dim=0)                                               # :274
    if stages:
        st = {k_: torch.stack(v) for k_, v in st.items()}
        st.update(b1=b1, b2=b2, thr=thr, bias=bias, cnt=cnt)
        return out, st
    return out


def _k_best(S: torch.Tensor, kk: int) -> torch.Tensor:
    """Indices of each row's kk largest scores (``torch.topk(S, kk, dim=1)``, GReccR2b_3mh_1-checkpoint.py:243-244).  torch.topk
    leaves the choice among EQUAL scores at the kk-th place unspecified; the oracle pins it -- the lower key index wins (a stable
    descending sort) -- so that outputs AND gradients are defined on degenerate maps; without such ties it is torch.topk's set."""
    return torch.sort(S, dim=1, descending=True, stable=True).indices[:, :kk]


def _graph_core(Wq, X, thr_n, bias_n, v_rows_n, mode, k, H, W, fold_pad, softmax_scale=SOFTMAX_SCALE, ksize=KSIZE, stride_q=STRIDE_Q):
    """One sample of dagl.py:250-267 from its feature rows: similarity, mask, edge softmax, aggregate, fold
    (not yet divided by the overlap count).  Returns (folded [1,c,H,W], dict of intermediates)."""
    dtype = Wq.dtype
    S = Wq @ X.t()                                                             # [L,N]   :250
    T = S.mean(dim=1) * thr_n - bias_n                # threshold of :256, per query
    if mode == "topk":
        kk = min(k, S.shape[1])
        sel = torch.zeros_like(S)
        sel.scatter_(1, _k_best(S, kk), 1.0)
        m, mb = sel, sel
    else:
        m = F.relu(S - S.mean(dim=1, keepdim=True) * thr_n.unsqueeze(1)
                   + bias_n.unsqueeze(1))                                      # :256
        mb = (m != 0).to(dtype)                                                # :257
        if mode == "adaptive_topk":
            kk = min(k, S.shape[1])
            keep = torch.zeros_like(S)
            keep.scatter_(1, _k_best(S, kk), 1.0)
            m, mb = m * keep, mb * keep
    A = F.softmax(S * m * softmax_scale, dim=1) * mb                           # :259-261 (self.softmax_scale, 10 unless the ctor says otherwise: :175)
    agg = A @ v_rows_n                                                         # [L,784] :263-264
    z = None
    if fold_pad is not None:                          # (None: a sample of the queries, nothing to fold)
        z = F.fold(agg.t().unsqueeze(0), (H, W), (ksize, ksize),
                   padding=fold_pad, stride=stride_q)                          # :265-267
    return z, dict(S=S, T=T, deg=mb.sum(dim=1), rowsum=A.sum(dim=1), agg=agg, mask_b=mb)


def ce_core_oracle(wq_rows: torch.Tensor, x_rows: torch.Tensor, b2: torch.Tensor,
                   thr: Optional[torch.Tensor], bias: Optional[torch.Tensor], *, mode: str = "adaptive",
                   k: Optional[int] = None, dtype: torch.dtype = torch.float64, stages: bool = False):
    """dagl.py:250-274 with the feature rows given (the boundary of ``dagl_ce_core_forward``):
    wq_rows [B,L,196] = relu(fc1(query patches)), x_rows [B,N,196] = relu(fc2(key patches)), b2 [B,c,H,W]
    the value map, thr / bias [B,L] the per-query heads (ignored by "topk").  Same loop body as
    ``ce_forward_oracle`` (``_graph_core``)."""
    B, c, H, W = b2.shape
    b2 = b2.to(dtype)
    v_rows = patch_rows(b2, KSIZE, STRIDE_KV)
    fold_pad = same_pad(b2[:1, :1], KSIZE, STRIDE_KV)[1][0]
    cnt = overlap_count(H, W, dtype, pad=fold_pad)
    cnt = cnt + (cnt == 0).to(dtype)
    L = wq_rows.shape[1]
    zero = torch.zeros(L, dtype=dtype)
    outs, sts = [], []
    for n in range(B):
        z, cst = _graph_core(wq_rows[n].to(dtype), x_rows[n].to(dtype),
                             thr[n].to(dtype) if thr is not None else zero,
                             bias[n].to(dtype) if bias is not None else zero, v_rows[n], mode, k, H, W, fold_pad)
        outs.append(z / cnt)
        sts.append(cst)
    out = torch.cat(outs, dim=0)
    if stages:
        return out, {k_: torch.stack([s[k_] for s in sts]) for k_ in sts[0]}
    return out


def ce_rows_oracle(x: torch.Tensor, params: Dict[str, torch.Tensor], rows: torch.Tensor, *, mode: str = "adaptive",
                   k: Optional[int] = None, dtype: torch.dtype = torch.float32, softmax_scale: float = 10.0):
    """``ce_forward_oracle`` restricted to the query patches ``rows`` (indices into the L queries) of ONE image
    ``x [1,C,H,W]``: every row of S is independent of the others (the mean of dagl.py:256 runs over the keys), so a
    sample of queries can be checked against all N keys at sizes where the full ``[L,N]`` matrix does not fit the host
    (512^2: 16 GiB, 1024^2: 256 GiB).  Returns dict(deg, rowsum, agg [len(rows),784] in (c,kh,kw) order, S, T)."""
    assert x.shape[0] == 1
    P = {n: t.to(dtype) for n, t in params.items()}
    x = x.to(dtype)
    b1 = F.conv2d(x, P["g.weight"], P["g.bias"], padding=1)
    b2 = F.conv2d(x, P["theta.weight"], P["theta.bias"])
    xq, _ = same_pad(x, KSIZE, STRIDE_Q)
    thr = F.conv2d(xq, P["thr_conv.weight"], P["thr_conv.bias"], stride=STRIDE_Q).reshape(-1)[rows]
    bias = F.conv2d(xq, P["bias_conv.weight"], P["bias_conv.bias"], stride=STRIDE_Q).reshape(-1)[rows]
    q_rows = patch_rows(b1, KSIZE, STRIDE_Q)[0][rows]
    Wq = F.relu(F.linear(q_rows, P["fc1.0.weight"], P["fc1.0.bias"]))
    N = x.shape[2] * x.shape[3]
    # keys / values in slabs of rows of the image would need halos; the unfolds of one image are affordable (N x 784)
    X = F.relu(F.linear(patch_rows(b1, KSIZE, STRIDE_KV)[0], P["fc2.0.weight"], P["fc2.0.bias"]))
    v_rows = patch_rows(b2, KSIZE, STRIDE_KV)[0]
    H, W = x.shape[2:]
    _, c = _graph_core(Wq, X, thr, bias, v_rows, mode, k, H, W, None, softmax_scale)
    return c


def gather_aggregate_oracle(idx: torch.Tensor, wgt: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
    """out[l,:] = sum_k wgt[l,k] * values[idx[l,k],:]  -- the sparse form of the
    dense ``torch.mm(yi, pi)`` at dagl.py:263-264 (idx < 0 marks an empty slot)."""
    ok = idx >= 0
    rows = values[idx.clamp(min=0).long()]              # [L,k,P]
    w = (wgt * ok.to(wgt.dtype)).unsqueeze(-1)
    return (rows * w).sum(dim=1)


def params_to_torch(np_params) -> Dict[str, torch.Tensor]:
    return {n: torch.from_numpy(a.copy()) for n, a in np_params.items()}
