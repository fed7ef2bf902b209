"""Stage-level host wrappers over the C ABI: torch supplies device memory and the stream, nothing else.

Every function takes contiguous fp32 CUDA(HIP) tensors, enqueues on torch's current stream and returns
freshly allocated outputs.  They mirror the stages of the reference method one to one
(DN_Gray/model/dagl.py:216-274); see include/dagl_ce.h for the exact contracts.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import DS, MODES, P, DaglError, check


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor")
    if not t.is_cuda:
        raise DaglError(f"{name}: must live on the GPU (dagl_amd has no CPU path)")
    if t.dtype != dtype:
        raise DaglError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise DaglError(f"{name}: must be contiguous")
    return t


def query_grid(H: int, W: int):
    return -(-H // 4), -(-W // 4)


def pad_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[B,16,H,W] -> zero-bordered channels-last [B,H+6,W+6,16]."""
    _need(x, "x")
    B, c, H, W = x.shape
    if c != 16:
        raise DaglError("pad_nhwc: 16 channels expected")
    out = torch.empty(B, H + 6, W + 6, 16, device=x.device, dtype=torch.float32)
    check(_lib.load().dagl_pad_nhwc(_stream(), B, H, W, x.data_ptr(), out.data_ptr()), "dagl_pad_nhwc")
    return out


def pack_fc_weight(w: torch.Tensor) -> torch.Tensor:
    _need(w, "w")
    if tuple(w.shape) != (196, 784):
        raise DaglError("pack_fc_weight: [196,784] expected")
    out = torch.empty(208 * 784, device=w.device, dtype=torch.float32)
    check(_lib.load().dagl_pack_fc_weight(_stream(), w.data_ptr(), out.data_ptr()), "dagl_pack_fc_weight")
    return out


def project_patches(map_nhwc: torch.Tensor, w_packed: torch.Tensor, fc_bias: torch.Tensor, H: int, W: int,
                    queries: bool, want_colsum: bool = False):
    """relu(Linear(patch)) for every patch -> ([B, rows_alloc, 204] features, optional [B,204] fp64 column sums)."""
    _need(map_nhwc, "map_nhwc"); _need(w_packed, "w_packed"); _need(fc_bias, "fc_bias")
    lib = _lib.load()
    B = map_nhwc.shape[0]
    Lh, Lw = query_grid(H, W)
    rows = Lh * Lw if queries else H * W
    ra = lib.dagl_feat_rows(rows)
    feat = torch.empty(B, ra, DS, device=map_nhwc.device, dtype=torch.float32)
    colsum = torch.empty(B, DS, device=map_nhwc.device, dtype=torch.float64) if (want_colsum and not queries) else None
    check(lib.dagl_project_patches(_stream(), B, H, W, int(queries), map_nhwc.data_ptr(), w_packed.data_ptr(),
                                   fc_bias.data_ptr(), feat.data_ptr(),
                                   colsum.data_ptr() if colsum is not None else None), "dagl_project_patches")
    return feat, colsum


def query_thresholds(wq: torch.Tensor, colsum: torch.Tensor, thr: torch.Tensor, L: int, N: int) -> torch.Tensor:
    _need(wq, "wq"); _need(colsum, "colsum", torch.float64); _need(thr, "thr")
    B = wq.shape[0]
    mt = torch.empty(B, L, device=wq.device, dtype=torch.float32)
    check(_lib.load().dagl_query_thresholds(_stream(), B, L, N, wq.data_ptr(), colsum.data_ptr(), thr.data_ptr(),
                                            mt.data_ptr()), "dagl_query_thresholds")
    return mt


def scores_dense(wq: torch.Tensor, x: torch.Tensor, L: int, N: int) -> torch.Tensor:
    _need(wq, "wq"); _need(x, "x")
    B = wq.shape[0]
    s = torch.empty(B, L, N, device=wq.device, dtype=torch.float32)
    check(_lib.load().dagl_scores_dense(_stream(), B, L, N, wq.data_ptr(), x.data_ptr(), s.data_ptr()),
          "dagl_scores_dense")
    return s


def gather_aggregate(idx: torch.Tensor, wgt: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
    """out[l,:] = sum_k wgt[l,k] * values[idx[l,k],:]   (idx < 0 = empty slot)."""
    _need(idx, "idx", torch.int32); _need(wgt, "wgt"); _need(values, "values")
    L, k = idx.shape
    if wgt.shape != idx.shape or values.dim() != 2:
        raise DaglError("gather_aggregate: shape mismatch")
    Pn = values.shape[1]
    out = torch.empty(L, Pn, device=values.device, dtype=torch.float32)
    check(_lib.load().dagl_gather_aggregate(_stream(), L, k, Pn, idx.data_ptr(), wgt.data_ptr(), values.data_ptr(),
                                            out.data_ptr()), "dagl_gather_aggregate")
    return out


def unfold_values(b2_nhwc: torch.Tensor, H: int, W: int) -> torch.Tensor:
    _need(b2_nhwc, "b2_nhwc")
    B = b2_nhwc.shape[0]
    rows = torch.empty(B, H * W, P, device=b2_nhwc.device, dtype=torch.float32)
    check(_lib.load().dagl_unfold_values(_stream(), B, H, W, b2_nhwc.data_ptr(), rows.data_ptr()),
          "dagl_unfold_values")
    return rows


def fold_normalize(agg: torch.Tensor, H: int, W: int) -> torch.Tensor:
    _need(agg, "agg")
    B = agg.shape[0]
    out = torch.empty(B, 16, H, W, device=agg.device, dtype=torch.float32)
    check(_lib.load().dagl_fold_normalize(_stream(), B, H, W, agg.data_ptr(), out.data_ptr()), "dagl_fold_normalize")
    return out


class StageProfile:
    """hipEvents at the stage boundaries of up to ``max_calls`` block forwards (include/dagl_ce.h, dagl_profile_*)."""

    def __init__(self, max_calls: int):
        self._h = C.c_void_p()
        check(_lib.load().dagl_profile_create(int(max_calls), C.byref(self._h)), "dagl_profile_create")
        self.max_calls = int(max_calls)

    def reset(self):
        check(_lib.load().dagl_profile_reset(self._h), "dagl_profile_reset")

    def select_stage(self, stage: "int | str"):
        """Record only the two events around one stage (index or name; -1 = every boundary again)."""
        if isinstance(stage, str):
            stage = list(_lib.STAGE_NAMES).index(stage)
        check(_lib.load().dagl_profile_select_stage(self._h, int(stage)), "dagl_profile_select_stage")

    def read(self):
        """-> list of per-call lists of N_STAGES milliseconds (waits for the recorded events)."""
        lib = _lib.load()
        n = C.c_int(0)
        buf = (C.c_float * (self.max_calls * _lib.N_STAGES))()
        check(lib.dagl_profile_read(self._h, C.byref(n), buf, self.max_calls), "dagl_profile_read")
        return [[buf[c * _lib.N_STAGES + s] for s in range(_lib.N_STAGES)] for c in range(n.value)]

    def __del__(self):
        try:
            if self._h:
                _lib.load().dagl_profile_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class Workspace:
    """Grow-only device scratch buffer reused across calls (allocated through torch's caching allocator, so it
    is stream-ordered and visible to torch's memory accounting)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device) -> torch.Tensor:
        if self.buf is None or self.buf.numel() < nbytes + 256 or self.buf.device != device:
            self.buf = None
            self.buf = torch.empty(int(nbytes * 1.05) + 4096, device=device, dtype=torch.uint8)
        return self.buf


def ce_forward(b1, b2, thr, bias, fc1_w, fc1_b, fc2_w, fc2_b, mode: str = "adaptive", k: int = 0,
               workspace: "Workspace | None" = None, return_info: bool = False, debug: bool = False,
               profile: "StageProfile | None" = None, exact_scan: bool = False):
    """Everything of CE.forward after its prologue convolutions (dagl.py:216-274) -> [B,16,H,W]."""
    lib = _lib.load()
    if mode not in MODES:
        raise DaglError(f"unknown mode {mode!r}")
    for n, t in (("b1", b1), ("b2", b2), ("fc1_w", fc1_w), ("fc1_b", fc1_b), ("fc2_w", fc2_w), ("fc2_b", fc2_b)):
        _need(t, n)
    B, c, H, W = b1.shape
    if c != 16 or b2.shape != b1.shape:
        raise DaglError("ce_forward: b1/b2 must both be [B,16,H,W]")
    if tuple(fc1_w.shape) != (196, 784) or tuple(fc2_w.shape) != (196, 784):
        raise DaglError("ce_forward: fc weights must be [196,784]")
    Lh, Lw = query_grid(H, W)
    if mode != "topk":
        _need(thr, "thr"); _need(bias, "bias")
        if thr.numel() != B * Lh * Lw or bias.numel() != B * Lh * Lw:
            raise DaglError("ce_forward: thr/bias must hold B*L values")
    ws = workspace if workspace is not None else Workspace()
    mode_flags = MODES[mode] | (_lib.FLAG_EXACT_SCAN if exact_scan else 0)
    need = lib.dagl_ce_workspace_bytes(B, H, W, mode_flags, int(k))
    if need == 0:
        check(-1, "dagl_ce_workspace_bytes")
    out = torch.empty(B, 16, H, W, device=b1.device, dtype=torch.float32)
    info = _lib.CeInfo()
    rc = 0
    dbg = None
    if debug:
        L = Lh * Lw
        dbg = dict(deg=torch.empty(B, L, device=b1.device, dtype=torch.int32),
                   rowsum=torch.empty(B, L, device=b1.device, dtype=torch.float32),
                   agg=torch.empty(B, L, P, device=b1.device, dtype=torch.float32))
    for _attempt in range(2):
        buf = ws.get(need, b1.device)
        base = buf.data_ptr()
        aligned = (base + 255) // 256 * 256
        args = (_stream(), B, H, W, b1.data_ptr(), b2.data_ptr(),
                thr.data_ptr() if thr is not None else None,
                bias.data_ptr() if bias is not None else None,
                fc1_w.data_ptr(), fc1_b.data_ptr(), fc2_w.data_ptr(), fc2_b.data_ptr(),
                mode_flags, int(k), out.data_ptr(), aligned, buf.numel() - (aligned - base), C.byref(info))
        if profile is not None:
            rc = lib.dagl_ce_forward_profiled(*args, profile._h)
        elif dbg is None:
            rc = lib.dagl_ce_forward(*args)
        else:
            rc = lib.dagl_ce_forward_debug(*args, dbg["deg"].data_ptr(), dbg["rowsum"].data_ptr(),
                                           dbg["agg"].data_ptr())
        if rc == _lib.ERR_WORKSPACE and info.required_bytes > need:
            need = int(info.required_bytes)      # dense neighbourhoods: the CSR fallback asked for more
            continue
        break
    check(rc, "dagl_ce_forward")
    meta = dict(required_bytes=info.required_bytes, total_edges=info.total_edges,
                max_degree=info.max_degree, path=info.path, redone_queries=info.redone_queries)
    if dbg is not None:
        meta.update(dbg)
    if return_info or debug:
        return out, meta
    return out


def ce_prologue(x, g_w, g_b, theta_w, theta_b, thr_w=None, thr_b=None, bias_w=None, bias_b=None):
    """The four prologue convolutions (dagl.py:208-215) -> (b1_nhwc, b2_nhwc, thr, bias); heads optional."""
    for n, t in (("x", x), ("g_w", g_w), ("g_b", g_b), ("theta_w", theta_w), ("theta_b", theta_b)):
        _need(t, n)
    B, c, H, W = x.shape
    if c != 64:
        raise DaglError("ce_prologue: 64 input channels expected")
    Lh, Lw = query_grid(H, W)
    b1p = torch.empty(B, H + 6, W + 6, 16, device=x.device, dtype=torch.float32)
    b2p = torch.empty_like(b1p)
    heads = thr_w is not None
    thr = torch.empty(B, Lh * Lw, device=x.device, dtype=torch.float32) if heads else None
    bias = torch.empty(B, Lh * Lw, device=x.device, dtype=torch.float32) if heads else None
    if heads:
        for n, t in (("thr_w", thr_w), ("thr_b", thr_b), ("bias_w", bias_w), ("bias_b", bias_b)):
            _need(t, n)
    p = lambda t: t.data_ptr() if t is not None else None
    check(_lib.load().dagl_ce_prologue(_stream(), B, H, W, x.data_ptr(), g_w.data_ptr(), g_b.data_ptr(),
                                       theta_w.data_ptr(), theta_b.data_ptr(), p(thr_w), p(thr_b), p(bias_w), p(bias_b),
                                       b1p.data_ptr(), b2p.data_ptr(), p(thr), p(bias)), "dagl_ce_prologue")
    return b1p, b2p, thr, bias


def ce_forward_fused(x, params: dict, mode: str = "adaptive", k: int = 0, workspace: "Workspace | None" = None,
                     profile: "StageProfile | None" = None, exact_scan: bool = False, weights_packed: bool = False,
                     dense_hint: bool = False, want_info: bool = True):
    """Whole CE.forward (dagl.py:207-275) from the block input ``x`` [B,64,H,W]; ``params`` maps the block's
    state_dict names to contiguous fp32 GPU tensors.  Returns (out, info).  ``dense_hint``: go straight to the streamed
    dense formulation (adaptive mode; same result, see DAGL_FLAG_DENSE_HINT); with ``want_info=False`` that path does
    not read its edge statistics back (no host synchronisation) and info is None."""
    lib = _lib.load()
    if mode not in MODES:
        raise DaglError(f"unknown mode {mode!r}")
    _need(x, "x")
    B, c, H, W = x.shape
    if c != 64:
        raise DaglError("ce_forward_fused: 64 input channels expected")
    names = ["g.weight", "g.bias", "theta.weight", "theta.bias", "thr_conv.weight", "thr_conv.bias",
             "bias_conv.weight", "bias_conv.bias", "fc1.0.weight", "fc1.0.bias", "fc2.0.weight", "fc2.0.bias"]
    ptrs = []
    for n in names:
        t = params[n]
        _need(t, n)
        ptrs.append(t.data_ptr())
    ws = workspace if workspace is not None else Workspace()
    mode_flags = MODES[mode] | (_lib.FLAG_EXACT_SCAN if exact_scan else 0)
    need = lib.dagl_ce_workspace_bytes(B, H, W, mode_flags, int(k))
    if need == 0:
        check(-1, "dagl_ce_workspace_bytes")
    if weights_packed and ws.buf is not None and ws.buf.numel() >= need + 256 and ws.buf.device == x.device:
        mode_flags |= _lib.FLAG_WEIGHTS_PACKED           # same buffer as last time: the packed weights are still in it
    if dense_hint and mode == "adaptive" and not exact_scan:
        mode_flags |= _lib.FLAG_DENSE_HINT
        if ws.buf is not None:
            need = max(need, ws.buf.numel() - 4096)      # keep the (larger) buffer the dense path asked for earlier
    quiet = dense_hint and not want_info
    out = torch.empty(B, 16, H, W, device=x.device, dtype=torch.float32)
    info = _lib.CeInfo()
    rc = 0
    for _attempt in range(3):
        buf = ws.get(need, x.device)
        base = buf.data_ptr()
        aligned = (base + 255) // 256 * 256
        rc = lib.dagl_ce_forward_fused(_stream(), B, H, W, x.data_ptr(), *ptrs, mode_flags, int(k), out.data_ptr(),
                                       aligned, buf.numel() - (aligned - base), None if quiet else C.byref(info),
                                       profile._h if profile is not None else None)
        if rc == _lib.ERR_WORKSPACE and quiet:
            quiet = False                                # ask again, this time for the size
            continue
        if rc == _lib.ERR_WORKSPACE and info.required_bytes > need:
            need = int(info.required_bytes)
            mode_flags &= ~_lib.FLAG_WEIGHTS_PACKED      # the buffer is about to be replaced
            continue
        break
    check(rc, "dagl_ce_forward_fused")
    if quiet:
        return out, None
    return out, dict(required_bytes=info.required_bytes, total_edges=info.total_edges,
                     max_degree=info.max_degree, path=info.path, redone_queries=info.redone_queries)


def ces_stage_forward(x, head_params, mix_w, mix_b, mode: str = "adaptive", k: int = 0,
                      workspace: "Workspace | None" = None, profile: "StageProfile | None" = None,
                      weights_packed: bool = False):
    """One CES stage in one launch set: ``conv1x1(cat(head_1(x)..head_4(x))) + x`` (dagl.py:114,116,118).
    ``head_params``: four dicts (state_dict names -> contiguous fp32 GPU tensors).  Returns (out [B,64,H,W], info), or
    (None, info) when a dense adaptive neighbourhood needs the per-head path."""
    lib = _lib.load()
    if mode not in MODES:
        raise DaglError(f"unknown mode {mode!r}")
    _need(x, "x"); _need(mix_w, "mix_w"); _need(mix_b, "mix_b")
    B, c, H, W = x.shape
    if c != 64 or len(head_params) != 4:
        raise DaglError("ces_stage_forward: x must be [B,64,H,W] and there must be four heads")
    arr = (_lib.CeWeights * 4)()
    for h, prm in enumerate(head_params):
        for field, name in zip([f for f, _ in _lib.CeWeights._fields_], _lib.CeWeights.NAMES):
            t = prm[name]
            _need(t, name)
            setattr(arr[h], field, t.data_ptr())
    ws = workspace if workspace is not None else Workspace()
    need = lib.dagl_ces_stage_workspace_bytes(B, H, W, MODES[mode], int(k))
    if need == 0:
        check(-1, "dagl_ces_stage_workspace_bytes")
    out = torch.empty(B, 64, H, W, device=x.device, dtype=torch.float32)
    info = _lib.CeInfo()
    buf = ws.get(need, x.device)
    base = buf.data_ptr()
    aligned = (base + 255) // 256 * 256
    rc = lib.dagl_ces_stage_forward(_stream(), B, H, W, x.data_ptr(), arr, mix_w.data_ptr(), mix_b.data_ptr(),
                                    MODES[mode] | (_lib.FLAG_WEIGHTS_PACKED if weights_packed else 0), int(k),
                                    out.data_ptr(), aligned, buf.numel() - (aligned - base),
                                    C.byref(info), profile._h if profile is not None else None)
    meta = dict(required_bytes=info.required_bytes, total_edges=info.total_edges, max_degree=info.max_degree,
                path=info.path, redone_queries=info.redone_queries)
    if rc == _lib.ERR_WORKSPACE and info.required_bytes == -1:
        return None, meta
    check(rc, "dagl_ces_stage_forward")
    return out, meta


def _aligned(buf: torch.Tensor):
    base = buf.data_ptr()
    a = (base + 255) // 256 * 256
    return a, buf.numel() - (a - base)


def ce_core_forward(wq_rows, x_rows, b2, thr, bias, mode: str = "adaptive", k: int = 0,
                    workspace: "Workspace | None" = None, exact_scan: bool = False):
    """Graph core with the projections given (training path, include/dagl_ce.h ``dagl_ce_core_forward``):
    wq_rows [B,L,196], x_rows [B,N,196], b2 [B,16,H,W], thr/bias [B,L] -> (out [B,16,H,W], saved lists dict)."""
    lib = _lib.load()
    if mode not in MODES:
        raise DaglError(f"unknown mode {mode!r}")
    for n, t in (("wq_rows", wq_rows), ("x_rows", x_rows), ("b2", b2)):
        _need(t, n)
    B, c, H, W = b2.shape
    Lh, Lw = query_grid(H, W)
    L, N = Lh * Lw, H * W
    if c != 16 or tuple(wq_rows.shape) != (B, L, 196) or tuple(x_rows.shape) != (B, N, 196):
        raise DaglError("ce_core_forward: expected wq_rows [B,L,196], x_rows [B,H*W,196], b2 [B,16,H,W]")
    adaptive = mode != "topk"
    if adaptive:
        _need(thr, "thr"); _need(bias, "bias")
        if thr.numel() != B * L or bias.numel() != B * L:
            raise DaglError("ce_core_forward: thr/bias must hold B*L values")
    mode_flags = MODES[mode] | (_lib.FLAG_EXACT_SCAN if exact_scan else 0)
    width = lib.dagl_ce_list_width(mode_flags, int(k))
    check(min(width, 0), "dagl_ce_list_width")
    need = lib.dagl_ce_workspace_bytes(B, H, W, mode_flags, int(k))
    if need == 0:
        check(-1, "dagl_ce_workspace_bytes")
    ws = workspace if workspace is not None else Workspace()
    dev = b2.device
    out = torch.empty(B, 16, H, W, device=dev, dtype=torch.float32)
    saved = dict(nb_idx=torch.zeros(B, L, width, device=dev, dtype=torch.int32),
                 nb_wgt=torch.zeros(B, L, width, device=dev, dtype=torch.float32),
                 nb_s=torch.zeros(B, L, width, device=dev, dtype=torch.float32),
                 nb_cnt=torch.zeros(B, L, device=dev, dtype=torch.int32),
                 mu=torch.zeros(B, L, device=dev, dtype=torch.float32) if adaptive else None)
    info = _lib.CeInfo()
    a, nbytes = _aligned(ws.get(need, dev))
    rc = lib.dagl_ce_core_forward(_stream(), B, H, W, wq_rows.data_ptr(), x_rows.data_ptr(), b2.data_ptr(),
                                  thr.data_ptr() if adaptive else None, bias.data_ptr() if adaptive else None,
                                  mode_flags, int(k), out.data_ptr(), saved["nb_idx"].data_ptr(),
                                  saved["nb_wgt"].data_ptr(), saved["nb_s"].data_ptr(), saved["nb_cnt"].data_ptr(),
                                  saved["mu"].data_ptr() if adaptive else None, a, nbytes, C.byref(info))
    check(rc, "dagl_ce_core_forward")
    saved["info"] = dict(total_edges=info.total_edges, max_degree=info.max_degree, path=info.path,
                         redone_queries=info.redone_queries)
    return out, saved


def ce_core_backward(d_out, wq_rows, x_rows, b2, thr, bias, saved: dict, mode: str = "adaptive", k: int = 0,
                     workspace: "Workspace | None" = None):
    """Gradients of the graph core (``dagl_ce_core_backward``) -> (d_wq_rows, d_x_rows, d_b2, d_thr, d_bias)."""
    lib = _lib.load()
    for n, t in (("d_out", d_out), ("wq_rows", wq_rows), ("x_rows", x_rows), ("b2", b2)):
        _need(t, n)
    B, _, H, W = b2.shape
    adaptive = mode != "topk"
    need = lib.dagl_ce_core_backward_workspace_bytes(B, H, W, MODES[mode], int(k))
    if need == 0:
        check(-1, "dagl_ce_core_backward_workspace_bytes")
    ws = workspace if workspace is not None else Workspace()
    dev = b2.device
    d_wq = torch.empty_like(wq_rows)
    d_x = torch.empty_like(x_rows)
    d_b2 = torch.empty_like(b2)
    d_thr = torch.empty(B, wq_rows.shape[1], device=dev, dtype=torch.float32) if adaptive else None
    d_bias = torch.empty_like(d_thr) if adaptive else None
    a, nbytes = _aligned(ws.get(need, dev))
    p = lambda t: t.data_ptr() if t is not None else None
    rc = lib.dagl_ce_core_backward(_stream(), B, H, W, MODES[mode], int(k), wq_rows.data_ptr(), x_rows.data_ptr(),
                                   b2.data_ptr(), p(thr) if adaptive else None, p(bias) if adaptive else None,
                                   saved["nb_idx"].data_ptr(), saved["nb_wgt"].data_ptr(), saved["nb_s"].data_ptr(),
                                   saved["nb_cnt"].data_ptr(), p(saved["mu"]), d_out.data_ptr(), d_wq.data_ptr(),
                                   d_x.data_ptr(), d_b2.data_ptr(), p(d_thr), p(d_bias), a, nbytes)
    check(rc, "dagl_ce_core_backward")
    return d_wq, d_x, d_b2, d_thr, d_bias
