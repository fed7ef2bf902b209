"""Stage-level host wrappers over the C ABI: torch supplies device memory and the stream, nothing else.

Every function takes contiguous fp32 CUDA(HIP) tensors, enqueues on torch's current stream and returns
freshly allocated outputs.  They mirror the stages of the reference method one to one
(DN_Gray/model/dagl.py:216-274); see include/dagl_ce.h for the exact contracts.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import DS, MODES, P, DaglError, check


# A/B switch (tests, bench.py --dense-backward fp32): the dense graph core's backward with its five matrix products on the fp32
# matrix cores.  An explicit attribute, not an environment variable: nothing outside the process can change gradient bits.
DENSE_BACKWARD_FP32 = False


def _stream() -> int:
    """torch's current stream ON THE CURRENT DEVICE -- every entry point below runs under ``_on_device`` which makes the
    tensors' device current first (the C ABI launches on the current HIP device)."""
    return torch.cuda.current_stream().cuda_stream


def _tensors_of(args, kwargs):
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor):
            yield a
        elif isinstance(a, dict):
            yield from (v for v in a.values() if isinstance(v, torch.Tensor))
        elif isinstance(a, (list, tuple)):
            for e in a:
                if isinstance(e, torch.Tensor):
                    yield e
                elif isinstance(e, dict):
                    yield from (v for v in e.values() if isinstance(v, torch.Tensor))


def _on_device(fn):
    """Run ``fn`` with the device of its tensor arguments as the current device (a module living on cuda:1 must not
    launch on cuda:0's stream) and refuse tensors spread over several devices."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = None
        for t in _tensors_of(args, kwargs):
            if not t.is_cuda:
                continue
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise DaglError(f"{fn.__name__}: tensors on different devices ({dev} and {t.device})")
        if dev is None:
            return fn(*args, **kwargs)          # the per-argument checks below report the CPU tensor
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


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


@_on_device
def pad_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[B,16,H,W] -> zero-bordered channels-last [B,H+6,W+6,16]."""
    _need(x, "x")
    B, c, H, W = x.shape
    if c != 16:
        raise DaglError("pad_nhwc: 16 channels expected")
    out = torch.empty(B, H + 6, W + 6, 16, device=x.device, dtype=torch.float32)
    check(_lib.load().dagl_pad_nhwc(_stream(), B, H, W, x.data_ptr(), out.data_ptr()), "dagl_pad_nhwc")
    return out


@_on_device
def pack_fc_weight(w: torch.Tensor) -> torch.Tensor:
    _need(w, "w")
    if tuple(w.shape) != (196, 784):
        raise DaglError("pack_fc_weight: [196,784] expected")
    out = torch.empty(208 * 784, device=w.device, dtype=torch.float32)
    check(_lib.load().dagl_pack_fc_weight(_stream(), w.data_ptr(), out.data_ptr()), "dagl_pack_fc_weight")
    return out


@_on_device
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


@_on_device
def query_thresholds(wq: torch.Tensor, colsum: torch.Tensor, thr: torch.Tensor, L: int, N: int) -> torch.Tensor:
    _need(wq, "wq"); _need(colsum, "colsum", torch.float64); _need(thr, "thr")
    B = wq.shape[0]
    mt = torch.empty(B, L, device=wq.device, dtype=torch.float32)
    check(_lib.load().dagl_query_thresholds(_stream(), B, L, N, wq.data_ptr(), colsum.data_ptr(), thr.data_ptr(),
                                            mt.data_ptr()), "dagl_query_thresholds")
    return mt


@_on_device
def scores_dense(wq: torch.Tensor, x: torch.Tensor, L: int, N: int) -> torch.Tensor:
    _need(wq, "wq"); _need(x, "x")
    B = wq.shape[0]
    s = torch.empty(B, L, N, device=wq.device, dtype=torch.float32)
    check(_lib.load().dagl_scores_dense(_stream(), B, L, N, wq.data_ptr(), x.data_ptr(), s.data_ptr()),
          "dagl_scores_dense")
    return s


@_on_device
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


@_on_device
def unfold_values(b2_nhwc: torch.Tensor, H: int, W: int) -> torch.Tensor:
    _need(b2_nhwc, "b2_nhwc")
    B = b2_nhwc.shape[0]
    rows = torch.empty(B, H * W, P, device=b2_nhwc.device, dtype=torch.float32)
    check(_lib.load().dagl_unfold_values(_stream(), B, H, W, b2_nhwc.data_ptr(), rows.data_ptr()),
          "dagl_unfold_values")
    return rows


@_on_device
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
    """Grow-only device scratch buffers reused across calls, one per device (allocated through torch's caching allocator,
    so they are stream-ordered and visible to torch's memory accounting).  ``nn.DataParallel`` replicas share the
    object across per-device threads: the buffer is built in a local and looked up by device, never handed across."""

    def __init__(self):
        self._bufs = {}

    @property
    def buf(self):
        """The buffer of the current device (None before the first call there)."""
        if not self._bufs:
            return None
        if torch.cuda.is_available():
            b = self._bufs.get(torch.device("cuda", torch.cuda.current_device()))
            if b is not None:
                return b
        return next(iter(self._bufs.values())) if len(self._bufs) == 1 else None

    def peek(self, device):
        return self._bufs.get(torch.device(device))

    def get(self, nbytes: int, device) -> torch.Tensor:
        device = torch.device(device)
        if device.index is None and device.type == "cuda":
            device = torch.device("cuda", torch.cuda.current_device())
        b = self._bufs.get(device)
        if b is None or b.numel() < nbytes + 256:
            self._bufs.pop(device, None)
            b = None                                              # release before the larger allocation
            b = torch.empty(int(nbytes * 1.05) + 4096, device=device, dtype=torch.uint8)
            self._bufs[device] = b
        return b


@_on_device
def ce_forward(b1, b2, thr, bias, fc1_w, fc1_b, fc2_w, fc2_b, mode: str = "adaptive", k: int = 0,
               workspace: "Workspace | None" = None, return_info: bool = False, debug: bool = False,
               profile: "StageProfile | None" = None, exact_scan: bool = False, tight_topk: bool = False,
               sampled_topk: bool = False):
    """Everything of CE.forward after its prologue convolutions (dagl.py:216-274) -> [B,16,H,W].  ``tight_topk`` /
    ``sampled_topk``: DAGL_FLAG_TIGHT_TOPK / DAGL_FLAG_SAMPLED_TOPK (top-k modes behind the screen), as in ``ce_forward_fused``."""
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
    if mode != "adaptive" and not exact_scan:
        mode_flags |= _lib.FLAG_TIGHT_TOPK if tight_topk else (_lib.FLAG_SAMPLED_TOPK if sampled_topk else 0)
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
                max_degree=info.max_degree, path=info.path, redone_queries=info.redone_queries,
                range_fallback=info.range_fallback, dense_rerun_blocks=info.dense_rerun_blocks)
    if dbg is not None:
        meta.update(dbg)
    if return_info or debug:
        return out, meta
    return out


@_on_device
def ce_forward_generic(x, params: dict, ksize: int, stride_1: int, stride_2: int, inter_channels: int, mode: str = "adaptive",
                       k: int = 0, softmax_scale: float = 10.0, workspace: "Workspace | None" = None, want_degree: bool = False):
    """``CE.forward`` for ANY patch geometry (``dagl_ce_generic_forward``, csrc/generic.hip; dagl.py:175-176 makes ksize, stride_1,
    stride_2 and inter_channels constructor arguments): x [B,Cin,H,W] fp32 -> [B,inter_channels,H,W].  ``params`` = the block's
    state_dict tensors under their own names (fp32, on the device); Cin and inter_channels multiples of 4."""
    lib = _lib.load()
    if mode not in MODES:
        raise DaglError(f"unknown mode {mode!r}")
    _need(x, "x")
    B, Cin, H, W = x.shape
    c, ks = int(inter_channels), int(ksize)
    P_, heads = ks * ks * c, mode != "topk"
    want = {"g.weight": (c, Cin, 3, 3), "g.bias": (c,), "theta.weight": (c, Cin, 1, 1), "theta.bias": (c,),
            "fc1.0.weight": (P_ // 4, P_), "fc1.0.bias": (P_ // 4,), "fc2.0.weight": (P_ // 4, P_), "fc2.0.bias": (P_ // 4,)}
    if heads:
        want.update({"thr_conv.weight": (1, Cin, ks, ks), "thr_conv.bias": (1,), "bias_conv.weight": (1, Cin, ks, ks), "bias_conv.bias": (1,)})
    for n, shp in want.items():
        _need(params[n], n)
        if tuple(params[n].shape) != shp:
            raise DaglError(f"ce_forward_generic: {n} is {tuple(params[n].shape)}, expected {shp}")
    need = lib.dagl_ce_generic_workspace_bytes(B, Cin, H, W, ks, int(stride_1), int(stride_2), c)
    if need == 0:
        raise DaglError(f"ce_forward_generic: unsupported shape / geometry B={B} Cin={Cin} H={H} W={W} ksize={ks} "
                        f"strides=({stride_1},{stride_2}) inter_channels={c}")
    ws = workspace if workspace is not None else Workspace()
    buf = ws.get(need, x.device)
    out = torch.empty(B, c, H, W, device=x.device, dtype=torch.float32)
    L = (-(-H // int(stride_1))) * (-(-W // int(stride_1)))
    deg = torch.empty(B, L, device=x.device, dtype=torch.int32) if want_degree else None
    ptr = lambda n: params[n].data_ptr() if n in want else None
    check(lib.dagl_ce_generic_forward(_stream(), B, Cin, H, W, ks, int(stride_1), int(stride_2), c, float(softmax_scale), MODES[mode], int(k),
                                      x.data_ptr(), ptr("g.weight"), ptr("g.bias"), ptr("theta.weight"), ptr("theta.bias"),
                                      ptr("thr_conv.weight"), ptr("thr_conv.bias"), ptr("bias_conv.weight"), ptr("bias_conv.bias"),
                                      ptr("fc1.0.weight"), ptr("fc1.0.bias"), ptr("fc2.0.weight"), ptr("fc2.0.bias"),
                                      out.data_ptr(), deg.data_ptr() if deg is not None else None, buf.data_ptr(), buf.numel()),
          "dagl_ce_generic_forward")
    return (out, deg) if want_degree else out


def _generic_geom(H, W, ksize, stride_1, stride_2):
    lib = _lib.load()
    pg = lib.dagl_ce_generic_border(int(ksize))
    L = (-(-H // int(stride_1))) * (-(-W // int(stride_1)))
    N = (-(-H // int(stride_2))) * (-(-W // int(stride_2)))
    return pg, L, N


@_on_device
def ce_generic_core_forward(wq_rows, x_rows, b2p, thr, bias, H: int, W: int, ksize: int, stride_1: int, stride_2: int, mode: str = "adaptive",
                            k: int = 0, softmax_scale: float = 10.0, workspace: "Workspace | None" = None, want_degree: bool = False):
    """dagl.py:250-272 from the feature rows for any patch geometry (``dagl_ce_generic_core_forward``): wq_rows [B,L,D], x_rows [B,N,D],
    b2p the zero-bordered NHWC value map [B,H+2pg,W+2pg,c] (pg = ``dagl_ce_generic_border(ksize)``), thr / bias [B,L] -> [B,c,H,W]."""
    lib = _lib.load()
    for n, t in (("wq_rows", wq_rows), ("x_rows", x_rows), ("b2p", b2p)):
        _need(t, n)
    B, c = b2p.shape[0], b2p.shape[3]
    pg, L, N = _generic_geom(H, W, ksize, stride_1, stride_2)
    D_ = ksize * ksize * c // 4
    if tuple(b2p.shape) != (B, H + 2 * pg, W + 2 * pg, c) or tuple(wq_rows.shape) != (B, L, D_) or tuple(x_rows.shape) != (B, N, D_):
        raise DaglError(f"ce_generic_core_forward: shapes {tuple(wq_rows.shape)}, {tuple(x_rows.shape)}, {tuple(b2p.shape)} do not fit "
                        f"L={L} N={N} D={D_} border={pg}")
    heads = mode != "topk"
    if heads:
        _need(thr, "thr"); _need(bias, "bias")
    need = lib.dagl_ce_generic_core_workspace_bytes(B, H, W, int(ksize), int(stride_1), int(stride_2), c, 0)
    ws = workspace if workspace is not None else Workspace()
    buf = ws.get(need, b2p.device)
    out = torch.empty(B, c, H, W, device=b2p.device, dtype=torch.float32)
    deg = torch.empty(B, L, device=b2p.device, dtype=torch.int32) if want_degree else None
    check(lib.dagl_ce_generic_core_forward(_stream(), B, H, W, int(ksize), int(stride_1), int(stride_2), c, float(softmax_scale), MODES[mode], int(k),
                                           wq_rows.data_ptr(), x_rows.data_ptr(), b2p.data_ptr(), thr.data_ptr() if heads else None,
                                           bias.data_ptr() if heads else None, out.data_ptr(), deg.data_ptr() if deg is not None else None,
                                           buf.data_ptr(), buf.numel()), "dagl_ce_generic_core_forward")
    return (out, deg) if want_degree else out


@_on_device
def ce_generic_core_backward(d_out, wq_rows, x_rows, b2p, thr, bias, H: int, W: int, ksize: int, stride_1: int, stride_2: int,
                             mode: str = "adaptive", k: int = 0, softmax_scale: float = 10.0, workspace: "Workspace | None" = None):
    """Gradients of ``ce_generic_core_forward`` w.r.t. (wq_rows, x_rows, b2p, thr, bias) (``dagl_ce_generic_core_backward``)."""
    lib = _lib.load()
    _need(d_out, "d_out")
    B, c = b2p.shape[0], b2p.shape[3]
    heads = mode != "topk"
    need = lib.dagl_ce_generic_core_workspace_bytes(B, H, W, int(ksize), int(stride_1), int(stride_2), c, 1)
    ws = workspace if workspace is not None else Workspace()
    buf = ws.get(need, b2p.device)
    d_wq, d_x, d_b2p = torch.empty_like(wq_rows), torch.empty_like(x_rows), torch.empty_like(b2p)
    d_thr = torch.empty_like(thr) if heads else None
    d_bias = torch.empty_like(bias) if heads else None
    check(lib.dagl_ce_generic_core_backward(_stream(), B, H, W, int(ksize), int(stride_1), int(stride_2), c, float(softmax_scale), MODES[mode], int(k),
                                            wq_rows.data_ptr(), x_rows.data_ptr(), b2p.data_ptr(), thr.data_ptr() if heads else None,
                                            bias.data_ptr() if heads else None, d_out.data_ptr(), d_wq.data_ptr(), d_x.data_ptr(), d_b2p.data_ptr(),
                                            d_thr.data_ptr() if heads else None, d_bias.data_ptr() if heads else None, buf.data_ptr(), buf.numel()),
          "dagl_ce_generic_core_backward")
    return d_wq, d_x, d_b2p, d_thr, d_bias


@_on_device
def ce_prologue(x, g_w, g_b, theta_w, theta_b, thr_w=None, thr_b=None, bias_w=None, bias_b=None, fast=False):
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
    if fast:
        # g / theta on the fp16 matrix cores with split operands (conv_pair16_kernel, the inference path's kernel, fp32 map out)
        lib = _lib.load()
        need = lib.dagl_ce_prologue16_scratch_bytes(B, H, W)
        scratch = torch.empty(need + 256, device=x.device, dtype=torch.uint8)
        base = (scratch.data_ptr() + 255) // 256 * 256
        check(lib.dagl_ce_prologue16(_stream(), B, H, W, x.data_ptr(), g_w.data_ptr(), g_b.data_ptr(), theta_w.data_ptr(), theta_b.data_ptr(),
                                     p(thr_w), p(thr_b), p(bias_w), p(bias_b), b1p.data_ptr(), b2p.data_ptr(), p(thr), p(bias), base, need),
              "dagl_ce_prologue16")
        return b1p, b2p, thr, bias
    scratch = torch.empty(8 * B * Lh * Lw, device=x.device, dtype=torch.float32) if heads else None
    check(_lib.load().dagl_ce_prologue(_stream(), B, H, W, x.data_ptr(), g_w.data_ptr(), g_b.data_ptr(),
                                       theta_w.data_ptr(), theta_b.data_ptr(), p(thr_w), p(thr_b), p(bias_w), p(bias_b),
                                       b1p.data_ptr(), b2p.data_ptr(), p(thr), p(bias), p(scratch)), "dagl_ce_prologue")
    return b1p, b2p, thr, bias


@_on_device
def ce_forward_fused(x, params: dict, mode: str = "adaptive", k: int = 0, workspace: "Workspace | None" = None,
                     profile: "StageProfile | None" = None, exact_scan: bool = False, weights_packed: bool = False,
                     dense_hint: bool = False, want_info: bool = True, no_wait: bool = False, tight_topk: bool = False,
                     sampled_topk: bool = False, no_redo: bool = False):
    """Whole CE.forward (dagl.py:207-275) from the block input ``x`` [B,64,H,W]; ``params`` maps the block's
    state_dict names to contiguous fp32 GPU tensors.  Returns (out, info).  ``dense_hint``: go straight to the streamed
    dense formulation (adaptive mode; same result, see DAGL_FLAG_DENSE_HINT); with ``want_info=False`` that path does
    not read its edge statistics back (no host synchronisation) and info is None.  ``no_wait`` (adaptive mode behind the
    screen, DAGL_FLAG_NO_WAIT): the verdict stays on the device, an unserved call is NaN-filled and ``ce_range_check``
    reports it; info is None.  ``tight_topk`` (top-k modes, DAGL_FLAG_TIGHT_TOPK): candidate threshold from every second key tile and
    eight times the candidate slots -- for maps whose sampled threshold lets too many keys through (natural images); same result.
    ``sampled_topk`` (DAGL_FLAG_SAMPLED_TOPK) forces the sampled threshold; with neither the workspace's own policy word decides
    on the device (sticky switch to the tight threshold once a call overflowed; a cold workspace re-runs tight in the same call).
    ``no_redo`` (top-k modes, DAGL_FLAG_NO_REDO): the fp32 redo pass behind the refine kernel is not queued; a call that flagged a
    query group anyway is NaN-filled and ``ce_range_check`` reports it (bit 4, sticky)."""
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
    if weights_packed and ws.peek(x.device) is not None and ws.peek(x.device).numel() >= need + 256:
        mode_flags |= _lib.FLAG_WEIGHTS_PACKED           # same buffer as last time: the packed weights are still in it
    if dense_hint and mode == "adaptive" and not exact_scan:
        mode_flags |= _lib.FLAG_DENSE_HINT
        if ws.peek(x.device) is not None:
            need = max(need, ws.peek(x.device).numel() - 4096)      # keep the (larger) buffer the dense path asked for earlier
    if mode != "adaptive" and not exact_scan:
        mode_flags |= _lib.FLAG_TIGHT_TOPK if tight_topk else (_lib.FLAG_SAMPLED_TOPK if sampled_topk else 0)
        if no_redo and (mode_flags & _lib.FLAG_WEIGHTS_PACKED):
            mode_flags |= _lib.FLAG_NO_REDO
    quiet = dense_hint and not want_info
    if no_wait and mode == "adaptive" and not exact_scan and not dense_hint and H * W >= 2048:
        mode_flags |= _lib.FLAG_NO_WAIT
        quiet = True
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
                     max_degree=info.max_degree, path=info.path, redone_queries=info.redone_queries,
                     range_fallback=info.range_fallback, dense_rerun_blocks=info.dense_rerun_blocks)


def ce_range_check(shape, mode: str, k: int, workspace: "Workspace", device) -> int:
    """True when a forward on ``workspace`` (input shape ``shape``) since the last check left the range of the split-fp16
    kernels and returned a NaN-filled output (``dagl_ce_range_check``; one host synchronisation; the word is sticky and
    cleared by the check that reports it).  ``shape[0]`` counts head x image pairs for a stage workspace."""
    lib = _lib.load()
    buf = workspace.peek(device)
    if buf is None:
        return False
    B, _, H, W = shape
    a, nbytes = _aligned(buf)
    out = C.c_int(0)
    with torch.cuda.device(device):
        check(lib.dagl_ce_range_check(_stream(), B, H, W, MODES[mode], int(k), a, nbytes, C.byref(out)), "dagl_ce_range_check")
    return int(out.value)            # bit 0: range, bit 1: an unserved no-wait adaptive call, bit 2: the last top-k call had a redo pass,
                                     # bit 3: the workspace's top-k threshold policy word says "tight", bit 4: a no-redo call went unserved


@_on_device
def ces_stage_forward(x, head_params, mix_w, mix_b, mode: str = "adaptive", k: int = 0,
                      workspace: "Workspace | None" = None, profile: "StageProfile | None" = None,
                      weights_packed: bool = False, tight_topk: bool = False, sampled_topk: bool = False):
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
                                    MODES[mode] | (_lib.FLAG_WEIGHTS_PACKED if weights_packed else 0) |
                                    ((_lib.FLAG_TIGHT_TOPK if tight_topk else (_lib.FLAG_SAMPLED_TOPK if sampled_topk else 0))
                                     if mode != "adaptive" else 0), int(k),
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


@_on_device
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
    if mode != "adaptive":
        k = min(int(k), N)                 # top_k = min(num_edge, N): the lists are that wide
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
                         redone_queries=info.redone_queries, range_fallback=info.range_fallback,
                         dense_rerun_blocks=info.dense_rerun_blocks)
    return out, saved


@_on_device
def ce_core_backward(d_out, wq_rows, x_rows, b2, thr, bias, saved: dict, mode: str = "adaptive", k: int = 0,
                     workspace: "Workspace | None" = None):
    """Gradients of the graph core (``dagl_ce_core_backward``) -> (d_wq_rows, d_x_rows, d_b2, d_thr, d_bias)."""
    lib = _lib.load()
    for n, t in (("d_out", d_out), ("wq_rows", wq_rows), ("x_rows", x_rows), ("b2", b2)):
        _need(t, n)
    B, _, H, W = b2.shape
    adaptive = mode != "topk"
    if mode != "adaptive":
        k = min(int(k), H * W)
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


@_on_device
def gemm_f32(A: torch.Tensor, B: torch.Tensor, a_k_contiguous: bool = True, b_k_contiguous: bool = False, out=None,
             alpha: float = 1.0, beta: float = 0.0, bias=None, relu: bool = False, chunk_tiles: int = 0,
             split_k: bool = True) -> torch.Tensor:
    """Batched fp32 matrix product on the matrix cores (``dagl_gemm_f32``).  A: [b,M,K] (a_k_contiguous) or [b,K,M];
    B: [b,N,K] (b_k_contiguous) or [b,K,N]; 2-D operands = batch of one.  Returns C [b,M,N] (= alpha A B + beta out)."""
    _need(A, "A"); _need(B, "B")
    squeeze = A.dim() == 2
    if squeeze:
        A, B = A[None], B[None]
    nb = A.shape[0]
    M, K = (A.shape[1], A.shape[2]) if a_k_contiguous else (A.shape[2], A.shape[1])
    N, K2 = (B.shape[1], B.shape[2]) if b_k_contiguous else (B.shape[2], B.shape[1])
    if K != K2 or B.shape[0] != nb:
        raise DaglError("gemm_f32: shape mismatch")
    if out is None:
        if beta != 0.0:
            raise DaglError("gemm_f32: beta needs an existing output")
        out = torch.empty(nb, M, N, device=A.device, dtype=torch.float32)
    else:
        _need(out, "out")
    if bias is not None:
        _need(bias, "bias")
    lib = _lib.load()
    scratch = None
    if split_k and nb == 1:
        nf = lib.dagl_gemm_f32_scratch_floats(nb, M, N, K)
        if nf:
            scratch = torch.empty(nf, device=A.device, dtype=torch.float32)
    check(lib.dagl_gemm_f32(_stream(), nb, M, N, K, A.data_ptr(), A.shape[2], A.shape[1] * A.shape[2],
                            int(a_k_contiguous), B.data_ptr(), B.shape[2], B.shape[1] * B.shape[2],
                            int(b_k_contiguous), out.data_ptr(), N, M * N, float(alpha), float(beta),
                            bias.data_ptr() if bias is not None else None, int(relu), int(chunk_tiles),
                            scratch.data_ptr() if scratch is not None else None), "dagl_gemm_f32")
    return out[0] if squeeze and out.dim() == 3 else out


@_on_device
def ce_core_dense_forward(wq_rows, x_rows, b2, thr, bias, workspace: "Workspace | None" = None, want_info: bool = True,
                          exact: bool = False):
    """Graph core in the dense regime under autograd (``dagl_ce_core_dense_forward``): same operands as
    ``ce_core_forward`` (adaptive mode) -> (out [B,16,H,W], saved dict(lse [B,L,2], mu [B,L], info)).  ``exact``: the
    chunked fp32 GEMM form whatever the size (modules with ``scan="exact"``: no split-fp16 range limit)."""
    lib = _lib.load()
    for n, t in (("wq_rows", wq_rows), ("x_rows", x_rows), ("b2", b2), ("thr", thr), ("bias", bias)):
        _need(t, n)
    B, c, H, W = b2.shape
    Lh, Lw = query_grid(H, W)
    L, N = Lh * Lw, H * W
    if c != 16 or tuple(wq_rows.shape) != (B, L, 196) or tuple(x_rows.shape) != (B, N, 196) or thr.numel() != B * L \
            or bias.numel() != B * L:
        raise DaglError("ce_core_dense_forward: expected wq_rows [B,L,196], x_rows [B,H*W,196], b2 [B,16,H,W], thr/bias [B,L]")
    need = lib.dagl_ce_core_dense_workspace_bytes(B, H, W, 0) + 256
    ws = workspace if workspace is not None else Workspace()
    dev = b2.device
    out = torch.empty(B, 16, H, W, device=dev, dtype=torch.float32)
    lse = torch.empty(B, L, 2, device=dev, dtype=torch.float32)
    mu = torch.empty(B, L, device=dev, dtype=torch.float32)
    info = _lib.CeInfo()
    a, nbytes = _aligned(ws.get(need, dev))
    check(lib.dagl_ce_core_dense_forward(_stream(), B, H, W, _lib.FLAG_EXACT_SCAN if exact else 0, wq_rows.data_ptr(),
                                         x_rows.data_ptr(), b2.data_ptr(), thr.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                         lse.data_ptr(), mu.data_ptr(), a, nbytes, C.byref(info) if want_info else None),
          "dagl_ce_core_dense_forward")
    meta = dict(total_edges=info.total_edges, max_degree=info.max_degree, path=5, redone_queries=-1,
                range_fallback=info.range_fallback, dense_rerun_blocks=info.dense_rerun_blocks) if want_info else None
    return out, dict(lse=lse, mu=mu, info=meta)


@_on_device
def ce_core_dense_backward(d_out, wq_rows, x_rows, b2, thr, bias, saved: dict, workspace: "Workspace | None" = None,
                           exact: bool = False):
    """Gradients of the dense graph core (``dagl_ce_core_dense_backward``) -> (d_wq_rows, d_x_rows, d_b2, d_thr, d_bias).
    ``exact``: the five matrix products on the fp32 matrix cores instead of the fp16 ones with split operands."""
    exact = exact or DENSE_BACKWARD_FP32
    lib = _lib.load()
    for n, t in (("d_out", d_out), ("wq_rows", wq_rows), ("x_rows", x_rows), ("b2", b2), ("thr", thr), ("bias", bias)):
        _need(t, n)
    B, _, H, W = b2.shape
    need = lib.dagl_ce_core_dense_workspace_bytes(B, H, W, 1)
    ws = workspace if workspace is not None else Workspace()
    dev = b2.device
    d_wq, d_x, d_b2 = torch.empty_like(wq_rows), torch.empty_like(x_rows), torch.empty_like(b2)
    d_thr = torch.empty(B, wq_rows.shape[1], device=dev, dtype=torch.float32)
    d_bias = torch.empty_like(d_thr)
    a, nbytes = _aligned(ws.get(need, dev))
    check(lib.dagl_ce_core_dense_backward(_stream(), B, H, W, _lib.FLAG_EXACT_SCAN if exact else 0, wq_rows.data_ptr(), x_rows.data_ptr(), b2.data_ptr(),
                                          thr.data_ptr(), bias.data_ptr(), saved["lse"].data_ptr(), saved["mu"].data_ptr(),
                                          d_out.data_ptr(), d_wq.data_ptr(), d_x.data_ptr(), d_b2.data_ptr(),
                                          d_thr.data_ptr(), d_bias.data_ptr(), a, nbytes), "dagl_ce_core_dense_backward")
    return d_wq, d_x, d_b2, d_thr, d_bias


@_on_device
def ce_core_wide_forward(wq_rows, x_rows, b2, thr, bias, mode: str, k: int, workspace: "Workspace | None" = None,
                         want_info: bool = False):
    """Graph core of the top-k modes whose neighbourhoods exceed the lists (min(k, N) > MAX_TOPK) under autograd
    (``dagl_ce_core_wide_forward``): the dense formulation with the row-wise selection of the k best scores as its mask
    (GReccR2b_3mh_1-checkpoint.py:242-250; CA_model-checkpoint.py:134-143 takes 500) -> (out [B,16,H,W], info | None).
    ``thr`` / ``bias`` are None in mode "topk"."""
    lib = _lib.load()
    if mode not in ("topk", "adaptive_topk"):
        raise DaglError(f"ce_core_wide_forward: mode {mode!r}: expected 'topk' or 'adaptive_topk'")
    for n, t in (("wq_rows", wq_rows), ("x_rows", x_rows), ("b2", b2)) + ((("thr", thr), ("bias", bias)) if mode != "topk" else ()):
        _need(t, n)
    B, c, H, W = b2.shape
    Lh, Lw = query_grid(H, W)
    L, N = Lh * Lw, H * W
    if c != 16 or tuple(wq_rows.shape) != (B, L, 196) or tuple(x_rows.shape) != (B, N, 196):
        raise DaglError("ce_core_wide_forward: expected wq_rows [B,L,196], x_rows [B,H*W,196], b2 [B,16,H,W]")
    need = lib.dagl_ce_core_dense_workspace_bytes(B, H, W, 0) + 256
    ws = workspace if workspace is not None else Workspace()
    out = torch.empty(B, 16, H, W, device=b2.device, dtype=torch.float32)
    info = _lib.CeInfo()
    a, nbytes = _aligned(ws.get(need, b2.device))
    heads = mode != "topk"
    check(lib.dagl_ce_core_wide_forward(_stream(), B, H, W, MODES[mode], int(k), wq_rows.data_ptr(), x_rows.data_ptr(), b2.data_ptr(),
                                        thr.data_ptr() if heads else None, bias.data_ptr() if heads else None, out.data_ptr(),
                                        a, nbytes, C.byref(info) if want_info else None), "dagl_ce_core_wide_forward")
    meta = dict(total_edges=info.total_edges, max_degree=info.max_degree, path=5, redone_queries=-1,
                range_fallback=0) if want_info else None
    return out, meta


@_on_device
def ce_core_wide_backward(d_out, wq_rows, x_rows, b2, thr, bias, mode: str, k: int, workspace: "Workspace | None" = None):
    """Gradients of ``ce_core_wide_forward`` (``dagl_ce_core_wide_backward``) -> (d_wq_rows, d_x_rows, d_b2, d_thr, d_bias);
    the last two are None in mode "topk" (a 0/1 mask has no threshold heads)."""
    lib = _lib.load()
    for n, t in (("d_out", d_out), ("wq_rows", wq_rows), ("x_rows", x_rows), ("b2", b2)):
        _need(t, n)
    B, _, H, W = b2.shape
    need = lib.dagl_ce_core_dense_workspace_bytes(B, H, W, 1)
    ws = workspace if workspace is not None else Workspace()
    dev = b2.device
    heads = mode != "topk"
    d_wq, d_x, d_b2 = torch.empty_like(wq_rows), torch.empty_like(x_rows), torch.empty_like(b2)
    d_thr = torch.empty(B, wq_rows.shape[1], device=dev, dtype=torch.float32) if heads else None
    d_bias = torch.empty_like(d_thr) if heads else None
    a, nbytes = _aligned(ws.get(need, dev))
    check(lib.dagl_ce_core_wide_backward(_stream(), B, H, W, MODES[mode], int(k), wq_rows.data_ptr(), x_rows.data_ptr(), b2.data_ptr(),
                                         thr.data_ptr() if heads else None, bias.data_ptr() if heads else None, d_out.data_ptr(),
                                         d_wq.data_ptr(), d_x.data_ptr(), d_b2.data_ptr(),
                                         d_thr.data_ptr() if heads else None, d_bias.data_ptr() if heads else None, a, nbytes),
          "dagl_ce_core_wide_backward")
    return d_wq, d_x, d_b2, d_thr, d_bias
