"""The prologue convolutions and the two patch projections of ``CE.forward`` (DN_Gray/model/dagl.py:208-249) as
differentiable ops on the HIP library: ``unfold -> fp32 matrix-core GEMM (+ bias, ReLU)`` forward, explicit
``d weight / d bias / d rows -> fold`` backward (include/dagl_ce.h: dagl_unfold_patches, dagl_gemm_f32,
dagl_fold_patches, dagl_copy4, dagl_relu_backward, dagl_col_sum).  No MIOpen / rocBLAS call is made on behalf of the
block when it trains; torch keeps the parameters, the autograd tape and trivial views (weight permutes).
"""
from __future__ import annotations

import os

import torch

from . import _lib, ops
from ._lib import DaglError, check

PAD = 3      # border of the NHWC maps: covers the 3x3 (pad 1), 7x7 (pad 3) and stride-4 SAME (top/left <= 3) windows


def _copy4(src, sizes, s_strides, dst, d_strides):
    check(_lib.load().dagl_copy4(ops._stream(), *sizes, src.data_ptr(), *s_strides, dst.data_ptr(), *d_strides), "dagl_copy4")


class _ToPaddedNHWC(torch.autograd.Function):
    """[B,C,H,W] (NCHW) or [B,H*W,C] rows -> zero-bordered channels-last map [B,H+6,W+6,C]; backward = crop."""

    @staticmethod
    def forward(ctx, x, H, W, from_rows):
        x = x.contiguous()
        B = x.shape[0]
        C = x.shape[2] if from_rows else x.shape[1]
        Hp, Wp = H + 2 * PAD, W + 2 * PAD
        with torch.cuda.device(x.device):
            out = torch.zeros(B, Hp, Wp, C, device=x.device, dtype=torch.float32)
            inner = (PAD * Wp + PAD) * C
            if from_rows:       # src index (b, y, x, c)
                _copy4(x, (B, H, W, C), (H * W * C, W * C, C, 1), out.view(-1)[inner:], (Hp * Wp * C, Wp * C, C, 1))
            else:               # src NCHW, iterate (b, y, x, c) with c fastest in the destination
                _copy4(x, (B, H, W, C), (C * H * W, W, 1, H * W), out.view(-1)[inner:], (Hp * Wp * C, Wp * C, C, 1))
        ctx.geom = (B, C, H, W, from_rows)
        return out

    @staticmethod
    def backward(ctx, d_out):
        B, C, H, W, from_rows = ctx.geom
        d_out = d_out.contiguous()
        Hp, Wp = H + 2 * PAD, W + 2 * PAD
        inner = (PAD * Wp + PAD) * C
        with torch.cuda.device(d_out.device):
            if from_rows:
                dx = torch.empty(B, H * W, C, device=d_out.device, dtype=torch.float32)
                _copy4(d_out.view(-1)[inner:], (B, H, W, C), (Hp * Wp * C, Wp * C, C, 1), dx, (H * W * C, W * C, C, 1))
            else:
                dx = torch.empty(B, C, H, W, device=d_out.device, dtype=torch.float32)
                # iterate (b, c, y, x): x fastest in the destination
                _copy4(d_out.view(-1)[inner:], (B, C, H, W), (Hp * Wp * C, 1, Wp * C, C), dx, (C * H * W, H * W, W, 1))
        return dx, None, None, None


def to_padded_nhwc(x, H, W, from_rows=False):
    return _ToPaddedNHWC.apply(x, H, W, from_rows)


FAST_FC_FORWARD = True          # forward of the two patch projections on the split-fp16 inference kernels (tests flip it)
FAST_FC_BACKWARD = True         # their two gradient products on the split-fp16 GEMM (gemm16s.hip; tests flip it)
FAST_PROLOGUE_FORWARD = True    # g / theta of the differentiable path's forward on the fp16 matrix cores (dagl_ce_prologue16; tests flip it)
FOLD_IN_PRODUCT = True          # ... with d rows of the stride-1 projection folded inside the product (dagl_fc_grad16_dmap; tests flip it)


def _fc_grid(k, C, O, relu, stride, oy, ox, oh, ow, H, W):
    """1 (queries) / 0 (keys) when the call is one of CE's two patch projections on its own grids, else None."""
    if not (k == 7 and C == 16 and O == 196 and relu):
        return None
    if (stride, oy, ox, oh, ow) == (1, 0, 0, H, W):
        return 0
    from .synth import same_pad_amounts
    t, l = same_pad_amounts(H, 7, 4)[0], same_pad_amounts(W, 7, 4)[0]
    if (stride, oy, ox, oh, ow) == (4, PAD - t, PAD - l, -(-H // 4), -(-W // 4)):
        return 1
    return None


class _PatchLinear(torch.autograd.Function):
    """y[b, patch, :] = act(W . unfold(map)[b, patch, :] + bias): a convolution (any kernel size / stride) or a Linear over
    extracted patches.  ``weight`` is [O, k*k*C] with the patch elements in (kh, kw, c) order."""

    @staticmethod
    def forward(ctx, pmap, weight, bias, k, stride, oy, ox, oh, ow, relu, allow_fast=True):
        pmap, weight, bias = pmap.contiguous(), weight.contiguous(), bias.contiguous()
        B, Hp, Wp, C = pmap.shape
        O, K = weight.shape
        if K != k * k * C:
            raise DaglError("patch_linear: weight does not match the patch size")
        lib = _lib.load()
        H, W = Hp - 2 * PAD, Wp - 2 * PAD
        fast = _fc_grid(k, C, O, relu, stride, oy, ox, oh, ow, H, W) if (FAST_FC_FORWARD and allow_fast) else None
        with torch.cuda.device(pmap.device):
            if fast is not None:
                # the two 7x7x16 -> 196 projections (dagl.py:248-249): the inference kernels (split-fp16 matrix cores on the
                # map itself, five times the fp32 rate, no [n, 784] patch rows); the backward below is unchanged
                need = lib.dagl_project_patches16_scratch_bytes(B, H, W, fast)
                scratch = torch.empty(need + 256, device=pmap.device, dtype=torch.uint8)
                base = (scratch.data_ptr() + 255) // 256 * 256
                y = torch.empty(B * oh * ow, O, device=pmap.device, dtype=torch.float32)
                check(lib.dagl_project_patches16(ops._stream(), B, H, W, fast, pmap.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                 y.data_ptr(), base, need), "dagl_project_patches16")
            else:
                rows = torch.empty(B * oh * ow, K, device=pmap.device, dtype=torch.float32)
                check(lib.dagl_unfold_patches(ops._stream(), B, Hp, Wp, C, k, stride, oy, ox, oh, ow, pmap.data_ptr(),
                                              rows.data_ptr()), "dagl_unfold_patches")
                y = ops.gemm_f32(rows, weight, a_k_contiguous=True, b_k_contiguous=True, bias=bias, relu=relu, chunk_tiles=7)
        ctx.geom = (B, Hp, Wp, C, k, stride, oy, ox, oh, ow, relu, O, K)
        ctx.fast_grad = bool(FAST_FC_BACKWARD and allow_fast and (k, C, O) == (7, 16, 196))
        ctx.save_for_backward(pmap, weight, y if relu else pmap.new_empty(0))
        return y.view(B, oh * ow, O)

    @staticmethod
    def backward(ctx, d_y):
        pmap, weight, y = ctx.saved_tensors
        d_map, d_w, d_b = _patch_linear_backward(pmap, weight, y, ctx.geom, d_y, *ctx.needs_input_grad[:3], fast=ctx.fast_grad)
        return d_map, d_w, d_b, None, None, None, None, None, None, None, None


def _patch_linear_backward(pmap, weight, y, geom, d_y, need_map, need_w, need_b, fast=False):
    """d map / d weight / d bias of ``y = act(W . unfold(map) + bias)`` (the patch rows are recomputed, not kept).  ``fast``: the
    two 7x7x16 -> 196 projections take the split-fp16 gradient GEMM (``dagl_fc_grad16``: no fp32 patch rows for d W)."""
    B, Hp, Wp, C, k, stride, oy, ox, oh, ow, relu, O, K = geom
    lib = _lib.load()
    n = B * oh * ow
    with torch.cuda.device(pmap.device):
        dz = d_y.contiguous().view(n, O).float()
        d_w = d_b = d_map = None
        if fast and (need_w or need_map):
            # one call: ReLU backward, the split's scale and d bias in one pass over d y, then the two split-fp16 products
            d_w = torch.empty(O, K, device=pmap.device, dtype=torch.float32) if need_w else None
            d_b = torch.empty(O, device=pmap.device, dtype=torch.float32) if need_b else None
            if need_map and FOLD_IN_PRODUCT and lib.dagl_fc_grad16_dmap_ok(stride, ow):
                # round 6: the key projection (stride 1) folds d rows inside the product -- the [n, 784] rows (411 MB at n = 131 072) are
                # neither written nor read back (dagl_fc_grad16_dmap, gemm16s.hip fold_tile)
                need = lib.dagl_fc_grad16_dmap_scratch_bytes(B, oh, ow)
                scratch = torch.empty(need + 256, device=pmap.device, dtype=torch.uint8)
                base = (scratch.data_ptr() + 255) // 256 * 256
                d_map = torch.empty_like(pmap)
                check(lib.dagl_fc_grad16_dmap(ops._stream(), B, Hp, Wp, stride, oy, ox, oh, ow, pmap.data_ptr(), weight.data_ptr(),
                                              y.data_ptr() if relu else None, dz.data_ptr(), d_w.data_ptr() if need_w else None,
                                              d_b.data_ptr() if need_b else None, d_map.data_ptr(), base, need), "dagl_fc_grad16_dmap")
                return d_map, d_w, d_b
            need = lib.dagl_fc_grad16_scratch_bytes(B, oh, ow)
            scratch = torch.empty(need + 256, device=pmap.device, dtype=torch.uint8)
            base = (scratch.data_ptr() + 255) // 256 * 256
            d_rows = torch.empty(n, K, device=pmap.device, dtype=torch.float32) if need_map else None
            check(lib.dagl_fc_grad16(ops._stream(), B, Hp, Wp, stride, oy, ox, oh, ow, pmap.data_ptr(), weight.data_ptr(),
                                     y.data_ptr() if relu else None, dz.data_ptr(), d_w.data_ptr() if need_w else None,
                                     d_b.data_ptr() if need_b else None, d_rows.data_ptr() if need_map else None, base, need),
                  "dagl_fc_grad16")
            if need_map:
                d_map = torch.empty_like(pmap)
                check(lib.dagl_fold_patches(ops._stream(), B, Hp, Wp, C, k, stride, oy, ox, oh, ow, d_rows.data_ptr(),
                                            d_map.data_ptr()), "dagl_fold_patches")
            return d_map, d_w, d_b
        if relu:
            dzr = torch.empty_like(dz)
            check(lib.dagl_relu_backward(ops._stream(), n * O, y.data_ptr(), dz.data_ptr(), dzr.data_ptr()),
                  "dagl_relu_backward")
            dz = dzr
        rows = torch.empty(n, K, device=pmap.device, dtype=torch.float32)          # recomputed, not kept
        check(lib.dagl_unfold_patches(ops._stream(), B, Hp, Wp, C, k, stride, oy, ox, oh, ow, pmap.data_ptr(),
                                      rows.data_ptr()), "dagl_unfold_patches")
        if need_w:
            # [O,n] x [n,K]: split-K keeps the fma chains at a few thousand products, no chunked accumulation needed
            # (its second accumulator set costs a third of the kernel's occupancy)
            d_w = ops.gemm_f32(dz, rows, a_k_contiguous=False, b_k_contiguous=False)
        if need_b:
            d_b = torch.empty(O, device=pmap.device, dtype=torch.float32)
            scr = torch.empty(lib.dagl_col_sum_scratch_bytes(n, O), device=pmap.device, dtype=torch.uint8)
            check(lib.dagl_col_sum(ops._stream(), n, O, dz.data_ptr(), d_b.data_ptr(), scr.data_ptr()), "dagl_col_sum")
        if need_map:
            d_rows = ops.gemm_f32(dz, weight, a_k_contiguous=True, b_k_contiguous=False, out=rows)   # [n,O] x [O,K]
            d_map = torch.empty_like(pmap)
            check(lib.dagl_fold_patches(ops._stream(), B, Hp, Wp, C, k, stride, oy, ox, oh, ow, d_rows.data_ptr(),
                                        d_map.data_ptr()), "dagl_fold_patches")
    return d_map, d_w, d_b


# tests / A-B runs (bench.py --prologue-backward unfold): the unfold + GEMM + fold backward of g / theta on every shape.  An explicit
# attribute, not an environment variable: nothing outside the process can change gradient bits.
_FORCE_UNFOLD_BACKWARD = False


class _PrologueConvs(torch.autograd.Function):
    """The four prologue convolutions of dagl.py:208-215 in one forward call of the library's fp32 prologue kernels
    (``dagl_ce_prologue``: zero-bordered NHWC maps of g(b) and theta(b), thr / bias per query) -- as unfold + GEMM a 64 -> 16
    convolution fills an eighth of the GEMM's 128-wide tile and writes 300 MB of patch rows first.  The backward is the
    unfold / GEMM / fold one, layer by layer."""

    @staticmethod
    def forward(ctx, x, g_w, g_b, th_w, th_b, thr_w, thr_b, bias_w, bias_b, fast=True):
        x = x.contiguous()
        heads = thr_w is not None
        c = lambda t: t.contiguous() if t is not None else None
        with torch.cuda.device(x.device):
            if x.shape[1] == 64:
                b1p, b2p, thr, bias = ops.ce_prologue(x, c(g_w), c(g_b), c(th_w), c(th_b), c(thr_w), c(thr_b), c(bias_w), c(bias_b),
                                                      fast=bool(fast) and FAST_PROLOGUE_FORWARD and x.dtype == torch.float32)
            else:
                b1p, b2p, thr, bias = prologue_forward_any_width(x, g_w, g_b, th_w, th_b, thr_w, thr_b, bias_w, bias_b)
        ctx.heads = heads
        ctx.save_for_backward(x, g_w, th_w, *( (thr_w, bias_w) if heads else () ))
        if heads:
            return b1p, b2p, thr, bias
        return b1p, b2p

    @staticmethod
    def backward(ctx, d_b1p, d_b2p, d_thr=None, d_bias=None):
        saved = ctx.saved_tensors
        x, g_w, th_w = saved[:3]
        B, C, H, W = x.shape
        Hp, Wp = H + 2 * PAD, W + 2 * PAD
        from .synth import same_pad_amounts
        t, l = same_pad_amounts(H, 7, 4)[0], same_pad_amounts(W, 7, 4)[0]
        Lh, Lw = -(-H // 4), -(-W // 4)
        need_x = ctx.needs_input_grad[0]
        with torch.cuda.device(x.device):
            inner = (PAD * Wp + PAD) * 16

            def crop_rows(d_map):                      # padded NHWC gradient -> rows [B, H*W, 16]
                d_map = d_map.contiguous()
                rows = torch.empty(B, H * W, 16, device=x.device, dtype=torch.float32)
                _copy4(d_map.view(-1)[inner:], (B, H, W, 16), (Hp * Wp * 16, Wp * 16, 16, 1), rows, (H * W * 16, W * 16, 16, 1))
                return rows

            d_xp = None
            lib = _lib.load()
            direct = (C == 64 and bool(lib.dagl_conv_pair_backward_supported(B, H, W)) and x.dtype == torch.float32
                      and not _FORCE_UNFOLD_BACKWARD)
            d_x_direct = None
            grads = {"g": (None, None), "theta": (None, None)}
            if direct:
                # g and theta on the maps themselves (conv_grad.hip): no patch rows, no layout copies
                need_p = any(ctx.needs_input_grad[1:5])
                d1, d2 = d_b1p.contiguous(), d_b2p.contiguous()
                gw, tw = g_w.detach().contiguous(), th_w.detach().contiguous()
                if need_x:
                    d_x_direct = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
                if need_p:
                    d_gw, d_gb = torch.empty_like(gw), torch.empty(16, device=x.device, dtype=torch.float32)
                    d_tw, d_tb = torch.empty_like(tw), torch.empty(16, device=x.device, dtype=torch.float32)
                    scr = torch.empty(max(16, lib.dagl_conv_pair_backward_scratch_bytes(B, H, W)), device=x.device, dtype=torch.uint8)
                    grads = {"g": (d_gw, d_gb), "theta": (d_tw, d_tb)}
                if need_x or need_p:
                    check(lib.dagl_conv_pair_backward(ops._stream(), B, H, W, x.data_ptr(), d1.data_ptr(), d2.data_ptr(), gw.data_ptr(),
                                                      tw.data_ptr(), d_x_direct.data_ptr() if need_x else None,
                                                      d_gw.data_ptr() if need_p else None, d_gb.data_ptr() if need_p else None,
                                                      d_tw.data_ptr() if need_p else None, d_tb.data_ptr() if need_p else None,
                                                      scr.data_ptr() if need_p else None), "dagl_conv_pair_backward")
            if not direct or ctx.heads:
                xp = _ToPaddedNHWC.apply(x.detach(), H, W, False)                              # [B,H+6,W+6,64], recomputed
            if not direct:
              # g (3x3) and theta (1x1 = the centre tap of the 3x3 window) share their patch rows: ONE 32-output layer over the
              # 3x3 patches, theta's weights sitting in the centre tap (a 16-output product fills an eighth of the GEMM's tile)
              w32 = torch.zeros(32, 9 * C, device=x.device, dtype=torch.float32)
              w32[:16] = conv_weight_rows(g_w.detach())
              w32[16:, 4 * C:5 * C] = th_w.detach().reshape(16, C)
              d32 = torch.cat([crop_rows(d_b1p), crop_rows(d_b2p)], dim=-1)                   # [B, H*W, 32]
              geom = (B, Hp, Wp, C, 3, 1, PAD - 1, PAD - 1, H, W, False, 32, 9 * C)
              need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
              need_b = ctx.needs_input_grad[2] or ctx.needs_input_grad[4]
              d_xp, d_w32, d_b32 = _patch_linear_backward(xp, w32, None, geom, d32, need_x, need_w, need_b)
              if d_w32 is not None:
                  grads["g"] = (d_w32[:16].view(16, 3, 3, C).permute(0, 3, 1, 2).contiguous(), None)
                  grads["theta"] = (d_w32[16:, 4 * C:5 * C].reshape(th_w.shape).contiguous(), None)
              if d_b32 is not None:
                  grads["g"] = (grads["g"][0], d_b32[:16].contiguous())
                  grads["theta"] = (grads["theta"][0], d_b32[16:].contiguous())
            d_thr_w = d_thr_b = d_bias_w = d_bias_b = None
            if ctx.heads:
                thr_w, bias_w = saved[3], saved[4]
                w_tb = torch.cat([conv_weight_rows(thr_w.detach()), conv_weight_rows(bias_w.detach())], dim=0).contiguous()
                zero = torch.zeros(B, Lh * Lw, device=x.device, dtype=torch.float32)
                d_tb = torch.stack([(d_thr if d_thr is not None else zero).reshape(B, Lh * Lw),
                                    (d_bias if d_bias is not None else zero).reshape(B, Lh * Lw)], dim=-1).contiguous()   # [B,L,2]
                geom = (B, Hp, Wp, C, 7, 4, PAD - t, PAD - l, Lh, Lw, False, 2, 49 * C)
                need_w = ctx.needs_input_grad[5] or ctx.needs_input_grad[7]
                need_b = ctx.needs_input_grad[6] or ctx.needs_input_grad[8]
                d_map, d_w, d_b = _patch_linear_backward(xp, w_tb, None, geom, d_tb, need_x, need_w, need_b)
                if d_map is not None:
                    d_xp = d_map if d_xp is None else d_xp.add_(d_map)
                if d_w is not None:
                    d_w = d_w.view(2, 7, 7, C).permute(0, 3, 1, 2).contiguous()
                    d_thr_w, d_bias_w = d_w[0:1], d_w[1:2]
                if d_b is not None:
                    d_thr_b, d_bias_b = d_b[0:1], d_b[1:2]
            d_x = None
            if need_x and d_xp is not None:
                d_x = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
                inner64 = (PAD * Wp + PAD) * C
                _copy4(d_xp.view(-1)[inner64:], (B, C, H, W), (Hp * Wp * C, 1, Wp * C, C), d_x, (C * H * W, H * W, W, 1))
            if d_x_direct is not None:
                d_x = d_x_direct if d_x is None else d_x_direct.add_(d_x)
        return (d_x, grads["g"][0], grads["g"][1], grads["theta"][0], grads["theta"][1], d_thr_w, d_thr_b, d_bias_w, d_bias_b, None)


class _PReLU1(torch.autograd.Function):
    """Single-parameter PReLU on the HIP library (``dagl_prelu_forward`` / ``_backward``): the activation of every ResBlock of the
    trunk (DN_Gray/model/common.py:59-79 with ``act = nn.PReLU()``, dagl.py:27-35,76-90).  torch's own backward for it ran at a
    tenth of the memory bandwidth and was 10 ms of the 62 ms training step (24 activations of [8,64,128,128])."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            check(_lib.load().dagl_prelu_forward(ops._stream(), x.numel(), x.data_ptr(), weight.data_ptr(), y.data_ptr()), "dagl_prelu_forward")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.data_ptr() % 16:                     # (a contiguous view at an odd storage offset: the kernel moves float4s)
            dy = dy.clone()
        lib = _lib.load()
        with torch.cuda.device(x.device):
            dx = torch.empty_like(x)
            da = torch.empty_like(weight)
            scratch = torch.empty(max(8, lib.dagl_prelu_scratch_bytes(x.numel())), device=x.device, dtype=torch.uint8)
            check(lib.dagl_prelu_backward(ops._stream(), x.numel(), x.data_ptr(), dy.data_ptr(), weight.data_ptr(), dx.data_ptr(),
                                          da.data_ptr(), scratch.data_ptr()), "dagl_prelu_backward")
        return dx, da


class PReLU(torch.nn.PReLU):
    """``nn.PReLU()`` (one shared slope; same parameter name, shape and initial value, so reference checkpoints load unchanged) whose
    fp32 GPU calls run on the HIP library; anything else (CPU, half precision, per-channel slopes, odd sizes) takes torch's path."""

    def forward(self, x):
        # (the kernels move float4s: contiguous views at an odd storage offset take torch's path like any other unsupported input)
        if (x.is_cuda and x.dtype == torch.float32 and self.weight.numel() == 1 and self.weight.dtype == torch.float32
                and x.numel() % 4 == 0 and x.numel() > 0 and x.is_contiguous() and x.data_ptr() % 16 == 0):
            return _PReLU1.apply(x, self.weight)
        return super().forward(x)


def prologue_forward_any_width(x, g_w, g_b, th_w, th_b, thr_w=None, thr_b=None, bias_w=None, bias_b=None):
    """The four prologue convolutions (dagl.py:208-215) for ANY input width -- ``CE(in_channels = n_feats)``, dagl.py:94-109 -- as
    unfold + fp32 matrix-core GEMM on the HIP library (the fused kernels of prologue.hip are laid out for 64 channels): g (3x3)
    and theta (1x1 = the centre tap) as one 32-output layer over the 3x3 patches, thr / bias as one 2-output layer over the
    stride-4 SAME 7x7 patches.  Same outputs as ``ops.ce_prologue``: zero-bordered NHWC maps [B,H+6,W+6,16] and [B,L] heads."""
    from .synth import same_pad_amounts
    lib = _lib.load()
    x = x.contiguous()
    B, C, H, W = x.shape
    Hp, Wp = H + 2 * PAD, W + 2 * PAD
    with torch.cuda.device(x.device), torch.no_grad():
        xp = _ToPaddedNHWC.apply(x.detach(), H, W, False)
        w32 = torch.zeros(32, 9 * C, device=x.device, dtype=torch.float32)
        w32[:16] = conv_weight_rows(g_w.detach())
        w32[16:, 4 * C:5 * C] = th_w.detach().reshape(16, C)
        b32 = torch.cat([g_b.detach(), th_b.detach()]).contiguous()
        rows = torch.empty(B * H * W, 9 * C, device=x.device, dtype=torch.float32)
        check(lib.dagl_unfold_patches(ops._stream(), B, Hp, Wp, C, 3, 1, PAD - 1, PAD - 1, H, W, xp.data_ptr(), rows.data_ptr()),
              "dagl_unfold_patches")
        y32 = ops.gemm_f32(rows, w32, a_k_contiguous=True, b_k_contiguous=True, bias=b32, relu=False, chunk_tiles=7)   # [B*H*W, 32]
        del rows
        maps = []
        for o in (0, 16):
            m = torch.zeros(B, Hp, Wp, 16, device=x.device, dtype=torch.float32)
            _copy4(y32.view(-1)[o:], (B, H, W, 16), (H * W * 32, W * 32, 32, 1), m.view(-1)[(PAD * Wp + PAD) * 16:], (Hp * Wp * 16, Wp * 16, 16, 1))
            maps.append(m)
        thr = bias = None
        if thr_w is not None:
            t, l = same_pad_amounts(H, 7, 4)[0], same_pad_amounts(W, 7, 4)[0]
            Lh, Lw = -(-H // 4), -(-W // 4)
            w_tb = torch.cat([conv_weight_rows(thr_w.detach()), conv_weight_rows(bias_w.detach())], dim=0).contiguous()
            b_tb = torch.cat([thr_b.detach(), bias_b.detach()]).contiguous()
            rows = torch.empty(B * Lh * Lw, 49 * C, device=x.device, dtype=torch.float32)
            check(lib.dagl_unfold_patches(ops._stream(), B, Hp, Wp, C, 7, 4, PAD - t, PAD - l, Lh, Lw, xp.data_ptr(), rows.data_ptr()),
                  "dagl_unfold_patches")
            y2 = ops.gemm_f32(rows, w_tb, a_k_contiguous=True, b_k_contiguous=True, bias=b_tb, relu=False, chunk_tiles=7).view(B, Lh * Lw, 2)
            thr, bias = y2[..., 0].contiguous(), y2[..., 1].contiguous()
    return maps[0], maps[1], thr, bias


def prologue_convs(x, g, theta, thr_conv=None, bias_conv=None, fast=True):
    """(b1p, b2p[, thr, bias]) of the block input: zero-bordered NHWC maps [B,H+6,W+6,16] and per-query thr / bias [B,L].
    ``fast=False`` (modules moved to ``scan="exact"``): g / theta on the fp32 matrix cores instead of the split-fp16 ones."""
    if thr_conv is None:
        return _PrologueConvs.apply(x, g.weight, g.bias, theta.weight, theta.bias, None, None, None, None, fast)
    return _PrologueConvs.apply(x, g.weight, g.bias, theta.weight, theta.bias, thr_conv.weight, thr_conv.bias,
                                bias_conv.weight, bias_conv.bias, fast)


def patch_linear(pmap, weight, bias, k, stride, oy, ox, oh, ow, relu=False, allow_fast=True):
    """``allow_fast=False`` keeps CE's two patch projections on the fp32 GEMM forward (modules moved to ``scan="exact"``)."""
    return _PatchLinear.apply(pmap, weight, bias, k, stride, oy, ox, oh, ow, relu, allow_fast)


def conv_weight_rows(w: torch.Tensor) -> torch.Tensor:
    """conv weight [O,C,kh,kw] -> [O, kh*kw*C] in the unfold's (kh,kw,c) element order (a torch view + copy: autograd
    carries the gradient back to the parameter)."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def fc_weight_rows(w: torch.Tensor, c: int, k: int) -> torch.Tensor:
    """Linear weight over Unfold's (c,kh,kw) patch order (dagl.py:196-203, :248-249) -> (kh,kw,c) order."""
    return w.view(w.shape[0], c, k, k).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
