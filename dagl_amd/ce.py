"""``CE`` -- drop-in replacement of the reference's patch-graph attention head.

Mirrors ``CE`` of /root/reference/DN_Gray/model/dagl.py:174-277 (and its forks in
CAR/, Demosaic/, DN_Real/): same constructor signature, same parameter names and
shapes (including the registered-but-unused ``W``, dagl.py:192), same
``forward(b: [B,in_channels,H,W]) -> [B,inter_channels,H,W]`` -- so ``CES``
(dagl.py:74-119) accepts it by class substitution and reference checkpoints load
unchanged.

The whole method -- the four prologue convolutions (dagl.py:208-215) and
everything from patch extraction to the batch concat (dagl.py:216-274) -- runs
in the HIP library behind ``include/dagl_ce.h``; torch only owns the parameters,
the device memory and the stream.  There is no eager fallback.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import ERR_UNSUPPORTED, FAST_CAP, MAX_TOPK, DaglError
from .synth import same_pad_amounts


class _GraphCore(torch.autograd.Function):
    """dagl.py:250-272 as one differentiable op: HIP forward (``dagl_ce_core_forward``) keeps each query's neighbour
    list, HIP backward (``dagl_ce_core_backward``) returns the gradients autograd would derive from the dense form."""

    @staticmethod
    def forward(ctx, wq_rows, x_rows, b2, thr, bias, mode, k, exact_scan, ws_f, ws_b, sink):
        wq_rows, x_rows, b2 = wq_rows.contiguous(), x_rows.contiguous(), b2.contiguous()
        adaptive = mode != "topk"
        thr_c = thr.contiguous() if adaptive else None
        bias_c = bias.contiguous() if adaptive else None
        out, saved = ops.ce_core_forward(wq_rows, x_rows, b2, thr_c, bias_c, mode=mode, k=k, workspace=ws_f,
                                         exact_scan=exact_scan)
        ctx.mode, ctx.k, ctx.ws_b, ctx.adaptive = mode, k, ws_b, adaptive
        ctx.thr_shape = thr.shape if adaptive else None
        tensors = [wq_rows, x_rows, b2, saved["nb_idx"], saved["nb_wgt"], saved["nb_s"], saved["nb_cnt"]]
        if adaptive:
            tensors += [thr_c, bias_c, saved["mu"]]
        ctx.save_for_backward(*tensors)
        if sink is not None:
            sink.update(saved["info"])
        return out

    @staticmethod
    def backward(ctx, d_out):
        t = ctx.saved_tensors
        wq_rows, x_rows, b2, nb_idx, nb_wgt, nb_s, nb_cnt = t[:7]
        thr, bias, mu = (t[7], t[8], t[9]) if ctx.adaptive else (None, None, None)
        saved = dict(nb_idx=nb_idx, nb_wgt=nb_wgt, nb_s=nb_s, nb_cnt=nb_cnt, mu=mu)
        d_wq, d_x, d_b2, d_thr, d_bias = ops.ce_core_backward(d_out.contiguous().float(), wq_rows, x_rows, b2, thr, bias,
                                                              saved, mode=ctx.mode, k=ctx.k, workspace=ctx.ws_b)
        if ctx.adaptive:
            d_thr, d_bias = d_thr.view(ctx.thr_shape), d_bias.view(ctx.thr_shape)
        return d_wq, d_x, d_b2, d_thr, d_bias, None, None, None, None, None, None


class _GraphCoreDense(torch.autograd.Function):
    """dagl.py:250-272 for dense neighbourhoods (adaptive masks that keep more keys than a fixed-width list holds --
    default-initialised heads keep ~95 %): the dense formulation (``dagl_ce_core_dense_forward`` / ``_backward``): the forward
    on the streamed split-fp16 kernel, the backward's matrix products on the fp16 matrix cores with split operands
    (``exact``: both on the fp32 matrix cores; dense_train.hip)."""

    @staticmethod
    def forward(ctx, wq_rows, x_rows, b2, thr, bias, ws_f, ws_b, sink, want_info, exact=False):
        wq_rows, x_rows, b2 = wq_rows.contiguous(), x_rows.contiguous(), b2.contiguous()
        thr_c, bias_c = thr.contiguous(), bias.contiguous()
        out, saved = ops.ce_core_dense_forward(wq_rows, x_rows, b2, thr_c, bias_c, workspace=ws_f, want_info=want_info,
                                               exact=exact)
        ctx.ws_b, ctx.thr_shape, ctx.exact = ws_b, thr.shape, bool(exact)
        ctx.save_for_backward(wq_rows, x_rows, b2, thr_c, bias_c, saved["lse"], saved["mu"])
        if sink is not None and saved["info"] is not None:
            sink.update(saved["info"])
        return out

    @staticmethod
    def backward(ctx, d_out):
        wq_rows, x_rows, b2, thr, bias, lse, mu = ctx.saved_tensors
        d_wq, d_x, d_b2, d_thr, d_bias = ops.ce_core_dense_backward(d_out.contiguous().float(), wq_rows, x_rows, b2, thr, bias,
                                                                    dict(lse=lse, mu=mu), workspace=ctx.ws_b, exact=ctx.exact)
        return d_wq, d_x, d_b2, d_thr.view(ctx.thr_shape), d_bias.view(ctx.thr_shape), None, None, None, None, None


class _GraphCoreGeneric(torch.autograd.Function):
    """dagl.py:250-272 under autograd for a module built with a non-default patch geometry (``dagl_ce_generic_core_forward`` /
    ``_backward``, csrc/generic.hip): the dense formulation on the fp32 matrix cores, S and A recomputed chunk by chunk in the backward."""

    @staticmethod
    def forward(ctx, wq_rows, x_rows, b2p, thr, bias, geom, mode, k, scale, ws_f, ws_b):
        H, W, ks, s1, s2 = geom
        heads = mode != "topk"
        wq_rows, x_rows, b2p = wq_rows.contiguous(), x_rows.contiguous(), b2p.contiguous()
        thr_c, bias_c = (thr.contiguous(), bias.contiguous()) if heads else (None, None)
        out = ops.ce_generic_core_forward(wq_rows, x_rows, b2p, thr_c, bias_c, H, W, ks, s1, s2, mode=mode, k=k, softmax_scale=scale, workspace=ws_f)
        ctx.geom, ctx.mode, ctx.k, ctx.scale, ctx.ws_b, ctx.heads = geom, mode, k, scale, ws_b, heads
        ctx.save_for_backward(*([wq_rows, x_rows, b2p] + ([thr_c, bias_c] if heads else [])))
        return out

    @staticmethod
    def backward(ctx, d_out):
        wq_rows, x_rows, b2p, *tb = ctx.saved_tensors
        thr, bias = tb if ctx.heads else (None, None)
        H, W, ks, s1, s2 = ctx.geom
        d_wq, d_x, d_b2p, d_thr, d_bias = ops.ce_generic_core_backward(d_out.contiguous().float(), wq_rows, x_rows, b2p, thr, bias, H, W, ks, s1, s2,
                                                                       mode=ctx.mode, k=ctx.k, softmax_scale=ctx.scale, workspace=ctx.ws_b)
        return d_wq, d_x, d_b2p, d_thr, d_bias, None, None, None, None, None, None


class _GraphCoreWide(torch.autograd.Function):
    """dagl.py:250-272 in the top-k modes when min(k, N) exceeds the lists' width (the fixed-k variant takes any num_edge,
    GReccR2b_3mh_1-checkpoint.py:242-250; CA_model-checkpoint.py:134-143 uses 500): the dense formulation with the row-wise
    selection of the k best scores as its mask (``dagl_ce_core_wide_forward`` / ``_backward``; dense_train.hip, wide_select.h)."""

    @staticmethod
    def forward(ctx, wq_rows, x_rows, b2, thr, bias, mode, k, ws_f, ws_b, sink):
        wq_rows, x_rows, b2 = wq_rows.contiguous(), x_rows.contiguous(), b2.contiguous()
        heads = mode != "topk"
        thr_c, bias_c = (thr.contiguous(), bias.contiguous()) if heads else (None, None)
        out, info = ops.ce_core_wide_forward(wq_rows, x_rows, b2, thr_c, bias_c, mode, k, workspace=ws_f, want_info=sink is not None)
        ctx.ws_b, ctx.mode, ctx.k, ctx.heads = ws_b, mode, k, heads
        ctx.thr_shape = thr.shape if heads else None
        ctx.save_for_backward(*([wq_rows, x_rows, b2] + ([thr_c, bias_c] if heads else [])))
        if sink is not None and info is not None:
            sink.update(info)
        return out

    @staticmethod
    def backward(ctx, d_out):
        wq_rows, x_rows, b2, *tb = ctx.saved_tensors
        thr, bias = tb if ctx.heads else (None, None)
        d_wq, d_x, d_b2, d_thr, d_bias = ops.ce_core_wide_backward(d_out.contiguous().float(), wq_rows, x_rows, b2, thr, bias,
                                                                   ctx.mode, ctx.k, workspace=ctx.ws_b)
        if ctx.heads:
            d_thr, d_bias = d_thr.view(ctx.thr_shape), d_bias.view(ctx.thr_shape)
        return d_wq, d_x, d_b2, d_thr, d_bias, None, None, None, None, None


class _EvalLazyGrad(torch.autograd.Function):
    """An eval() block inside an autograd-enabled forward (the reference's test loop, DN_Gray/trainer.py:128-140, builds a
    graph it never uses): forward on the inference kernels, nothing saved but the input; a backward -- rare -- recomputes
    the block on the differentiable path (``CE._forward_train``) and hands its gradients on."""

    @staticmethod
    def forward(ctx, module, k_eff, n_params, b, *params):
        ctx.module, ctx.n_params = module, n_params
        ctx.save_for_backward(b, *params)
        with torch.no_grad():
            return module._forward_infer(b, k_eff)

    @staticmethod
    def backward(ctx, d_out):
        b, *params = ctx.saved_tensors
        with torch.enable_grad():
            bb = b.detach().requires_grad_(True)
            out = ctx.module._forward_train(bb)
            used = [p for p in params]
            grads = torch.autograd.grad(out, [bb] + used, d_out.contiguous().float(), allow_unused=True)
        return (None, None, None) + tuple(grads)


class CE(nn.Module):
    def __init__(self, ksize=7, stride_1=4, stride_2=1, softmax_scale=10, shape=64, p_len=64, in_channels=64,
                 inter_channels=16, use_multiple_size=False, use_topk=False, add_SE=False, num_edge=50):
        super().__init__()
        # Patch geometry (dagl.py:175-176).  The reference's own builders never override it (dagl.py:94-109) and every tuned kernel has
        # (7, 4, 1, 16) compiled in; any other geometry runs the reference's dense formulation with the geometry as run-time arguments
        # (dagl_ce_generic_forward, csrc/generic.hip: fp32 matrix cores, inference only).
        self._generic = (int(ksize), int(stride_1), int(stride_2), int(inter_channels)) != (7, 4, 1, 16)
        if self._generic:
            if not (1 <= int(ksize) <= 31 and int(stride_1) >= 1 and int(stride_2) >= 1):
                raise DaglError(f"CE: ksize={ksize} (1..31), stride_1={stride_1}, stride_2={stride_2} (>= 1)")
            if int(inter_channels) < 4 or int(inter_channels) % 4:
                raise DaglError(f"CE: inter_channels={inter_channels} must be a multiple of 4 (16-byte pixels of the NHWC maps)")
        if not float(softmax_scale) > 0.0:
            raise DaglError(f"CE: softmax_scale={softmax_scale} must be positive")
        self.ksize, self.shape, self.p_len = ksize, shape, p_len
        self.stride_1, self.stride_2 = stride_1, stride_2
        self.softmax_scale = softmax_scale
        self.inter_channels, self.in_channels = inter_channels, in_channels
        self.use_multiple_size, self.use_topk, self.add_SE = use_multiple_size, use_topk, add_SE
        self.num_edge = num_edge
        # same registration order and names as dagl.py:190-205
        self.g = nn.Conv2d(in_channels, inter_channels, kernel_size=3, stride=1, padding=1)
        self.W = nn.Conv2d(inter_channels, in_channels, kernel_size=1, stride=1, padding=0)   # never applied
        self.theta = nn.Conv2d(in_channels, inter_channels, kernel_size=1, stride=1, padding=0)
        feat = ksize ** 2 * inter_channels
        self.fc1 = nn.Sequential(nn.Linear(feat, feat // 4), nn.ReLU())
        self.fc2 = nn.Sequential(nn.Linear(feat, feat // 4), nn.ReLU())
        self.thr_conv = nn.Conv2d(in_channels, 1, kernel_size=ksize, stride=stride_1, padding=0)
        self.bias_conv = nn.Conv2d(in_channels, 1, kernel_size=ksize, stride=stride_1, padding=0)
        # Neighbour selection.  The shipped forward ignores use_topk/num_edge (dead arguments,
        # dagl.py:176,187,189) and always applies the adaptive mask; ``select_mode`` makes the fixed-k
        # variant (GReccR2b_3mh_1-checkpoint.py:242-250) and the intersection available:
        #   "adaptive" | "topk" | "adaptive_topk", with k = ``select_k``.
        self.select_mode = "adaptive"
        self.select_k = num_edge       # the fixed-k variant's own default is 50; k > MAX_TOPK: row-wise dense form (no lists)
        # "screened": bf16 matrix-core screen of all L*N scores + exact refinement of the survivors (default);
        # "exact": every score on the fp32 matrix cores.  Same neighbours either way.
        self.scan = "screened"
        self._ws = ops.Workspace()
        self._ws_bwd = ops.Workspace()
        self._pack_key = None
        self._pack_epoch = 0           # bumped by invalidate_packed()
        self._f32_cache = {}              # fp32 copies of half-precision parameters (model.half(), DN_Gray/model/__init__.py:98-99)
        self._scaled_cache = {}           # softmax_scale != 10: scaled copies of fc1 / the bias head (_scale_c)
        self._train_calls = 0
        self._wide_calls = 0
        self._last_call = None
        self._calls_since_range_check = 0
        self._train_dense = False      # the differentiable path met dense neighbourhoods last time (dense_train.hip)
        self._train_dense_calls = 0
        self._dense_hint = False       # the last adaptive call ended in the dense formulation: start there next time
        self._dense_calls = 0
        # adaptive mode: "always" (default) = read the call's verdict on the host every call: any neighbourhood structure is
        # served (lists + per-query redo, dense formulation, CSR lists).  "auto" (opt-in, for steady sparse workloads and HIP-graph
        # capture) = once four calls in a row were served in-stream, stop waiting (DAGL_FLAG_NO_WAIT: no synchronisation at
        # all); the device-side verdict NaN-fills a call the in-stream kernels could not serve -- never wrong numbers -- and
        # the sticky word is polled every 16th call, which sends the module back to waiting.
        self.adaptive_sync = "always"
        # top-k modes: where the candidate threshold comes from.  "sparse" = every 8th key tile (DAGL_FLAG_SAMPLED_TOPK: enough on
        # maps whose scores are spread evenly, e.g. the synthetic benchmark features); "full" = every second key tile + eight times
        # the candidate slots (DAGL_FLAG_TIGHT_TOPK: +25 us at 256^2) -- on natural-image features the sampled threshold lets
        # hundreds of keys per query through, the slots overflow and the call lands on the fp32 redo pass (2.7 ms instead of 0.25);
        # "auto" (default) = the workspace's own policy word, read by the kernels themselves (round 4: no host poll, valid under
        # HIP-graph replay): sampled until any query of a call overflows (a full segment spills into the query's shared area first),
        # tight from then on; the first call of a shape re-runs tight in-stream; maps of <= 16 384 keys start tight.
        self.topk_threshold = "auto"
        # Top-k modes behind the screen: "always" (default) = every call queues the fp32 redo pass for query groups whose candidate slots
        # overflowed (a launch that finds nothing on a warm workspace: 4.7 us under rocprofv3, which serialises dispatches); "auto" =
        # once three polls of the workspace in a row (every 64th call, as for the range word) have found the last call without redo work, identical
        # calls (same shape, weights, workspace) go without the launch (DAGL_FLAG_NO_REDO) and the workspace is polled every 32nd call;
        # a call that flags a group after all returns NaN -- never wrong numbers --, the next poll reports it and the module queues
        # the pass again from then on.  Not under HIP-graph capture (a replayed graph is never polled).  Measured on the headline
        # (profiles/r05_ab_topk_redo_launch.log, same box, interleaved): 0.2250 ms with "auto" against 0.2254 with "always" -- in the
        # un-profiled stream the empty launch runs under its neighbours' dispatch and the polls cost what is left; hence not the default.
        self.topk_redo = "always"
        self._redo_skip = False        # the last three polls found no redo work
        self._redo_clean_polls = 0
        self._redo_banned = False      # a no-redo call went unserved once: never again on this module
        self._served_streak = 0
        self._served_streaks = {}      # the same for adaptive_sync = "auto"
        self._served_shape = None
        self._nowait_calls = 0
        self.last_info = None
        self.profile = None            # optional ops.StageProfile (benchmark instrumentation)

    def reset_topk_policy(self):
        """Forget what ``topk_threshold = "auto"`` has learnt: the policy word lives in the workspace and starts afresh with a cold
        one (the next call re-packs the weights there)."""
        self._pack_key = None

    def topk_policy_is_tight(self) -> bool:
        """Whether this module's workspace has switched to the tight threshold (bit 3 of ``dagl_ce_range_check``; one host
        synchronisation; reads -- and clears, if set -- the sticky range word like ``range_ok``)."""
        if self._last_call is None or self.select_mode == "adaptive" or self.scan == "exact":
            return False
        shape, dev = self._last_call
        bad = ops.ce_range_check(shape, self.select_mode, min(int(self.select_k), shape[2] * shape[3]), self._ws, dev)
        if bad & 1:
            self._note_range_violation("a call left the split-fp16 range (its output is NaN-filled)")
        return bool(bad & 8)

    def range_ok(self) -> bool:
        """False when an inference call on this module since the last poll met operands the split-fp16 kernels do not hold
        (include/dagl_ce.h "Range": since round 6 that is a non-finite input or |g(x)| >= 1.5e7 -- the input itself has no limit
        and the key / query map two tiers --, or dense-regime features >= 937): that call's output is NaN-filled.  Switches the
        module to ``scan = "exact"`` (the fp32 path, no range limit) in that case.  One host synchronisation."""
        self._calls_since_range_check = 0
        if self.scan == "exact" or self._last_call is None:
            return True
        shape, dev = self._last_call
        bad = ops.ce_range_check(shape, self.select_mode, min(int(self.select_k), shape[2] * shape[3]) if self.select_mode != "adaptive" else 0,
                                 self._ws, dev)
        if self.select_mode != "adaptive":
            if bad & 16:
                import warnings
                warnings.warn("dagl_amd.CE: a top-k call that went without the fp32 redo pass (topk_redo = 'auto') met query groups whose "
                              "candidate slots overflowed: its output is NaN-filled; this module queues the pass again from now on")
                self._redo_banned, self._redo_skip, self._redo_clean_polls = True, False, 0
            else:
                # three polls in a row without redo work (not one: inputs that overflow the candidate slots now and then would hand
                # out NaN-filled calls until the next poll)
                self._redo_clean_polls = 0 if (bad & 4) else self._redo_clean_polls + 1
                self._redo_skip = self._redo_clean_polls >= 3 and not self._redo_banned
        if bad & 2:
            import warnings
            warnings.warn("dagl_amd.CE: an adaptive call that did not wait for its verdict met neighbourhoods the in-stream kernels "
                          "could not serve (its output is NaN-filled); the module reads the verdict on the host again")
            self._served_streak = 0
        if bad & 1:
            self._note_range_violation("a call left the split-fp16 range (its output is NaN-filled)")
        return not (bad & 19)

    def _note_range_violation(self, what):
        import warnings
        warnings.warn(f"dagl_amd.CE: {what}; this module now uses scan='exact' (fp32 matrix cores, no range limit)")
        self.scan = "exact"
        self._pack_key = None

    def invalidate_packed(self):
        """Forget the packed copies of fc1 / fc2 / g / theta kept in the workspace.  The cache is keyed on the weights' storage and
        torch's version counter, which every in-place op and optimizer step bumps -- but edits through ``.data``
        (``w.data.mul_()``, EMA / clipping written that way) do NOT: call this after such an edit.  ``.to()`` / ``.cuda()`` /
        ``.half()`` (``_apply``) and ``load_state_dict`` call it themselves."""
        self._pack_epoch += 1
        self._pack_key = None
        self._f32_cache = {}
        self._scaled_cache = {}

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_packed()
        return super()._load_from_state_dict(*args, **kwargs)

    def extra_repr(self):
        return f"select_mode={self.select_mode!r}, select_k={self.select_k}"

    def _scale_c(self) -> float:
        """softmax_scale other than the kernels' 10 (dagl.py:175, 260), without touching a kernel: the logits are
        softmax_scale * S * m with m = relu(S - mean(S) thr + bias) (top-k variant: m in {0, 1}).  Scaling the query features by c
        (fc1's weight and bias: ReLU is positively homogeneous) scales S and mean(S) thr by c; scaling the bias head by c as well
        scales m by c, the mask (m > 0) and the top-k order do not move, and the kernels' 10 * S' * m' is 10 c^2 S m (10 c S for the
        top-k variant).  So c = sqrt(softmax_scale / 10), or softmax_scale / 10 in "topk" mode."""
        r = float(self.softmax_scale) / 10.0
        if r == 1.0:
            return 1.0
        return r if self.select_mode == "topk" else math.sqrt(r)

    def _params_f32(self):
        """The block's parameters as contiguous fp32 tensors.  fp32 modules: the parameters themselves.  ``model.half()`` /
        ``.bfloat16()`` modules (the reference's ``--precision half`` test path, DN_Gray/model/__init__.py:98-99,
        option.py:76-78): converted once and kept until the parameter changes (storage or version counter) -- the block
        computes in fp32 on the values the half-precision weights hold."""
        out = {}
        for n, p in self.named_parameters():
            if n.startswith("W."):
                continue
            t = p.detach()
            if t.dtype == torch.float32:
                out[n] = t.contiguous()
                continue
            if t.dtype not in (torch.float16, torch.bfloat16):
                raise DaglError(f"CE: parameter {n} has dtype {t.dtype}; fp32, fp16 or bf16 expected")
            tag = (t.data_ptr(), t._version, t.dtype, t.device)
            hit = self._f32_cache.get(n)
            if hit is None or hit[0] != tag:
                hit = (tag, t.float().contiguous())
                self._f32_cache[n] = hit
            out[n] = hit[1]
        c = self._scale_c()
        if c == 1.0:
            self._scaled_cache = {}
        else:                           # (see _scale_c; scaled copies kept until the PARAMETER -- not its fp32 copy, whose version
                                        # counter never moves -- or the factor changes)
            own = dict(self.named_parameters())
            for n in ("fc1.0.weight", "fc1.0.bias") + (() if self.select_mode == "topk" else ("bias_conv.weight", "bias_conv.bias")):
                src = out[n]
                tag = (own[n].data_ptr(), own[n]._version, own[n].dtype, own[n].device, c)
                hit = self._scaled_cache.get(n)
                if hit is None or hit[0] != tag:
                    hit = (tag, (src * c).contiguous())
                    self._scaled_cache[n] = hit
                out[n] = hit[1]
        return out

    def _prologue(self, b):
        """The four prologue convolutions of dagl.py:208-215 as stock torch ops (MIOpen) -- kept for the
        stage-level parity tests; forward() computes them inside the HIP library (prologue.hip)."""
        b1 = self.g(b)
        b2 = self.theta(b)
        H, W = b.shape[-2:]
        t, bo = same_pad_amounts(H, self.ksize, self.stride_1)
        l, r = same_pad_amounts(W, self.ksize, self.stride_1)
        b4 = F.pad(b, (l, r, t, bo))
        thr = self.thr_conv(b4).reshape(b.shape[0], -1)
        bias = self.bias_conv(b4).reshape(b.shape[0], -1)
        return b1, b2, thr, bias

    def _forward_train(self, b: torch.Tensor) -> torch.Tensor:
        """Differentiable path (DN_Gray/trainer.py:44-50), every stage on the HIP library: the four prologue convolutions
        and the two patch projections as unfold + fp32 matrix-core GEMM with explicit backward (train_ops.py; fc(unfold(.))
        = a product of the patch rows with the Linear weight, dagl.py:240-249), the graph core (dagl.py:250-272) as a HIP
        op with its own backward: neighbour lists (top-k modes, adaptive masks keeping <= 64 keys per query) or, for
        denser masks, the dense formulation."""
        from . import train_ops as T
        if any(p.dtype != torch.float32 for p in self.parameters()):
            raise DaglError("CE: the differentiable path needs fp32 parameters (the reference's --precision half is a test-time "
                            "switch, DN_Gray/model/__init__.py:98-99); run half-precision modules under torch.no_grad()")
        self._last_call = None         # the shared workspace is about to be reused with the training layout: no range word to poll
        B, _, H, W = b.shape
        ks, c = self.ksize, self.inter_channels
        t, _bo = same_pad_amounts(H, ks, self.stride_1)
        l, _r = same_pad_amounts(W, ks, self.stride_1)
        Lh, Lw = -(-H // self.stride_1), -(-W // self.stride_1)
        # dagl.py:208-215  g (3x3, pad 1), theta (1x1), thr_conv / bias_conv (7x7 stride 4 on the SAME-padded input): one forward
        # call of the library's fp32 prologue kernels, layer-by-layer unfold / GEMM backward (train_ops._PrologueConvs)
        thr = bias = None
        convs = (self.g, self.theta, self.thr_conv, self.bias_conv)
        padc = (-self.in_channels) % 4                  # (float4 channel groups: zero channels, zero weight columns -- autograd slices them off)
        if padc:
            from types import SimpleNamespace
            b = F.pad(b, (0, 0, 0, 0, 0, padc))
            convs = tuple(SimpleNamespace(weight=F.pad(m.weight, (0, 0, 0, 0, 0, padc)), bias=m.bias) for m in convs)
        if self.select_mode != "topk":                 # (the fixed-k variant has no threshold heads)
            b1p, b2p, thr, bias = T.prologue_convs(b, *convs, fast=self.scan != "exact")
        else:
            b1p, b2p = T.prologue_convs(b, convs[0], convs[1], fast=self.scan != "exact")
        b2 = b2p[:, T.PAD:T.PAD + H, T.PAD:T.PAD + W, :].permute(0, 3, 1, 2)                  # NCHW view of the value map
        # dagl.py:216-249  patches of b1 (stride 4 SAME / stride 1) through fc1 / fc2 + ReLU
        wq_rows = T.patch_linear(b1p, T.fc_weight_rows(self.fc1[0].weight, c, ks), self.fc1[0].bias, ks, self.stride_1,
                                 T.PAD - t, T.PAD - l, Lh, Lw, relu=True, allow_fast=self.scan != "exact")   # [B,L,196]
        x_rows = T.patch_linear(b1p, T.fc_weight_rows(self.fc2[0].weight, c, ks), self.fc2[0].bias, ks, self.stride_2,
                                0, 0, H, W, relu=True, allow_fast=self.scan != "exact")             # [B,N,196]
        sc = self._scale_c()
        if sc != 1.0:                   # softmax_scale != 10: see _scale_c (plain torch ops: autograd carries the factor)
            wq_rows = wq_rows * sc
            if bias is not None:
                bias = bias * sc
        info = {}
        self._pack_key = None          # the shared workspace is reused with another layout
        out = None
        k_eff = min(int(self.select_k), H * W) if self.select_mode != "adaptive" else 0
        if k_eff > MAX_TOPK:
            # more neighbours than the lists hold: the dense formulation with the row-wise selection as its mask
            want = self._wide_calls % 64 == 0           # (statistics cost a host synchronisation: every 64th call)
            self._wide_calls += 1
            out = _GraphCoreWide.apply(wq_rows, x_rows, b2, thr, bias, self.select_mode, k_eff, self._ws, self._ws_bwd,
                                       info if want else None)
        elif self.select_mode == "adaptive" and self._train_dense:
            # the last training call met dense neighbourhoods: start in the dense formulation; every 16th call reads the
            # degrees back (one host synchronisation) to notice when the masks have become sparse enough for the lists
            self._train_dense_calls += 1
            probe = self._train_dense_calls % 16 == 1
            out = _GraphCoreDense.apply(wq_rows, x_rows, b2, thr, bias, self._ws, self._ws_bwd, info, probe, self.scan == "exact")
            if probe and 0 <= info.get("max_degree", -1) <= FAST_CAP:
                self._train_dense = False
        else:
            try:
                out = _GraphCore.apply(wq_rows, x_rows, b2, thr, bias, self.select_mode, min(int(self.select_k), H * W), self.scan == "exact",
                                       self._ws, self._ws_bwd, info)
            except DaglError as e:
                if self.select_mode != "adaptive" or e.code != ERR_UNSUPPORTED:
                    raise
                self._train_dense, self._train_dense_calls = True, 1
                out = _GraphCoreDense.apply(wq_rows, x_rows, b2, thr, bias, self._ws, self._ws_bwd, info, True, self.scan == "exact")
        if info:
            self.last_info = info
        # range guard of the training path: its split-fp16 forward kernels (the two patch projections: |b1| < 4094, |w_fc| < 64; the
        # streamed dense core: |feature| < 937) NaN-fill what they hand out once an operand leaves that range -- never wrong
        # numbers -- and the dense core re-runs itself in fp32 whenever it reads statistics back.  The module finds out from
        # that flag, or by looking at its output every 64th call (one synchronisation), and moves to scan = "exact" (fp32
        # GEMM forward, no range limit); the call at hand is then repeated on that path.
        if self.scan != "exact":
            self._train_calls += 1
            left = bool(info.get("range_fallback"))
            if not left and self._train_calls % 64 == 1:
                left = bool(torch.isnan(out).any())
            if left:
                self._note_range_violation("a training call left the split-fp16 range")
                return self._forward_train(b)
        return out

    def forward(self, b: torch.Tensor) -> torch.Tensor:
        if b.dim() != 4 or b.shape[1] != self.in_channels:
            raise DaglError(f"CE.forward: expected [B,{self.in_channels},H,W], got {tuple(b.shape)}")
        if not b.is_cuda:
            raise DaglError("CE.forward: input must be on the GPU; dagl_amd has no CPU path")
        if self.select_mode not in ("adaptive", "topk", "adaptive_topk"):
            raise DaglError(f"CE.select_mode {self.select_mode!r}: expected 'adaptive', 'topk' or 'adaptive_topk'")
        k_eff = 0
        if self.select_mode != "adaptive":
            # no silent clamp: the fixed-k variant takes exactly min(num_edge, N) neighbours (GReccR2b_3mh_1-checkpoint.py:243)
            if int(self.select_k) < 1:
                raise DaglError(f"CE: select_k={self.select_k} < 1")
            k_eff = min(int(self.select_k), b.shape[2] * b.shape[3])
            # k_eff > MAX_TOPK (include/dagl_ce.h DAGL_MAX_TOPK): no per-query lists -- the inference entry points take every
            # query's score row in the dense form (csrc/topk_wide.hip), the differentiable path the dense formulation with the
            # row-wise selection as its mask (_GraphCoreWide)
        in_dtype = b.dtype
        if in_dtype in (torch.bfloat16, torch.float16):
            # reduced-precision feature maps (BASELINE config 3): the block itself computes in fp32 with the bf16
            # matrix-core screen; I/O is converted at the boundary
            b = b.float()
        elif in_dtype != torch.float32:
            raise DaglError(f"CE.forward: unsupported dtype {in_dtype}")
        # differentiable route: the module is training (its own parameters or a gradient for its input).  An eval() module
        # stays on the inference kernels even when autograd is on -- the reference's test loop relies on the long-dead
        # ``volatile`` flag (DN_Gray/trainer.py:132), i.e. evaluates the whole network with autograd enabled, and inside
        # RR / CES the block's input then "requires grad" although nobody will ask for one.  Such a call keeps its place in
        # the autograd graph (_EvalLazyGrad): should a backward arrive after all, it recomputes the block on the
        # differentiable path then -- same gradients, paid only when used.
        if self._generic:
            if torch.is_grad_enabled() and (b.requires_grad or (self.training and any(p.requires_grad for p in self.parameters()))):
                out = self._forward_train_generic(b.contiguous(), k_eff)
            else:
                out = self._forward_infer_generic(b, k_eff)
            return out if in_dtype == torch.float32 else out.to(in_dtype)
        if torch.is_grad_enabled():
            if self.training and (b.requires_grad or any(p.requires_grad for p in self.parameters())):
                out = self._forward_train(b.contiguous())
                return out if in_dtype == torch.float32 else out.to(in_dtype)
            if b.requires_grad:
                ps = [p for p in self.parameters() if p.requires_grad]
                out = _EvalLazyGrad.apply(self, k_eff, len(ps), b.contiguous(), *ps)
                return out if in_dtype == torch.float32 else out.to(in_dtype)
        out = self._forward_infer(b, k_eff)
        return out if in_dtype == torch.float32 else out.to(in_dtype)

    def _forward_infer_generic(self, b: torch.Tensor, k_eff: int) -> torch.Tensor:
        """A module built with non-default ``ksize / stride_1 / stride_2 / inter_channels`` (dagl.py:175-176): the whole method through
        ``dagl_ce_generic_forward`` (csrc/generic.hip).  ``softmax_scale`` goes to the kernel as it is (no scaled parameter copies);
        an input width that is not a multiple of 4 gets zero channels (and zero weight columns), which change nothing."""
        p = {}
        for n, t in self.named_parameters():
            if n.startswith("W."):
                continue
            t = t.detach()
            if t.dtype not in (torch.float32, torch.float16, torch.bfloat16):
                raise DaglError(f"CE: parameter {n} has dtype {t.dtype}; fp32, fp16 or bf16 expected")
            p[n] = t.float().contiguous()
        padc = (-self.in_channels) % 4
        if padc:
            b = F.pad(b, (0, 0, 0, 0, 0, padc))
            for n in ("g.weight", "theta.weight", "thr_conv.weight", "bias_conv.weight"):
                p[n] = F.pad(p[n], (0, 0, 0, 0, 0, padc)).contiguous()
        with torch.no_grad():
            out, deg = ops.ce_forward_generic(b.contiguous(), p, self.ksize, self.stride_1, self.stride_2, self.inter_channels,
                                              mode=self.select_mode, k=k_eff, softmax_scale=float(self.softmax_scale),
                                              workspace=self._ws, want_degree=True)
        self._last_call = None
        self._pack_key = None                      # the shared workspace holds another layout now
        self.last_info = dict(path=7, degree=deg)  # (device tensor [B,L]: reading it is the caller's synchronisation)
        return out

    def _forward_train_generic(self, b: torch.Tensor, k_eff: int) -> torch.Tensor:
        """Differentiable route of a module with a non-default patch geometry: every convolution / Linear-over-patches as unfold + fp32
        matrix-core product with explicit adjoints (train_ops.patch_linear: any window, stride, channel count), the graph core as
        ``_GraphCoreGeneric``.  Layout plumbing (NCHW -> zero-bordered NHWC) is torch's: autograd carries it."""
        from . import train_ops as T
        if any(p.dtype != torch.float32 for p in self.parameters()):
            raise DaglError("CE: the differentiable path needs fp32 parameters")
        from ._lib import load
        B, Cin, H, W = b.shape
        ks, s1, s2, c = int(self.ksize), int(self.stride_1), int(self.stride_2), int(self.inter_channels)
        pg = load().dagl_ce_generic_border(ks)
        t1, l1 = same_pad_amounts(H, ks, s1)[0], same_pad_amounts(W, ks, s1)[0]
        t2, l2 = same_pad_amounts(H, ks, s2)[0], same_pad_amounts(W, ks, s2)[0]
        Lh, Lw, Nh, Nw = -(-H // s1), -(-W // s1), -(-H // s2), -(-W // s2)
        padc = (-Cin) % 4                             # float4 channel groups: zero channels, zero weight columns
        def wrows(m):
            w = F.pad(m.weight, (0, 0, 0, 0, 0, padc)) if padc else m.weight
            return T.conv_weight_rows(w)
        def nhwc_bordered(t_nchw):                    # [B,C,H,W] -> zero-bordered NHWC (border pg)
            return F.pad(t_nchw.permute(0, 2, 3, 1), (0, 0, pg, pg, pg, pg)).contiguous()
        xp = nhwc_bordered(F.pad(b, (0, 0, 0, 0, 0, padc)) if padc else b)
        heads = self.select_mode != "topk"
        b1 = T.patch_linear(xp, wrows(self.g), self.g.bias, 3, 1, pg - 1, pg - 1, H, W, allow_fast=False)             # [B, H*W, c]   dagl.py:208
        b2 = T.patch_linear(xp, wrows(self.theta), self.theta.bias, 1, 1, pg, pg, H, W, allow_fast=False)               # dagl.py:209
        thr = bias = None
        if heads:                                                                                                        # dagl.py:213-215
            thr = T.patch_linear(xp, wrows(self.thr_conv), self.thr_conv.bias, ks, s1, pg - t1, pg - l1, Lh, Lw, allow_fast=False).reshape(B, -1)
            bias = T.patch_linear(xp, wrows(self.bias_conv), self.bias_conv.bias, ks, s1, pg - t1, pg - l1, Lh, Lw, allow_fast=False).reshape(B, -1)
        b1p = F.pad(b1.view(B, H, W, c), (0, 0, pg, pg, pg, pg)).contiguous()
        b2p = F.pad(b2.view(B, H, W, c), (0, 0, pg, pg, pg, pg)).contiguous()
        wq_rows = T.patch_linear(b1p, T.fc_weight_rows(self.fc1[0].weight, c, ks), self.fc1[0].bias, ks, s1, pg - t1, pg - l1, Lh, Lw,
                                 relu=True, allow_fast=False)                                                          # dagl.py:248
        x_rows = T.patch_linear(b1p, T.fc_weight_rows(self.fc2[0].weight, c, ks), self.fc2[0].bias, ks, s2, pg - t2, pg - l2, Nh, Nw,
                                relu=True, allow_fast=False)                                                           # dagl.py:249
        self._last_call = None
        self._pack_key = None
        return _GraphCoreGeneric.apply(wq_rows, x_rows, b2p, thr, bias, (H, W, ks, s1, s2), self.select_mode, k_eff,
                                       float(self.softmax_scale), self._ws, self._ws_bwd)

    def _forward_infer_any_width(self, b: torch.Tensor, k_eff: int) -> torch.Tensor:
        """``CE(in_channels = n_feats)`` for n_feats != 64 (CES builds every head that way, dagl.py:94-109; ``--n_feats``,
        DN_Gray/option.py:70): the fused prologue kernels are laid out for 64 input channels, so the four prologue convolutions
        run as unfold + fp32 matrix-core GEMM on the HIP library (train_ops.prologue_forward_any_width) and everything behind
        them -- projections, selection, edge softmax, gather, fold: the 16-channel maps do not know the input width -- through
        ``dagl_ce_forward`` (include/dagl_ce.h), the C ABI's channel-agnostic entry point."""
        from . import train_ops as T
        p = self._params_f32()
        heads = self.select_mode != "topk"
        padc = (-self.in_channels) % 4          # the library's unfold works on float4 channel groups: zero channels change nothing
        if padc:
            b = F.pad(b, (0, 0, 0, 0, 0, padc))
            p = dict(p)
            for n in ("g.weight", "theta.weight", "thr_conv.weight", "bias_conv.weight"):
                p[n] = F.pad(p[n], (0, 0, 0, 0, 0, padc)).contiguous()
        hw = (p["thr_conv.weight"], p["thr_conv.bias"], p["bias_conv.weight"], p["bias_conv.bias"]) if heads else (None,) * 4
        b1p, b2p, thr, bias = T.prologue_forward_any_width(b.contiguous(), p["g.weight"], p["g.bias"], p["theta.weight"],
                                                           p["theta.bias"], *hw)
        H, W = b.shape[-2:]
        b1 = b1p[:, T.PAD:T.PAD + H, T.PAD:T.PAD + W, :].permute(0, 3, 1, 2).contiguous()
        b2 = b2p[:, T.PAD:T.PAD + H, T.PAD:T.PAD + W, :].permute(0, 3, 1, 2).contiguous()
        if self.topk_threshold not in ("auto", "full", "sparse"):
            raise DaglError(f"CE.topk_threshold {self.topk_threshold!r}: expected 'auto', 'full' or 'sparse'")
        topk_screen = self.select_mode != "adaptive" and self.scan != "exact"
        out, info = ops.ce_forward(b1, b2, thr, bias, p["fc1.0.weight"], p["fc1.0.bias"], p["fc2.0.weight"], p["fc2.0.bias"],
                                   mode=self.select_mode, k=k_eff, workspace=self._ws, return_info=True,
                                   exact_scan=(self.scan == "exact"), profile=self.profile,
                                   tight_topk=topk_screen and self.topk_threshold == "full",
                                   sampled_topk=topk_screen and self.topk_threshold == "sparse")
        self._last_call = None                     # (this entry point reads its statistics back every call: nothing to poll)
        self.last_info = info
        if info.get("range_fallback"):
            self._note_range_violation("a call left the split-fp16 range and was re-run on the fp32 path")
        return out

    def _forward_infer(self, b: torch.Tensor, k_eff: int) -> torch.Tensor:
        """The inference kernels (``dagl_ce_forward_fused``): fp32 [B,64,H,W] in, fp32 [B,16,H,W] out, no autograd."""
        if self.in_channels != 64:
            return self._forward_infer_any_width(b, k_eff)
        params = self._params_f32()
        # the packed copies of fc1/fc2 and of the g / theta convolutions live in this module's private workspace: skip
        # repacking while neither the weights (torch bumps ._version on every in-place update) nor the call geometry changed
        wsb = self._ws.peek(b.device)
        src = dict(self.named_parameters())       # (keyed on the module's own tensors: the fp32 copies of half weights follow them)
        key = (tuple(b.shape), self.select_mode, k_eff, self.scan, self._pack_epoch, self._scale_c(),
               tuple((src[n].data_ptr(), src[n]._version) for n in ("fc1.0.weight", "fc2.0.weight", "g.weight", "theta.weight")),
               wsb.data_ptr() if wsb is not None else 0)
        # dense regime: the edge statistics (and with them a host synchronisation) are only fetched every 16th call, to
        # notice when the neighbourhoods have become sparse again
        hint = self._dense_hint and self.select_mode == "adaptive" and self.scan != "exact"
        self._dense_calls = self._dense_calls + 1 if hint else 0
        want_info = (not hint) or (self._dense_calls % 16 == 1) or self.profile is not None
        if self._served_shape != tuple(b.shape):
            if self._served_shape is not None:
                self._served_streaks[self._served_shape] = self._served_streak
            self._served_shape = tuple(b.shape)
            self._served_streak = self._served_streaks.get(self._served_shape, 0)
        no_wait = (self.select_mode == "adaptive" and self.scan != "exact" and not hint and self.adaptive_sync == "auto"
                   and self._served_streak >= 4 and key == self._pack_key and self.profile is None)
        if self.topk_threshold not in ("auto", "full", "sparse"):
            raise DaglError(f"CE.topk_threshold {self.topk_threshold!r}: expected 'auto', 'full' or 'sparse'")
        topk_screen = self.select_mode != "adaptive" and self.scan != "exact"
        if self.topk_redo not in ("auto", "always"):
            raise DaglError(f"CE.topk_redo {self.topk_redo!r}: expected 'auto' or 'always'")
        no_redo = (topk_screen and self.topk_redo == "auto" and self._redo_skip and not self._redo_banned and key == self._pack_key
                   and not torch.cuda.is_current_stream_capturing())
        if key != self._pack_key:
            self._redo_skip, self._redo_clean_polls = False, 0     # another shape / weights / workspace: what the polls saw no longer applies
            if self.scan != "exact" and key[6] != (self._pack_key[6] if self._pack_key else None) and not torch.cuda.is_current_stream_capturing():
                # new weights are about to be packed as fp16 pairs (256 w_conv, 1024 w_fc): beyond |w_conv| < 234 / |w_fc| < 58 -- never
                # seen on a trained DAGL -- the module takes the fp32 path from its first call (one synchronisation per weight set; the
                # activations have no such limit since round 6: include/dagl_ce.h)
                wmax = torch.stack([params[n].abs().max() for n in ("g.weight", "theta.weight", "fc1.0.weight", "fc2.0.weight")]).tolist()
                if not (wmax[0] < 230.0 and wmax[1] < 230.0 and wmax[2] < 57.0 and wmax[3] < 57.0):
                    self._note_range_violation("weights beyond the split-fp16 range (|w_conv| < 234, |w_fc| < 58)")
                    return self._forward_infer(b, k_eff)
        out, info = ops.ce_forward_fused(b.contiguous(), params, mode=self.select_mode, k=k_eff,
                                         workspace=self._ws, profile=self.profile,
                                         exact_scan=(self.scan == "exact"), weights_packed=(key == self._pack_key),
                                         dense_hint=hint, want_info=want_info, no_wait=no_wait,
                                         tight_topk=topk_screen and self.topk_threshold == "full",
                                         sampled_topk=topk_screen and self.topk_threshold == "sparse", no_redo=no_redo)
        self._pack_key = key[:-1] + (self._ws.peek(b.device).data_ptr(),)
        self._last_call = (tuple(b.shape), b.device)
        if no_wait:
            self._nowait_calls += 1
            if self._nowait_calls >= 16 and not torch.cuda.is_current_stream_capturing():
                self._nowait_calls = 0
                self.range_ok()                    # (one synchronisation: sticky verdict + range word since the last poll)
        elif self.select_mode == "adaptive" and info is not None:
            served = info.get("path") == 3 and not info.get("range_fallback")
            self._served_streak = self._served_streak + 1 if served else 0
        if info is not None and info.get("range_fallback"):
            self._note_range_violation("an adaptive call left the split-fp16 range and was re-run on the fp32 path")
        elif self.select_mode != "adaptive" and self.scan != "exact":
            # no host round trip in the top-k modes: look at the range word every 64th call (one synchronisation); a call
            # that left the range has returned NaN (never wrong numbers), the module moves to the fp32 path from here on
            self._calls_since_range_check += 1
            if self._calls_since_range_check >= (32 if no_redo else 64) and not torch.cuda.is_current_stream_capturing():
                self.range_ok()
        if info is not None:
            self.last_info = info
            if self.select_mode == "adaptive":
                n_pairs = b.shape[0] * (-(-b.shape[2] // 4)) * (-(-b.shape[3] // 4)) * b.shape[2] * b.shape[3]
                # stay on the dense formulation while the optimistic lists would end there again: most queries beyond the
                # lists' width (the dense call counts them), or a mask that keeps more than 1/96 of all pairs (the library's
                # own limit for redoing the overflowed queries one by one)
                n_q = b.shape[0] * (-(-b.shape[2] // 4)) * (-(-b.shape[3] // 4))
                self._dense_hint = info["path"] == 4 and (96 * info["total_edges"] > n_pairs or
                                                          2 * info.get("redone_queries", 0) > n_q)
        return out
