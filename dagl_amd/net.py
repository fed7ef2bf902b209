"""The caller of the block and the inference driver around it, restated so that the
HIP block can be exercised the way the reference exercises ``CE``: inside the 3-stage x 4-head ``CES`` module
of the EDSR-style trunk ``RR`` and under the recursive 4-way tiling of ``forward_chop``.

Everything here except ``CE`` and the ResBlocks' PReLU (train_ops.PReLU: torch's backward for it was a sixth of the training
step) is stock PyTorch-ROCm convolutions.  Module / parameter names and registration order
follow the reference (``RR`` DN_Gray/model/dagl.py:11-54, ``CES`` :74-119, ``ResBlock`` DN_Gray/model/common.py:59-79),
so a reference ``state_dict`` loads strictly.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .ce import CE


def _conv(cin, cout, k):
    return nn.Conv2d(cin, cout, k, padding=k // 2)       # common.default_conv


class ResBlock(nn.Module):
    """conv3x3 - PReLU - conv3x3 + skip (common.py:59-79, bn=False, res_scale=1)."""

    def __init__(self, n_feats, res_scale=1.0):
        super().__init__()
        from .train_ops import PReLU                      # nn.PReLU() whose fp32 GPU calls run on the HIP library (same state_dict)
        self.body = nn.Sequential(_conv(n_feats, n_feats, 3), PReLU(), _conv(n_feats, n_feats, 3))
        self.res_scale = res_scale

    def forward(self, x):
        return self.body(x).mul(self.res_scale) + x


class CES(nn.Module):
    """Three stages of four patch-graph heads, each stage mixed by a 1x1 conv + residual (dagl.py:74-119)."""

    def __init__(self, in_channels, num=4, ce_cls=CE):
        super().__init__()
        self.RBS1 = nn.Sequential(*[ResBlock(in_channels) for _ in range(num)])
        self.RBS2 = nn.Sequential(*[ResBlock(in_channels) for _ in range(num)])
        for stage in (1, 2, 3):
            for head in (1, 2, 3, 4):
                setattr(self, f"c{stage}_{head}", ce_cls(in_channels=in_channels))
            setattr(self, f"c{stage}_c", nn.Conv2d(in_channels, in_channels, 1, 1, 0))
        self.fuse_stage = in_channels == 64       # stage-level launch set (include/dagl_ce.h: dagl_ces_stage_forward)
        self.last_info = None
        from . import ops
        self._ws = {1: ops.Workspace(), 2: ops.Workspace(), 3: ops.Workspace()}     # one per stage: packed weights persist
        self._pack_key = {1: None, 2: None, 3: None}
        self._skip_fused = {1: 0, 2: 0, 3: 0}     # calls for which a stage stays on the per-head path after it met dense masks
        self._fused_calls = {1: 0, 2: 0, 3: 0}    # fused calls since the stage's range word was last looked at

    def _stage(self, s, x):
        heads = [getattr(self, f"c{s}_{h}") for h in (1, 2, 3, 4)]
        mix = getattr(self, f"c{s}_c")
        from ._lib import MAX_TOPK
        mode = heads[0].select_mode if isinstance(heads[0], CE) else None
        k_eff = 0 if mode in (None, "adaptive") else min(int(heads[0].select_k), x.shape[-2] * x.shape[-1])
        # (an eval() stage fed a constant input under an autograd-enabled test loop counts as "no gradient wanted", like CE)
        no_grad = not torch.is_grad_enabled() or (not self.training and not x.requires_grad)
        if self.fuse_stage and all(isinstance(hd, CE) for hd in heads) and x.is_cuda and x.dtype == torch.float32 \
                and no_grad and len({(hd.select_mode, hd.select_k, hd.scan) for hd in heads}) == 1 \
                and heads[0].scan == "screened" and self._skip_fused[s] == 0 and 0 <= k_eff <= MAX_TOPK \
                and all(float(hd.softmax_scale) == 10.0 for hd in heads) \
                and all(p.dtype == torch.float32 for hd in heads for p in hd.parameters()):
            # the four heads share x: one launch set with the heads as a batch dimension + the 1x1 mix + residual
            from . import ops
            prm = [{n: p.detach().contiguous() for n, p in hd.named_parameters() if not n.startswith("W.")} for hd in heads]
            ws = self._ws[s]
            fcs = [prm[h][n] for h in range(4) for n in ("fc1.0.weight", "fc2.0.weight", "g.weight", "theta.weight")]
            wsb = ws.peek(x.device)
            key = (tuple(x.shape), heads[0].select_mode, k_eff, tuple(hd._pack_epoch for hd in heads),
                   tuple((t.data_ptr(), t._version) for t in fcs), wsb.data_ptr() if wsb is not None else 0)
            if key != self._pack_key[s] and (self._pack_key[s] is None or key[4] != self._pack_key[s][4]) \
                    and not torch.cuda.is_current_stream_capturing():
                # new weights are about to be packed as fp16 pairs: beyond |w_conv| < 234 / |w_fc| < 58 (never seen on a trained DAGL) the
                # heads take the fp32 path -- which takes them off the fused launch set -- before their first call (CE._forward_infer)
                wmax = torch.stack([t.abs().max() for t in fcs]).view(4, 4).max(dim=0).values.tolist()
                if not (wmax[0] < 57.0 and wmax[1] < 57.0 and wmax[2] < 230.0 and wmax[3] < 230.0):
                    for hd in heads:
                        hd._note_range_violation("weights beyond the split-fp16 range (|w_conv| < 234, |w_fc| < 58)")
                    return self._stage(s, x)
            # the top-k threshold policy ("auto") lives in the stage workspace and is read by the kernels themselves (CE.topk_threshold)
            thr = heads[0].topk_threshold
            out, info = ops.ces_stage_forward(x.contiguous(), prm, mix.weight.detach().contiguous(),
                                              mix.bias.detach().contiguous(), mode=heads[0].select_mode,
                                              k=k_eff, workspace=ws,
                                              weights_packed=(key == self._pack_key[s]),
                                              tight_topk=(mode != "adaptive" and thr == "full"),
                                              sampled_topk=(mode != "adaptive" and thr == "sparse"))
            self._pack_key[s] = key[:-1] + (ws.peek(x.device).data_ptr(),) if out is not None else None
            self.last_info = info
            violated = False
            if out is not None and mode != "adaptive":
                # the top-k modes never read anything back: look at the stage workspace's (sticky) range word every 64th
                # fused call -- one synchronisation.  A call that left the split-fp16 range has NaN-filled outputs (never
                # wrong numbers); the heads move to scan = "exact" (which takes them off the fused path) and this call is redone
                self._fused_calls[s] += 1
                if self._fused_calls[s] >= 64 and not torch.cuda.is_current_stream_capturing():
                    self._fused_calls[s] = 0
                    verdict = ops.ce_range_check((4 * x.shape[0],) + tuple(x.shape[1:]), mode, k_eff, ws, x.device)
                    if verdict & 3:
                        for hd in heads:
                            hd._note_range_violation("a fused CES stage call left the split-fp16 range (outputs NaN-filled)")
                        self._pack_key[s] = None
                        out, violated = None, True
            if out is not None:
                return out
            if violated:
                return self._stage(s, x)                    # (heads now on scan = "exact": per-head path)
            self._skip_fused[s] = 16                        # dense masks: the heads' own dense path serves the next calls
        elif self._skip_fused[s] > 0:
            self._skip_fused[s] -= 1
        outs = [hd(x) for hd in heads]                      # dense adaptive neighbourhoods / foreign heads
        return mix(torch.cat(outs, dim=1)) + x

    def forward(self, x):
        out = self._stage(1, x)
        out = self.RBS1(out)
        out = self._stage(2, out)
        out = self.RBS2(out)
        return self._stage(3, out)


class _MeanShift(nn.Conv2d):
    """Registered by the reference (``add_mean``, dagl.py:41) but never applied; kept for state_dict compatibility."""

    def __init__(self, rgb_range, rgb_mean=(0.4488, 0.4371, 0.4040), rgb_std=(1.0, 1.0, 1.0), sign=1):
        super().__init__(3, 3, kernel_size=1)
        std = torch.tensor(rgb_std)
        self.weight.data = torch.eye(3).view(3, 3, 1, 1) / std.view(3, 1, 1, 1)
        self.bias.data = sign * rgb_range * torch.tensor(rgb_mean) / std
        for p in self.parameters():
            p.requires_grad = False


class RR(nn.Module):
    """head conv - 8 ResBlocks - CES - 8 ResBlocks - conv - tail conv, global residual (dagl.py:11-54)."""

    def __init__(self, n_resblocks=16, n_feats=64, n_colors=1, res_scale=1.0, rgb_range=1.0, ce_cls=CE):
        super().__init__()
        body = [ResBlock(n_feats, res_scale) for _ in range(n_resblocks // 2)]
        body.append(CES(n_feats, ce_cls=ce_cls))
        body += [ResBlock(n_feats, res_scale) for _ in range(n_resblocks // 2)]
        body.append(_conv(n_feats, n_feats, 3))
        self.add_mean = _MeanShift(rgb_range)
        self.head = nn.Sequential(_conv(n_colors, n_feats, 3))
        self.body = nn.Sequential(*body)
        self.tail = nn.Sequential(_conv(n_feats, n_colors, 3))

    def forward(self, x):
        return x + self.tail(self.body(self.head(x)))

    def load_state_dict(self, state_dict, strict=True):
        """The reference network's own loader rule (``dagl.py:56-73``), which is NOT torch's: checkpoint entries are
        copied in place into the tensors this network owns; an entry whose shape does not fit raises ``RuntimeError``
        and an entry this network does not know raises ``KeyError`` (``strict`` only) -- unless its name contains
        ``tail``: the reference fine-tunes a gray checkpoint into an ``n_colors = 3`` network (and back) that way, so
        anything about the last convolution is tolerated.  Keys the checkpoint lacks are never an error.  Returns
        ``None`` like the reference."""
        own = self.state_dict()
        for name, value in state_dict.items():
            tolerated = "tail" in name
            if name not in own:
                if strict and not tolerated:
                    raise KeyError(f'unexpected key "{name}" in state_dict')
                continue
            src = value.data if isinstance(value, nn.Parameter) else value
            try:
                with torch.no_grad():
                    own[name].copy_(src)          # shares storage (and version counter) with the parameter / buffer
            except Exception:
                if not tolerated:
                    raise RuntimeError(
                        f"While copying the parameter named {name}, whose dimensions in the model are "
                        f"{tuple(own[name].shape)} and whose dimensions in the checkpoint are {tuple(src.shape)}.")
        for m in self.modules():                  # packed / converted weight copies of the heads follow the new values
            if hasattr(m, "invalidate_packed"):
                m.invalidate_packed()


def seeded_state_dict(template: "OrderedDict[str, torch.Tensor]", seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Regenerable stand-in for a trained checkpoint (none ships with the reference): every tensor of ``template`` in
    state_dict order from ONE numpy PCG64 stream -- weights uniform(+-1/sqrt(fan_in)), biases uniform(+-1/sqrt(fan_in)
    of their layer) approximated by +-0.05, PReLU slopes 0.25, ``add_mean`` left as built."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, t in template.items():
        if name.startswith("add_mean"):
            out[name] = t.clone()
        elif t.dim() >= 2:
            fan_in = int(np.prod(t.shape[1:]))
            b = 1.0 / math.sqrt(fan_in)
            out[name] = torch.from_numpy(rng.uniform(-b, b, size=tuple(t.shape)).astype(np.float32))
        elif name.endswith("bias"):
            out[name] = torch.from_numpy(rng.uniform(-0.05, 0.05, size=tuple(t.shape)).astype(np.float32))
        else:                                             # PReLU slope
            out[name] = torch.full(tuple(t.shape), 0.25)
    return out


CHOP_PRESETS = {       # (min_size, shave_size_max) of the four forks' forward_chop
    "dn_gray": (10000, 24),      # DN_Gray/model/__init__.py:179,187
    "car": (10000, 24),          # CAR/model/__init__.py:190,199
    "demosaic": (10000, 12),     # Demosaic/model/__init__.py:179,188
    "dn_real": (70000, 12),      # DN_Real/model/__init__.py:135,144  (no self-ensemble in that fork)
}


def sparse_heads_state_dict(sd: "OrderedDict[str, torch.Tensor]", seed: int, gain: float = 1.65,
                            prefix: str = "body.8.") -> "OrderedDict[str, torch.Tensor]":
    """Replace the parameters of the 12 ``CE`` heads in ``sd`` (keys ``<prefix>c<stage>_<head>.*``) by
    ``synth.make_ce_params(seed + index, "sparse", gain)``: thr / bias heads that keep a handful of neighbours per query
    (mean degrees of ~5-10, long-tailed) -- the regime a trained DAGL works in, as opposed to default-initialised heads
    that keep ~95 % of the keys.  Regenerable anywhere (numpy PCG64)."""
    from .synth import make_ce_params
    out = OrderedDict(sd)
    idx = 0
    for s in (1, 2, 3):
        for h in (1, 2, 3, 4):
            for n, a in make_ce_params(seed + idx, variant="sparse", sparse_gain=gain).items():
                out[f"{prefix}c{s}_{h}.{n}"] = torch.from_numpy(a)
            idx += 1
    return out


def tile_boxes(h: int, w: int, shave_scale: int = 4, shave_size_max: int = 24):
    """Geometry of ONE 4-way split of the reference's ``forward_chop`` (DN_Gray/model/__init__.py:194-198, :222-229) for an
    h x w tile: the four overlapping corner boxes ``(y0, y1, x0, x1)`` of size (h//2//4*4 + 24) x (w//2//4*4 + 24), and per corner
    the pair (destination box in the h x w output, source box inside the corner tile's output) that stitches the inner
    quadrants back.  Every user of the tiling (sequential driver, batched driver, leaf enumeration) takes it from here."""
    hh, wh = h // 2, w // 2
    hs = (hh // shave_scale) * shave_scale + shave_size_max
    ws = (wh // shave_scale) * shave_scale + shave_size_max
    corners = [(0, hs, 0, ws), (0, hs, w - ws, w), (h - hs, h, 0, ws), (h - hs, h, w - ws, w)]
    stitch = [((0, hh, 0, wh), (0, hh, 0, wh)),
              ((0, hh, wh, w), (0, hh, ws - w + wh, ws)),
              ((hh, h, 0, wh), (hs - h + hh, hs, 0, wh)),
              ((hh, h, wh, w), (hs - h + hh, hs, ws - w + wh, ws))]
    return corners, stitch, hs * ws


def _stitch(x: torch.Tensor, outs, stitch) -> torch.Tensor:
    out = x.new_empty(x.shape)
    for o, ((dy0, dy1, dx0, dx1), (sy0, sy1, sx0, sx1)) in zip(outs, stitch):
        out[:, :, dy0:dy1, dx0:dx1] = o[:, :, sy0:sy1, sx0:sx1]
    return out


def chop_leaf_boxes(h: int, w: int, min_size: int = 10000, shave_size_max: int = 24, shave_scale: int = 4, y: int = 0, x: int = 0):
    """The leaf tiles ``(y0, y1, x0, x1)`` (image coordinates, depth-first order) the reference's tiling cuts an h x w image into:
    256 x 256 -> 64 tiles of 72 x 72."""
    corners, _, area = tile_boxes(h, w, shave_scale, shave_size_max)
    boxes = [(y + y0, y + y1, x + x0, x + x1) for (y0, y1, x0, x1) in corners]
    if area < min_size:
        return boxes
    out = []
    for (y0, y1, x0, x1) in boxes:
        out.extend(chop_leaf_boxes(y1 - y0, x1 - x0, min_size, shave_size_max, shave_scale, y0, x0))
    return out


def chop_forward(model, x: torch.Tensor, min_size: int = 10000, shave_size_max: int = 24, shave_scale: int = 4,
                 ensemble: bool = False):
    """Recursive 4-way tiled inference: the reference's ``Model.forward_chop`` for scale 1
    (DN_Gray/model/__init__.py:179-231); ``ensemble`` = its ``--ensemble`` switch: ``test_x8`` on every leaf (:205-208).  A tile of
    h x w is split into four overlapping corner tiles (``tile_boxes``) until the corner area drops below ``min_size``; the outputs'
    inner quadrants are stitched back."""
    corners, stitch, area = tile_boxes(x.shape[2], x.shape[3], shave_scale, shave_size_max)
    tiles = [x[:, :, y0:y1, x0:x1] for (y0, y1, x0, x1) in corners]
    if area < min_size:
        # the reference runs the four leaves one by one with n_GPUs == 1 (__init__.py:203-209)
        outs = [forward_x8(model, t) if ensemble else model(t.contiguous()) for t in tiles]
    else:
        outs = [chop_forward(model, t, min_size, shave_size_max, shave_scale, ensemble) for t in tiles]
    return _stitch(x, outs, stitch)


def chop_forward_batched(model, x: torch.Tensor, min_size: int = 10000, shave_size_max: int = 24, shave_scale: int = 4,
                         max_batch: int = 64, ensemble: bool = False):
    """Same tiling and stitching as ``chop_forward``, but all leaf tiles (they share one shape) go through the network in
    batches of up to ``max_batch`` instead of one by one: tiles are independent, and a batch is just
    another grid dimension of the HIP block.  Device-side equivalent of the reference's per-leaf loop."""
    boxes = chop_leaf_boxes(x.shape[2], x.shape[3], min_size, shave_size_max, shave_scale)   # the order chop_forward visits them
    if len({(y1 - y0, x1 - x0) for (y0, y1, x0, x1) in boxes}) != 1:   # ragged leaves (odd sizes): the sequential driver
        return chop_forward(model, x, min_size, shave_size_max, shave_scale, ensemble)
    plan = [x[:, :, y0:y1, x0:x1] for (y0, y1, x0, x1) in boxes]
    outs = []
    for i in range(0, len(plan), max_batch):
        batch = torch.cat([t for t in plan[i:i + max_batch]], dim=0).contiguous()
        # self-ensemble: each of the 8 variants of ALL leaves of the batch is one network call (the leaves of a variant
        # share a shape); per leaf this is exactly test_x8's stack and mean
        res = forward_x8(model, batch) if ensemble else model(batch)
        outs.extend(res.split(x.shape[0], dim=0))
    return stitch_leaves(x, outs, min_size, shave_size_max, shave_scale)


def stitch_leaves(x: torch.Tensor, outs, min_size: int = 10000, shave_size_max: int = 24, shave_scale: int = 4) -> torch.Tensor:
    """Put the leaf outputs ``outs`` (the order of ``chop_leaf_boxes`` = the order ``chop_forward`` visits them) back together:
    the reference's recursion (DN_Gray/model/__init__.py:216-231) bottom-up, inner quadrants of every 4-way split."""
    it = iter(outs)

    def stitch_tree(h, w):
        corners, stitch, area = tile_boxes(h, w, shave_scale, shave_size_max)
        if area < min_size:
            o = [next(it) for _ in range(4)]
        else:
            o = [stitch_tree(y1 - y0, x1 - x0) for (y0, y1, x0, x1) in corners]
        return _stitch(o[0].new_empty(o[0].shape[0], o[0].shape[1], h, w), o, stitch)

    return stitch_tree(x.shape[2], x.shape[3])


def chop_forward_sharded(model, x: torch.Tensor, dist=None, min_size: int = 10000, shave_size_max: int = 24, shave_scale: int = 4,
                         max_batch: int = 64, ensemble: bool = False):
    """Tile-level multi-GPU inference: the reference hands ``min(n_GPUs, 4)`` leaf tiles per call to ``nn.DataParallel``
    (DN_Gray/model/__init__.py:181, 203-209); here -- one process per GPU, SURVEY.md section 8(e) "tiles/8 leaf tiles at inference"
    -- the leaf tiles of ``chop_leaf_boxes`` are dealt to the ranks as contiguous slices (``shard.shard_range``), every rank runs
    its slice in batches of up to ``max_batch``, ONE ``all_gather`` (RCCL on GPUs, any backend of the process group) brings the
    leaf outputs to every rank, and every rank stitches the full image (``stitch_leaves``).  ``x`` is the same full image on
    every rank.  Without an initialised process group (or at world size 1) this is ``chop_forward_batched``.  Tiles are independent:
    no halo, no exchange besides the gather."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return chop_forward_batched(model, x, min_size, shave_size_max, shave_scale, max_batch, ensemble)
    from .shard import shard_range
    rank, world = dist.get_rank(), dist.get_world_size()
    boxes = chop_leaf_boxes(x.shape[2], x.shape[3], min_size, shave_size_max, shave_scale)
    shapes = {(y1 - y0, x1 - x0) for (y0, y1, x0, x1) in boxes}
    if len(shapes) != 1:
        # ragged leaves (odd sizes) cannot share one gather buffer: every rank runs the sequential driver (same result everywhere)
        return chop_forward(model, x, min_size, shave_size_max, shave_scale, ensemble)
    lo, hi = shard_range(len(boxes), rank, world)
    mine = []
    for i in range(lo, hi, max_batch):
        batch = torch.cat([x[:, :, y0:y1, x0:x1] for (y0, y1, x0, x1) in boxes[i:min(i + max_batch, hi)]], dim=0).contiguous()
        res = forward_x8(model, batch) if ensemble else model(batch)
        mine.extend(res.split(x.shape[0], dim=0))
    # fixed-size slices (they differ by at most one leaf): pad, gather once, trim
    per = -(-len(boxes) // world)
    if mine:
        like = mine[0]
    else:                                                  # more ranks than leaves: the output shape of a leaf from one probe call
        (y0, y1, x0, x1) = boxes[0]
        like = model(x[:, :, y0:y1, x0:x1].contiguous())
    buf = like.new_zeros((per,) + tuple(like.shape))
    for j, o in enumerate(mine):
        buf[j] = o
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    outs = []
    for r in range(world):
        l, h = shard_range(len(boxes), r, world)
        outs.extend(gathered[r][j] for j in range(h - l))
    return stitch_leaves(x, outs, min_size, shave_size_max, shave_scale)


# The 8 symmetries of the square in the order the reference enumerates them (``augment_img`` modes 0..7,
# DN_Gray/model/__init__.py:35-51): (quarter turns counter-clockwise in the (H, W) plane, then flip the rows?)
_X8_MODES = ((0, False), (1, True), (0, True), (3, False), (2, True), (1, False), (2, False), (3, True))
_X8_INVERSE = (0, 1, 2, 5, 4, 3, 6, 7)      # modes 3 and 5 undo each other, the rest are involutions (test_x8, :56-59)


def _x8_apply(t: torch.Tensor, mode: int) -> torch.Tensor:
    k, flip = _X8_MODES[mode]
    if k:
        t = t.rot90(k, (-2, -1))
    if flip:
        t = t.flip(-2)
    return t


def forward_x8(forward_fn, x: torch.Tensor) -> torch.Tensor:
    """Geometric self-ensemble of the reference (``test_x8``, DN_Gray/model/__init__.py:53-62): the 8 flip / rotate
    variants go through ``forward_fn``, the outputs are transformed back and averaged -- same variants, same order of the
    stack that is averaged.  The reference round-trips every variant through numpy on the host (``augment_img_tensor``,
    :18-32); here the transforms stay on the device."""
    outs = [_x8_apply(forward_fn(_x8_apply(x, m).contiguous()), _X8_INVERSE[m]) for m in range(8)]
    return torch.stack(outs, dim=0).mean(dim=0, keepdim=False)


def psnr(img: torch.Tensor, ref: torch.Tensor, data_range: float = 1.0) -> float:
    """Per-image PSNR as ``batch_PSNR`` computes it (DN_Gray/utils.py:18-24: skimage compare_psnr on the float images)."""
    mse = torch.mean((img.double() - ref.double()) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * math.log10(data_range ** 2 / mse)


def set12_protocol_noise(clean: torch.Tensor, sigma: float = 50.0, rgb_range: float = 1.0) -> torch.Tensor:
    """Noise of the reference test script (DN_Gray/test.py:55-58): torch.manual_seed(1), CPU normal_()."""
    torch.manual_seed(1)
    noise = torch.FloatTensor(clean.size()).normal_(mean=0, std=sigma / (255.0 / rgb_range))
    return clean + noise
