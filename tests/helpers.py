"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def load_golden(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in ("out", "deg", "rowsum", "agg_sub")}


def load_geometry_golden(path):
    z = np.load(path, allow_pickle=False)
    return json.loads(str(z["meta"])), {k: z[k] for k in ("out", "deg")}


def case_inputs(meta):
    """Regenerate (x, params) of a golden case from its seed (numpy PCG64)."""
    from dagl_amd.synth import make_ce_params, make_features
    p = make_ce_params(meta["seed"], in_channels=meta["C"], inter_channels=meta.get("inter_channels", 16), ksize=meta.get("ksize", 7),
                       variant=meta["variant"], sparse_gain=meta["sparse_gain"])
    x = make_features(meta["seed"], meta["B"], meta["C"], meta["H"], meta["W"])
    return torch.from_numpy(x), {n: torch.from_numpy(a) for n, a in p.items()}


def normwise(a, b):
    """max|a-b| / max|b| -- the normwise relative error SURVEY.md section 7 prescribes."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / den) if den > 0 else float(np.abs(a - b).max())


def query_block(Lw: int, i0: int, j0: int, n: int):
    """Row-major indices of the n x n block of queries whose top-left query is (i0, j0) on a grid Lw queries wide."""
    import torch as _t
    ii, jj = _t.meshgrid(_t.arange(i0, i0 + n), _t.arange(j0, j0 + n), indexing="ij")
    return (ii * Lw + jj).reshape(-1)


def fold_block(agg_rows, H: int, W: int, i0: int, j0: int, n: int):
    """Fold the aggregated patches ``agg_rows [n*n, 784]`` (oracle order (c, kh, kw)) of the query block ``query_block(Lw, i0, j0, n)``
    the way dagl.py:265-272 folds all of them (stride 4, 7x7 windows, padding 3 -- the stride-1 SAME padding of dagl.py:243, not the
    padding the query patches were cut with --, overlap count) and return ``(mask [H, W],
    expected [16, H, W])``: ``mask`` marks the pixels ALL of whose covering queries lie inside the block -- there the block's fold IS
    the module's output, so a sample of queries checks the output of a call itself (gather + weighted sum + fold + normalise) and
    not only a debug read-out of another call."""
    import torch as _t
    from dagl_amd.synth import query_grid
    Lh, Lw, _, _ = query_grid(H, W)
    pt = pl = 3
    canvas = _t.zeros(16, H + 12, W + 12, dtype=agg_rows.dtype)
    cnt_blk = _t.zeros(H + 12, W + 12)
    cnt_all = _t.zeros(H + 12, W + 12)
    for i in range(Lh):
        for j in range(Lw):
            cnt_all[4 * i - pt + 6:4 * i - pt + 13, 4 * j - pl + 6:4 * j - pl + 13] += 1
    rows = agg_rows.reshape(n, n, 16, 7, 7)
    for a in range(n):
        for b in range(n):
            y, x = 4 * (i0 + a) - pt + 6, 4 * (j0 + b) - pl + 6
            canvas[:, y:y + 7, x:x + 7] += rows[a, b]
            cnt_blk[y:y + 7, x:x + 7] += 1
    canvas, cnt_blk, cnt_all = canvas[:, 6:6 + H, 6:6 + W], cnt_blk[6:6 + H, 6:6 + W], cnt_all[6:6 + H, 6:6 + W]
    mask = (cnt_blk == cnt_all) & (cnt_all > 0)
    return mask, canvas / cnt_all.clamp(min=1)
