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


def case_inputs(meta):
    """Regenerate (x, params) of a golden case from its seed (numpy PCG64)."""
    from dagl_amd.synth import make_ce_params, make_features
    p = make_ce_params(meta["seed"], in_channels=meta["C"], variant=meta["variant"], sparse_gain=meta["sparse_gain"])
    x = make_features(meta["seed"], meta["B"], meta["C"], meta["H"], meta["W"])
    return torch.from_numpy(x), {n: torch.from_numpy(a) for n, a in p.items()}


def normwise(a, b):
    """max|a-b| / max|b| -- the normwise relative error SURVEY.md section 7 prescribes."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / den) if den > 0 else float(np.abs(a - b).max())
