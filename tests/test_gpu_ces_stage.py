"""SURVEY 8f-1 pinned on the reference: one CES stage -- conv1x1(cat(c1_1(x)..c1_4(x))) + x, DN_Gray/model/dagl.py:114 --
and the whole CES.forward (:112-119), from tests/golden/ces_stage_48x48.npz (reference modules, sparse thr/bias heads):
the fused launch set ``dagl_ces_stage_forward`` is compared with the reference's output directly."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN_DIR, normwise

pytestmark = pytest.mark.gpu


def _ces(meta, dev):
    from dagl_amd.net import CES, seeded_state_dict
    from dagl_amd.synth import make_ce_params
    ces = CES(64).eval()
    sd = seeded_state_dict(ces.state_dict(), meta["seed"])
    idx = 0
    for s in (1, 2, 3):
        for h in (1, 2, 3, 4):
            for n, a in make_ce_params(meta["seed"] + idx, variant="sparse", sparse_gain=meta["gain"]).items():
                sd[f"c{s}_{h}.{n}"] = torch.from_numpy(a)
            idx += 1
    ces.load_state_dict(sd, strict=True)
    return ces.to(dev)


def test_fused_stage_and_whole_ces_match_the_reference_modules():
    from dagl_amd import ops
    from dagl_amd.synth import make_features
    z = np.load(os.path.join(GOLDEN_DIR, "ces_stage_48x48.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    dev = torch.device("cuda:0")
    ces = _ces(meta, dev)
    x = (torch.from_numpy(make_features(meta["seed"], 1, 64, meta["H"], meta["W"])) * meta["scale"]).to(dev)
    heads = [getattr(ces, f"c1_{h}") for h in (1, 2, 3, 4)]
    prm = [{n: p.detach().contiguous() for n, p in hd.named_parameters() if not n.startswith("W.")} for hd in heads]
    with torch.no_grad():
        got, info = ops.ces_stage_forward(x, prm, ces.c1_c.weight.detach().contiguous(), ces.c1_c.bias.detach().contiguous(),
                                          mode="adaptive")
        assert got is not None and info["path"] == 3, info          # the fused launch set ran (no hand-back)
        assert info["max_degree"] == int(z["deg"].max()) and info["total_edges"] == int(z["deg"].sum())
        assert normwise(got[0, :, ::2, ::2].cpu().numpy(), z["stage1"]) <= 1e-4
        # and through the module (stage 1 fused, stage 3 has degrees > 64 and goes head by head)
        stage1 = ces._stage(1, x)
        assert normwise(stage1[0, :, ::2, ::2].cpu().numpy(), z["stage1"]) <= 1e-4
        whole = ces(x)
        assert normwise(whole[0, :, 1::2, ::2].cpu().numpy(), z["whole"]) <= 2e-4
