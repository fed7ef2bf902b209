import sys, os, torch, numpy as np
sys.path.insert(0, "/root/repo")
from tests.conftest import golden_cases
from tests.helpers import case_inputs, load_golden, normwise
from dagl_amd.ce import CE
for path in golden_cases():
    meta, g = load_golden(path)
    x, params = case_inputs(meta)
    ce = CE(in_channels=64); ce.load_state_dict(params, strict=True)
    ce.select_mode = meta["mode"]
    if meta["k"]: ce.select_k = meta["k"]
    ce = ce.cuda().eval()
    with torch.no_grad(): out = ce(x.cuda()).cpu().numpy()
    print(f"{meta['name']:26s} {normwise(out, g['out']):.3e}")
