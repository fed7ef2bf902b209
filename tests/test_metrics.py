"""dagl_amd.metrics against values produced by the reference's own metric code (tests/golden/make_metrics_golden.py)."""
import json
import os

import numpy as np
import torch

from tests.helpers import GOLDEN_DIR


def _inputs():
    rng = np.random.default_rng(123)
    a = torch.from_numpy(rng.random((2, 3, 40, 36)).astype(np.float32))
    b = (a + torch.from_numpy(rng.normal(0, 0.05, (2, 3, 40, 36)).astype(np.float32))).clamp(0, 1)
    return a, b


def test_metrics_match_reference_values():
    from dagl_amd import metrics
    g = json.load(open(os.path.join(GOLDEN_DIR, "metrics.json")))
    a, b = _inputs()
    assert abs(float(metrics.ssim(a, b)) - g["ssim_mean"]) < 1e-6
    assert np.allclose(metrics.ssim(a, b, size_average=False).numpy(), g["ssim_per_image"], atol=1e-6)
    assert abs(metrics.calc_psnr(b, a, 1, 1.0) - g["calc_psnr_train"]) < 1e-5
    assert abs(metrics.calc_psnr(b, a, 2, 1.0, benchmark=True) - g["calc_psnr_bench_rgb"]) < 1e-5
    assert abs(float(metrics.quantize(b * 0.7, 1.0).double().sum()) - g["quantize_sum"]) < 1e-6


def test_batch_psnr_is_the_per_image_mean():
    from dagl_amd import metrics
    clean = torch.zeros(2, 1, 4, 4)
    img = torch.stack([torch.full((1, 4, 4), 0.1), torch.full((1, 4, 4), 0.01)])
    assert abs(metrics.batch_psnr(img, clean, 1.0) - (20.0 + 40.0) / 2) < 1e-4
