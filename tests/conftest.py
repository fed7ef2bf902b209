import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "tests", "golden", "*.npz")))
    return [f for f in files if not os.path.basename(f).startswith(("set12", "grad_", "ces_stage", "x8_protocol", "quality_", "geom_", "geomgrad_"))      # CE block cases only
            and "_scale" not in os.path.basename(f)]                        # (softmax_scale != 10: scale_cases())


def scale_cases():
    """Goldens minted with a softmax_scale other than the default 10 (meta["softmax_scale"])."""
    import glob
    return [f for f in sorted(glob.glob(os.path.join(REPO, "tests", "golden", "*_scale*.npz"))) if not os.path.basename(f).startswith(("geom_", "geomgrad_"))]


def geometry_cases():
    """Goldens minted from the reference built with non-default ksize / stride_1 / stride_2 / inter_channels
    (tests/golden/make_golden_geometry.py; meta carries the geometry)."""
    import glob
    return sorted(glob.glob(os.path.join(REPO, "tests", "golden", "geom_*.npz")))


@pytest.fixture(scope="session")
def repo_root():
    return REPO
