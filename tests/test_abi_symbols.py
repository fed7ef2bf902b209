"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

from tests.helpers import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "dagl_ce.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dagl_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from dagl_amd.build import build
    return build()


def test_header_declares_the_path():
    syms = _declared_symbols()
    for must in ("dagl_ce_forward", "dagl_ce_workspace_bytes", "dagl_gather_aggregate", "dagl_last_error",
                 "dagl_project_patches", "dagl_fold_normalize"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for s in _declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/dagl_ce.h but not exported"


def test_python_binding_covers_header(lib_path):
    from dagl_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    lib = _lib.load()
    assert lib.dagl_version() >= 100


def test_host_side_argument_errors(lib_path):
    """No-GPU behaviour of the boundary: planning works, bad arguments give error codes + messages."""
    from dagl_amd import _lib
    lib = _lib.load()
    assert lib.dagl_ce_workspace_bytes(1, 64, 64, 0, 0) > 0
    assert lib.dagl_ce_workspace_bytes(1, 256, 256, 1, 8) > lib.dagl_ce_workspace_bytes(1, 64, 64, 1, 8)
    assert lib.dagl_ce_workspace_bytes(1, 64, 64, 1, 0) == 0          # k missing in top-k mode
    assert b"k=0" in lib.dagl_last_error()
    assert lib.dagl_ce_workspace_bytes(1, 64, 64, 7, 0) == 0          # unknown mode
    assert lib.dagl_ce_workspace_bytes(0, 64, 64, 0, 0) == 0
    assert lib.dagl_gather_aggregate(None, 4, 8, 783, None, None, None, None) == -1
    assert lib.dagl_feat_rows(100) == 160


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dagl_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DaglError, match="no fallback"):
        _lib.load()


def test_info_struct_layout_matches_header():
    """ctypes mirror of dagl_ce_info: same field order and size as the C struct (3 x int64 + 2 x int32)."""
    import ctypes as C
    from dagl_amd import _lib
    text = open(os.path.join(REPO, "include", "dagl_ce.h")).read()
    body = text[text.index("typedef struct dagl_ce_info {"):text.index("} dagl_ce_info;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int64_t|int32_t)\s+(\w+)\s*;", body)
    assert [n for _, n in fields] == [n for n, _ in _lib.CeInfo._fields_]
    assert [t for t, _ in fields] == ["int64_t" if f is C.c_int64 else "int32_t" for _, f in _lib.CeInfo._fields_]
    assert C.sizeof(_lib.CeInfo) == 40
