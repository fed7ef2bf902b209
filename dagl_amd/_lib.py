"""ctypes binding of the C-ABI device library (include/dagl_ce.h).

The library is built in-tree by ``dagl_amd.build`` (hipcc, gfx950).  There is no
CPU or pure-torch fallback: if the shared object is missing or a call fails the
error is raised to the caller.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdagl_ce.so")

MODE_ADAPTIVE, MODE_TOPK, MODE_ADAPTIVE_TOPK = 0, 1, 2
MODES = {"adaptive": MODE_ADAPTIVE, "topk": MODE_TOPK, "adaptive_topk": MODE_ADAPTIVE_TOPK}
MAX_TOPK = 64
ABI_VERSION = 406          # include/dagl_ce.h DAGL_ABI_VERSION this binding was written against
FAST_CAP = 64
P = 784
D = 196
DS = 204
ERR_WORKSPACE = -2
ERR_UNSUPPORTED = -5
N_STAGES = 8
STAGE_NAMES = ("layout", "project", "thresholds", "screen_sample", "select", "edge_softmax", "gather", "fold")
FLAG_EXACT_SCAN = 0x100
FLAG_WEIGHTS_PACKED = 0x200
FLAG_DENSE_HINT = 0x400
FLAG_NO_WAIT = 0x800
FLAG_TIGHT_TOPK = 0x1000
FLAG_SAMPLED_TOPK = 0x2000
FLAG_NO_REDO = 0x4000


class DaglError(RuntimeError):
    code = 0          # the C ABI's status code when the error came from the library (include/dagl_ce.h DAGL_ERR_*)


class CeInfo(C.Structure):
    _fields_ = [("required_bytes", C.c_int64), ("total_edges", C.c_int64), ("redone_queries", C.c_int64),
                ("max_degree", C.c_int32), ("path", C.c_int32), ("range_fallback", C.c_int32), ("dense_rerun_blocks", C.c_int32)]


class CeWeights(C.Structure):
    """dagl_ce_weights: the 12 parameter tensors of one head (device pointers)."""
    NAMES = ("g.weight", "g.bias", "theta.weight", "theta.bias", "thr_conv.weight", "thr_conv.bias",
             "bias_conv.weight", "bias_conv.bias", "fc1.0.weight", "fc1.0.bias", "fc2.0.weight", "fc2.0.bias")
    _fields_ = [(n, C.c_void_p) for n in ("g_w", "g_b", "theta_w", "theta_b", "thr_w", "thr_b", "bias_w", "bias_b",
                                          "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t

# name -> (restype, argtypes); the one table every declared symbol of include/dagl_ce.h appears in
SIGNATURES = {
    "dagl_version": (_i, []),
    "dagl_last_error": (C.c_char_p, []),
    "dagl_device_check": (_i, []),
    "dagl_ce_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "dagl_ce_forward": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz,
                             C.POINTER(CeInfo)]),
    "dagl_ce_forward_debug": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz,
                                   C.POINTER(CeInfo), _vp, _vp, _vp]),
    "dagl_ce_list_width": (_i, [_i, _i]),
    "dagl_ce_range_check": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _sz, C.POINTER(_i)]),
    "dagl_ce_core_forward": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                  C.POINTER(CeInfo)]),
    "dagl_ce_core_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "dagl_ce_core_backward": (_i, [_vp, _i, _i, _i, _i, _i] + [_vp] * 17 + [_sz]),
    "dagl_ce_core_dense_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dagl_ce_core_dense_forward": (_i, [_vp, _i, _i, _i, _i] + [_vp] * 9 + [_sz, C.POINTER(CeInfo)]),
    "dagl_ce_core_dense_backward": (_i, [_vp, _i, _i, _i, _i] + [_vp] * 14 + [_sz]),
    "dagl_ce_core_wide_forward": (_i, [_vp, _i, _i, _i, _i, _i] + [_vp] * 7 + [_sz, C.POINTER(CeInfo)]),
    "dagl_ce_core_wide_backward": (_i, [_vp, _i, _i, _i, _i, _i] + [_vp] * 12 + [_sz]),
    "dagl_gemm_f32_scratch_floats": (_sz, [_i, _i, _i, _i]),
    "dagl_gemm_f32": (_i, [_vp, _i, _i, _i, _i, _vp, C.c_longlong, C.c_longlong, _i, _vp, C.c_longlong, C.c_longlong, _i,
                           _vp, C.c_longlong, C.c_longlong, C.c_float, C.c_float, _vp, _i, _i, _vp]),
    "dagl_probe_mfma_bf16": (_i, [_vp, _i, _i, _vp, _vp]),
    "dagl_selftest_wave_ops": (_i, [_vp, _vp]),
    "dagl_fc_grad16_scratch_bytes": (_sz, [_i, _i, _i]),
    "dagl_fc_grad16": (_i, [_vp] + [_i] * 8 + [_vp] * 8 + [_sz]),
    "dagl_fc_grad16_dmap_ok": (_i, [_i, _i]),
    "dagl_fc_grad16_dmap_scratch_bytes": (_sz, [_i, _i, _i]),
    "dagl_fc_grad16_dmap": (_i, [_vp] + [_i] * 8 + [_vp] * 8 + [_sz]),
    "dagl_unfold_patches": (_i, [_vp] + [_i] * 10 + [_vp, _vp]),
    "dagl_fold_patches": (_i, [_vp] + [_i] * 10 + [_vp, _vp]),
    "dagl_copy4": (_i, [_vp, _i, _i, _i, _i, _vp] + [C.c_longlong] * 4 + [_vp] + [C.c_longlong] * 4),
    "dagl_relu_backward": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "dagl_prelu_forward": (_i, [_vp, _sz, _vp, _vp, _vp]),
    "dagl_prelu_scratch_bytes": (_sz, [_sz]),
    "dagl_prelu_backward": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dagl_col_sum_scratch_bytes": (_sz, [_sz, _i]),
    "dagl_conv_pair_backward_supported": (_i, [_i, _i, _i]),
    "dagl_conv_pair_backward_scratch_bytes": (_sz, [_i, _i, _i]),
    "dagl_conv_pair_backward": (_i, [_vp, _i, _i, _i] + [_vp] * 11),
    "dagl_col_sum": (_i, [_vp, _sz, _i, _vp, _vp, _vp]),
    "dagl_profile_create": (_i, [_i, C.POINTER(_vp)]),
    "dagl_profile_destroy": (_i, [_vp]),
    "dagl_profile_reset": (_i, [_vp]),
    "dagl_profile_select_stage": (_i, [_vp, _i]),
    "dagl_profile_read": (_i, [_vp, C.POINTER(_i), C.POINTER(C.c_float), _i]),
    "dagl_ce_forward_profiled": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz,
                                      C.POINTER(CeInfo), _vp]),
    "dagl_ce_forward_fused": (_i, [_vp, _i, _i, _i] + [_vp] * 13 + [_i, _i, _vp, _vp, _sz, C.POINTER(CeInfo), _vp]),
    "dagl_ces_stage_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "dagl_ces_stage_forward": (_i, [_vp, _i, _i, _i, _vp, C.POINTER(CeWeights), _vp, _vp, _i, _i, _vp, _vp, _sz,
                                    C.POINTER(CeInfo), _vp]),
    "dagl_ce_prologue": (_i, [_vp, _i, _i, _i] + [_vp] * 14),
    "dagl_ce_prologue16_scratch_bytes": (_sz, [_i, _i, _i]),
    "dagl_ce_prologue16": (_i, [_vp, _i, _i, _i] + [_vp] * 14 + [_sz]),
    "dagl_pad_nhwc": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "dagl_pack_fc_weight": (_i, [_vp, _vp, _vp]),
    "dagl_project_patches": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "dagl_project_patches16_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "dagl_project_patches16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz]),
    "dagl_feat_rows": (_i, [_i]),
    "dagl_query_thresholds": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dagl_gather_aggregate": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "dagl_unfold_values": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "dagl_fold_normalize": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "dagl_scores_dense": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "dagl_ce_generic_workspace_bytes": (_sz, [_i] * 8),
    "dagl_ce_generic_forward": (_i, [_vp] + [_i] * 8 + [C.c_float, _i, _i] + [_vp] * 16 + [_sz]),
    "dagl_ce_generic_border": (_i, [_i]),
    "dagl_ce_generic_core_workspace_bytes": (_sz, [_i] * 8),
    "dagl_ce_generic_core_forward": (_i, [_vp] + [_i] * 7 + [C.c_float, _i, _i] + [_vp] * 8 + [_sz]),
    "dagl_ce_generic_core_backward": (_i, [_vp] + [_i] * 7 + [C.c_float, _i, _i] + [_vp] * 12 + [_sz]),
}

_lib = None


def load():
    """dlopen the device library (once) and type its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one this process
    # initialises, so it is loaded before the device library's DT_NEEDED libamdhip64.so.7 is resolved
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DaglError(
            f"{LIB_PATH} is missing: build it with `python -m dagl_amd.build` "
            "(hipcc --offload-arch=gfx950); dagl_amd has no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError => the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.dagl_version() != ABI_VERSION:
        raise DaglError(f"{LIB_PATH} reports ABI version {lib.dagl_version()}, this binding expects {ABI_VERSION}: "
                        "rebuild it (`python -m dagl_amd.build --force`)")
    if torch.cuda.is_available():
        # once per process: the kernels are gfx950 code objects (MFMA shapes, LDS-DMA, 160 KiB LDS)
        rc = lib.dagl_device_check()
        if rc != 0:
            raise DaglError("dagl_amd needs an MI355X (gfx950): " + lib.dagl_last_error().decode("utf-8", "replace"))
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().dagl_last_error().decode("utf-8", "replace")
        err = DaglError(f"{what} failed (code {rc}): {msg}")
        err.code = rc
        raise err
