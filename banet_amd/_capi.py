"""ctypes binding of libbanet_hip.so (the C ABI declared in include/banet_hip.h).

PyTorch is used only as plumbing here: device memory (`tensor.data_ptr()`), the current
HIP stream and the caching allocator for workspaces.  There is NO fallback: if the shared
library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BANET_HIP_LIB: development override (same-box A/B of two builds of the library, tools/ab_builds.sh); the in-tree build otherwise
LIB_PATH = os.environ.get("BANET_HIP_LIB") or os.path.join(_HERE, "lib", "libbanet_hip.so")

LEGACY_LM, LEGACY_FIXED, BUNDLE_CAMERA, BUNDLE = 0, 1, 2, 3

_FP = ctypes.c_void_p


class BanetError(RuntimeError):
    pass


class Level(ctypes.Structure):
    """mirror of banet_level_t"""
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("C", ctypes.c_int32), ("K", ctypes.c_int32),
                ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("variant", ctypes.c_int32),
                ("dense", ctypes.c_int32), ("tgt_has_grad", ctypes.c_int32), ("normalize_rays", ctypes.c_int32),
                ("scale", ctypes.c_float), ("pairs", ctypes.c_int32), ("flags", ctypes.c_int32), ("policy", ctypes.c_int32),
                ("src", _FP), ("tgt", _FP), ("depth", _FP), ("basis", _FP), ("rays", _FP),
                ("fx", _FP), ("fy", _FP), ("ox", _FP), ("oy", _FP), ("intr", _FP)]

    # the field's name until round 4 (tools/ and older tests still say `reserved_`): same storage
    reserved_ = property(lambda self: self.flags, lambda self, v: setattr(self, "flags", v))


POLICY_THROUGHPUT, POLICY_BATCH_INVARIANT = 0, 1   # banet_hip.h: BANET_POLICY_*
CANONICAL_BATCH = 32


class Mlp(ctypes.Structure):
    """mirror of banet_mlp_t"""
    _fields_ = [("w", _FP * 5), ("b", _FP * 5)]


class State(ctypes.Structure):
    """mirror of banet_state_t"""
    _fields_ = [("R", _FP), ("T", _FP), ("Wc", _FP), ("iters", _FP), ("ratio", _FP), ("lambda_out", _FP),
                ("delta", _FP)]


class LmParams(ctypes.Structure):
    """mirror of banet_lm_params_t (the run-time LM configuration of legacy/ba.py:5-9)"""
    _fields_ = [("angle_change", ctypes.c_float), ("translation_change", ctypes.c_float),
                ("residual_ratio", ctypes.c_float), ("solver", ctypes.c_int32)]


SOLVER_QR, SOLVER_INVERSE = 0, 1

EXPORTS = {
    "banet_version": (ctypes.c_int, []),
    "banet_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "banet_equation_construction_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "banet_equation_construction_f32": (ctypes.c_int, [_FP] * 5 + [ctypes.c_int] * 4 + [_FP, ctypes.c_size_t, _FP]),
    "banet_equation_construction_grad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "banet_equation_construction_grad_f32": (ctypes.c_int, [_FP] * 8 + [ctypes.c_int] * 4 + [_FP, ctypes.c_size_t, _FP]),
    "banet_resample_f32": (ctypes.c_int, [_FP] * 3 + [ctypes.c_int] * 6 + [_FP]),
    "banet_target_map_f32": (ctypes.c_int, [_FP] * 2 + [ctypes.c_int] * 4 + [_FP]),
    "banet_depth_output_f32": (ctypes.c_int, [_FP] * 4 + [ctypes.c_int] * 3 + [_FP]),
    "banet_sample_stats_blocks": (ctypes.c_int, [ctypes.c_int]),
    "banet_sample_stats_f32": (ctypes.c_int, [_FP] * 4 + [ctypes.c_int] * 5 + [_FP] * 3),
    "banet_sample_stats_grad_f32": (ctypes.c_int, [_FP] * 4 + [ctypes.c_int] * 5 + [_FP] * 6),
    "banet_ba_assemble_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Level)]),
    "banet_ba_assemble_f32": (ctypes.c_int, [ctypes.POINTER(Level)] + [_FP] * 7 + [_FP, ctypes.c_size_t, _FP]),
    "banet_ba_assemble_mask_f32": (ctypes.c_int, [ctypes.POINTER(Level)] + [_FP] * 8 + [_FP, ctypes.c_size_t, _FP]),
    "banet_ba_solve_update_f32": (ctypes.c_int, [ctypes.POINTER(Level), ctypes.POINTER(Mlp), ctypes.c_float] + [_FP] * 4 +
                                  [ctypes.POINTER(State), _FP]),
    "banet_ba_solve_update_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Level)]),
    "banet_ba_solve_update_ws_f32": (ctypes.c_int, [ctypes.POINTER(Level), ctypes.POINTER(Mlp), ctypes.c_float] + [_FP] * 4 +
                                     [ctypes.POINTER(State), _FP, ctypes.c_size_t, _FP]),
    "banet_lm_level_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Level)]),
    "banet_lm_level_f32": (ctypes.c_int, [ctypes.POINTER(Level), ctypes.POINTER(Mlp), ctypes.c_float, ctypes.c_int,
                                          ctypes.c_int, ctypes.POINTER(State), _FP, ctypes.c_size_t, _FP]),
    "banet_lm_params_default": (None, [ctypes.POINTER(LmParams)]),
    "banet_lm_level_ex_f32": (ctypes.c_int, [ctypes.POINTER(Level), ctypes.POINTER(Mlp), ctypes.c_float, ctypes.c_int,
                                             ctypes.c_int, ctypes.POINTER(LmParams), ctypes.POINTER(State), _FP,
                                             ctypes.c_size_t, _FP]),
    "banet_sample_stats_grad_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "banet_sample_stats_grad_det_f32": (ctypes.c_int, [_FP] * 4 + [ctypes.c_int] * 5 + [_FP] * 5 + [_FP, ctypes.c_size_t, _FP]),
    "banet_spd_solve_f32": (ctypes.c_int, [_FP] * 3 + [ctypes.c_int] * 2 + [_FP]),
    "banet_dense_adjoint_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(Level)]),
    "banet_dense_adjoint_workspace_bytes_ex": (ctypes.c_size_t, [ctypes.POINTER(Level), ctypes.c_int]),
    "banet_dense_adjoint_f32": (ctypes.c_int, [ctypes.POINTER(Level)] + [_FP] * 11 + [_FP, ctypes.c_size_t, _FP]),
    "banet_dense_adjoint_ex_f32": (ctypes.c_int, [ctypes.POINTER(Level)] + [_FP] * 11 + [ctypes.c_int, _FP, ctypes.c_size_t, _FP]),
    "banet_small_step_adjoint_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "banet_small_step_adjoint_f32": (ctypes.c_int, [ctypes.c_int] * 6 + [ctypes.c_float, ctypes.POINTER(Mlp)] + [_FP] * 14 +
                                     [ctypes.POINTER(Mlp), _FP, ctypes.c_size_t, _FP]),
    "banet_target_map_adjoint_f32": (ctypes.c_int, [_FP] * 2 + [ctypes.c_int] * 4 + [_FP]),
    "banet_target_map_adjoint_ex_f32": (ctypes.c_int, [_FP] * 2 + [ctypes.c_int] * 5 + [_FP]),
    "banet_build_id": (ctypes.c_char_p, []),
    "banet_gather_selection": (ctypes.c_int, [_FP]),
    "banet_syrk_selection": (ctypes.c_int, [_FP]),
    "banet_profile_ranges": (ctypes.c_int, [ctypes.c_int]),
    "banet_profile_begin": (ctypes.c_int, [ctypes.c_int]),
    "banet_profile_end": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                         ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]),
}

_lib = None


def lib():
    """Load libbanet_hip.so (once).  Raises BanetError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BanetError("%s not found: build it with banet_amd/csrc/build.sh (or __graft_entry__.build())"
                             % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise BanetError("libbanet_hip: %s (%d)" % (lib().banet_error_string(rc).decode(), rc))


def ptr(t):
    """device pointer of a contiguous float32/int32 CUDA(HIP) tensor (None -> NULL)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise BanetError("banet_amd runs on the GPU only: got a %s tensor" % t.device)
    if not t.is_contiguous():
        raise BanetError("tensor must be contiguous")
    if t.dtype not in (torch.float32, torch.int32):
        raise BanetError("tensor must be float32/int32, got %s" % t.dtype)
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes, device):
    """256-byte aligned scratch from the caching allocator"""
    n = max(int(nbytes), 256)
    buf = torch.empty(n + 256, dtype=torch.uint8, device=device)
    off = (-buf.data_ptr()) % 256
    return buf[off:off + n]


def f32c(t):
    return t.contiguous().to(torch.float32)
