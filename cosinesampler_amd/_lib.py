"""ctypes binding of libcosine_sampler_hip.so (include/cosine_sampler.h).

There is no fallback: if the HIP library is missing or stale this raises, loudly.
"""
import ctypes
import os

from . import build as _build

_c_f = ctypes.c_void_p      # device pointers travel as integers (tensor.data_ptr())
_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_sz = ctypes.c_size_t

class CotangentLayout(ctypes.Structure):
    """cs_cotangent_layout (include/cosine_sampler.h): n-strides, in elements, of grad_output / grad_out_ggout."""
    _fields_ = [("grad_output_stride_n", ctypes.c_int64), ("grad_out_ggout_stride_n", ctypes.c_int64),
                ("sorted_grad_output_valid", ctypes.c_int32), ("leave_sorted_grad_output", ctypes.c_int32),
                ("grad_grad_out_stride_n", ctypes.c_int64), ("accumulate_grad_input", ctypes.c_int32),
                ("reserved_", ctypes.c_int32)]


# name -> number of leading pointer args; then (N, C, [D], H, W, P), 4 int flags, [layout*, backward stages only],
# input_cl, plan, workspace, bytes, stream
_STAGES = {
    "forward": 4,
    "backward": 6,
    "backward_backward": 9,
    "backward_backward_backward": 8,
    "bbb_fused": 9,
}
EXPORTS = (["cs_abi_version", "cs_error_string", "cs_workspace_bytes", "cs_half_streams_supported", "cs_pack_bytes", "cs_pack_input",
            "cs2d_plan_bytes", "cs2d_plan_build", "cs3d_plan_bytes", "cs3d_plan_build", "cs_debug_force_path",
            "cs2d_plan_keeps_sorted_copy", "cs_sort_points_bytes", "cs2d_sort_points", "cs3d_sort_points",
            "cs_points_tile_changes", "cs_points_tile_changes_sampled", "cs_debug_coherent_tuning", "cs2d_sum_over_n_supported",
            "cs_accumulator_kind", "cs_accumulator_bytes", "cs_accumulator_finish", "cs_carry_points", "cs_sum_over_n_supported"]
           + ["cs%dd_%s" % (d, s) for d in (2, 3) for s in _STAGES] + ["cs2d_bbb_grid", "cs3d_bbb_grid"])

ABI_VERSION = 15
ERR_UNSUPPORTED = -2       # CS_ERR_UNSUPPORTED
STAGE_NO_GRAD_INPUT = 0x10   # CS_STAGE_NO_GRAD_INPUT
STREAM_F16, STREAM_BF16 = 0x1000, 0x2000   # CS_STREAM_F16 / CS_STREAM_BF16, OR-ed into `kernel`
GRID_BROADCAST = 0x4000                   # CS_GRID_BROADCAST, OR-ed into `kernel` / the plan builders' `flags`
SUM_OVER_N = 0x10000                      # CS_SUM_OVER_N, OR-ed into `kernel` with GRID_BROADCAST: per-point results summed over the tables
POINTS_COHERENT = 0x8000                  # CS_POINTS_COHERENT, OR-ed into `kernel`: consecutive points share cells (a hint)
STAGE_POINTS_COHERENT = 0x20              # CS_STAGE_POINTS_COHERENT, OR-ed into the stage id of cs_workspace_bytes
STAGE_ACCUMULATE = 0x40                   # CS_STAGE_ACCUMULATE: the call adds into the caller's step accumulator
ACC_NONE, ACC_NCHW, ACC_CHANNELS_LAST = 0, 1, 2   # CS_ACC_*: cs_accumulator_kind
STAGE_ID = {"forward": 0, "backward": 1, "backward_backward": 2, "backward_backward_backward": 3, "bbb_fused": 3}
_lib = None


def lib_path():
    # COSINESAMPLER_LIB: development override to A/B a differently-built library
    return os.environ.get("COSINESAMPLER_LIB", _build.LIB)


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "cosinesampler_amd: %s not found. Build it with `python -m cosinesampler_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU or PyTorch fallback." % path)
    lib = ctypes.CDLL(path)
    lib.cs_abi_version.restype = _c_int
    lib.cs_error_string.restype = ctypes.c_char_p
    lib.cs_error_string.argtypes = [_c_int]
    lib.cs_workspace_bytes.restype = _c_sz
    lib.cs_workspace_bytes.argtypes = [_c_int, _c_int] + [_c_i64] * 6 + [_c_int] * 3
    lib.cs_half_streams_supported.restype = _c_int
    lib.cs_half_streams_supported.argtypes = [_c_int] + [_c_i64] * 6
    lib.cs_pack_bytes.restype = _c_sz
    lib.cs_pack_bytes.argtypes = [_c_int] + [_c_i64] * 6
    lib.cs_pack_input.restype = _c_int
    lib.cs_pack_input.argtypes = [_c_int, _c_f, _c_f] + [_c_i64] * 5 + [_c_f]
    lib.cs2d_plan_bytes.restype = _c_sz
    lib.cs2d_plan_bytes.argtypes = [_c_i64] * 5
    lib.cs2d_plan_build.restype = _c_int
    lib.cs2d_plan_build.argtypes = [_c_f, _c_f, _c_f, _c_sz] + [_c_i64] * 5 + [_c_int] * 4 + [_c_f]
    lib.cs3d_plan_bytes.restype = _c_sz
    lib.cs3d_plan_bytes.argtypes = [_c_i64] * 6
    lib.cs3d_plan_build.restype = _c_int
    lib.cs3d_plan_build.argtypes = [_c_f, _c_f, _c_f, _c_sz] + [_c_i64] * 6 + [_c_int] * 4 + [_c_f]
    lib.cs_debug_force_path.restype = _c_int
    lib.cs_debug_force_path.argtypes = [_c_int]
    lib.cs2d_plan_keeps_sorted_copy.restype = _c_int
    lib.cs2d_plan_keeps_sorted_copy.argtypes = [_c_i64] * 5
    lib.cs_sort_points_bytes.restype = _c_sz
    lib.cs_sort_points_bytes.argtypes = [_c_i64]
    lib.cs2d_sort_points.restype = _c_int
    lib.cs2d_sort_points.argtypes = [_c_f, _c_f, _c_f] + [_c_i64] * 3 + [_c_int] * 3 + [_c_f, _c_sz, _c_f]
    lib.cs3d_sort_points.restype = _c_int
    lib.cs3d_sort_points.argtypes = [_c_f, _c_f, _c_f] + [_c_i64] * 4 + [_c_int] * 3 + [_c_f, _c_sz, _c_f]
    lib.cs_points_tile_changes.restype = _c_int
    lib.cs_points_tile_changes.argtypes = [_c_int, _c_f, _c_f] + [_c_i64] * 4 + [_c_int] * 3 + [_c_f]
    lib.cs2d_sum_over_n_supported.restype = _c_int
    lib.cs2d_sum_over_n_supported.argtypes = [_c_i64] * 5 + [_c_int] * 2
    lib.cs_points_tile_changes_sampled.restype = _c_int
    lib.cs_points_tile_changes_sampled.argtypes = [_c_int, _c_f, _c_f] + [_c_i64] * 4 + [_c_int] * 4 + [_c_f]
    lib.cs_debug_coherent_tuning.restype = _c_int
    lib.cs_debug_coherent_tuning.argtypes = [_c_int, _c_int]
    lib.cs_sum_over_n_supported.restype = _c_int
    lib.cs_sum_over_n_supported.argtypes = [_c_int] + [_c_i64] * 6 + [_c_int] * 2
    lib.cs_carry_points.restype = _c_int
    lib.cs_carry_points.argtypes = [_c_f, _c_f, _c_f, _c_i64, _c_i64, _c_int, _c_f]
    lib.cs_accumulator_kind.restype = _c_int
    lib.cs_accumulator_kind.argtypes = [_c_int] + [_c_i64] * 6 + [_c_int]
    lib.cs_accumulator_bytes.restype = _c_sz
    lib.cs_accumulator_bytes.argtypes = [_c_int, _c_int] + [_c_i64] * 5
    lib.cs_accumulator_finish.restype = _c_int
    lib.cs_accumulator_finish.argtypes = [_c_int, _c_int, _c_f, _c_f] + [_c_i64] * 5 + [_c_f]
    if lib.cs_abi_version() != ABI_VERSION:
        raise RuntimeError("cosinesampler_amd: %s has ABI %d, host code wants %d -- rebuild"
                           % (path, lib.cs_abi_version(), ABI_VERSION))
    for dim in (2, 3):
        for stage, nptr in _STAGES.items():
            fn = getattr(lib, "cs%dd_%s" % (dim, stage))
            fn.restype = _c_int
            layout = [] if stage == "forward" else [ctypes.POINTER(CotangentLayout)]
            fn.argtypes = ([_c_f] * nptr + [_c_i64] * (3 + dim) + [_c_int] * 4 + layout
                           + [_c_f, _c_f, _c_f, _c_sz, _c_f])
    for dim in (2, 3):   # opt-in third-order grid gradient: 8 pointers, sizes, flags, layout*, stream
        fn = getattr(lib, "cs%dd_bbb_grid" % dim)
        fn.restype = _c_int
        fn.argtypes = [_c_f] * 8 + [_c_i64] * (3 + dim) + [_c_int] * 4 + [ctypes.POINTER(CotangentLayout), _c_f]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().cs_error_string(rc).decode()
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg, rc))
