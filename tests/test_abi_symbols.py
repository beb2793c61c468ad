"""CPU: the C-ABI library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
every symbol include/cosine_sampler.h declares.  No compute call is made here."""
import ctypes
import os
import re

from cosinesampler_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "cosine_sampler.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cs[0-9a-z_]*)\s*\(", text)))


def test_header_declares_the_reference_entry_points():
    names = declared_functions()
    for dim in (2, 3):
        for stage in ("forward", "backward", "backward_backward", "backward_backward_backward", "bbb_fused"):
            assert "cs%dd_%s" % (dim, stage) in names
    assert sorted(_lib.EXPORTS) == names


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.cs_abi_version.restype = ctypes.c_int
    assert lib.cs_abi_version() == _lib.ABI_VERSION


def test_loader_binds_and_reports_errors_without_gpu():
    lib = _lib.load()
    assert lib.cs_error_string(0) == b"ok"
    assert b"invalid" in lib.cs_error_string(-1)
    # headline config: the tiled path wants scratch for the channels-last copy (64 MiB) in forward ...
    assert lib.cs_workspace_bytes(2, 0, 16, 16, 1, 256, 256, 1 << 20, 0, 0, 0) == 16 * 16 * 256 * 256 * 4
    assert lib.cs_workspace_bytes(2, 0, 16, 16, 1, 256, 256, 1 << 20, 1, 0, 0) == 0
    # ... one fat row (C payload floats + 4 coefficients) per sample in backward (plus the plan unless one is passed in);
    # the plan has room for the sorted copy of one grad_output (64 B per sample) next to the ids
    S = 16 << 20
    plan = lib.cs2d_plan_bytes(16, 16, 256, 256, 1 << 20)
    assert plan > S * (4 + 4 + 64)
    assert lib.cs_workspace_bytes(2, 1, 16, 16, 1, 256, 256, 1 << 20, 1, 1, 0) == S * 80
    assert lib.cs_workspace_bytes(2, 1, 16, 16, 1, 256, 256, 1 << 20, 1, 0, 0) == S * 80 + plan
    # a backward call that does not want grad_input (CS_STAGE_NO_GRAD_INPUT) scatters nothing: at most channels-last tables
    T2 = 16 * 16 * 256 * 256 * 4
    assert lib.cs_workspace_bytes(2, 1 | 0x10, 16, 16, 1, 256, 256, 1 << 20, 1, 0, 0) == 0
    assert lib.cs_workspace_bytes(2, 2 | 0x10, 16, 16, 1, 256, 256, 1 << 20, 0, 0, 1) == 2 * T2
    assert lib.cs_workspace_bytes(3, 1 | 0x10, 8, 8, 128, 128, 128, 1 << 19, 1, 0, 0) == 0
    # a crowded table (96 tables of 16x16 cells, 10^5 points) keeps round 1's path: one fat row per sample
    assert lib.cs_workspace_bytes(2, 1, 96, 4, 1, 16, 16, 100000, 1, 1, 0) == 96 * 100000 * 32
    # 3D, BASELINE configs[3] (not crowded): the tile path -- a plan (the per-tile sample lists) and p-ordered rows of
    # 8 + 8 floats, no accumulator
    T3 = 8 * 8 * 128 ** 3 * 4
    S3 = 8 << 19
    plan3 = lib.cs3d_plan_bytes(8, 8, 128, 128, 128, 1 << 19)
    assert plan3 > S3 * 8 * 4                                                               # 8 list slots per sample
    assert lib.cs_workspace_bytes(3, 1, 8, 8, 128, 128, 128, 1 << 19, 1, 1, 0) == S3 * 64
    assert lib.cs_workspace_bytes(3, 3, 8, 8, 128, 128, 128, 1 << 19, 1, 1, 0) == S3 * 128  # fused third backward: two payloads
    assert lib.cs_workspace_bytes(3, 1, 8, 8, 128, 128, 128, 1 << 19, 0, 0, 0) == 2 * T3 + plan3 + S3 * 64   # the 3D table copy is z-paired
    assert lib.cs_workspace_bytes(3, 0, 8, 8, 128, 128, 128, 1 << 19, 1, 0, 0) == 0
    # a table too large for the tile histogram keeps the row atomics: an accumulator of input's size
    T4 = 2 * 4 * 512 * 512 * 256 * 4
    assert lib.cs3d_plan_bytes(2, 4, 256, 512, 512, 1 << 20) == 0
    assert lib.cs_workspace_bytes(3, 1, 2, 4, 256, 512, 512, 1 << 20, 1, 0, 0) == T4
    assert lib.cs_pack_bytes(3, 8, 8, 128, 128, 128, 1 << 19) == 2 * T3
    assert lib.cs_workspace_bytes(2, 3, 16, 64, 1, 256, 256, 1 << 20, 0, 0, 0) == 16 * 64 * 256 * 256 * 4
    # 5 channels run zero-padded as 8 on the tiled path: table copy (8 ch), plan, rows of 8 + 4 floats
    assert lib.cs_workspace_bytes(2, 1, 16, 5, 1, 256, 256, 1 << 20, 0, 0, 0) == (
        16 * 8 * 256 * 256 * 4 + lib.cs2d_plan_bytes(16, 5, 256, 256, 1 << 20) + (16 << 20) * 48)
    assert lib.cs_workspace_bytes(2, 1, 16, 33, 1, 256, 256, 1 << 20, 0, 0, 0) == 0     # beyond 32 channels: direct kernels
    assert lib.cs_workspace_bytes(2, 1, 1, 16, 1, 32, 32, 1024, 0, 0, 0) == 0
    # the step accumulator (cs_cotangent_layout.accumulate_grad_input): which layout the stages of a problem add in, its
    # size, and the scratch a coherent stage no longer needs when the accumulator is the caller's
    assert lib.cs_accumulator_kind(2, 16, 16, 1, 256, 256, 1 << 20, 0) == _lib.ACC_NCHW
    assert lib.cs_accumulator_kind(2, 16, 16, 1, 256, 256, 1 << 20, _lib.POINTS_COHERENT) == _lib.ACC_CHANNELS_LAST
    assert lib.cs_accumulator_kind(2, 16, 16, 1, 256, 256, 1 << 20, 0x100) == _lib.ACC_NONE       # '+mixed' can leave the fast path
    assert lib.cs_accumulator_kind(2, 16, 64, 1, 256, 256, 1 << 20, 0) == _lib.ACC_NONE          # beyond 32 channels
    assert lib.cs_accumulator_kind(3, 8, 8, 128, 128, 128, 1 << 19, 0) == _lib.ACC_NONE
    assert lib.cs_accumulator_bytes(2, _lib.ACC_NCHW, 16, 5, 1, 256, 256) == 16 * 5 * 256 * 256 * 4
    assert lib.cs_accumulator_bytes(2, _lib.ACC_CHANNELS_LAST, 16, 5, 1, 256, 256) == 16 * 8 * 256 * 256 * 4
    assert lib.cs_workspace_bytes(2, 1 | 0x20, 16, 16, 1, 256, 256, 1 << 20, 1, 0, 0) == T2
    assert lib.cs_workspace_bytes(2, 1 | 0x20 | 0x40, 16, 16, 1, 256, 256, 1 << 20, 1, 0, 0) == 0
    bad = _lib.CotangentLayout(16, 16, 0, 0, 0, 9, 0)
    rc = lib.cs2d_backward(None, None, None, None, None, None, 1, 1, 4, 4, 8, 0, 1, 0, 1, bad, None, None, None, 0, None)
    assert rc == -1
    # argument validation happens before any device work: callable without a GPU
    rc = lib.cs2d_forward(None, None, None, None, 1, 1, 4, 4, 8, 7, 1, 0, 1, None, None, None, 0, None)
    assert rc == -1  # padding_mode 7 is not a mode
    # a cotangent layout with a negative n-stride is refused; NULL pointers for live sizes are, too
    bad = _lib.CotangentLayout(-1, 16)
    rc = lib.cs2d_backward(None, None, None, None, None, None, 1, 1, 4, 4, 8, 0, 1, 0, 1, bad, None, None, None, 0, None)
    assert rc == -1
    ok = _lib.CotangentLayout(0, 0)          # n-expanded cotangents: fine as a layout, but the pointers are missing
    rc = lib.cs2d_backward(None, None, None, None, None, None, 1, 1, 4, 4, 8, 0, 1, 0, 1, ok, None, None, None, 0, None)
    assert rc == -1
    # zero-sized problems are a no-op, null pointers and all (empty tensors come with null data pointers)
    rc = lib.cs2d_backward(None, None, None, None, None, None, 0, 1, 4, 4, 8, 0, 1, 0, 1, None, None, None, None, 0, None)
    assert rc == 0


def test_debug_knobs_are_inert_without_the_environment_variable():
    """include/cosine_sampler.h: 'no mutable global state' -- cs_debug_* act only in a process started with
    COSINESAMPLER_DEBUG=1 (the test suite is one: conftest.py); a child process without it gets 0 back from both"""
    import subprocess
    import sys
    code = ("import ctypes,sys; lib=ctypes.CDLL(sys.argv[1]); lib.cs_debug_force_path.restype=ctypes.c_int; "
            "lib.cs_debug_coherent_tuning.restype=ctypes.c_int; "
            "print(lib.cs_debug_force_path(2), lib.cs_debug_coherent_tuning(128, 7))")
    env = dict(os.environ)
    env.pop("COSINESAMPLER_DEBUG", None)
    out = subprocess.run([sys.executable, "-c", code, build.LIB], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["0", "0"]
    env["COSINESAMPLER_DEBUG"] = "1"
    out = subprocess.run([sys.executable, "-c", code, build.LIB], env=env, capture_output=True, text=True, timeout=120)
    assert out.stdout.split() == ["1", "1"]


def test_code_object_is_gfx950():
    data = open(build.LIB, "rb").read()
    assert b"gfx950" in data
