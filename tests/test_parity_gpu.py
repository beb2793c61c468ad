"""GPU (MI355X): the HIP kernels, called through the C ABI, against
  * the CPU oracle (oracle/cs_oracle.c) on the same seeded inputs,
  * the golden vectors from the reference's ground truth,
  * torch.nn.functional.grid_sample for the linear kernel (bit-exact forward),
  * size-independent properties at BASELINE.json's full sizes.
Tolerance: max|a-b|/max|b| <= 1e-5 per tensor (helpers.REL_TOL; north_star's "within 1e-5 fp32")."""
import os

import pytest
import torch
import torch.nn.functional as F

from cosinesampler_amd import CosineSampler2d, CosineSampler3d, _lib, multicell_offset, ops
from helpers import (KERNEL_ENUM, PAD, PIXEL_KEYS_2D, PIXEL_KEYS_3D, assert_close, axis_only, load, offsets,
                     parse_stage_name, pixel_pipeline, rel_err, stage_fixtures)
from oracle import cs_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_native_library_is_the_loaded_one():
    lib = _lib.load()
    assert lib.cs_abi_version() == _lib.ABI_VERSION
    assert torch.cuda.is_available()
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def _g(t):
    return None if t is None else t.to(DEV)


def _case(d, N, C, sp, P, seed, spread=1.3):
    g = torch.Generator().manual_seed(seed)
    inp = torch.rand((N, C) + tuple(sp), generator=g)
    grid = torch.rand((N,) + (1,) * (d - 1) + (P, d), generator=g) * (2 * spread) - spread
    if P >= 4:  # exact corners and centre
        grid.view(N, P, d)[:, 0] = -1.0
        grid.view(N, P, d)[:, 1] = 1.0
        grid.view(N, P, d)[:, 2] = 0.0
    oshape = (N, C) + (1,) * (d - 1) + (P,)
    t = dict(inp=inp, grid=grid, gOut=torch.randn(oshape, generator=g), cI=torch.randn(inp.shape, generator=g),
             cG=torch.randn(grid.shape, generator=g), hG=torch.randn(grid.shape, generator=g),
             hO=torch.randn(oshape, generator=g))
    return t


def _run_all_stages(mod, t, off, pad, align, ke, mc, dev):
    """mod = ops (GPU, through the C ABI) or cs_oracle (CPU).  Same call sequence for both."""
    x = {k: v.to(dev) for k, v in t.items()}
    off = off.to(dev)
    r = {}
    r["out"] = mod.forward(x["inp"], x["grid"], off, pad, align, ke, mc)
    r["gI"], r["gG"] = mod.backward(x["gOut"], x["inp"], x["grid"], off, pad, align, True, ke, mc)
    none_gi, gG2 = mod.backward(x["gOut"], x["inp"], x["grid"], off, pad, align, False, ke, mc)
    assert none_gi is None
    r["gG_noinput"] = gG2
    r["bbI"], r["bbG"], r["bbO"] = mod.backward_backward(x["cI"], x["cG"], x["inp"], x["grid"], x["gOut"], off, pad,
                                                         align, True, ke, mc)
    r["bbI0"], r["bbG0"], r["bbO0"] = mod.backward_backward(None, x["cG"], x["inp"], x["grid"], x["gOut"], off,
                                                            pad, align, False, ke, mc)
    r["k4I"], r["k4O"] = mod.backward_backward_backward(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], off, pad,
                                                        align, True, ke, mc)
    r["fI"], r["fO"] = mod.bbb_fused(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], off, pad, align, ke,
                                     mc)
    return r


CASES = []
for _d in (2, 3):
    for _kernel in (0, 1, 2):
        for _mc in (True, False):
            for _pad in (0, 1, 2):
                for _align in (True, False):
                    CASES.append((_d, _kernel, _mc, _pad, _align))


@pytest.mark.parametrize("d,ke,mc,pad,align", CASES)
def test_every_stage_matches_cpu_oracle(d, ke, mc, pad, align):
    N, C, P = 3, 5, 777                                # ragged: not a multiple of the block size
    sp = (9, 14) if d == 2 else (5, 7, 6)
    t = _case(d, N, C, sp, P, seed=100 * d + 10 * ke + pad)
    off = offsets(N, mc)
    want = _run_all_stages(cs_oracle, t, off, pad, align, ke, mc, "cpu")
    got = _run_all_stages(ops, t, off, pad, align, ke, mc, DEV)
    torch.cuda.synchronize()
    for k in want:
        assert_close(got[k], want[k], "d=%d kernel=%d mc=%s pad=%d align=%s: %s" % (d, ke, mc, pad, align, k))


TILED_CASES = []
for _C in (4, 8, 16):
    for _ke in (0, 1, 2):
        for _pad, _align, _mc in ((0, True, True), (0, False, False), (1, True, False), (2, True, True),
                                  (2, False, False)):
            TILED_CASES.append((_C, _ke, _pad, _align, _mc))
# 32 channels = 8 quads per sample (LDS of the walkers and of the fused third backward beyond 64 KiB)
TILED_CASES += [(32, 0, 0, True, True), (32, 2, 1, False, False), (32, 1, 2, True, True)]
# channel counts below 4 run zero-padded to one quad on the same path
TILED_CASES += [(1, 0, 0, True, True), (2, 0, 0, True, True), (2, 2, 1, False, False), (3, 1, 2, True, True),
                (3, 0, 0, False, True)]


# every other channel count runs zero-padded up to the next supported one (5..7 as 8, 9..15 as 16, 17..31 as 32): the
# reference loops over any C (2d.cu:340-354)
TILED_CASES += [(5, 0, 0, True, True), (6, 2, 1, False, False), (7, 1, 2, True, True), (12, 0, 0, True, True),
                (12, 2, 2, False, True), (24, 0, 0, True, True), (31, 0, 1, True, False)]


# more than 32 channels: channel ranges of at most 32 through the same path (ops: channel groups), results concatenated /
# grad_grid summed -- no cliff to the direct kernels (reference: any C in one loop, 2d.cu:340-354)
TILED_CASES += [(33, 0, 0, True, True), (48, 2, 1, False, False), (64, 0, 0, True, True), (64, 1, 2, True, False)]


@pytest.mark.parametrize("C,ke,pad,align,mc", TILED_CASES)
@pytest.mark.parametrize("shared", [False, True])
def test_tiled_path_matches_cpu_oracle(C, ke, pad, align, mc, shared):
    """The fast 2D path (channels-last gathers + plan + payload rows + tile walkers), forced on
    for a small problem that still spans several tiles, ragged edges and out-of-range points."""
    N, P, sp = 3, 3001, (37, 50)
    t = _case(2, N, C, sp, P, seed=7000 + C + 10 * ke + pad, spread=1.15)
    off = offsets(N, mc)
    want = _run_all_stages(cs_oracle, t, off, pad, align, ke, mc, "cpu")
    ops.force_path(2)
    try:
        if shared:      # one StepContext across the stages, as the autograd chain uses it
            step = ops.StepContext()

            class Shared(object):
                def __getattr__(self, name):
                    fn = getattr(ops, name)
                    return lambda *a: fn(*a, ctx=step)
            got = _run_all_stages(Shared(), t, off, pad, align, ke, mc, DEV)
        else:
            got = _run_all_stages(ops, t, off, pad, align, ke, mc, DEV)
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for k in want:
        assert_close(got[k], want[k], "tiled C=%d kernel=%d pad=%d align=%s mc=%s shared=%s: %s"
                     % (C, ke, pad, align, mc, shared, k))


@pytest.mark.parametrize("N,C,P,sp,crowded", [(8, 16, 3001, (37, 50), False), (16, 4, 1029, (20, 33), False), (24, 8, 700, (18, 18), False),
                                               (8, 4, 20000, (9, 12), True), (16, 16, 255, (37, 50), False), (2, 8, 257, (37, 50), False),
                                               (6, 16, 2303, (20, 33), False)])
def test_xcd_aware_workgroup_order_of_the_point_kernels(N, C, P, sp, crowded):
    """With fp32 streams and an even N the tiled backward point kernels take their workgroups in another order -- the XCDs
    form two groups that own whole tables (n = g, g + 2, ..), the second group starting half the points further on
    (cs_tiled.cuh pblk, Dims::xcd): a speed choice that must not change a value.  Ragged P (a last workgroup with few points,
    fewer workgroups than XCDs), the tile walkers and the crowded-table path (which keeps the launch order), every stage
    against the oracle.  (Most 2D cases of this file with an even N run in that order too.)"""
    ke, pad, align, mc = 0, 0, True, True
    t = _case(2, N, C, sp, P, seed=8800 + N + C, spread=1.1)
    off = offsets(N, mc)
    want = _run_all_stages(cs_oracle, t, off, pad, align, ke, mc, "cpu")
    ops.force_path(2)
    try:
        got = _run_all_stages(_Shared(), t, off, pad, align, ke, mc, DEV)
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for k in want:
        assert_close(got[k], want[k], "N=%d C=%d P=%d (XCD-aware order): %s" % (N, C, P, k))


ROW_CASES = [(3, 8, 0, 0, True, True), (3, 8, 2, 0, True, True), (3, 2, 1, 1, False, False), (3, 16, 2, 2, True, False),
             (3, 4, 0, 2, False, True), (2, 2, 0, 0, True, True), (2, 32, 2, 1, True, False), (2, 64, 1, 0, False, False),
             (3, 3, 0, 0, True, True), (3, 1, 2, 1, True, False),   # 3D with 1..3 channels: one zero-padded quad
             (3, 5, 0, 0, True, True), (3, 6, 2, 1, False, True), (3, 12, 0, 0, True, True)]   # padded to 8 / 16


class _Shared(object):
    """ops with one StepContext threaded through every call (what the autograd chain does)."""

    def __init__(self):
        self.step = ops.StepContext()

    def __getattr__(self, name):
        fn = getattr(ops, name)
        return lambda *a: fn(*a, ctx=self.step)


@pytest.mark.parametrize("d,C,ke,pad,align,mc", ROW_CASES)
@pytest.mark.parametrize("shared,force", [(False, 2), (True, 2), (True, 3)])
def test_row_scatter_path_matches_cpu_oracle(d, C, ke, pad, align, mc, shared, force):
    """Shapes outside the tiled path (3D; 2D with C = 2, 32, 64): p-ordered outputs from the direct or channels-last
    point kernels, input-shaped gradients by row atomics into a channels-last scratch (force 3) or, for the small
    crowded 3D tables these cases are with C in {4, 8}, by the plan-by-cell + wave-per-cell path (force 2)."""
    N, P = 2, 5000   # 3D: 7*10*8 = 560 cells, 8 samples per cell and more -> the dense path applies (force 2)
    sp = (11, 13) if d == 2 else (6, 9, 7)
    t = _case(d, N, C, sp, P, seed=8100 + 10 * C + ke, spread=1.2)
    off = offsets(N, mc)
    want = _run_all_stages(cs_oracle, t, off, pad, align, ke, mc, "cpu")
    ops.force_path(force)
    try:
        # shared: the channels-last copy of `input` (and the 3D plan) is built once per step
        got = _run_all_stages(_Shared() if shared else ops, t, off, pad, align, ke, mc, DEV)
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for k in want:
        assert_close(got[k], want[k], "rows d=%d C=%d kernel=%d pad=%d align=%s mc=%s: %s" % (d, C, ke, pad, align, mc, k))


TILES3_CASES = [   # C, (W, H, D) as sp = (D, H, W) below, P, kernel, pad, align, multicell
    (8, (20, 9, 37), 30000, 2, 0, True, True),      # the config-3 kind: sparse, several tiles per axis, W % 4 != 0
    (8, (16, 16, 32), 3000, 0, 0, True, True),      # W % 4 == 0: float4 row stores; ~0.3 samples per cell
    (3, (7, 21, 18), 20000, 0, 1, False, False),    # one zero-padded quad, border padding, align_corners = False
    (16, (12, 6, 40), 25000, 2, 2, True, False),    # 4 quads (two waves per workgroup), reflection
    (5, (33, 5, 17), 20000, 1, 0, True, True),      # padded to 8, linear kernel
    (4, (4, 4, 16), 2500, 0, 0, True, True),        # a single tile per (y, z), crowded: many entries with the same code
    (1, (9, 3, 5), 300, 2, 0, False, True),         # tiny
    (24, (20, 9, 21), 20000, 2, 0, True, True),     # more than 16 channels: ranges of 16 + 8 through the same path
    (32, (16, 8, 24), 15000, 0, 1, False, True),    # 16 + 16
]


@pytest.mark.parametrize("C,dims,P,ke,pad,align,mc", TILES3_CASES)
@pytest.mark.parametrize("shared", [False, True])
def test_tiles3_path_matches_cpu_oracle(C, dims, P, ke, pad, align, mc, shared):
    """3D tables that are not crowded enough for the wave-per-cell path: grad_input cut into 16x4x4-node tiles, every
    sample listed in the tiles that own its corners, one wave per tile, no atomics (cs_dense3d.cuh tiles3).  Partial
    tiles at every face, widths that are and are not multiples of 4, every quad count, all paddings."""
    N = 3
    sp = dims
    D, H, W = sp
    t = _case(3, N, C, sp, P, seed=3300 + C + P, spread=1.15)
    off = offsets(N, mc)
    want = _run_all_stages(cs_oracle, t, off, pad, align, ke, mc, "cpu")
    lib = _lib.load()
    ops.force_path(2)
    try:
        assert lib.cs3d_plan_bytes(N, min(C, 16), D, H, W, P) > 0 and P < 8 * (D + 1) * (H + 1) * (W + 1)   # a plan, and not the cell one
        got = _run_all_stages(_Shared() if shared else ops, t, off, pad, align, ke, mc, DEV)
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for k in want:
        assert_close(got[k], want[k], "tiles3 C=%d %s P=%d kernel=%d pad=%d align=%s mc=%s: %s" % (C, sp, P, ke, pad, align, mc, k))


@pytest.mark.parametrize("d,C,force", [(2, 16, 2), (2, 4, 2), (2, 3, 0), (2, 32, 2), (3, 8, 2), (3, 3, 0), (3, 2, 2)])
@pytest.mark.parametrize("with_cI", [False, True])
def test_second_backward_without_table_gradient(d, C, force, with_cI):
    """grad_input == NULL in cs{2,3}d_backward_backward (the engine told the autograd layer that d/d input of this
    stage is not used): grad_grid and grad_grad_out must be what the full call gives, on every path."""
    N, P = 2, 2500
    sp = (37, 50) if d == 2 else (6, 9, 7)
    t = _case(d, N, C, sp, P, seed=8800 + C, spread=1.15)
    off = offsets(N, True).to(DEV)
    inp, grid, gO, cG = _g(t["inp"]), _g(t["grid"]), _g(t["gOut"]), _g(t["cG"])
    cI = _g(t["cI"]) if with_cI else None
    ops.force_path(force)
    try:
        for shared in (False, True):
            step = ops.StepContext() if shared else None
            full = ops.backward_backward(cI, cG, inp, grid, gO, off, 0, True, with_cI, 0, True, ctx=step)
            part = ops.backward_backward(cI, cG, inp, grid, gO, off, 0, True, with_cI, 0, True, ctx=step,
                                         want_grad_input=False)
            torch.cuda.synchronize()
            assert part[0] is None
            # separately compiled kernel variants: same formulas, the compiler may fuse multiply-adds differently
            assert_close(part[1], full[1], "grad_grid without grad_input")
            assert_close(part[2], full[2], "grad_grad_out without grad_input")
    finally:
        ops.force_path(0)


@pytest.mark.parametrize("C", [1, 3, 4, 16])
@pytest.mark.parametrize("force", [0, 2])
def test_exact_mixed_second_backward_with_grad_out_input(C, force):
    """'+mixed' second backward WITH a grad_out_input while d/d input is wanted (a double backward through grad_input):
    the tiled 2D path hands this call to the direct kernel for grad_grid and to the row-atomic scatter for grad_input,
    which exists only for power-of-two channel counts >= 2 -- C = 1 and 3 must stay on the direct kernel (round-1
    advisor finding: they reached row_scatter with a channel shift of -1, out of bounds).  N*P >= 2^16 so that the
    fast paths are chosen without forcing; the direct kernels (force = 1) are the reference."""
    N, P = 2, 40000
    t = _case(2, N, C, (33, 29), P, seed=4100 + C, spread=1.1)
    off = offsets(N, True).to(DEV)
    inp, grid, gO, cG, cI = (_g(t[k]) for k in ("inp", "grid", "gOut", "cG", "cI"))
    ke = 0 | ops.EXACT_MIXED
    ops.force_path(1)
    try:
        want = ops.backward_backward(cI, cG, inp, grid, gO, off, 0, True, True, ke, True)
        ops.force_path(force)
        got = ops.backward_backward(cI, cG, inp, grid, gO, off, 0, True, True, ke, True, ctx=ops.StepContext())
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for a, b, nm in zip(got, want, ("grad_input", "grad_grid", "grad_grad_out")):
        assert torch.isfinite(a).all()
        assert_close(a, b, "exact-mixed bb with grad_out_input C=%d force=%d: %s" % (C, force, nm))


@pytest.mark.parametrize("d,C,force", [(2, 16, 2), (2, 8, 2), (2, 3, 0), (2, 32, 2), (3, 8, 2), (3, 5, 0), (3, 2, 2)])
def test_expanded_cotangents_equal_contiguous_ones(d, C, force):
    """cs_cotangent_layout: grad_output / grad_out_ggout expanded along n (stride 0, what the backward of PIXEL's
    sum over n produces) must give what their `.contiguous()` copies give, in every stage, on every path."""
    N, P = 3, 2500
    sp = (37, 50) if d == 2 else (6, 9, 7)
    t = _case(d, N, C, sp, P, seed=9900 + C, spread=1.15)
    off = offsets(N, True).to(DEV)
    inp, grid, cI, cG, hG = (_g(t[k]) for k in ("inp", "grid", "cI", "cG", "hG"))
    gO_e = _g(t["gOut"])[:1].expand(N, *t["gOut"].shape[1:])
    hO_e = _g(t["hO"])[1:2].expand(N, *t["hO"].shape[1:])
    assert gO_e.stride(0) == 0 and not gO_e.is_contiguous()
    gO_c, hO_c = gO_e.contiguous(), hO_e.contiguous()
    ops.force_path(force)
    try:
        for shared in (False, True):
            res = []
            for gO, hO in ((gO_e, hO_e), (gO_c, hO_c)):
                step = ops.StepContext() if shared else None
                r = list(ops.backward(gO, inp, grid, off, 0, True, True, 0, True, ctx=step))
                r += list(ops.backward(gO, inp, grid, off, 0, True, False, 0, True, ctx=step))[1:]
                r += list(ops.backward_backward(cI, cG, inp, grid, gO, off, 0, True, True, 0, True, ctx=step))
                r += list(ops.backward_backward(None, cG, inp, grid, gO, off, 0, True, False, 0, True, ctx=step,
                                                want_grad_input=False))[1:]
                r += list(ops.backward_backward_backward(inp, grid, gO, cG, hG, off, 0, True, True, 0, True, ctx=step))
                r += list(ops.bbb_fused(inp, grid, gO, cG, hG, hO, off, 0, True, 0, True, ctx=step))
                r += list(ops.bbb_fused(inp, grid, gO, cG, hG, None, off, 0, True, 0, True, ctx=step))
                res.append(r)
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(*res)):
                assert a.is_contiguous() and a.shape == b.shape
                # same kernels, same values: equal up to the run-to-run order of scattered sums (grad_input) and
                # of the plan's within-cell order (grad_grid from the tile walkers): last-bit differences only
                assert_close(a, b, "output %d, expanded vs contiguous cotangents" % i, tol=2e-6)
    finally:
        ops.force_path(0)
    with pytest.raises(RuntimeError):     # any other non-contiguous layout is still refused
        ops.backward(gO_c.transpose(0, 1).contiguous().transpose(0, 1), inp, grid, off, 0, True, True, 0, True)


@pytest.mark.parametrize("d,C,force,P", [(2, 16, 2, 30000), (2, 6, 2, 30000), (2, 16, 4, 30000), (2, 3, 0, 3000), (2, 40, 0, 3000),
                                         (2, 4, 3, 40000), (3, 8, 2, 30000), (3, 3, 0, 2000), (3, 2, 2, 30000), (2, 5, 1, 3000),
                                         (3, 4, 1, 3000)])
@pytest.mark.parametrize("mc", [True, False])
def test_broadcast_grid_equals_repeated_grid(d, C, force, P, mc):
    """One set of P points for every n -- a (1, ..., dim) grid, CS_GRID_BROADCAST -- against the same points repeated N
    times as PIXEL does (reference test/test_2d.py:38), on every execution path and every stage.  Per-sample results must
    be the very same numbers; everything grid-shaped has a leading 1 and is the sum over n of the repeated run's."""
    N = 3
    sp = ((40, 33) if force != 3 else (16, 16)) if d == 2 else (9, 11, 7)
    t = _case(d, N, C, sp, P, seed=77 + C + d)
    off = offsets(N, mc).to(DEV)
    inp, gOut, hO, cI = (_g(t[k]) for k in ("inp", "gOut", "hO", "cI"))
    g1, cG1, hG1 = (_g(t[k][:1].contiguous()) for k in ("grid", "cG", "hG"))
    rep = lambda x: x.repeat((N,) + (1,) * (x.dim() - 1))
    gN, cGN, hGN = rep(g1), rep(cG1), rep(hG1)
    ops.force_path(force)
    try:
        for shared in (False, True):
            a, b = (ops.StepContext() if shared else None), (ops.StepContext() if shared else None)
            assert torch.equal(ops.forward(inp, g1, off, 0, True, 0, mc, ctx=a), ops.forward(inp, gN, off, 0, True, 0, mc, ctx=b))
            gI1, gG1 = ops.backward(gOut, inp, g1, off, 0, True, True, 0, mc, ctx=a)
            gIN, gGN = ops.backward(gOut, inp, gN, off, 0, True, True, 0, mc, ctx=b)
            assert gG1.shape == g1.shape
            assert_close(gI1, gIN, "broadcast grid: grad_input")
            assert_close(gG1, gGN.sum(0, keepdim=True), "broadcast grid: grad_grid")
            assert_close(ops.backward(gOut, inp, g1, off, 0, True, False, 0, mc, ctx=a)[1], gGN.sum(0, keepdim=True),
                         "broadcast grid: grad_grid alone")
            b1 = ops.backward_backward(cI, cG1, inp, g1, gOut, off, 0, True, True, 0, mc, ctx=a)
            bN = ops.backward_backward(cI, cGN, inp, gN, gOut, off, 0, True, True, 0, mc, ctx=b)
            assert_close(b1[0], bN[0], "broadcast grid: second-backward grad_input")
            assert_close(b1[1], bN[1].sum(0, keepdim=True), "broadcast grid: second-backward grad_grid")
            assert torch.equal(b1[2], bN[2])
            b1 = ops.backward_backward(None, cG1, inp, g1, gOut, off, 0, True, False, 0, mc, ctx=a, want_grad_input=False)
            bN = ops.backward_backward(None, cGN, inp, gN, gOut, off, 0, True, False, 0, mc, ctx=b, want_grad_input=False)
            assert b1[0] is None and torch.equal(b1[2], bN[2])
            assert_close(b1[1], bN[1].sum(0, keepdim=True), "broadcast grid: second-backward grad_grid, no table gradient")
            f1 = ops.bbb_fused(inp, g1, gOut, cG1, hG1, hO, off, 0, True, 0, mc, ctx=a)
            fN = ops.bbb_fused(inp, gN, gOut, cGN, hGN, hO, off, 0, True, 0, mc, ctx=b)
            assert_close(f1[0], fN[0], "broadcast grid: third-backward grad_input")
            assert torch.equal(f1[1], fN[1])
            k1 = ops.backward_backward_backward(inp, g1, gOut, cG1, hG1, off, 0, True, True, 0, mc, ctx=a)
            kN = ops.backward_backward_backward(inp, gN, gOut, cGN, hGN, off, 0, True, True, 0, mc, ctx=b)
            assert_close(k1[0], kN[0], "broadcast grid: K4 grad_input")
            assert torch.equal(k1[1], kN[1])
        e1 = ops.bbb_grid(inp, g1, gOut, cG1, hG1, hO, off, 0, True, 0 | ops.EXACT_MIXED, mc)
        eN = ops.bbb_grid(inp, gN, gOut, cGN, hGN, hO, off, 0, True, 0 | ops.EXACT_MIXED, mc)
        assert e1.shape == g1.shape
        assert_close(e1, eN.sum(0, keepdim=True), "broadcast grid: third-order grid gradient")
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)


@pytest.mark.parametrize("d", [2, 3])
def test_broadcast_grid_through_autograd(d):
    """The PIXEL pattern without the repeat: CosineSampler.apply(cells, grid_1) == CosineSampler.apply(cells,
    grid_1.repeat(N, ...)) through three levels of autograd, gradients w.r.t. the points included."""
    N, C, P = 4, 8, 20000
    sp = (24, 31) if d == 2 else (7, 9, 8)
    t = _case(d, N, C, sp, P, seed=4242 + d, spread=0.95)
    Fn = CosineSampler2d if d == 2 else CosineSampler3d
    res = []
    for bc in (True, False):
        cells = _g(t["inp"]).clone().requires_grad_(True)
        pts = _g(t["grid"][:1]).clone().requires_grad_(True)
        grid = pts if bc else pts.repeat((N,) + (1,) * (pts.dim() - 1))
        out = Fn.apply(cells, grid, "zeros", True, "cosine", True)
        u = torch.tanh(out.sum(0)).sum(0)                     # (1.., P)
        (u_g,) = torch.autograd.grad(u.sum(), pts, create_graph=True)
        (u_gg,) = torch.autograd.grad(u_g[..., 0].sum(), pts, create_graph=True)
        loss = (u_gg[..., 0] ** 2).mean() + (u ** 2).mean()
        (gc,) = torch.autograd.grad(loss, cells)
        res.append((out.detach(), u_g.detach(), u_gg.detach(), gc))
    for a, b, what in zip(res[0], res[1], ("output", "u_x", "u_xx", "d loss / d cells")):
        assert_close(a, b, "broadcast grid through autograd: " + what, tol=2e-5)


@pytest.mark.parametrize("d,C", [(2, 16), (2, 4), (3, 8)])
def test_autograd_chain_on_fast_paths_vs_composite(d, C):
    """Large enough for the fast paths (2D: tiled, 3D: channels-last + row scatter) to be chosen by the
    library itself; driven through torch.autograd to third order with one StepContext per forward, and
    compared with the exact-derivative composite evaluated on the GPU (axis-aligned second/third
    order, where the op and the exact derivative agree: DESIGN.md section 2)."""
    from oracle import composite
    torch.manual_seed(5)
    N, S, P = 4, 48, 40000
    cells = torch.rand((N, C) + (S,) * d, device=DEV, requires_grad=True)
    coords = [(torch.rand(P, 1, device=DEV) * 2 - 1).requires_grad_(True) for _ in range(d)]
    grid = torch.cat(coords, -1).view((1,) * d + (P, d)).repeat((N,) + (1,) * (d + 1))
    w = torch.randn((N, C) + (1,) * (d - 1) + (P,), device=DEV)
    Fn = CosineSampler2d if d == 2 else CosineSampler3d

    def quantities(sample):
        out = sample(cells, grid)
        u = (out * w).sum(dim=(0, 1)).view(P, 1)
        res = {"u": u}
        g = lambda y, x: torch.autograd.grad(y, x, torch.ones_like(y), retain_graph=True, create_graph=True)[0]
        res["u_cell"] = g(u, cells)
        for j, nm in enumerate("xyz"[:d]):
            uj = g(u, coords[j])
            ujj = g(uj, coords[j])
            res["u_" + nm], res["u_" + nm * 2] = uj, ujj
            res["u_%s_cell" % nm], res["u_%s_cell" % (nm * 2)] = g(uj, cells), g(ujj, cells)
        return {k: v.detach() for k, v in res.items()}

    got = quantities(lambda c, g_: Fn.apply(c, g_, "zeros", True, "cosine", True))
    want = quantities(lambda c, g_: composite.grid_sample_nd(c, g_, "cosine", True, True))
    for k in want:
        assert_close(got[k], want[k], "autograd %dD C=%d %s" % (d, C, k), tol=2e-5)


@pytest.mark.parametrize("d,C,kern,force", [(2, 16, "cosine", 0), (2, 4, "smooth-step", 0), (2, 3, "cosine", 0),
                                            (2, 16, "bilinear", 0), (2, 8, "cosine", 1), (2, 32, "cosine", 2),
                                            (3, 8, "smooth-step", 0), (3, 3, "cosine", 0)])
def test_mixed_second_derivatives_with_the_exact_flag(d, C, kern, force):
    """kernel + '+mixed' (CS_KERNEL_EXACT_MIXED, not in the reference): u_xy, u_yx and d(u_xy)/d(cells) through
    autograd are those of the interpolant -- compared with the exact-derivative composite.  Without the suffix the 2D
    op returns u_xy = 0, as the reference does."""
    from oracle import composite
    torch.manual_seed(6)
    N, S, P = 4, 48, 40000
    cells = torch.rand((N, C) + (S,) * d, device=DEV, requires_grad=True)
    # keep clear of the cell boundaries, where k'' of cosine/smooth-step jumps
    coords = [(torch.rand(P, 1, device=DEV) * 1.9 - 0.95).requires_grad_(True) for _ in range(d)]
    grid = torch.cat(coords, -1).view((1,) * d + (P, d)).repeat((N,) + (1,) * (d + 1))
    w = torch.randn((N, C) + (1,) * (d - 1) + (P,), device=DEV)
    Fn = CosineSampler2d if d == 2 else CosineSampler3d
    name = {"cosine": "cosine", "smooth-step": "smoothstep", "bilinear": "linear"}[kern]
    if d == 3 and kern == "bilinear":
        kern = "trilinear"
    g = lambda y, x: torch.autograd.grad(y, x, torch.ones_like(y), retain_graph=True, create_graph=True)[0]

    def quantities(sample):
        u = (sample(cells, grid) * w).sum(dim=(0, 1)).view(P, 1)
        u_x, u_y = g(u, coords[0]), g(u, coords[1])
        u_xy, u_yx, u_xx = g(u_x, coords[1]), g(u_y, coords[0]), g(u_x, coords[0])
        res = {"u_xy": u_xy, "u_yx": u_yx, "u_xx": u_xx, "u_xy_cell": g(u_xy, cells), "u_xx_cell": g(u_xx, cells),
               "mixed_loss_cell": g(((u_xy + 0.5 * u_xx) ** 2).mean().view(1, 1), cells)}
        # third order w.r.t. the coordinates (the reference returns None there): only with '+mixed'
        def g1(y, x):   # a derivative that is identically zero (linear kernel) has no graph: read None as zeros
            if not y.requires_grad:
                return torch.zeros_like(x)
            r = torch.autograd.grad(y, x, torch.ones_like(y), retain_graph=True, allow_unused=True)[0]
            return torch.zeros_like(x) if r is None else r
        res.update({"u_xxx": g1(u_xx, coords[0]), "u_xxy": g1(u_xx, coords[1]), "u_xyx": g1(u_xy, coords[0]),
                    "u_xyy": g1(u_xy, coords[1])})
        if d == 3:
            res["u_xyz"] = g1(u_xy, coords[2])
        return {k: v.detach() for k, v in res.items()}

    ops.force_path(force)
    try:
        got = quantities(lambda c, g_: Fn.apply(c, g_, "zeros", True, kern + "+mixed", True))
        plain = g((Fn.apply(cells, grid, "zeros", True, kern, True) * w).sum(dim=(0, 1)).view(P, 1), coords[0])
        plain_xy = torch.autograd.grad(plain, coords[1], torch.ones_like(plain), allow_unused=True)[0]
    finally:
        ops.force_path(0)
    want = quantities(lambda c, g_: composite.grid_sample_nd(c, g_, name, True, True))
    for k in want:
        assert_close(got[k], want[k], "exact-mixed %dD C=%d %s %s" % (d, C, kern, k), tol=3e-5)
    if d == 2:   # the reference's behaviour without the flag: no mixed term at all
        assert plain_xy is None or float(plain_xy.abs().max()) == 0.0


@pytest.mark.parametrize("d", [2, 3])
def test_reference_test_scripts_at_their_own_shapes(d):
    """The reference's two test scripts at their real sizes (test/test_2d.py:20-40: 96 cells of 4x16x16,
    100 000 points; test/test_3d.py:14-34: 50 cells of 4x16^3), same quantities, with this repo's composite
    (validated against the reference's ground truth, tests/test_oracle_golden.py) standing in for
    test/grid_sampler.py, and the reference's own acceptance test: rtol 1e-4 on d loss / d cells."""
    from oracle import composite
    torch.manual_seed(51 if d == 2 else 6)
    n_cell, cell_dim, numb = (96, 4, 100000) if d == 2 else (50, 4, 100000)
    cells = torch.rand((n_cell, cell_dim) + (16,) * d, device=DEV, requires_grad=True)
    # Second derivatives of the cosine kernel JUMP at cell boundaries (k''(0) = -k''(1)), so a sample whose
    # source index lands within rounding error of an integer may legitimately fall on either side: the op
    # fuses x*(size-2)+offset into one fmaf like a GPU build of the reference, torch's composite rounds twice.
    # With n_cell offsets n/n_cell per point, random points always have some n that close to a boundary; draw
    # the points on the lattice i = (k + 1/2)/n_cell instead, half a lattice step away from every boundary.
    def lattice():
        k = torch.randint(0, 14 * n_cell, (numb, 1), device=DEV).double()
        return (2.0 * ((k + 0.5) / n_cell) / 14.0 - 1.0).float().requires_grad_(True)
    coords = [lattice() for _ in range(d)]
    net = torch.nn.Sequential(torch.nn.Linear(cell_dim, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1)).to(DEV)
    Fn = CosineSampler2d if d == 2 else CosineSampler3d

    def run(sample):
        grid = torch.cat(coords, -1).view((1,) * d + (numb, d)).repeat((n_cell,) + (1,) * (d + 1))
        val = net(sample(cells, grid).sum(0).view(cell_dim, -1).t())
        g = lambda y, x: torch.autograd.grad(y, x, torch.ones_like(y), retain_graph=True, create_graph=True)[0]
        firsts = [g(val, c) for c in coords]
        seconds = [g(f, c) for f, c in zip(firsts, coords)]
        if d == 2:   # test_2d.py:221
            f_pred = firsts[1] * 2 + 5 * (val ** 3) - 5 * val - 0.0001 * seconds[0]
        else:        # test_3d.py:270
            f_pred = seconds[0] + seconds[1] + seconds[2] + val
        res = dict(val=val, u_cell=g(val, cells), u_x=firsts[0], u_xx=seconds[0], u_xx_cell=g(seconds[0], cells))
        res["dloss"] = torch.autograd.grad(torch.mean(f_pred ** 2), cells)[0]      # last: frees the graph
        return {k: v.detach() for k, v in res.items()}

    got = run(lambda c, g_: Fn.apply(c, g_, "zeros", True, "cosine", True))
    want = run(lambda c, g_: composite.grid_sample_nd(c, g_, "cosine", True, True))
    for k in want:
        assert_close(got[k], want[k], "reference test_%dd shapes: %s" % (d, k), tol=2e-5)
    a, b = got["dloss"].detach().cpu(), want["dloss"].detach().cpu()
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))     # test_2d.py:244 / test_3d.py:293


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16, torch.bfloat16])
def test_other_float_dtypes_on_gpu(dtype):
    """double tensors are converted at the autograd boundary; half / bfloat16 tensors too on the direct path (a problem
    this small), with fp32 kernels in between -- values checked against the fp32 op on the same (rounded) numbers."""
    torch.manual_seed(9)
    cells32 = torch.rand(3, 4, 12, 12, device=DEV).to(dtype).float()      # representable in `dtype`
    grid32 = (torch.rand(3, 1, 500, 2, device=DEV) * 2 - 1).to(dtype).float()
    ref = CosineSampler2d.apply(cells32, grid32, "zeros", True, "cosine", True)
    cells = cells32.to(dtype).requires_grad_(True)
    grid = grid32.to(dtype).requires_grad_(True)
    out = CosineSampler2d.apply(cells, grid, "zeros", True, "cosine", True)
    assert out.dtype == dtype
    eps = {torch.float64: 1e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert rel_err(out, ref) <= eps
    gI, gG = torch.autograd.grad(out.sum(), (cells, grid), create_graph=True)
    assert gI.dtype == dtype and gG.dtype == dtype and torch.isfinite(gI.float()).all()
    (gc,) = torch.autograd.grad(gG[..., 1].sum(), cells)
    assert gc.dtype == dtype and gc.shape == cells.shape


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d,C,P", [(2, 16, 40000), (2, 6, 40000), (3, 8, 40000), (2, 16, 39999), (2, 7, 40002), (2, 3, 40000)])
def test_native_half_streams(dtype, d, C, P):
    """float16 / bfloat16 tensors on the fast paths: the channel-major streams (output, grad_output, grad_grad_out,
    grad_out_ggout) are read and written in that type by the kernels (CS_STREAM_F16 / CS_STREAM_BF16; the reference
    dispatches half too, 2d.cu:905, :948, :1009, :1076), fp32 arithmetic in between.  Every stage is compared with the
    fp32 op on the same numbers: the inputs are representable in `dtype`, so results differ by the rounding of the
    16-bit OUTPUTS only (fp32 outputs -- grad_grid, the input-shaped gradients -- agree to fp32 accuracy)."""
    eps = 1e-3 if dtype == torch.float16 else 8e-3
    N = 2
    sp = (40, 33) if d == 2 else (9, 11, 7)
    t = _case(d, N, C, sp, P, seed=5150 + C, spread=1.1)
    off = offsets(N, True).to(DEV)
    inp, grid, cG, hG = (_g(t[k]) for k in ("inp", "grid", "cG", "hG"))
    gO16, hO16 = _g(t["gOut"]).to(dtype), _g(t["hO"]).to(dtype)
    gO32, hO32 = gO16.float(), hO16.float()
    assert ops.half_streams_ok(inp, grid)
    ops.force_path(2)
    try:
        for shared in (False, True):
            s16 = ops.StepContext() if shared else None
            s32 = ops.StepContext() if shared else None
            out16 = ops.forward(inp, grid, off, 0, True, 0, True, ctx=s16, out_dtype=dtype)
            out32 = ops.forward(inp, grid, off, 0, True, 0, True, ctx=s32)
            assert out16.dtype == dtype and rel_err(out16, out32) <= eps
            # same fp32 arithmetic, one rounding at the store: bit-identical to rounding the fp32 result (this pins the
            # lane-pair dword path of even P -- cs_tiled.cuh st_pair16 -- as much as the element path of odd P)
            assert d == 3 or torch.equal(out16, out32.to(dtype))   # (3D: the fp32 call may take another fast path)
            gI16, gG16 = ops.backward(gO16, inp, grid, off, 0, True, True, 0, True, ctx=s16)
            gI32, gG32 = ops.backward(gO32, inp, grid, off, 0, True, True, 0, True, ctx=s32)
            assert gI16.dtype == torch.float32
            assert_close(gI16, gI32, "half streams: grad_input")
            assert_close(gG16, gG32, "half streams: grad_grid")
            # a cotangent that starts on an odd 16-bit element cannot be moved as dwords: the element path must give
            # exactly what the lane-pair path gave
            odd = torch.empty(gO16.numel() + 1, dtype=dtype, device=DEV)[1:].view(gO16.shape).copy_(gO16)
            assert odd.data_ptr() % 4 == 2
            gIo, gGo = ops.backward(odd, inp, grid, off, 0, True, True, 0, True)
            gIa, gGa = ops.backward(gO16, inp, grid, off, 0, True, True, 0, True)
            assert torch.equal(gGo, gGa) and rel_err(gIo, gIa) <= 1e-6
            b16 = ops.backward_backward(None, cG, inp, grid, gO16, off, 0, True, False, 0, True, ctx=s16)
            b32 = ops.backward_backward(None, cG, inp, grid, gO32, off, 0, True, False, 0, True, ctx=s32)
            assert_close(b16[0], b32[0], "half streams: second-backward grad_input")
            assert_close(b16[1], b32[1], "half streams: second-backward grad_grid")
            assert b16[2].dtype == dtype and rel_err(b16[2], b32[2]) <= eps
            f16 = ops.bbb_fused(inp, grid, gO16, cG, hG, hO16, off, 0, True, 0, True, ctx=s16)
            f32 = ops.bbb_fused(inp, grid, gO32, cG, hG, hO32, off, 0, True, 0, True, ctx=s32)
            assert_close(f16[0], f32[0], "half streams: third-backward grad_input")
            assert f16[1].dtype == dtype and rel_err(f16[1], f32[1]) <= eps
            k16 = ops.backward_backward_backward(inp, grid, gO16, cG, hG, off, 0, True, True, 0, True, ctx=s16)
            k32 = ops.backward_backward_backward(inp, grid, gO32, cG, hG, off, 0, True, True, 0, True, ctx=s32)
            assert_close(k16[0], k32[0], "half streams: K4 grad_input")
            assert rel_err(k16[1], k32[1]) <= eps
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    # and through the autograd layer: half tensors in, half tensors out, no conversion of the big tensors
    Fn = CosineSampler2d if d == 2 else CosineSampler3d
    cells_h = inp.to(dtype).requires_grad_(True)
    ops.force_path(2)
    try:
        out = Fn.apply(cells_h, grid.to(dtype), "zeros", True, "cosine", True)
        ref = Fn.apply(cells_h.float(), grid.to(dtype).float(), "zeros", True, "cosine", True)
        assert out.dtype == dtype and rel_err(out, ref) <= eps
        (gc,) = torch.autograd.grad((out.float() * gO32).sum(), cells_h)
        (gr,) = torch.autograd.grad((ref * gO32).sum(), cells_h)
        assert gc.dtype == dtype and rel_err(gc, gr) <= eps
    finally:
        ops.force_path(0)


@pytest.mark.parametrize("d,C,force,P", [(2, 16, 2, 9000), (2, 6, 2, 9000), (2, 4, 3, 20000), (2, 5, 1, 3000), (3, 8, 2, 9000),
                                         (3, 4, 1, 3000), (3, 2, 2, 9000)])
@pytest.mark.parametrize("mc", [True, False])
def test_broadcast_grid_vs_oracle(d, C, force, P, mc):
    """SURVEY 8(f)1, against the ORACLE (the self-comparison above shows the two forms agree; this shows they are right):
    one (1, ..., dim) set of points on every path -- tiled / crowded / direct in 2D, channels-last point kernels with tile
    and row-atomic scatter / direct in 3D -- against cs_oracle on the points repeated N times (the reference's form,
    test/test_2d.py:38); everything grid-shaped is the oracle's sum over n."""
    N = 3
    sp = ((40, 33) if force != 3 else (16, 16)) if d == 2 else (9, 11, 7)
    t = _case(d, N, C, sp, P, seed=177 + C + d)
    off = offsets(N, mc)
    rep = lambda x: x[:1].repeat((N,) + (1,) * (x.dim() - 1)).contiguous()
    tr = dict(t)
    for k in ("grid", "cG", "hG"):
        tr[k] = rep(t[k])
    want = {}
    want["out"] = cs_oracle.forward(tr["inp"], tr["grid"], off, 0, True, 0, mc)
    want["gI"], want["gG"] = cs_oracle.backward(tr["gOut"], tr["inp"], tr["grid"], off, 0, True, True, 0, mc)
    want["bbI"], want["bbG"], want["bbO"] = cs_oracle.backward_backward(None, tr["cG"], tr["inp"], tr["grid"], tr["gOut"], off,
                                                                        0, True, False, 0, mc)
    want["fI"], want["fO"] = cs_oracle.bbb_fused(tr["inp"], tr["grid"], tr["gOut"], tr["cG"], tr["hG"], tr["hO"], off, 0,
                                                 True, 0, mc)
    x = {k: _g(v) for k, v in t.items()}
    g1, cG1, hG1 = (x[k][:1].contiguous() for k in ("grid", "cG", "hG"))
    offd = off.to(DEV)
    ops.force_path(force)
    try:
        for shared in (False, True):
            sc = ops.StepContext(points_order="random") if shared else None
            got = {}
            got["out"] = ops.forward(x["inp"], g1, offd, 0, True, 0, mc, ctx=sc)
            got["gI"], got["gG"] = ops.backward(x["gOut"], x["inp"], g1, offd, 0, True, True, 0, mc, ctx=sc)
            got["bbI"], got["bbG"], got["bbO"] = ops.backward_backward(None, cG1, x["inp"], g1, x["gOut"], offd, 0, True, False,
                                                                       0, mc, ctx=sc)
            got["fI"], got["fO"] = ops.bbb_fused(x["inp"], g1, x["gOut"], cG1, hG1, x["hO"], offd, 0, True, 0, mc, ctx=sc)
            torch.cuda.synchronize()
            for k in want:
                w = want[k].sum(0, keepdim=True) if k in ("gG", "bbG") else want[k]
                assert_close(got[k], w, "broadcast grid vs oracle d=%d C=%d force=%d mc=%s shared=%s: %s" % (d, C, force, mc, shared, k))
    finally:
        ops.force_path(0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d,C,P", [(2, 16, 9000), (2, 6, 9001), (3, 8, 9000), (2, 3, 9000)])
def test_native_half_streams_vs_oracle(dtype, d, C, P):
    """SURVEY 8(f)3, against the ORACLE: 16-bit streams on the fast paths vs cs_oracle on the ROUNDED inputs (fp32
    arithmetic on both sides, so fp32 outputs agree to 1e-5 and 16-bit outputs to the rounding of the type: 2^-11 relative
    for float16, 2^-8 for bfloat16, taken per tensor against its largest element)."""
    N = 2
    sp = (40, 33) if d == 2 else (9, 11, 7)
    t = _case(d, N, C, sp, P, seed=6150 + C, spread=1.1)
    for k in ("gOut", "hO"):
        t[k] = t[k].to(dtype).float()
    off = offsets(N, True)
    want = {}
    want["out"] = cs_oracle.forward(t["inp"], t["grid"], off, 0, True, 0, True)
    want["gI"], want["gG"] = cs_oracle.backward(t["gOut"], t["inp"], t["grid"], off, 0, True, True, 0, True)
    want["bbI"], want["bbG"], want["bbO"] = cs_oracle.backward_backward(None, t["cG"], t["inp"], t["grid"], t["gOut"], off,
                                                                        0, True, False, 0, True)
    want["fI"], want["fO"] = cs_oracle.bbb_fused(t["inp"], t["grid"], t["gOut"], t["cG"], t["hG"], t["hO"], off, 0, True, 0, True)
    x = {k: _g(v) for k, v in t.items()}
    g16, h16 = x["gOut"].to(dtype), x["hO"].to(dtype)
    offd = off.to(DEV)
    eps16 = 2.0 ** (-10 if dtype == torch.float16 else -7)
    ops.force_path(2)
    try:
        sc = ops.StepContext(points_order="random")
        got = {}
        got["out"] = ops.forward(x["inp"], x["grid"], offd, 0, True, 0, True, ctx=sc, out_dtype=dtype)
        got["gI"], got["gG"] = ops.backward(g16, x["inp"], x["grid"], offd, 0, True, True, 0, True, ctx=sc)
        got["bbI"], got["bbG"], got["bbO"] = ops.backward_backward(None, x["cG"], x["inp"], x["grid"], g16, offd, 0, True, False,
                                                                   0, True, ctx=sc)
        got["fI"], got["fO"] = ops.bbb_fused(x["inp"], x["grid"], g16, x["cG"], x["hG"], h16, offd, 0, True, 0, True, ctx=sc)
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for k in want:
        is16 = got[k].dtype == dtype
        assert is16 == (k in ("out", "bbO", "fO"))
        assert_close(got[k].float(), want[k], "%s streams vs oracle d=%d C=%d: %s" % (dtype, d, C, k), tol=eps16 if is16 else 1e-5)


def test_step_context_follows_in_place_updates():
    """The channels-last copy is keyed on the tensor's version counter: an optimizer step on `cells`
    between two uses of one context must not serve stale values."""
    torch.manual_seed(6)
    N, C, S, P = 4, 16, 32, 30000
    cells = torch.rand(N, C, S, S, device=DEV)
    grid = torch.rand(N, 1, P, 2, device=DEV) * 2 - 1
    off = multicell_offset(N, True, DEV)
    step = ops.StepContext()
    a = ops.forward(cells, grid, off, 0, True, 0, True, ctx=step)
    cells.mul_(2.0)
    b = ops.forward(cells, grid, off, 0, True, 0, True, ctx=step)
    assert rel_err(b, a * 2.0) <= 1e-6
    grid2 = grid.clone()
    grid2[..., 0] = -grid2[..., 0]
    gOut = torch.randn_like(a)
    g1, _ = ops.backward(gOut, cells, grid, off, 0, True, True, 0, True, ctx=step)
    g2, _ = ops.backward(gOut, cells, grid2, off, 0, True, True, 0, True, ctx=step)     # new grid -> new plan
    r2, _ = ops.backward(gOut, cells, grid2, off, 0, True, True, 0, True)
    assert rel_err(g2, r2) <= 1e-5 and rel_err(g1, r2) > 1e-2


@pytest.mark.parametrize("C", [2, 4, 8, 16, 32])
@pytest.mark.parametrize("P,pad,mode", [(40000, 0, 2), (40000, 1, 2), (40000, 2, 2), (18000, 0, 2), (40000, 0, 3)])
def test_tiled_path_crowded_tables(C, P, pad, mode):
    """PIXEL-like shape (reference test/test_2d.py: 16x16 cells, 1e5 points): hundreds of samples per cell and the
    whole table inside one tile.  From 128 samples per cell on (P >= 128 * 17 * 17 = 36992) the plan bins by cell and
    a wave per (n, cell) bucket does the sums (cell_scatter); below that, or in mode 3, the tile walkers do."""
    N, sp = 3, (16, 16)
    t = _case(2, N, C, sp, P, seed=4242 + C, spread=1.1)
    off = offsets(N, True)
    want = _run_all_stages(cs_oracle, t, off, pad, True, 0, True, "cpu")
    ops.force_path(mode)   # 3: the walkers on a crowded table, too
    try:
        got = _run_all_stages(_Shared(), t, off, pad, True, 0, True, DEV)
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
    for k in want:
        assert_close(got[k], want[k], "crowded C=%d P=%d pad=%d: %s" % (C, P, pad, k))


@pytest.mark.parametrize("shape", [(4, 16, 48, 30000), (96, 4, 16, 40000), (2, 3, 20, 3000), (3, 32, 40, 30000),
                                   (4, 8, (12, 20, 36), 70000), (6, 4, (8, 8, 8), 60000)])
def test_stages_can_be_captured_in_a_hip_graph(shape):
    """Nothing in a stage allocates through HIP, synchronises or touches the host, so a whole
    step -- channels-last copy, plan, forward and the three backward stages -- can be captured into a HIP graph and
    replayed on new data in the same buffers (what a launch-bound PIXEL loop wants).  Tiled, crowded and direct paths in
    2D; in 3D (a tuple of sizes) the tile path and the wave-per-cell path."""
    N, C, S, P = shape
    sp = S if isinstance(S, tuple) else (S, S)
    dm = len(sp)
    gshape = (N,) + (1,) * (dm - 1) + (P, dm)
    oshape = (N, C) + (1,) * (dm - 1) + (P,)
    g = torch.Generator().manual_seed(77)
    cells = torch.rand((N, C) + sp, generator=g).to(DEV)
    grid = (torch.rand(gshape, generator=g) * 2.2 - 1.1).to(DEV)
    gO = torch.randn(oshape, generator=g).to(DEV)
    cG = torch.randn(gshape, generator=g).to(DEV)
    hG = torch.randn(gshape, generator=g).to(DEV)
    hO = torch.randn(oshape, generator=g).to(DEV)
    off = offsets(N, True).to(DEV)

    def step():
        sc = ops.StepContext()
        out = ops.forward(cells, grid, off, 0, True, 0, True, ctx=sc)
        gI, gG = ops.backward(gO, cells, grid, off, 0, True, True, 0, True, ctx=sc)
        bI, bG, bO = ops.backward_backward(None, cG, cells, grid, gO, off, 0, True, False, 0, True, ctx=sc)
        tI, tO = ops.bbb_fused(cells, grid, gO, cG, hG, hO, off, 0, True, 0, True, ctx=sc)
        return [out, gI, gG, bI, bG, bO, tI, tO]

    step()                                   # warm up: library load, allocator pools
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    for trial in range(2):                   # new data in the captured input buffers
        cells.copy_(torch.rand((N, C) + sp, generator=g))
        grid.copy_(torch.rand(gshape, generator=g) * 2.2 - 1.1)
        gO.copy_(torch.randn(oshape, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in captured]
        want = step()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(got, want)):
            assert_close(a, b, "graph replay %d, output %d" % (trial, i), tol=2e-6)


@pytest.mark.parametrize("d,C,force", [(2, 16, 2), (2, 3, 0), (2, 32, 2), (3, 8, 2), (3, 3, 0)])
def test_non_finite_coordinates_give_zeros(d, C, force):
    """NaN / +-Inf / absurdly large grid coordinates touch no node: every output of every stage is finite, and the
    rows of those samples are exactly zero, on every path (the reference leaves this case to undefined casts)."""
    N, P = 2, 3000
    sp = (37, 50) if d == 2 else (6, 9, 7)
    t = _case(d, N, C, sp, P, seed=1234 + C, spread=1.1)
    bad = torch.tensor([float("nan"), float("inf"), -float("inf"), 3e38, -1e30])
    gv = t["grid"].view(N, P, d)
    for k in range(5):
        gv[:, 10 + k, k % d] = bad[k]
    off = offsets(N, True)
    ops.force_path(force)
    try:
        for pad in (0, 1, 2):
            got = _run_all_stages(_Shared(), t, off, pad, True, 0, True, DEV)
            torch.cuda.synchronize()
            for k, v in got.items():
                assert bool(torch.isfinite(v).all()), "pad %d: %s has non-finite values" % (pad, k)
            if pad == 0:
                assert float(got["out"].view(N, C, P)[:, :, 10:15].abs().max()) == 0.0
                assert float(got["gG"].view(N, P, d)[:, 10:15].abs().max()) == 0.0
    finally:
        ops.force_path(0)


def test_tiled_path_empty_and_clustered_points():
    """Degenerate point sets for the plan: every point in one cell, every point out of range."""
    N, C, sp = 2, 16, (40, 33)
    off = offsets(N, True)
    ops.force_path(2)
    try:
        for kind in ("one_cell", "all_outside", "two_points"):
            P = 2 if kind == "two_points" else 5000
            t = _case(2, N, C, sp, P, seed=99)
            if kind == "one_cell":
                t["grid"] = (torch.rand_like(t["grid"]) * 0.01 + 0.3)
            elif kind == "all_outside":
                t["grid"] = torch.rand_like(t["grid"]) + 3.0
            want = _run_all_stages(cs_oracle, t, off, 0, True, 0, True, "cpu")
            got = _run_all_stages(ops, t, off, 0, True, 0, True, DEV)
            torch.cuda.synchronize()
            for k in want:
                assert_close(got[k], want[k], "tiled %s: %s" % (kind, k))
    finally:
        ops.force_path(0)


@pytest.mark.parametrize("name", stage_fixtures())
def test_stage_golden_vectors(name):
    d, kernel, mc = parse_stage_name(name)
    fx = load(name)
    ke = KERNEL_ENUM[kernel]
    N = fx["cells"].shape[0]
    off = multicell_offset(N, mc, DEV)
    cells, grid, gOut = _g(fx["cells"]), _g(fx["grid"]), _g(fx["gOut"])
    out = ops.forward(cells, grid, off, 0, True, ke, mc)
    assert_close(out, fx["out"], name + " out")
    gI, gG = ops.backward(gOut, cells, grid, off, 0, True, True, ke, mc)
    assert_close(gI, fx["gI"], name + " gI")
    assert_close(gG, fx["gG"], name + " gG")
    bbI, bbG, bbO = ops.backward_backward(_g(fx["cI"]), _g(fx["cG"]), cells, grid, gOut, off, 0, True, True, ke, mc)
    assert_close(bbI, fx["bbI"], name + " bbI")
    assert_close(bbO, fx["bbO"], name + " bbO")
    if d == 3:
        assert_close(bbG, fx["bbG"], name + " bbG")
    for j in range(d):
        cGj, hGj = _g(axis_only(fx["cG"], j)), _g(axis_only(fx["hG"], j))
        bI, bG, bO = ops.backward_backward(None, cGj, cells, grid, gOut, off, 0, True, False, ke, mc)
        assert_close(bI, fx["bbj%d_I" % j], name + " bbj I")
        assert_close(bO, fx["bbj%d_O" % j], name + " bbj O")
        ref = fx["bbj%d_G" % j]
        den = max(float(ref.abs().max()), 1e-30)
        assert float((bG.cpu()[..., j] - ref[..., j]).abs().max()) / den <= 1e-5, name + " bbj G_j"
        tI, tO = ops.bbb_fused(cells, grid, gOut, cGj, hGj, _g(fx["hO"]), off, 0, True, ke, mc)
        assert_close(tI, fx["bbbj%d_I" % j], name + " bbbj I")
        assert_close(tO, fx["bbbj%d_O" % j], name + " bbbj O")


@pytest.mark.parametrize("d", [2, 3])
def test_pixel_pipeline_golden(d):
    """End to end through torch.autograd on the GPU, the way the reference tests use the op."""
    fx = load("pixel_%dd" % d)
    Fn = CosineSampler2d if d == 2 else CosineSampler3d
    got = pixel_pipeline(lambda cells, grid: Fn.apply(cells, grid, "zeros", True, "cosine", True), fx, d, DEV)
    for k in (PIXEL_KEYS_2D if d == 2 else PIXEL_KEYS_3D):
        assert_close(got[k], fx[k], "pixel_%dd %s" % (d, k))
    atol = 1e-5 * float(fx["dloss"].abs().max())
    torch.testing.assert_close(got["dloss"], fx["dloss"], rtol=1e-4, atol=atol)  # test_2d.py:244


@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("shape", [(1, 1, 32, 32, 1024), (4, 3, 17, 29, 5000)])
def test_linear_forward_bit_matches_torch_grid_sample_2d(pad, shape):
    """BASELINE.json configs[0] + north_star: 'linear kernel bit-matches torch.grid_sample'."""
    N, C, H, W, P = shape
    g = torch.Generator().manual_seed(7)
    inp = torch.rand(N, C, H, W, generator=g).to(DEV)
    grid = (torch.rand(N, 1, P, 2, generator=g) * 2.6 - 1.3).to(DEV)
    want = F.grid_sample(inp, grid, mode="bilinear", padding_mode=pad, align_corners=True)
    got = CosineSampler2d.apply(inp, grid, pad, True, "bilinear", False)
    assert torch.equal(got, want), "max diff %g" % float((got - want).abs().max())
    # first backward: same maths, float atomics in both -> tolerance, not bits
    inp.requires_grad_(True)
    grid.requires_grad_(True)
    gOut = torch.randn(N, C, 1, P, generator=g).to(DEV)
    wI, wG = torch.autograd.grad(F.grid_sample(inp, grid, mode="bilinear", padding_mode=pad, align_corners=True),
                                 (inp, grid), gOut)
    gI, gG = torch.autograd.grad(CosineSampler2d.apply(inp, grid, pad, True, "bilinear", False), (inp, grid), gOut)
    assert rel_err(gI, wI) <= 1e-5 and rel_err(gG, wG) <= 1e-5


@pytest.mark.parametrize("pad", ["zeros", "border"])
@pytest.mark.parametrize("align", [True, False])
def test_linear_forward_bit_matches_torch_grid_sample_3d(pad, align):
    N, C, D, H, W, P = 2, 3, 9, 11, 10, 4001
    g = torch.Generator().manual_seed(8)
    inp = torch.rand(N, C, D, H, W, generator=g).to(DEV)
    grid = (torch.rand(N, 1, 1, P, 3, generator=g) * 2.6 - 1.3).to(DEV)
    want = F.grid_sample(inp, grid, mode="bilinear", padding_mode=pad, align_corners=align)
    got = CosineSampler3d.apply(inp, grid, pad, align, "trilinear", False)
    assert torch.equal(got, want), "max diff %g" % float((got - want).abs().max())


@pytest.mark.parametrize("d", [2, 3])
def test_edge_shapes(d):
    sp = (6, 5) if d == 2 else (4, 6, 5)
    Fn = CosineSampler2d if d == 2 else CosineSampler3d
    # empty point set, single point, single channel, far out-of-range points
    for N, C, P in ((2, 3, 0), (1, 1, 1), (2, 1, 65), (1, 7, 256)):
        inp = torch.rand((N, C) + sp, device=DEV, requires_grad=True)
        grid = (torch.rand((N,) + (1,) * (d - 1) + (P, d), device=DEV) * 8 - 4).requires_grad_(True)
        out = Fn.apply(inp, grid, "zeros", True, "cosine", True)
        assert out.shape == (N, C) + (1,) * (d - 1) + (P,)
        gI, gG = torch.autograd.grad(out.sum(), (inp, grid), create_graph=True)
        assert gI.shape == inp.shape and gG.shape == grid.shape
        assert torch.isfinite(out).all() and torch.isfinite(gI).all() and torch.isfinite(gG).all()
        if P:
            t = dict(inp=inp.detach().cpu(), grid=grid.detach().cpu())
            want = cs_oracle.forward(t["inp"], t["grid"], offsets(N, True), 0, True, 0, True)
            assert_close(out, want, "edge N=%d C=%d P=%d" % (N, C, P))
    # non-contiguous / wrong dtype inputs are rejected, not silently copied
    inp = torch.rand((2, 3) + sp, device=DEV)
    grid = torch.rand((2,) + (1,) * (d - 1) + (9, d), device=DEV)
    with pytest.raises(RuntimeError, match="contiguous"):
        Fn.apply(inp.transpose(-1, -2), grid)
    with pytest.raises(RuntimeError, match="float32"):       # the op layer itself is fp32-only ...
        ops.forward(inp.double(), grid.double(), multicell_offset(2, True, DEV), 0, True, 0, True)
    with pytest.raises(RuntimeError, match="floating-point"):   # ... and the autograd layer converts floats only
        Fn.apply(inp.long(), grid)
    with pytest.raises(RuntimeError, match="grid must be"):   # N of the grid: the table's, or 1 (the same points for every n)
        Fn.apply(inp, torch.cat([grid, grid[:1]]))
    assert Fn.apply(inp, grid[:1]).shape == (2, 3) + (1,) * (d - 1) + (9,)


def _full_size_inputs(N=16, C=16, H=256, P=1 << 20):
    # default: BASELINE.json configs[1]: 2D cosine, multicell, N=16 C=16 H=W=256 P=2^20
    torch.manual_seed(0)
    inp = torch.rand(N, C, H, H, device=DEV)
    xy = torch.rand(P, 2, device=DEV) * 2 - 1
    grid = xy.view(1, 1, P, 2).repeat(N, 1, 1, 1).contiguous()
    return inp, grid


@pytest.mark.parametrize("N,C,H,P", [(16, 16, 256, 1 << 20), (96, 4, 16, 100000), (1, 4, 256, 1 << 24), (2, 2, 64, 1 << 22),
                                     (1, 4, 96, (1 << 24) + 1), (2, 12, 128, 1 << 20)])
def test_full_size_2d_properties(N, C, H, P):
    """At full size the oracle is too slow; check identities that do not depend on size:
       partition of unity, linearity, and the adjoint identities linking each stage pair.
       BASELINE config 2 (tile walkers), the reference test scripts' own shapes (crowded tables: wave per cell), the
       largest P the plan's packed keys hold (2^24 points of one table) and one point more (separate cell bytes; the
       reference takes any size through its 64-bit index instantiation, 2d.cu:920-933), a padded 2-channel crowded
       table and a 12-channel table (runs padded to 16)."""
    inp, grid = _full_size_inputs(N, C, H, P)
    off = multicell_offset(N, True, DEV)
    args = (0, True, 0, True)
    # 1. weights sum to one: a constant field samples to the constant (all nodes in range here)
    ones = torch.ones_like(inp)
    out1 = ops.forward(ones, grid, off, *args)
    assert float((out1 - 1).abs().max()) <= 2e-6
    # 2. linearity in input
    inp2 = torch.rand_like(inp)
    a = ops.forward(inp, grid, off, *args)
    b = ops.forward(inp2, grid, off, *args)
    ab = ops.forward(inp * 0.5 + inp2, grid, off, *args)
    assert rel_err(ab, a * 0.5 + b) <= 1e-6
    del ab, b, out1, ones
    # Inner products of 2^24..2^28 signed terms: compare them relative to the sum of |terms|
    # (their condition number), not to the heavily cancelled total.
    def ip(x, y):
        prod = x.double() * y.double()
        return float(prod.sum()), float(prod.abs().sum())

    def same(*pairs):
        vals = [ip(x, y) for x, y in pairs]
        scale = max(v[1] for v in vals)
        return all(abs(v[0] - vals[0][0]) <= 1e-6 * scale for v in vals)

    # 3. <out, gOut> == <input, grad_input>   (backward is the adjoint of forward)
    gOut = torch.randn(N, C, 1, P, device=DEV)
    gI, gG = ops.backward(gOut, inp, grid, off, 0, True, True, 0, True)
    assert same((a, gOut), (inp, gI))
    # 4. conservation: sum of grad_input == sum of gOut (weights sum to one)
    assert abs(float(gI.double().sum()) - float(gOut.double().sum())) <= 1e-6 * float(gOut.double().abs().sum())
    # 5. grad_grid is linear in input: <gG(inp), cG> == <ggOut(cG), gOut> == <gInput_bb(cG), inp>
    cG = torch.randn_like(grid)
    bbI, bbG, bbO = ops.backward_backward(None, cG, inp, grid, gOut, off, 0, True, False, 0, True)
    assert same((gG, cG), (bbO, gOut), (bbI, inp))
    # 6. third order: gGrid is linear in input and in gOut: <bbG, hG> == <k4O, gOut> == <k4I, inp>
    hG = torch.randn_like(grid)
    k4I, k4O = ops.backward_backward_backward(inp, grid, gOut, cG, hG, off, 0, True, True, 0, True)
    assert same((bbG, hG), (k4O, gOut), (k4I, inp))
    # 7. fused third backward == K4 + the gInput of a K3 run on hO
    hO = torch.randn_like(gOut)
    fI, fO = ops.bbb_fused(inp, grid, gOut, cG, hG, hO, off, 0, True, 0, True)
    extra, _, _ = ops.backward_backward(None, cG, inp, grid, hO, off, 0, True, False, 0, True)
    assert rel_err(fI, k4I + extra) <= 1e-5
    assert rel_err(fO, k4O) <= 1e-6
    # 8. the two GPU paths agree at full size (the direct kernels are the ones checked point by
    #    point against the oracle above)
    ops.force_path(1)
    try:
        dI, dG = ops.backward(gOut, inp, grid, off, 0, True, True, 0, True)
        dfI, dfO = ops.bbb_fused(inp, grid, gOut, cG, hG, hO, off, 0, True, 0, True)
    finally:
        ops.force_path(0)
    assert rel_err(gI, dI) <= 1e-5 and rel_err(gG, dG) <= 1e-5
    assert rel_err(fI, dfI) <= 1e-5 and rel_err(fO, dfO) <= 1e-5
    # 9. a slice of every p-ordered full-size result against the CPU oracle (n = 5, first 4096 points): these depend on
    #    the slice's own points only; the input-shaped gradients (sums over all points) rest on 3.-8.
    sl, k = slice(0, min(4096, P)), min(5, N - 1)
    kk = slice(k, k + 1)
    inp_k, grid_s, off_k = inp[kk].cpu(), grid[kk, :, sl].contiguous().cpu(), off[kk].cpu()
    ps = lambda t: t[kk][..., sl].contiguous().cpu()           # (N,C,1,P) streams
    gs = lambda t: t[kk, :, sl].contiguous().cpu()             # (N,1,P,2) grid-shaped
    want = cs_oracle.forward(inp_k, grid_s, off_k, 0, True, 0, True)
    assert_close(a[kk, :, :, sl], want, "full-size slice vs oracle: out")
    _, w_gG = cs_oracle.backward(ps(gOut), inp_k, grid_s, off_k, 0, True, True, 0, True)
    assert_close(gG[kk, :, sl], w_gG, "full-size slice vs oracle: grad_grid")
    _, w_bbG, w_bbO = cs_oracle.backward_backward(None, gs(cG), inp_k, grid_s, ps(gOut), off_k, 0, True, False, 0, True)
    assert_close(bbG[kk, :, sl], w_bbG, "full-size slice vs oracle: second-backward grad_grid")
    assert_close(bbO[kk, :, :, sl], w_bbO, "full-size slice vs oracle: grad_grad_out")
    _, w_fO = cs_oracle.bbb_fused(inp_k, grid_s, ps(gOut), gs(cG), gs(hG), ps(hO), off_k, 0, True, 0, True)
    assert_close(fO[kk, :, :, sl], w_fO, "full-size slice vs oracle: third-backward grad_grad_out")
    # 10. the stages of ONE step sharing a StepContext (prepared table copy and plan, the sorted grad_output copy the
    #     first backward leaves for the walkers of the later stages) give what the independent calls above gave
    sc = ops.StepContext()
    s_out = ops.forward(inp, grid, off, *args, ctx=sc)
    s_gI, s_gG = ops.backward(gOut, inp, grid, off, 0, True, True, 0, True, ctx=sc)
    s_bbI, s_bbG, s_bbO = ops.backward_backward(None, cG, inp, grid, gOut, off, 0, True, False, 0, True, ctx=sc)
    s_fI, s_fO = ops.bbb_fused(inp, grid, gOut, cG, hG, hO, off, 0, True, 0, True, ctx=sc)
    for got, ref, nm in ((s_out, a, "out"), (s_gI, gI, "gI"), (s_gG, gG, "gG"), (s_bbI, bbI, "bbI"), (s_bbG, bbG, "bbG"),
                         (s_bbO, bbO, "bbO"), (s_fI, fI, "fI"), (s_fO, fO, "fO")):
        assert rel_err(got, ref) <= 1e-5, "shared StepContext vs independent calls: %s" % nm


def test_full_size_extensions():
    """BASELINE configs[1] sizes (N=16 C=16 256^2 P=2^20) for the two opt-in input forms: one set of points for every
    table (CS_GRID_BROADCAST) against the repeated grid, and bfloat16 streams against the fp32 op -- same kernels as the
    headline step, so same numbers: per-sample outputs bit for bit, sums to rounding."""
    N, C, H, P = 16, 16, 256, 1 << 20
    g = torch.Generator(device="cpu").manual_seed(99)
    cells = torch.rand(N, C, H, H, generator=g).to(DEV)
    pts = (torch.rand(1, 1, P, 2, generator=g) * 2.1 - 1.05).to(DEV)
    gridN = pts.repeat(N, 1, 1, 1)
    gO = torch.randn(N, C, 1, P, generator=g).to(DEV)
    cG1 = torch.randn(1, 1, P, 2, generator=g).to(DEV)
    hG1 = torch.randn(1, 1, P, 2, generator=g).to(DEV)
    hO = torch.randn(N, C, 1, P, generator=g).to(DEV)
    off = offsets(N, True).to(DEV)
    a, b = ops.StepContext(), ops.StepContext()
    out1 = ops.forward(cells, pts, off, 0, True, 0, True, ctx=a)
    outN = ops.forward(cells, gridN, off, 0, True, 0, True, ctx=b)
    assert torch.equal(out1, outN)
    gI1, gG1 = ops.backward(gO, cells, pts, off, 0, True, True, 0, True, ctx=a)
    gIN, gGN = ops.backward(gO, cells, gridN, off, 0, True, True, 0, True, ctx=b)
    assert gG1.shape == pts.shape
    assert_close(gI1, gIN, "full size, broadcast grid: grad_input", tol=2e-6)
    assert_close(gG1, gGN.sum(0, keepdim=True), "full size, broadcast grid: grad_grid", tol=2e-6)
    t1 = ops.bbb_fused(cells, pts, gO, cG1, hG1, hO, off, 0, True, 0, True, ctx=a)
    tN = ops.bbb_fused(cells, gridN, gO, cG1.repeat(N, 1, 1, 1), hG1.repeat(N, 1, 1, 1), hO, off, 0, True, 0, True, ctx=b)
    assert torch.equal(t1[1], tN[1])
    assert_close(t1[0], tN[0], "full size, broadcast grid: third-backward grad_input", tol=2e-6)
    del a, b, gridN, tN, gIN, gGN
    torch.cuda.empty_cache()
    # bfloat16 streams: the forward rounds the same fp32 sums once; the cotangent is read as bfloat16 exactly
    assert ops.half_streams_ok(cells, pts)
    out16 = ops.forward(cells, pts, off, 0, True, 0, True, out_dtype=torch.bfloat16)
    assert torch.equal(out16, out1.to(torch.bfloat16))
    gO16 = gO.to(torch.bfloat16)
    gI16, gG16 = ops.backward(gO16, cells, pts, off, 0, True, True, 0, True)
    gI32, gG32 = ops.backward(gO16.float(), cells, pts, off, 0, True, True, 0, True)
    assert_close(gI16, gI32, "full size, bfloat16 cotangent: grad_input", tol=2e-6)
    assert_close(gG16, gG32, "full size, bfloat16 cotangent: grad_grid", tol=2e-6)
    torch.cuda.synchronize()


def test_full_size_3d_properties():
    # BASELINE.json configs[3]: 3D smoothstep, N=8 C=8 128^3, P=2^19
    torch.manual_seed(1)
    N, C, S, P = 8, 8, 128, 1 << 19
    inp = torch.rand(N, C, S, S, S, device=DEV)
    grid = (torch.rand(N, 1, 1, P, 3, device=DEV) * 2 - 1)
    off = multicell_offset(N, True, DEV)
    out = ops.forward(inp, grid, off, 0, True, 2, True)
    gOut = torch.randn_like(out)
    gI, gG = ops.backward(gOut, inp, grid, off, 0, True, True, 2, True)
    def ip(x, y):
        prod = x.double() * y.double()
        return float(prod.sum()), float(prod.abs().sum())

    (l, la), (r, ra) = ip(out, gOut), ip(inp, gI)
    assert abs(l - r) <= 1e-6 * max(la, ra)
    cG = torch.randn_like(grid)
    bbI, bbG, bbO = ops.backward_backward(None, cG, inp, grid, gOut, off, 0, True, False, 2, True)
    vals = [ip(gG, cG), ip(bbO, gOut), ip(bbI, inp)]
    scale = max(v[1] for v in vals)
    assert all(abs(v[0] - vals[0][0]) <= 1e-6 * scale for v in vals)
    same = lambda vals: all(abs(v[0] - vals[0][0]) <= 1e-6 * max(w[1] for w in vals) for v in vals)
    # third order at full size (3d.cu:875-1071): gGrid is linear in input and in gOut.  K8 keeps the pure second
    # derivatives only while K7's gGrid has the mixed ones too, so the adjoint identity holds for
    # axis-aligned cotangents (one non-zero component, same axis) -- what grad(u_xx, cells) produces
    from helpers import axis_only
    cGa, hG = axis_only(cG, 1), torch.randn_like(grid)
    hGa = axis_only(hG, 1)
    _, bbGa, _ = ops.backward_backward(None, cGa, inp, grid, gOut, off, 0, True, False, 2, True, want_grad_input=False)
    k4I, k4O = ops.backward_backward_backward(inp, grid, gOut, cGa, hGa, off, 0, True, True, 2, True)
    assert same([ip(bbGa, hGa), ip(k4O, gOut), ip(k4I, inp)])
    del bbGa
    k4I, k4O = ops.backward_backward_backward(inp, grid, gOut, cG, hG, off, 0, True, True, 2, True)
    # ... and the fused third backward is K4 plus the grad_input of a second backward run on hO (modules_3d.py:95-100)
    hO = torch.randn_like(gOut)
    fI, fO = ops.bbb_fused(inp, grid, gOut, cG, hG, hO, off, 0, True, 2, True)
    extra, _, _ = ops.backward_backward(None, cG, inp, grid, hO, off, 0, True, False, 2, True)
    assert rel_err(fI, k4I + extra) <= 1e-5
    assert rel_err(fO, k4O) <= 1e-6
    del extra, k4I, k4O
    # slices of every p-ordered result against the CPU oracle (table 3, first 2048 points)
    sl, kk = slice(0, 2048), slice(3, 4)
    inp_k, grid_s, off_k = inp[kk].cpu(), grid[kk, :, :, sl].contiguous().cpu(), off[kk].cpu()
    ps = lambda t: t[kk][..., sl].contiguous().cpu()
    gs = lambda t: t[kk, :, :, sl].contiguous().cpu()
    want = cs_oracle.forward(inp_k, grid_s, off_k, 0, True, 2, True)
    assert_close(out[kk, :, :, :, sl], want, "3D full-size slice vs oracle: out")
    _, w_gG = cs_oracle.backward(ps(gOut), inp_k, grid_s, off_k, 0, True, True, 2, True)
    assert_close(gG[kk, :, :, sl], w_gG, "3D full-size slice vs oracle: grad_grid")
    _, w_bbG, w_bbO = cs_oracle.backward_backward(None, gs(cG), inp_k, grid_s, ps(gOut), off_k, 0, True, False, 2, True)
    assert_close(bbG[kk, :, :, sl], w_bbG, "3D full-size slice vs oracle: second-backward grad_grid")
    assert_close(bbO[kk, :, :, :, sl], w_bbO, "3D full-size slice vs oracle: grad_grad_out")
    _, w_fO = cs_oracle.bbb_fused(inp_k, grid_s, ps(gOut), gs(cG), gs(hG), ps(hO), off_k, 0, True, 2, True)
    assert_close(fO[kk, :, :, :, sl], w_fO, "3D full-size slice vs oracle: third-backward grad_grad_out")


def test_rccl_path_runs_on_one_gpu():
    """The multi-GPU job's communication path -- init_process_group("nccl") (= RCCL), the asynchronous per-stage
    all-reduces of GradReducer, the barrier and MAX-reduce of bench.py's timing -- executed for real in fresh child
    processes on the one GPU a test box has (a one-rank group): tests/rccl_alone_child.py checks the numbers,
    `bench.py --rccl-alone` the distributed branch of the benchmark and its extra fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_alone_child.py")], capture_output=True,
                       text=True, timeout=300, env=dict(env, MASTER_PORT="29533"))
    assert r.returncode == 0 and "RCCL_ALONE_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rccl-alone", "--points", "131072", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-helmholtz"], capture_output=True, text=True,
                       timeout=300, env=dict(env, MASTER_PORT="29534"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["ms_per_step_no_reduce"] > 0 and "allreduce_ms" in line and "RCCL" in line["reduce"]
    # the default schedule: the stages add into one step accumulator, ONE all-reduce per step on its sum
    assert line["reduce_schedule"] == "once" and "ONE RCCL all-reduce" in line["reduce"]
    # the other schedule: the three gradients kept apart, each reduced asynchronously as soon as its stage is enqueued
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rccl-alone", "--reduce", "per_stage", "--points", "131072",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-helmholtz"], capture_output=True,
                       text=True, timeout=300, env=dict(env, MASTER_PORT="29535"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["reduce_schedule"] == "per_stage" and "3 RCCL all-reduces" in line["reduce"] and line["value"] > 0


def test_integration_stub_from_the_document():
    """INTEGRATION.md, Option A: the ctypes stand-in for the reference's pybind module `_cosine_2d` -- the code block is
    taken from the document as it stands, executed, and its four functions (the reference's names and argument order,
    2d.cpp:130-135) are compared with cosinesampler_amd.ops on the GPU."""
    import re
    import types
    from cosinesampler_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# cosine_sampler_2d/_cosine_2d\.py.*?)```", text, re.S)
    assert m, "Option A code block not found"
    os.environ["COSINESAMPLER_LIB"] = build.LIB
    try:
        stub = types.ModuleType("_cosine_2d")
        exec(compile(m.group(1), "INTEGRATION.md:_cosine_2d.py", "exec"), stub.__dict__)
    finally:
        del os.environ["COSINESAMPLER_LIB"]
    N, C, P = 3, 16, 40000
    t = _case(2, N, C, (48, 40), P, seed=99, spread=1.1)
    off = offsets(N, True).to(DEV)
    inp, grid, gO, cG, hG, cI = (_g(t[k]) for k in ("inp", "grid", "gOut", "cG", "hG", "cI"))
    out = stub.forward(inp, grid, off, 0, True, 0, True)
    assert torch.equal(out, ops.forward(inp, grid, off, 0, True, 0, True))
    gi, gg = stub.backward(gO, inp, grid, off, 0, True, True, 0, True)
    w_gi, w_gg = ops.backward(gO, inp, grid, off, 0, True, True, 0, True)
    assert_close(gi, w_gi, "stub backward grad_input")
    assert torch.equal(gg, w_gg)
    assert stub.backward(gO, inp, grid, off, 0, True, False, 0, True)[0] is None
    for irg in (True, False):
        got = stub.backward_backward(cI if irg else torch.zeros(1), cG, inp, grid, gO, off, 0, True, irg, 0, True)
        want = ops.backward_backward(cI if irg else None, cG, inp, grid, gO, off, 0, True, irg, 0, True)
        for a, b, nm in zip(got, want, ("gInput", "gGrid", "ggOut")):
            assert_close(a, b, "stub backward_backward(%s) %s" % (irg, nm))
    got = stub.backward_backward_backward(inp, grid, gO, cG, hG, off, 0, True, True, 0, True)
    want = ops.backward_backward_backward(inp, grid, gO, cG, hG, off, 0, True, True, 0, True)
    assert_close(got[0], want[0], "stub third backward gInput")
    assert_close(got[1], want[1], "stub third backward ggOut")



@pytest.mark.parametrize("N,C,D,H,W", [(2, 8, 19, 32, 16), (1, 4, 5, 32, 16), (3, 16, 17, 16, 16), (2, 6, 33, 32, 32),
                                       (1, 8, 16, 64, 64)])
def test_column_wise_z_paired_pack_is_the_layout(N, C, D, H, W):
    """cs_pack_input for 3D tables: the column-wise kernel (one read of the table, round 3) against the layout it must
    produce, built with torch -- node v holds [its own channel row | the row of the node one z-plane above, zeros past the
    last plane], channels padded with zeros (cs_kernels_direct.cuh, pack_cl4) -- and against the two-reads kernel."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    inp = torch.rand(N, C, D, H, W, generator=g).to(DEV)
    CP = min(c for c in (4, 8, 16, 32) if c >= C)          # cs_abi.hip cpad
    nbytes = N * D * H * W * 2 * CP * 4
    assert lib.cs_pack_bytes(3, N, C, D, H, W, 1 << 18) >= nbytes
    cl = torch.zeros(N, D, H, W, 2, CP, device=DEV)
    cl[..., 0, :C] = inp.permute(0, 2, 3, 4, 1)
    cl[:, :-1, :, :, 1, :C] = inp.permute(0, 2, 3, 4, 1)[:, 1:]
    got = {}
    try:
        for mode in (2, 6):
            ops.force_path(mode)
            buf = torch.full((nbytes,), 0xAB, dtype=torch.uint8, device=DEV)
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.cs_pack_input(3, inp.data_ptr(), buf.data_ptr(), N, C, D, H, W, st), "cs_pack_input")
            got[mode] = buf.view(torch.float32).view(N, D, H, W, 2, CP).clone()
    finally:
        ops.force_path(0)
    assert torch.equal(got[2], cl), "column-wise pack differs from the z-paired layout"
    assert torch.equal(got[6], cl), "two-reads pack differs from the z-paired layout"
