"""CPU: the Python host side (autograd Function chain, enums, offsets, argument checks) with the
kernels replaced by the CPU oracle (tests/oracle_backend.py).  Compared against the golden vectors
from the reference's ground truth."""
import pytest
import torch

import oracle_backend
from cosinesampler_amd import CosineSampler2d, CosineSampler3d, kernel_enum, multicell_offset, padding_mode_enum
from helpers import (PIXEL_KEYS_2D, PIXEL_KEYS_3D, assert_close, axis_only, load, parse_stage_name, pixel_pipeline,
                     stage_fixtures)


def test_enums_follow_reference():
    # reference modules_2d.py:4-18, modules_3d.py:4-18
    assert [padding_mode_enum(s) for s in ("zeros", "border", "reflection", "anything")] == [0, 1, 2, 2]
    assert [kernel_enum(s) for s in ("cosine", "bilinear", "trilinear", "smooth-step")] == [0, 1, 1, 2]
    assert kernel_enum("cubic") is None


def test_mixed_suffix_is_an_extension_of_the_kernel_names():
    """'+mixed' (not in the reference) ORs CS_KERNEL_EXACT_MIXED into the enum; every reference name is unchanged."""
    from cosinesampler_amd import ops
    assert ops.EXACT_MIXED == 0x100
    for name, base in (("cosine", 0), ("bilinear", 1), ("trilinear", 1), ("smooth-step", 2)):
        assert kernel_enum(name) == base
        assert kernel_enum(name + "+mixed") == base | ops.EXACT_MIXED
    assert kernel_enum("cosine+exact") is None and kernel_enum("mixed") is None


def test_expanded_cotangents_are_left_alone():
    """ops.keep_expanded: an n-expanded view over one contiguous block stays a view (the kernels take its stride);
    every other non-contiguous layout is made contiguous, as the reference's .contiguous() would."""
    from cosinesampler_amd import ops
    base = torch.rand(1, 4, 1, 50)
    e = base.expand(6, 4, 1, 50)
    assert ops.keep_expanded(e) is e and e.stride(0) == 0
    c = torch.rand(6, 4, 1, 50)
    assert ops.keep_expanded(c) is c
    t = torch.rand(4, 6, 1, 50).transpose(0, 1)                  # channel-strided: not supported in place
    assert ops.keep_expanded(t).is_contiguous() and torch.equal(ops.keep_expanded(t), t)
    inner = torch.rand(1, 4, 1, 100)[..., ::2].expand(6, 4, 1, 50)   # expanded, but the block itself is strided
    assert ops.keep_expanded(inner).is_contiguous()
    assert ops.keep_expanded(None) is None


def test_offset_bits_match_reference_construction():
    for N in (1, 3, 16, 96):
        assert torch.equal(multicell_offset(N, True, "cpu"), torch.linspace(0, 1 - (1 / N), N))
        assert torch.equal(multicell_offset(N, False, "cpu"), torch.zeros(N))


def test_cpu_tensors_are_rejected_like_check_cuda():
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        CosineSampler2d.apply(torch.rand(1, 1, 4, 4), torch.rand(1, 1, 3, 2))


def test_unknown_kernel_is_a_type_error(monkeypatch):
    oracle_backend.install(monkeypatch)
    with pytest.raises(TypeError):
        CosineSampler2d.apply(torch.rand(1, 1, 4, 4), torch.rand(1, 1, 3, 2), "zeros", True, "cubic", True)


@pytest.mark.parametrize("d", [2, 3])
def test_pixel_pipeline_through_function_chain(monkeypatch, d):
    oracle_backend.install(monkeypatch)
    fx = load("pixel_%dd" % d)
    Fn = CosineSampler2d if d == 2 else CosineSampler3d
    got = pixel_pipeline(lambda cells, grid: Fn.apply(cells, grid, "zeros", True, "cosine", True), fx, d)
    for k in (PIXEL_KEYS_2D if d == 2 else PIXEL_KEYS_3D):
        assert_close(got[k], fx[k], "pixel_%dd %s" % (d, k))
    # the reference's own (only) assertion is elementwise rtol 1e-4 on d loss / d cells
    # (test_2d.py:244) at 100 000 points; with this fixture's 300 points a few cells see almost no
    # samples, so the elementwise check gets an absolute floor of 1e-5 of the tensor's max.
    atol = 1e-5 * float(fx["dloss"].abs().max())
    torch.testing.assert_close(got["dloss"], fx["dloss"], rtol=1e-4, atol=atol)


@pytest.mark.parametrize("name", stage_fixtures())
def test_stage_chain_by_autograd(monkeypatch, name):
    """Drive the three Function levels with explicit cotangents, as tests/golden/make_golden.py
    drove the reference ground truth."""
    oracle_backend.install(monkeypatch)
    d, kernel, mc = parse_stage_name(name)
    fx = load(name)
    Fn = CosineSampler2d if d == 2 else CosineSampler3d
    cells = fx["cells"].clone().requires_grad_(True)
    grid = fx["grid"].clone().requires_grad_(True)
    gOut = fx["gOut"].clone().requires_grad_(True)
    out = Fn.apply(cells, grid, "zeros", True, kernel, mc)
    assert_close(out, fx["out"], name + " out")
    gI, gG = torch.autograd.grad(out, (cells, grid), gOut, create_graph=True)
    assert_close(gI, fx["gI"], name + " gI")
    assert_close(gG, fx["gG"], name + " gG")
    s = (gI * fx["cI"]).sum() + (gG * fx["cG"]).sum()
    bbI, bbG, bbO = torch.autograd.grad(s, (cells, grid, gOut), retain_graph=True)
    assert_close(bbI, fx["bbI"], name + " bbI")
    assert_close(bbO, fx["bbO"], name + " bbO")
    for j in range(d):
        sj = (gG * axis_only(fx["cG"], j)).sum()      # cI absent -> None -> null pointer path
        bI, bG, bO = torch.autograd.grad(sj, (cells, grid, gOut), create_graph=True)
        assert_close(bI, fx["bbj%d_I" % j], name + " bbj I")
        assert_close(bO, fx["bbj%d_O" % j], name + " bbj O")
        s3 = (bG * axis_only(fx["hG"], j)).sum() + (bO * fx["hO"]).sum()
        tI, tG, tO = torch.autograd.grad(s3, (cells, grid, gOut), retain_graph=True, allow_unused=True)
        assert tG is None                               # third order has no d/dgrid (modules_2d.py:111)
        assert_close(tI, fx["bbbj%d_I" % j], name + " bbbj I")
        assert_close(tO, fx["bbbj%d_O" % j], name + " bbbj O")


def test_grad_input_skipped_when_input_does_not_require_grad(monkeypatch):
    oracle_backend.install(monkeypatch)
    cells = torch.rand(2, 2, 6, 6)
    grid = (torch.rand(2, 1, 20, 2) * 2 - 1).requires_grad_(True)
    out = CosineSampler2d.apply(cells, grid)
    (gG,) = torch.autograd.grad(out.sum(), grid)
    assert gG.shape == grid.shape


def test_unused_table_gradients_are_not_computed(monkeypatch):
    """PINN pattern: u_x = grad(u, x, create_graph=True), u_xx = grad(u_x, x, create_graph=True), loss.backward().
    Only loss.backward() uses d/d cells; the derivative calls must tell the op layer so (the reference computes
    and drops them), and the final gradient must be what it is when nothing is skipped."""
    oracle_backend.install(monkeypatch)
    from cosinesampler_amd import ops
    calls = []
    real_backward = ops.backward

    def spy_backward(gO, inp, grid, off, pad, align, input_requires_grad, kern, mc, ctx=None, go_owner=None):
        calls.append(bool(input_requires_grad))
        return real_backward(gO, inp, grid, off, pad, align, input_requires_grad, kern, mc, ctx=ctx)

    monkeypatch.setattr(ops, "backward", spy_backward)
    torch.manual_seed(3)
    cells0 = torch.rand(2, 2, 6, 6)
    x0 = torch.rand(20, 1) * 1.8 - 0.9
    y0 = torch.rand(20, 1) * 1.8 - 0.9

    def step(leaf):
        cells = cells0.clone().requires_grad_(True)
        table = cells if leaf else cells * 1.0
        x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
        grid = torch.cat([x, y], -1).view(1, 1, -1, 2).repeat(2, 1, 1, 1)
        u = CosineSampler2d.apply(table, grid, "zeros", True, "cosine", True).sum((0, 1)).view(-1, 1)
        del calls[:], oracle_backend.SKIPPED[:]
        (u_x,) = torch.autograd.grad(u.sum(), x, create_graph=True)
        assert calls == [False]                       # d/d cells of the first backward: dropped by the engine
        (u_xx,) = torch.autograd.grad(u_x.sum(), x, create_graph=True)
        assert oracle_backend.SKIPPED == [True]       # likewise for the second backward
        loss = ((u_xx + u) ** 2).mean()
        loss.backward()
        assert calls[1:] and all(calls[1:]) and not any(oracle_backend.SKIPPED[1:])   # now they are wanted
        return cells.grad.clone()

    g_leaf, g_nonleaf = step(True), step(False)
    monkeypatch.setattr("cosinesampler_amd.functions._engine_wants", lambda ctx, i: ctx.needs_input_grad[i])
    cells = cells0.clone().requires_grad_(True)
    x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
    grid = torch.cat([x, y], -1).view(1, 1, -1, 2).repeat(2, 1, 1, 1)
    u = CosineSampler2d.apply(cells, grid, "zeros", True, "cosine", True).sum((0, 1)).view(-1, 1)
    (u_x,) = torch.autograd.grad(u.sum(), x, create_graph=True)
    (u_xx,) = torch.autograd.grad(u_x.sum(), x, create_graph=True)
    ((u_xx + u) ** 2).mean().backward()
    assert torch.equal(g_leaf, cells.grad) and torch.equal(g_nonleaf, cells.grad)


def test_table_gradient_is_computed_when_asked_for_directly(monkeypatch):
    oracle_backend.install(monkeypatch)
    cells = torch.rand(2, 2, 6, 6, requires_grad=True)
    grid = (torch.rand(2, 1, 20, 2) * 1.8 - 0.9).requires_grad_(True)
    out = CosineSampler2d.apply(cells, grid)
    gI, gG = torch.autograd.grad(out.sum(), [cells, grid], create_graph=True)
    assert gI is not None and gI.shape == cells.shape
    (gI2,) = torch.autograd.grad(gG.sum(), cells)      # second order w.r.t. the table only
    assert gI2.shape == cells.shape and float(gI2.abs().sum()) > 0


@pytest.mark.parametrize("dtype", [torch.float64, torch.float16])
def test_other_float_dtypes_convert_at_the_boundary(monkeypatch, dtype):
    """The reference dispatches double/float/half (2d.cu:905); here they are served through fp32."""
    oracle_backend.install(monkeypatch)
    g = torch.Generator().manual_seed(1)
    cells32 = torch.rand(2, 3, 8, 8, generator=g)
    grid32 = torch.rand(2, 1, 40, 2, generator=g) * 2 - 1
    ref = CosineSampler2d.apply(cells32, grid32)
    cells = cells32.to(dtype).requires_grad_(True)
    grid = grid32.to(dtype).requires_grad_(True)
    out = CosineSampler2d.apply(cells, grid)
    assert out.dtype == dtype
    tol = 1e-6 if dtype == torch.float64 else 2e-3
    assert float((out.detach().float() - ref).abs().max()) <= tol * float(ref.abs().max())
    gI, gG = torch.autograd.grad(out.sum(), (cells, grid), create_graph=True)
    assert gI.dtype == dtype and gG.dtype == dtype
    (gg,) = torch.autograd.grad(gG[..., 0].sum(), cells)
    assert gg.dtype == dtype and gg.shape == cells.shape


def test_drop_in_package_names():
    import cosine_sampler_2d
    import cosine_sampler_3d
    assert cosine_sampler_2d.CosineSampler2d is CosineSampler2d
    assert cosine_sampler_3d.CosineSampler3d is CosineSampler3d


@pytest.mark.parametrize("d", [2, 3])
def test_broadcast_grid_through_the_function_chain(monkeypatch, d):
    """A (1, ..., dim) grid = the same points for every n (PIXEL's grid.repeat(N, 1, 1, 1), reference test/test_2d.py:38,
    without the repeat): the Function chain must give what the repeated grid gives, gradients w.r.t. the points summed
    over n -- here with the oracle as the backend, on the GPU in tests/test_parity_gpu.py."""
    oracle_backend.install(monkeypatch)
    from cosinesampler_amd import CosineSampler3d
    Fn = CosineSampler2d if d == 2 else CosineSampler3d
    torch.manual_seed(5 + d)
    N, C, P = 3, 2, 40
    cells0 = torch.rand((N, C) + (6,) * d)
    pts0 = torch.rand((1,) + (1,) * (d - 1) + (P, d)) * 1.8 - 0.9
    res = []
    for bc in (True, False):
        cells = cells0.clone().requires_grad_(True)
        pts = pts0.clone().requires_grad_(True)
        grid = pts if bc else pts.repeat((N,) + (1,) * (d + 1))
        out = Fn.apply(cells, grid, "zeros", True, "cosine", True)
        assert out.shape == (N, C) + (1,) * (d - 1) + (P,)
        u = torch.tanh(out.sum(0)).sum(0)
        (u_g,) = torch.autograd.grad(u.sum(), pts, create_graph=True)
        assert u_g.shape == pts.shape
        (u_gg,) = torch.autograd.grad(u_g[..., 0].sum(), pts, create_graph=True)
        loss = (u_gg[..., 0] ** 2).mean() + (u ** 2).mean()
        (gc,) = torch.autograd.grad(loss, cells)
        res.append((out.detach(), u_g.detach(), u_gg.detach(), gc))
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)


def test_held_identity_is_the_object_not_the_address():
    """ops._Held (round 3): what a cached plan / channels-last copy / sorted grad_output copy is remembered by.  Equal
    address + shape + version is NOT an identity once the tensor is gone; while it is held the address cannot be re-used,
    and an alias of the held memory (a view with the same layout) is the same bytes."""
    from cosinesampler_amd import ops
    a = torch.rand(4, 8, 1, 50)
    h = ops._Held(a)
    assert h.same(a)
    assert h.same(a.view(4, 8, 1, 50))                    # an alias of the very same memory
    assert not h.same(a.clone())                          # other memory
    assert not h.same(a[:2])                              # other shape
    a.add_(1.0)                                           # in-place update: another version
    assert not h.same(a)
    assert ops._Held(None).same(None) and not ops._Held(None).same(a) and not h.same(None)
    # the failure the strong reference prevents: free the tensor, allocate an equal one -- the allocator may hand out the
    # same block with version 0 again; a _Held of the FIRST tensor still says no (it holds it, so the block is not re-used)
    b = torch.rand(4, 8, 1, 50)
    hb = ops._Held(b)
    ptr = b.data_ptr()
    del b
    c = torch.rand(4, 8, 1, 50)
    assert c.data_ptr() != ptr and not hb.same(c)


def test_channel_groups_cover_every_channel_once():
    """ops._channel_groups: tables wider than the fast paths run as ranges of at most 32 (2D) / 16 (3D) channels."""
    from cosinesampler_amd import ops
    for dim, cap in ((2, 32), (3, 16)):
        for C in (1, cap - 1, cap):
            assert ops._channel_groups(torch.empty((2, C) + (4,) * dim), dim) is None
        for C in (cap + 1, 2 * cap, 2 * cap + 7, 5 * cap):
            groups = ops._channel_groups(torch.empty((2, C) + (4,) * dim), dim)
            assert groups[0][0] == 0 and groups[-1][1] == C
            assert all(b - a <= cap and b > a for a, b in groups)
            assert all(groups[i][1] == groups[i + 1][0] for i in range(len(groups) - 1))


def test_order_and_cache_knobs_validate_their_arguments():
    from cosinesampler_amd import ops
    assert ops.points_order() in ("auto", "coherent", "random")
    with pytest.raises(ValueError):
        ops.points_order("sorted")
    try:
        assert ops.points_order("random") == "random" and ops.points_order() == "random"
    finally:
        ops.points_order("auto")
    assert ops.plan_cache() == 0
    try:
        assert ops.plan_cache(3) == 3 and ops.plan_cache(-5) == 0
    finally:
        ops.plan_cache(0)


def test_grad_reducer_sums_locally_without_a_process_group():
    """dist.GradReducer with no process group: both schedules only add the pushed gradients up -- into `out` without a
    copy pass when there are two or more -- and an empty reducer zeroes `out`."""
    from cosinesampler_amd.dist import GradReducer
    g = [torch.rand(3, 5) for _ in range(3)]
    for schedule in ("per_stage", "once"):
        for k in (1, 2, 3):
            r = GradReducer(schedule=schedule)
            for t in g[:k]:
                r.push(t)
            out = torch.full((3, 5), 7.0)
            tot = r.finish(out=out)
            assert tot is out and torch.allclose(out, sum(g[:k]))
            r2 = GradReducer(schedule=schedule)
            for t in g[:k]:
                r2.push(t)
            assert torch.allclose(r2.finish(), sum(g[:k]))
        assert torch.equal(GradReducer(schedule=schedule).finish(out=torch.ones(2)), torch.zeros(2))
    with pytest.raises(ValueError):
        GradReducer(schedule="sometimes")


@pytest.mark.parametrize("d", [2, 3])
def test_summed_op_equals_the_repeat_sum_pattern_through_autograd(monkeypatch, d):
    """CosineSampler{2,3}dSum.apply(cells, points) = CosineSampler{2,3}d.apply(cells, points.repeat(N, ...)).sum(0, keepdim=True)
    (reference test/test_2d.py:38, :51 as one op), and so are its first, second and third derivatives -- the Function chain
    with the oracle as the backend; the kernels behind it are compared with the oracle on the GPU (tests/test_coherent_gpu.py)."""
    oracle_backend.install(monkeypatch)
    from cosinesampler_amd import CosineSampler2dSum, CosineSampler3d, CosineSampler3dSum
    Plain, Summed = (CosineSampler2d, CosineSampler2dSum) if d == 2 else (CosineSampler3d, CosineSampler3dSum)
    torch.manual_seed(11 + d)
    N, C, P = 3, 2, 40
    cells0 = torch.rand((N, C) + (6,) * d)
    pts0 = torch.rand((1,) + (1,) * (d - 1) + (P, d)) * 1.8 - 0.9
    W = torch.rand(C, 1)
    res = []
    for summed in (True, False):
        cells = cells0.clone().requires_grad_(True)
        pts = pts0.clone().requires_grad_(True)
        if summed:
            feat = Summed.apply(cells, pts, "zeros", True, "cosine", True)
            assert feat.shape == (1, C) + (1,) * (d - 1) + (P,)
        else:
            feat = Plain.apply(cells, pts.repeat((N,) + (1,) * (d + 1)), "zeros", True, "cosine", True).sum(0, keepdim=True)
        u = torch.tanh(feat.reshape(C, P).t() @ W)
        (u_g,) = torch.autograd.grad(u.sum(), pts, create_graph=True)
        (u_gg,) = torch.autograd.grad(u_g[..., 0].sum(), pts, create_graph=True)
        loss = ((u_gg[..., 0] + u_gg[..., 1]) ** 2).mean() + (u ** 2).mean()
        (gc,) = torch.autograd.grad(loss, cells)
        res.append((feat.detach(), u_g.detach(), u_gg.detach(), gc))
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)
