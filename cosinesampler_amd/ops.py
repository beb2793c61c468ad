"""Stage-level ops on torch CUDA tensors -> C ABI -> HIP kernels.

Mirror of the reference's pybind module `_cosine_2d` / `_cosine_3d`
(reference cosine_sampler_2d/csrc/cosine_sampler_2d.cpp:47-135, cosine_sampler_3d/csrc/cosine_sampler_3d.cpp:50-138):
same four functions, same argument order and meaning, one module for both dimensionalities
(dispatch on input.dim()), plus `bbb_fused` = everything CosineSamplerBackwardBackward.backward
does (reference modules_2d.py:98-111) in one launch.

PyTorch is used for device memory and the current stream only.  Differences from the reference,
all deliberate: outputs are torch.empty (the kernels define every element, nothing relies on
zeros_like); shapes/dtypes are validated (the reference has no checks beyond CUDA+contiguous);
no host synchronisation anywhere.
"""
import torch

from . import _lib


EXACT_MIXED = 0x100   # CS_KERNEL_EXACT_MIXED (include/cosine_sampler.h): keep the mixed second derivatives


def _check(t, name, allow_none=False):
    if t is None:
        if allow_none:
            return
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)         # reference 2d.cpp:4
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)            # reference 2d.cpp:5
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be float32 (got %s): only the fp32 path is built" % (name, t.dtype))


STREAM_DTYPES = {torch.float32: 0, torch.float16: _lib.STREAM_F16, torch.bfloat16: _lib.STREAM_BF16}


def half_streams_ok(input, grid):
    """Can the channel-major streams of this problem be float16 / bfloat16 natively (CS_STREAM_F16 / CS_STREAM_BF16)?
    Only the fast paths take them; elsewhere the caller converts."""
    if not (input.is_cuda and grid.is_cuda):
        return False
    dim, shape, P = _problem(input, grid)
    D = shape[2] if dim == 3 else 1
    return bool(_lib.load().cs_half_streams_supported(dim, shape[0], shape[1], D, shape[-2], shape[-1], P))


def _check_stream(t, name):
    """A channel-major stream (N,C,[Do,]Ho,Wo), fp32 / float16 / bfloat16: contiguous, or -- beyond the reference,
    which insists on contiguous (2d.cpp:5) -- expanded along n (stride 0) over one contiguous (C,[Do,]Ho,Wo) block, which
    is what the backward of PIXEL's `features.sum(0)` hands over.  -> the n-stride in elements."""
    if t is None or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if t.dtype not in STREAM_DTYPES:
        raise RuntimeError("%s must be float32, float16 or bfloat16 (got %s)" % (name, t.dtype))
    if t.is_contiguous():
        return t[0].numel() if t.shape[0] else 0
    if t.dim() >= 2 and t.stride(0) == 0 and t[0].is_contiguous():
        return 0
    if t.dim() >= 2 and t.shape[0] and t[0].is_contiguous() and t.stride(0) >= t[0].numel():
        return t.stride(0)        # a channel range of a larger stream (channel groups below): the kernels take any n-stride
    raise RuntimeError("%s must be contiguous (or expanded along n only)" % name)


def keep_expanded(t):
    """What the autograd layer does to an incoming cotangent instead of `.contiguous()`: leave an n-expanded
    tensor alone (the kernels take its stride), make anything else contiguous."""
    if t is None or t.is_contiguous():
        return t
    if t.dim() >= 2 and t.shape[0] > 1 and t.stride(0) == 0 and t[0].is_contiguous():
        return t
    return t.contiguous()


def _al(*ts):
    """The C ABI wants 16-byte aligned tensors (include/cosine_sampler.h, Alignment); a contiguous view can start anywhere
    in its storage, so such a tensor is copied to fresh memory here (the reference takes any contiguous tensor)."""
    out = tuple(t.clone(memory_format=torch.contiguous_format) if t is not None and t.is_cuda and t.data_ptr() % 16 else t
                for t in ts)
    return out[0] if len(out) == 1 else out


def _ptr(t):
    return None if t is None else t.data_ptr()


def _problem(input, grid):
    _check(input, "input")
    _check(grid, "grid")
    dim = input.dim() - 2
    if dim not in (2, 3):
        raise RuntimeError("input must be (N,C,H,W) or (N,C,D,H,W), got %s" % (tuple(input.shape),))
    if grid.dim() != dim + 2 or grid.shape[-1] != dim or grid.shape[0] not in (1, input.shape[0]):
        raise RuntimeError("grid must be (N,%s%d) with N=%d (or N=1: one set of points for every n), got %s"
                           % ("Ho,Wo," if dim == 2 else "Do,Ho,Wo,", dim, input.shape[0], tuple(grid.shape)))
    if grid.device != input.device:
        raise RuntimeError("input and grid must be on the same device")
    P = 1
    for s in grid.shape[1:-1]:
        P *= int(s)
    return dim, [int(s) for s in input.shape], P


def grid_is_broadcast(input, grid):
    """A (1, ..., dim) grid with N > 1 tables: the same points for every n (CS_GRID_BROADCAST) -- what PIXEL builds with
    grid.repeat(N, 1, 1, 1) (reference test/test_2d.py:38), without the repeat.  Everything grid-shaped then has a
    leading 1: grad_out_grid / grad_out_ggrid as given, and the grad_grid results, which are summed over n (the
    gradient w.r.t. the shared points, i.e. what autograd makes of the repeat)."""
    return grid.shape[0] == 1 and input.shape[0] > 1


def _grid_result(grid, N, bc):
    """where the kernels write a per-point result: always one row per (n, p)"""
    return torch.empty((N,) + tuple(grid.shape[1:]), dtype=grid.dtype, device=grid.device) if bc else torch.empty_like(grid)


def _grid_reduce(t, bc):
    return t.sum(0, keepdim=True) if bc else t


def _same(t, like_shape, name, device, stream=False):
    """-> n-stride in elements when `stream` (an n-expanded tensor is allowed), else None."""
    ns = _check_stream(t, name) if stream else _check(t, name)
    _shape_dev(t, like_shape, name, device)
    return ns


def _shape_dev(t, like_shape, name, device):
    if tuple(t.shape) != tuple(like_shape):
        raise RuntimeError("%s must have shape %s, got %s" % (name, tuple(like_shape), tuple(t.shape)))
    if t.device != device:
        raise RuntimeError("%s must be on %s" % (name, device))


def _offset_ok(offset, N, device):
    _check(offset, "offset")
    if offset.numel() != N or offset.device != device:
        raise RuntimeError("offset must hold N=%d floats on %s" % (N, device))


class _Held(object):
    """Identity of a tensor's bytes for as long as we HOLD the tensor: the object itself (a strong reference, so that its
    storage cannot be freed and handed to another tensor with the same address, shape and version 0 -- the caching
    allocator does exactly that with same-sized temporaries) plus its version counter (in-place updates)."""
    __slots__ = ("t", "version")

    def __init__(self, t):
        self.t = t
        self.version = None if t is None else t._version

    def same(self, t):
        if t is None or self.t is None:
            return t is self.t
        # while self.t is held its storage cannot be re-used, so an equal address means the same memory (an alias)
        return (t is self.t or (t.data_ptr() == self.t.data_ptr() and t.shape == self.t.shape and t.dtype == self.t.dtype
                                and t.stride() == self.t.stride())) and t._version == self.version


class _PlanEntry(object):
    """A built point plan: its buffer, the grid it was made from (held, so that an equal address means the same memory),
    the configuration it was built for, and whose cell-sorted grad_output copy it currently holds."""
    __slots__ = ("buf", "of", "off", "cfg", "sorted_go")

    def __init__(self, buf, of, off, cfg):
        self.buf, self.of, self.off, self.cfg, self.sorted_go = buf, of, off, cfg, None


# Plans that outlive a step.  The plan is a function of the grid alone ("cached per grid"): a caller that hands the
# SAME grid tensor to every step -- fixed collocation points, reference test/test_2d.py:28-38 -- need not have it rebuilt
# (0.35 ms per step at BASELINE configs[1], 0.21 ms at configs[3]).  Off by default: an entry keeps its grid and its plan
# (150 MiB + the 1 GiB sorted copy at configs[1]) alive; ops.plan_cache(n) keeps the n most recent ones.
_plan_cache = []
_plan_cache_size = 0


def plan_cache(entries=None):
    """Get / set how many point plans are kept across steps, keyed on the grid tensor (identity + version counter, the
    tensor held while its plan is), the offsets and the problem.  0 (default): a plan lives as long as its StepContext."""
    global _plan_cache_size
    if entries is not None:
        _plan_cache_size = max(0, int(entries))
        del _plan_cache[_plan_cache_size:]
    return _plan_cache_size


class StepContext(object):
    """Prepared objects that the stages of ONE training step share (include/cosine_sampler.h,
    `input_cl` / `plan`): the channels-last copy of `input` and the point-binning plan of `grid`.
    Built lazily on first use and dropped with the context.  Every derived object is tied to the tensor it was made from
    by a strong reference and the tensor's version counter (`_Held`): a different tensor object -- even one that the
    allocator placed at the same address -- or an in-place update invalidates it (the reference reads its own arguments
    on every call, modules_2d.py:55-62, :89-95).  The autograd Functions create one context per forward call, so nothing
    outlives the graph it belongs to; pass `ctx=None` to let every stage work from scratch."""

    def __init__(self, reuse_grad_output=True, points_order=None, accumulate=False):
        self._cl = None
        self._cl_of = None
        self._pe = None                  # _PlanEntry
        # Whose cell-sorted copy the plan holds (include/cosine_sampler.h, cs_cotangent_layout.sorted_grad_output_valid)
        # and when a stage is asked to leave one (leave_sorted_grad_output: +0.25 ms at config 2, repaid by the next
        # stage that streams it).  reuse_grad_output=True -- a caller driving the stages of one step itself (bench.py,
        # the reference's pybind use): every backward stage is handed the same grad_output, the first one that
        # scatters leaves the copy.  False -- the autograd layer: the engine interleaves nodes with different
        # grad_outputs (a Helmholtz step: five scatter stages, four different tensors), so a copy is left only for a
        # tensor that `expect()` has announced at least twice.
        self.reuse_grad_output = reuse_grad_output
        self.half_ok = False             # set by the autograd layer: 16-bit streams go to the kernels as they are
        self._expected = []              # [_Held, count]
        # (_sorted_go: _Held of the tensor whose sorted copy the plan holds -- kept with the plan, which may be shared)
        # 'coherent' / 'random' / None (= ops.points_order(), by default measured): do consecutive points share cells?
        self.points_order = points_order
        # accumulate=True (a caller driving the stages of one step itself: bench.py, the reference's pybind use): what a
        # training step wants from its backward stages is the SUM of their input-shaped gradients (the autograd engine
        # adds them up into cells.grad; a multi-GPU step all-reduces that sum once).  The stages then ADD into one
        # accumulator held here (cs_cotangent_layout.accumulate_grad_input) and return None in place of grad_input;
        # grad_input_sum() hands the total over: one clear and one layout conversion per step instead of one per stage,
        # and no adding passes.  A stage that cannot add natively (3D, channel groups, a path with the other accumulator
        # layout) computes its gradient the plain way and it is added here with torch.
        self.accumulate = bool(accumulate)
        self._acc = None                 # [kind, buffer, shape of input, dim]
        self._acc_extra = None
        self.acc_native = self.acc_fallback = 0     # how many stages added natively / through the fallback (tests)

    def _acc_target(self, lib, dim, input, shape, P, kernel):
        """-> the CS_ACC_* kind this stage can add with into the step's accumulator (made on first use), or 0"""
        D = shape[2] if dim == 3 else 1
        kind = lib.cs_accumulator_kind(dim, shape[0], shape[1], D, shape[-2], shape[-1], P, int(kernel))
        if not kind:
            return 0
        if self._acc is None:
            if kind == _lib.ACC_NCHW:        # the caller's own layout: the accumulator IS the final tensor
                buf = torch.zeros_like(input)
            else:
                nbytes = lib.cs_accumulator_bytes(dim, kind, shape[0], shape[1], D, shape[-2], shape[-1])
                buf = torch.zeros(nbytes // 4, dtype=torch.float32, device=input.device)
            self._acc = [kind, buf, tuple(input.shape), dim]
        a = self._acc
        if a[0] != kind or a[2] != tuple(input.shape) or a[1].device != input.device:
            return 0
        return kind

    def _add_extra(self, grad_input):
        if grad_input is not None:
            self._acc_extra = grad_input if self._acc_extra is None else self._acc_extra.add_(grad_input)

    def grad_input_sum(self):
        """The sum of the input-shaped gradients of every backward stage run with this (accumulating) context since the
        last call, in the caller's (N,C,[D,]H,W) layout; None if there was none.  Resets the accumulator."""
        total = None
        if self._acc is not None:
            kind, buf, shape, dim = self._acc
            if kind == _lib.ACC_NCHW:
                total = buf
            else:
                total = torch.empty(shape, dtype=torch.float32, device=buf.device)
                D = shape[2] if dim == 3 else 1
                with torch.cuda.device(buf.device):
                    _lib.check(_lib.load().cs_accumulator_finish(
                        dim, kind, buf.data_ptr(), total.data_ptr(), shape[0], shape[1], D, shape[-2], shape[-1],
                        torch.cuda.current_stream(buf.device).cuda_stream), "cs_accumulator_finish")
        if self._acc_extra is not None:
            total = self._acc_extra if total is None else total.add_(self._acc_extra)
        self._acc = self._acc_extra = None
        return total

    @property
    def _sorted_go(self):
        return None if self._pe is None else self._pe.sorted_go

    @_sorted_go.setter
    def _sorted_go(self, held):
        if self._pe is not None:
            self._pe.sorted_go = held

    def expect(self, grad_output):
        """The autograd layer announces that a node holding this grad_output (the CALLER's tensor, before any dtype
        conversion) exists and will hand it to a scatter stage."""
        for e in self._expected:
            if e[0].same(grad_output):
                e[1] += 1
                return
        self._expected.append([_Held(grad_output), 1])

    def _expected_count(self, owner):
        for e in self._expected:
            if e[0].same(owner):
                return e[1]
        return 0

    def input_cl(self, lib, input, dim, shape, P, stream):
        if self._cl_of is None or not self._cl_of.same(input):
            D = shape[2] if dim == 3 else 1
            nbytes = lib.cs_pack_bytes(dim, shape[0], shape[1], D, shape[-2], shape[-1], P)
            self._cl, self._cl_of = None, _Held(input)
            if nbytes:
                buf = torch.empty(nbytes, dtype=torch.uint8, device=input.device)
                _lib.check(lib.cs_pack_input(dim, input.data_ptr(), buf.data_ptr(), shape[0], shape[1], D, shape[-2],
                                             shape[-1], stream), "cs_pack_input")
                self._cl = buf
        return self._cl

    def plan(self, lib, grid, offset, dim, shape, P, padding_mode, align_corners, multicell, stream):
        bc = grid.shape[0] == 1 and shape[0] > 1
        # (the plan bins samples by the SHIFTED point: the offsets are part of its identity, held and version-checked like
        # the grid -- an address alone could be a freed tensor's, re-used, or the same tensor updated in place)
        cfg = tuple(shape) + (int(padding_mode), bool(align_corners), bool(multicell), _force_epoch)
        pe = self._pe
        if pe is None or not pe.of.same(grid) or not pe.off.same(offset) or pe.cfg != cfg:
            pe = None
            for i, e in enumerate(_plan_cache):              # a plan of this very grid from an earlier step
                if e.cfg == cfg and e.of.same(grid) and e.off.same(offset):
                    pe = e
                    pe.sorted_go = None      # ... but never that step's sorted grad_output copy: a new step, new cotangents
                    _plan_cache.insert(0, _plan_cache.pop(i))
                    break
        if pe is None:
            sizes = shape[:2] + list(shape[2:]) + [P]          # N, C, [D,] H, W, P
            nbytes = getattr(lib, "cs%dd_plan_bytes" % dim)(*sizes)
            buf = None
            if nbytes:
                buf = torch.empty(nbytes, dtype=torch.uint8, device=grid.device)
                _lib.check(getattr(lib, "cs%dd_plan_build" % dim)(
                    grid.data_ptr(), offset.data_ptr(), buf.data_ptr(), nbytes, *sizes, int(padding_mode),
                    int(bool(align_corners)), int(bool(multicell)), _lib.GRID_BROADCAST if bc else 0, stream),
                    "cs%dd_plan_build" % dim)
            pe = _PlanEntry(buf, _Held(grid), _Held(offset), cfg)
            if _plan_cache_size and buf is not None:
                _plan_cache.insert(0, pe)
                del _plan_cache[_plan_cache_size:]
        self._pe = pe
        return pe.buf

    def prepare_plan(self, input, grid, offset, padding_mode, align_corners, multicell):
        """Build the point plan of `grid` now, on the current stream (otherwise the first stage that scatters builds it).
        -> True if this problem has a plan.  Lets a caller account for the plan separately from the stage that happens to
        need it first: it depends on the grid alone and serves every backward stage of the step."""
        dim, shape, P = _problem(input, grid)
        _offset_ok(offset, shape[0], input.device)
        lib = _lib.load()
        with torch.cuda.device(input.device):
            stream = torch.cuda.current_stream(input.device).cuda_stream
            return self.plan(lib, grid, offset, dim, shape, P, padding_mode, align_corners, multicell, stream) is not None


_force_epoch = 0


_force_mode = 0
_FLAGS = EXACT_MIXED | _lib.STREAM_F16 | _lib.STREAM_BF16 | _lib.GRID_BROADCAST | _lib.POINTS_COHERENT | _lib.SUM_OVER_N


# number of C-ABI calls by stage, and how many of them scattered (produced an input-shaped gradient): a rocprof-free way for
# tests to see that a graph ran the stages it should (e.g. that the autograd layer skipped the scatters nobody asked for)
call_counts = {}


def _count(stage, scattered, coherent):
    c = call_counts.setdefault(stage, [0, 0, 0])      # calls, of which scattered, of which on the coherent-points kernels
    c[0] += 1
    c[1] += int(bool(scattered))
    c[2] += int(bool(coherent))


def _call(stage, dim, ptrs, shape, P, padding_mode, align_corners, kernel, multicell, device, ctx=None, input=None,
          grid=None, offset=None, want_grad_input=False, have_cI=False, go_ns=None, ho_ns=None, grad_output=None,
          go_owner=None, sum_n=False, out_ns=None, gi_index=None, _define=False):
    """want_grad_input: the stage produces an input-shaped gradient (it scatters); it is allocated here (ptrs[gi_index]
    is its place in the argument list) and returned -- or, with an accumulating context, added to the step's accumulator
    and None is returned.  go_owner: the caller's tensor that `grad_output` was made from (itself unless the autograd layer
    converted the dtype): what the sorted copy in the plan is remembered by."""
    kernel_in = kernel
    if not isinstance(kernel, int) or (kernel & ~_FLAGS) not in (0, 1, 2):
        # the reference's kernel_enum returns None for unknown names and pybind then rejects it
        raise TypeError("kernel enum must be 0 (cosine), 1 (linear) or 2 (smooth-step), optionally | EXACT_MIXED, "
                        "got %r" % (kernel,))
    if grid is not None and grid.shape[0] == 1 and shape[0] > 1:
        kernel |= _lib.GRID_BROADCAST
    lib = _lib.load()
    fn = getattr(lib, "cs%dd_%s" % (dim, stage))
    D = shape[2] if dim == 3 else 1
    owner = grad_output if go_owner is None else go_owner
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        # coherent points (CS_POINTS_COHERENT): the 2D fast path reads the table through on-chip windows and needs no plan
        # (sum_n in 2D: the summing kernels ARE coherent-points kernels, the caller has decided; in 3D the summing mode runs
        # on the plain op's point kernels and scatter -- plan and all)
        coherent = dim == 2 and bool(sum_n or (grid is not None and _order_is_coherent(
            ctx, lib, grid, shape, P, padding_mode, align_corners, multicell, stream)))
        if sum_n:
            kernel |= _lib.SUM_OVER_N
        if coherent:
            kernel |= _lib.POINTS_COHERENT
        cl = plan = None
        if ctx is not None:
            cl = ctx.input_cl(lib, input, dim, shape, P, stream)
            if want_grad_input and not coherent:
                plan = ctx.plan(lib, grid, offset, dim, shape, P, padding_mode, align_corners, multicell, stream)
        stage_id = _lib.STAGE_ID[stage]
        if stage in ("backward", "backward_backward") and not want_grad_input:
            stage_id |= _lib.STAGE_NO_GRAD_INPUT       # grad_input is not wanted: no plan, no scatter scratch
        if coherent:
            stage_id |= _lib.STAGE_POINTS_COHERENT
        # the input-shaped gradient: a fresh tensor the call defines, or the step's accumulator it adds to
        grad_input, acc_kind = None, 0
        if want_grad_input:
            ptrs = list(ptrs)
            if ctx is not None and ctx.accumulate and not _define:
                acc_kind = ctx._acc_target(lib, dim, input, shape, P, kernel)
            if acc_kind:
                ptrs[gi_index] = ctx._acc[1].data_ptr()
                stage_id |= _lib.STAGE_ACCUMULATE
            else:
                grad_input = torch.empty_like(input)
                ptrs[gi_index] = grad_input.data_ptr()
        need = lib.cs_workspace_bytes(dim, stage_id, shape[0], shape[1], D, shape[-2], shape[-1], P,
                                      int(cl is not None), int(plan is not None), int(have_cI))
        ws = torch.empty(need, dtype=torch.uint8, device=device) if need else None
        tail = (_ptr(cl), _ptr(plan), _ptr(ws), need, stream)
        g_leave = 0
        if stage != "forward":
            CP = shape[1] * P
            g_valid = 0
            # the sorted copy of grad_output lives in walker plans of the 2D fast path only, and the '+mixed' second
            # backward with grad_out_input runs on kernels that neither read nor leave it
            keeps = (plan is not None and dim == 2 and not ((kernel & EXACT_MIXED) and have_cI)
                     and bool(lib.cs2d_plan_keeps_sorted_copy(shape[0], shape[1], shape[-2], shape[-1], P)))
            if keeps:
                g_valid = int(ctx._sorted_go is not None and ctx._sorted_go.same(owner))
                if not g_valid:
                    g_leave = int(ctx.reuse_grad_output or ctx._expected_count(owner) >= 2)
            layout = _lib.CotangentLayout(CP if go_ns is None else go_ns, CP if ho_ns is None else ho_ns, g_valid,
                                          g_leave, 0 if out_ns is None else out_ns, acc_kind, 0)
            tail = (layout,) + tail
        rc = fn(*ptrs, *shape, P, int(padding_mode), int(bool(align_corners)), int(kernel), int(bool(multicell)),
                *tail)
    if rc == _lib.ERR_UNSUPPORTED and acc_kind:
        # this stage's path keeps its sums in the other layout (e.g. the general second backward with grad_out_input in
        # a step on the coherent kernels); nothing has been written: run it the plain way and add the result below
        return _call(stage, dim, ptrs, shape, P, padding_mode, align_corners, kernel_in, multicell, device, ctx, input, grid,
                     offset, want_grad_input, have_cI, go_ns, ho_ns, grad_output, go_owner, sum_n, out_ns, gi_index, True)
    _lib.check(rc, "cs%dd_%s" % (dim, stage))
    _count(stage, want_grad_input, coherent)
    if g_leave:   # the plan now holds this one's sorted copy: remember whose, and keep it alive
        ctx._sorted_go = _Held(owner)
    if want_grad_input and ctx is not None and ctx.accumulate:
        if acc_kind:
            ctx.acc_native += 1
        else:
            ctx.acc_fallback += 1
            ctx._add_extra(grad_input)
        return None
    return grad_input


# ---- the order of the points ------------------------------------------------------------------------------------
# Whether consecutive points share cells is a property of the caller's DATA, and the coherent-points kernels are 20-40x
# slower than the general path on unordered points (13-17 ms per scatter stage at BASELINE configs[1]): 'auto' (default)
# therefore runs them ONLY on a grid tensor that has ITSELF been measured as ordered -- never on the strength of another
# tensor's measurement (an equal shape says nothing: train / eval sets, two models, an ordered set followed by an unordered
# one) and over ALL its points (every table of an (N,P,2) grid, not table 0).  The measurement is a 5 us kernel
# (cs_points_tile_changes_sampled) whose count comes back through a pinned word behind an event:
#   * a tensor seen before (identity + version counter, held while remembered): its decision, no work at all;
#   * a new tensor: the measurement is enqueued.  If the LAST tensor of this problem signature turned out ordered -- the one
#     case where waiting can pay: a caller whose steps rebuild the grid tensor from the same ordered points, as PIXEL's
#     torch.cat([x, y]).repeat(...) does (reference test/test_2d.py:36-38) -- the host waits for the count (one
#     event wait per new tensor; `order_waits` counts them) and then knows; otherwise nothing waits, the call takes the
#     general path (always right) and the count is picked up by a later call with the same tensor.
# So a wrong guess costs a missed speed-up or one wait, never the cliff.  Both paths give the same results for any order.
_points_order = "auto"
MIN_COHERENT_SAMPLES = 1 << 16
ORDER_SAMPLE_SEGMENTS = 64         # the order is judged on this many runs of 1024 consecutive points spread over the set (a 5 us kernel)
ORDER_CACHE = 3                    # grid tensors whose decision is remembered (each is held alive while it is)
ORDER_HISTORY = 64                 # problem signatures whose last decision is remembered
order_waits = 0                    # how often the host waited for a measurement (tests, diagnostics)
_order_known = []                  # _OrderEntry, most recent first
_order_history = {}                # signature -> was the last measured tensor of this signature ordered?
_order_pool = []                   # (device word, pinned host word, event) not in use


class _OrderEntry(object):
    __slots__ = ("held", "sig", "points", "decision", "slot")


def points_order(mode=None):
    """Get / set how the op decides whether the points are coherent: 'auto' (measure), 'coherent', 'random'."""
    global _points_order
    if mode is not None:
        if mode not in ("auto", "coherent", "random"):
            raise ValueError("points order must be 'auto', 'coherent' or 'random', got %r" % (mode,))
        _points_order = mode
        _order_reset()
    return _points_order


def _order_reset():
    for e in _order_known:
        if e.slot is not None:
            _order_pool.append(e.slot)
    del _order_known[:]
    _order_history.clear()


def _order_arrived(e):
    e.decision = int(e.slot[1].item()) * 256 <= min(e.points, ORDER_SAMPLE_SEGMENTS * 1024)
    _order_pool.append(e.slot)
    e.slot = None
    _order_history.pop(e.sig, None)
    _order_history[e.sig] = e.decision
    while len(_order_history) > ORDER_HISTORY:
        _order_history.pop(next(iter(_order_history)))


def _order_is_coherent(ctx, lib, grid, shape, P, padding_mode, align_corners, multicell, stream):
    global order_waits
    mode = (ctx.points_order if ctx is not None and ctx.points_order else None) or _points_order
    if mode != "auto":
        return mode == "coherent"
    if shape[0] * P < MIN_COHERENT_SAMPLES:
        return False
    ent = None
    for e in _order_known:                             # counts that have come back since the last call
        if e.decision is None and e.slot[2].query():
            _order_arrived(e)
    for i, e in enumerate(_order_known):
        if e.held.same(grid):
            ent = e
            if i:
                _order_known.insert(0, _order_known.pop(i))
            break
    if ent is None:
        if torch.cuda.is_current_stream_capturing():      # nothing can be read back inside a capture: the general path
            return False
        ent = _OrderEntry()
        ent.held, ent.decision = _Held(grid), None
        ent.sig = (grid.device, tuple(shape), tuple(grid.shape), int(padding_mode), bool(align_corners), bool(multicell))
        ent.points = grid.numel() // 2               # every table's points, in the order the kernels would walk them
        ent.slot = _order_pool.pop() if _order_pool else (torch.empty(1, dtype=torch.int32, device=grid.device),
                                                          torch.empty(1, dtype=torch.int32, pin_memory=True), torch.cuda.Event())
        if ent.slot[0].device != grid.device:
            ent.slot = (torch.empty(1, dtype=torch.int32, device=grid.device), ent.slot[1], ent.slot[2])
        _lib.check(lib.cs_points_tile_changes_sampled(2, grid.data_ptr(), ent.slot[0].data_ptr(), ent.points, 1, shape[-2],
                                                      shape[-1], int(padding_mode), int(bool(align_corners)),
                                                      int(bool(multicell)), ORDER_SAMPLE_SEGMENTS, stream),
                   "cs_points_tile_changes_sampled")
        ent.slot[1].copy_(ent.slot[0], non_blocking=True)
        ent.slot[2].record()
        _order_known.insert(0, ent)
        for old in _order_known[ORDER_CACHE:]:
            if old.slot is not None:
                # (its measurement may still be in flight: the words are only re-used behind their own event)
                old.slot[2].synchronize()
                _order_pool.append(old.slot)
        del _order_known[ORDER_CACHE:]
    if ent.decision is None:
        if ent.slot[2].query():
            _order_arrived(ent)
        elif _order_history.get(ent.sig, False) and not torch.cuda.is_current_stream_capturing():
            ent.slot[2].synchronize()                  # the last tensor of this signature was ordered: worth knowing now
            order_waits += 1
            _order_arrived(ent)
        else:
            return False                               # in flight: the general path is right for any order
    return ent.decision


def points_tile_changes(points, size, padding_mode=0, align_corners=True, multicell=True):
    """How often the 8-cell tile changes between consecutive points ((P, dim) tensor; size = (H, W) or (D, H, W)):
    about the number of occupied tiles for an ordered set, about P for an unordered one.  Synchronises (returns an int)."""
    _check(points, "points")
    dim = points.shape[-1]
    pts = points.reshape(-1, dim)
    D, H, W = ([1] + [int(x) for x in size])[-3:]
    count = torch.empty(1, dtype=torch.int32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.load().cs_points_tile_changes(dim, pts.data_ptr(), count.data_ptr(), pts.shape[0], D, H, W,
                                                      int(padding_mode), int(bool(align_corners)), int(bool(multicell)),
                                                      torch.cuda.current_stream(points.device).cuda_stream),
                   "cs_points_tile_changes")
    return int(count.item())


def sort_points(points, size, padding_mode=0, align_corners=True, multicell=True):
    """Order a point set by the cell it falls into (cs2d_sort_points / cs3d_sort_points): what a PIXEL-style caller does ONCE
    with its collocation points (reference test/test_2d.py:28-38 draws them once and re-uses them every step) so that the
    backward stages run on the coherent-points kernels.  points: (..., dim) fp32 CUDA tensor in [-1, 1]; size: (H, W) or
    (D, H, W) of the tables.  -> (sorted_points like points, perm (P,) int64 with sorted = points.reshape(-1, dim)[perm])."""
    _check(points, "points")
    dim = points.shape[-1]
    if dim not in (2, 3) or len(size) != dim:
        raise RuntimeError("points must be (..., 2) with size (H, W) or (..., 3) with size (D, H, W)")
    pts = points.reshape(-1, dim)
    P = pts.shape[0]
    out = torch.empty_like(pts)
    perm = torch.empty(P, dtype=torch.int32, device=points.device)
    lib = _lib.load()
    need = lib.cs_sort_points_bytes(P)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=points.device)
    sizes = [int(x) for x in size]
    with torch.cuda.device(points.device):
        stream = torch.cuda.current_stream(points.device).cuda_stream
        rc = getattr(lib, "cs%dd_sort_points" % dim)(pts.data_ptr(), out.data_ptr(), perm.data_ptr(), P, *sizes,
                                                     int(padding_mode), int(bool(align_corners)), int(bool(multicell)),
                                                     ws.data_ptr(), need, stream)
    _lib.check(rc, "cs%dd_sort_points" % dim)
    return out.reshape(points.shape), perm.long()


def force_path(mode):
    """Testing knob (cs_debug_force_path): 0 auto, 1 direct kernels only, 2 fast paths wherever they exist,
    3 = 2 without the wave-per-cell kernel for crowded tables, 4 = 2 without the re-use of the sorted grad_output copy
    between the stages of a step, 5 = 0 with the coherent-points hint ignored, 6 = 2 with the two-reads pack of 3D tables."""
    global _force_epoch, _force_mode
    if not _lib.load().cs_debug_force_path(int(mode)):
        raise RuntimeError("the testing knobs of libcosine_sampler_hip.so are inert in this process: start it with "
                           "COSINESAMPLER_DEBUG=1 in the environment (tests/conftest.py does)")
    _force_epoch += 1
    _force_mode = int(mode)


def out_shape(input, grid):
    return tuple(input.shape[:2]) + tuple(grid.shape[1:-1])


def _stream_kernel(kernel, *streams):
    """kernel enum | the stream-dtype flag; the streams of one call share one element type."""
    dts = {t.dtype for t in streams if t is not None}
    if len(dts) > 1:
        raise RuntimeError("the channel-major tensors of one call must share a dtype, got %s" % sorted(map(str, dts)))
    dt = dts.pop() if dts else torch.float32
    if not isinstance(kernel, int):
        return kernel, dt           # rejected with a TypeError by _call, as before
    return kernel | STREAM_DTYPES[dt], dt


# ---- channel groups ---------------------------------------------------------------------------------------------
# The fast paths hold one node's channels in registers / one LDS row: up to 32 channels in 2D, 16 in 3D.  The reference
# loops over any C (2d.cu:340-354); beyond those counts the direct kernels would take over (51 ms per backward stage at
# BASELINE configs[1] sizes).  Instead a table with more channels is run as channel ranges of at most that many through the
# fast paths: the sampler is linear and separable over channels -- `output`, `grad_input`, `grad_grad_out` are per channel
# (concatenated), `grad_grid` sums over channels (added up).  Input streams are handed over as views (the kernels take their
# n-stride), the table range is copied once per step (StepContext), per-channel results are concatenated.
GROUP_CHANNELS = {2: 32, 3: 16}


def _channel_groups(input, dim):
    C, g = int(input.shape[1]), GROUP_CHANNELS[dim]
    if C <= g or _force_mode == 1:
        return None
    return [(a, min(a + g, C)) for a in range(0, C, g)]


def _group_ctx(ctx, input, a, b):
    """-> (the contiguous channel range a:b of `input`, the child context of that range): both live in `ctx`, so that the
    stages of a step share the range's channels-last copy and plan exactly as they share the whole table's."""
    if ctx is None:
        return input[:, a:b].contiguous(), None
    kids = ctx.__dict__.setdefault("_kids", {})
    ent = kids.get((a, b))
    if ent is None or not ent[0].same(input):
        ent = kids[(a, b)] = (_Held(input), input[:, a:b].contiguous(),
                              StepContext(reuse_grad_output=ctx.reuse_grad_output, points_order=ctx.points_order))
    return ent[1], ent[2]


def _out_view(view, like, name):
    """A caller-provided place for a written stream -- the channel range of a wider (N, C_total, ..., P) tensor that a channel
    group's result goes to directly (no torch.cat afterwards): contiguous within one n, 16-byte aligned.  -> its n-stride, or
    None when the view cannot be written in place (the caller then copies)."""
    if view is None or tuple(view.shape) != tuple(like.shape) or view.dtype != like.dtype or view.device != like.device:
        return None
    if not view.shape[0] or not view[0].is_contiguous() or view.stride(0) < view[0].numel() or view.data_ptr() % 16:
        return None
    return view.stride(0)


def _rng(t, a, b):
    return None if t is None else t[:, a:b]


def _cat(parts):
    return None if parts[0] is None else torch.cat(parts, 1)


def _add(parts):
    out = parts[0]
    for x in parts[1:]:
        out = out + x
    return out


def _grouped(ctx, grad_input):
    """the input-shaped gradient of a table run as channel groups: returned, or (accumulating context) added to the step's sum"""
    if ctx is not None and ctx.accumulate and grad_input is not None:
        ctx.acc_fallback += 1
        ctx._add_extra(grad_input)
        return None
    return grad_input


def forward(input, grid, offset, padding_mode, align_corners, kernel, multicell, ctx=None, out_dtype=None):
    """out_dtype (not in the reference's signature): torch.float16 / torch.bfloat16 to have `output` written in that
    type by the kernel (fast paths only: half_streams_ok); default fp32."""
    input, grid = _al(input, grid)
    dim, shape, P = _problem(input, grid)
    groups = _channel_groups(input, dim)
    if groups:
        outs = []
        for a, b in groups:
            ig, cg = _group_ctx(ctx, input, a, b)
            outs.append(forward(ig, grid, offset, padding_mode, align_corners, kernel, multicell, cg, out_dtype))
        return torch.cat(outs, 1)
    _offset_ok(offset, shape[0], input.device)
    out_dtype = out_dtype or input.dtype
    if out_dtype not in STREAM_DTYPES:
        raise RuntimeError("output dtype must be float32, float16 or bfloat16, got %s" % out_dtype)
    output = torch.empty(out_shape(input, grid), dtype=out_dtype, device=input.device)
    kernel, _ = _stream_kernel(kernel, output)
    _call("forward", dim, [_ptr(input), _ptr(grid), _ptr(offset), _ptr(output)], shape, P,
          padding_mode, align_corners, kernel, multicell, input.device, ctx, input, grid, offset)
    return output


def backward(grad_output, input, grid, offset, padding_mode, align_corners, input_requires_grad, kernel, multicell,
             ctx=None, go_owner=None):
    """-> (grad_input | None, grad_grid); grad_input is None when input_requires_grad is False
    (the reference returns an undefined Tensor, 2d.cpp:73-79) -- and, in every backward stage, when `ctx` accumulates
    (StepContext(accumulate=True): the gradient has been added to the step's accumulator, ctx.grad_input_sum())."""
    if go_owner is None:
        go_owner = grad_output
    grad_output, input, grid = _al(grad_output, input, grid)
    dim, shape, P = _problem(input, grid)
    groups = _channel_groups(input, dim)
    if groups:
        _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
        gI, gG = [], []
        for a, b in groups:
            ig, cg = _group_ctx(ctx, input, a, b)
            r = backward(_rng(grad_output, a, b), ig, grid, offset, padding_mode, align_corners, input_requires_grad,
                         kernel, multicell, cg)
            gI.append(r[0])
            gG.append(r[1])
        return _grouped(ctx, _cat(gI)), _add(gG)
    _offset_ok(offset, shape[0], input.device)
    go_ns = _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
    kernel, _ = _stream_kernel(kernel, grad_output)
    bc = grid_is_broadcast(input, grid)
    grad_grid = _grid_result(grid, shape[0], bc)
    grad_input = _call("backward", dim, [_ptr(grad_output), _ptr(input), _ptr(grid), _ptr(offset), None,
                                         _ptr(grad_grid)], shape, P, padding_mode, align_corners, kernel, multicell,
                       input.device, ctx, input, grid, offset, want_grad_input=bool(input_requires_grad), go_ns=go_ns,
                       grad_output=grad_output, go_owner=go_owner, gi_index=4)
    return grad_input, _grid_reduce(grad_grid, bc)


def backward_backward(grad_out_input, grad_out_grid, input, grid, grad_output, offset, padding_mode, align_corners,
                      input_requires_grad, kernel, multicell, ctx=None, want_grad_input=True, go_owner=None, ggo_out=None):
    """-> (grad_input, grad_grid, grad_grad_out).  grad_out_input is only read when
    input_requires_grad (reference 2d.cu:654-656); grad_out_grid may be None (= zeros).
    want_grad_input=False (not in the reference): grad_input is not computed and comes back as None --
    the scatter half of the stage, and the point plan it needs, are skipped."""
    if go_owner is None:
        go_owner = grad_output
    grad_out_input, grad_out_grid, input, grid, grad_output = _al(grad_out_input, grad_out_grid, input, grid, grad_output)
    dim, shape, P = _problem(input, grid)
    groups = _channel_groups(input, dim)
    if groups:
        _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
        if input_requires_grad:
            _same(grad_out_input, input.shape, "grad_out_input", input.device)
        gI, gG = [], []
        # every group writes its channels of grad_grad_out in place (cs_cotangent_layout.grad_grad_out_stride_n): no cat
        ggO = torch.empty(grad_output.shape, dtype=grad_output.dtype, device=grad_output.device)
        for a, b in groups:
            ig, cg = _group_ctx(ctx, input, a, b)
            ci = _rng(grad_out_input, a, b).contiguous() if input_requires_grad else None
            r = backward_backward(ci, grad_out_grid, ig, grid, _rng(grad_output, a, b), offset, padding_mode,
                                  align_corners, input_requires_grad, kernel, multicell, cg, want_grad_input,
                                  ggo_out=ggO[:, a:b])
            gI.append(r[0])
            gG.append(r[1])
            if r[2].data_ptr() != ggO[:, a:b].data_ptr():
                ggO[:, a:b].copy_(r[2])
        return _grouped(ctx, _cat(gI)), _add(gG), ggO
    _offset_ok(offset, shape[0], input.device)
    go_ns = _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
    if input_requires_grad:
        _same(grad_out_input, input.shape, "grad_out_input", input.device)
    else:
        grad_out_input = None
    if grad_out_grid is not None:
        _same(grad_out_grid, grid.shape, "grad_out_grid", input.device)
    kernel, _ = _stream_kernel(kernel, grad_output)
    bc = grid_is_broadcast(input, grid)
    grad_grid = _grid_result(grid, shape[0], bc)
    out_ns = _out_view(ggo_out, grad_output, "grad_grad_out")
    grad_grad_out = ggo_out if out_ns is not None else torch.empty(grad_output.shape, dtype=grad_output.dtype,
                                                                   device=grad_output.device)
    grad_input = _call("backward_backward", dim,
                       [_ptr(grad_out_input), _ptr(grad_out_grid), _ptr(input), _ptr(grid), _ptr(grad_output), _ptr(offset),
                        None, _ptr(grad_grid), _ptr(grad_grad_out)],
                       shape, P, padding_mode, align_corners, kernel, multicell, input.device, ctx, input, grid, offset,
                       want_grad_input=bool(want_grad_input), have_cI=grad_out_input is not None, go_ns=go_ns,
                       grad_output=grad_output, go_owner=go_owner, out_ns=out_ns, gi_index=6)
    return grad_input, _grid_reduce(grad_grid, bc), grad_grad_out


def backward_backward_backward(input, grid, grad_output, grad_out_grid, grad_out_ggrid, offset, padding_mode,
                               align_corners, input_requires_grad, kernel, multicell, ctx=None):
    """-> (grad_input, grad_grad_out).  `input_requires_grad` is accepted and ignored, as in the
    reference kernel (2d.cu:736)."""
    go_owner = grad_output
    input, grid, grad_output, grad_out_grid, grad_out_ggrid = _al(input, grid, grad_output, grad_out_grid, grad_out_ggrid)
    dim, shape, P = _problem(input, grid)
    groups = _channel_groups(input, dim)
    if groups:
        _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
        gI, gO = [], []
        for a, b in groups:
            ig, cg = _group_ctx(ctx, input, a, b)
            r = backward_backward_backward(ig, grid, _rng(grad_output, a, b), grad_out_grid, grad_out_ggrid, offset,
                                           padding_mode, align_corners, input_requires_grad, kernel, multicell, cg)
            gI.append(r[0])
            gO.append(r[1])
        return _grouped(ctx, torch.cat(gI, 1)), torch.cat(gO, 1)
    _offset_ok(offset, shape[0], input.device)
    go_ns = _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
    _same(grad_out_grid, grid.shape, "grad_out_grid", input.device)
    _same(grad_out_ggrid, grid.shape, "grad_out_ggrid", input.device)
    kernel, _ = _stream_kernel(kernel, grad_output)
    grad_grad_out = torch.empty(grad_output.shape, dtype=grad_output.dtype, device=grad_output.device)
    grad_input = _call("backward_backward_backward", dim,
                       [_ptr(input), _ptr(grid), _ptr(grad_output), _ptr(grad_out_grid), _ptr(grad_out_ggrid), _ptr(offset),
                        None, _ptr(grad_grad_out)],
                       shape, P, padding_mode, align_corners, kernel, multicell, input.device, ctx, input, grid, offset,
                       want_grad_input=True, go_ns=go_ns, grad_output=grad_output, go_owner=go_owner, gi_index=6)
    return grad_input, grad_grad_out


def bbb_fused(input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset, padding_mode,
              align_corners, kernel, multicell, ctx=None, go_owner=None, ggo_out=None):
    """-> (grad_input, grad_grad_out) of the whole third backward (reference modules_2d.py:98-111):
    grad_input = K4.gInput + K3(gOut := grad_out_ggout, gOutInput := ones).gInput in one pass.
    grad_out_grid / grad_out_ggrid / grad_out_ggout may each be None (= zeros)."""
    if go_owner is None:
        go_owner = grad_output
    input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout = _al(
        input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout)
    dim, shape, P = _problem(input, grid)
    groups = _channel_groups(input, dim)
    if groups:
        _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
        if grad_out_ggout is not None:
            _same(grad_out_ggout, grad_output.shape, "grad_out_ggout", input.device, stream=True)
        gI = []
        ggO = torch.empty(grad_output.shape, dtype=grad_output.dtype, device=grad_output.device)   # written in place, group by group
        for a, b in groups:
            ig, cg = _group_ctx(ctx, input, a, b)
            r = bbb_fused(ig, grid, _rng(grad_output, a, b), grad_out_grid, grad_out_ggrid, _rng(grad_out_ggout, a, b),
                          offset, padding_mode, align_corners, kernel, multicell, cg, ggo_out=ggO[:, a:b])
            gI.append(r[0])
            if r[1].data_ptr() != ggO[:, a:b].data_ptr():
                ggO[:, a:b].copy_(r[1])
        return _grouped(ctx, torch.cat(gI, 1)), ggO
    _offset_ok(offset, shape[0], input.device)
    go_ns = _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
    for t, nm in ((grad_out_grid, "grad_out_grid"), (grad_out_ggrid, "grad_out_ggrid")):
        if t is not None:
            _same(t, grid.shape, nm, input.device)
    ho_ns = None
    if grad_out_ggout is not None:
        ho_ns = _same(grad_out_ggout, grad_output.shape, "grad_out_ggout", input.device, stream=True)
    kernel, _ = _stream_kernel(kernel, grad_output, grad_out_ggout)
    out_ns = _out_view(ggo_out, grad_output, "grad_grad_out")
    grad_grad_out = ggo_out if out_ns is not None else torch.empty(grad_output.shape, dtype=grad_output.dtype,
                                                                   device=grad_output.device)
    grad_input = _call("bbb_fused", dim,
                       [_ptr(input), _ptr(grid), _ptr(grad_output), _ptr(grad_out_grid), _ptr(grad_out_ggrid),
                        _ptr(grad_out_ggout), _ptr(offset), None, _ptr(grad_grad_out)],
                       shape, P, padding_mode, align_corners, kernel, multicell, input.device, ctx, input, grid, offset,
                       want_grad_input=True, go_ns=go_ns, ho_ns=ho_ns, grad_output=grad_output, go_owner=go_owner,
                       out_ns=out_ns, gi_index=7)
    return grad_input, grad_grad_out



# ---- the PIXEL pattern as one op: per-point results summed over the tables (CS_SUM_OVER_N) -----------------------------
# features = sampler(cells, grid.repeat(N,1,1,1)).sum(0) (reference test/test_2d.py:38, :51): one set of points, and -- in
# every derivative the caller takes -- one cotangent for all N tables and results that are summed over them.  These four
# functions take and return the n-free tensors ((1,C,..,P) streams, (1,..,P,dim) grid-shaped); input-shaped gradients stay
# (N,C,H,W).  Where the summing kernels apply (2D fast path, fp32, zeros padding with align_corners, N > 1, C <= 32; 3D:
# C <= 16, fp32, every padding mode -- cs_sum_over_n_supported) no (N,C,P) stream is ever materialised; anywhere else the
# same values come from the plain op on expanded inputs followed by torch sums.
# The 2D summing kernels walk the points in cell order.  Points in the order they were DRAWN are put into that order INSIDE the
# op (round 4): with one set of points for all N tables every per-point tensor is N times smaller than the plain op's --
# (1,C,P) cotangents and results, 64 MiB where the plain op has 1 GiB -- so sorting the points once per step
# (cs2d_sort_points: 0.3 ms for 2^20) and carrying the cotangents into that order and the results back with index
# selections costs a fraction of what the (N,C,P) streams, their records and the sums over n cost the plain op: BASELINE
# configs[2] through autograd on drawn points 15.3 -> see profiles/round4_ablation.txt.  (The plain op cannot do the same:
# its streams are per table, permuting them moves more than it saves -- DESIGN.md 4.4.)


def sum_over_n_mode(input, grid, padding_mode, align_corners, multicell, ctx=None, kernel=0):
    """How the *_sum_n functions will run this problem: 'kernels' (the summing kernels on the points as they are: in cell
    order), 'sorted' (the same kernels, the points put into cell order inside the op) or None (plain op + sums)."""
    if not (input.is_cuda and grid.is_cuda) or input.dim() not in (4, 5) or input.dtype != torch.float32:
        return None
    if isinstance(kernel, int) and (kernel & EXACT_MIXED):
        return None
    dim, shape, P = _problem(input, grid)
    if grid.shape[0] != 1 or shape[0] < 2 or _channel_groups(input, dim) or _force_mode == 1:
        return None
    lib = _lib.load()
    D = shape[2] if dim == 3 else 1
    if not lib.cs_sum_over_n_supported(dim, shape[0], shape[1], D, shape[-2], shape[-1], P, int(padding_mode),
                                       int(bool(align_corners))):
        return None
    if dim == 3:          # the 3D point kernels walk the tables per point: any order of the points, nothing to sort
        return "kernels"
    with torch.cuda.device(input.device):
        stream = torch.cuda.current_stream(input.device).cuda_stream
        coherent = _order_is_coherent(ctx, lib, grid, shape, P, padding_mode, align_corners, multicell, stream)
    if coherent or shape[0] * P < MIN_COHERENT_SAMPLES:
        return "kernels"
    return None if torch.cuda.is_current_stream_capturing() and _sorted_points_of(ctx, grid) is None else "sorted"


def sum_over_n_fused(input, grid, padding_mode, align_corners, multicell, ctx=None, kernel=0):
    """Will the *_sum_n functions run on the summing kernels for this problem (else: plain op + sums)?"""
    return sum_over_n_mode(input, grid, padding_mode, align_corners, multicell, ctx, kernel) is not None


sum_n_sorts = 0          # how often the summed op put a point set into cell order itself (tests, diagnostics)


def _sorted_points_of(ctx, grid):
    ent = getattr(ctx, "_sorted_pts", None) if ctx is not None else None
    return ent if ent is not None and ent[0].same(grid) else None


def _sum_n_order(ctx, grid, shape, padding_mode, align_corners, multicell):
    """The summed op's points in cell order -> (grid in that order, perm, inverse perm): once per grid tensor and step
    (kept in the step's context)."""
    global sum_n_sorts
    ent = _sorted_points_of(ctx, grid)
    if ent is None:
        P = grid.numel() // 2
        pts_s, perm = sort_points(grid.reshape(P, 2), tuple(shape[2:]), padding_mode, align_corners, multicell)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(P, dtype=perm.dtype, device=perm.device)
        perm, inv = perm.int(), inv.int()          # (cs_carry_points takes 32-bit indices)
        ent = (_Held(grid), pts_s.view(grid.shape), perm, inv, [])
        sum_n_sorts += 1
        if ctx is not None:
            ctx._sorted_pts = ent
    return ent[1], ent[2], ent[3], ent[4]


def _carry_points(t, index, dim):
    """t (an fp32 CUDA tensor) with its point axis `dim` (-1: a (.., C, .., P) stream; -2: a (.., P, d) grid-shaped tensor)
    taken in the order `index` (int32): cs_carry_points -- rows of 4 MiB gathered one after the other, a third of the time
    of torch.index_select (91 us per call in the Helmholtz step: profiles/round4_ablation.txt)"""
    width = 1 if dim == -1 else int(t.shape[-1])
    P = int(t.shape[dim])
    if t.dtype != torch.float32 or not t.is_contiguous() or width > 4 or t.numel() != (t.numel() // (P * width)) * P * width \
            or (dim == -1 and t.stride(-1) != 1):
        return t.index_select(dim, index.long())
    rows = t.numel() // (P * width)
    # every axis between the rows and the point axis must be 1 (streams are (N,C,1,..,P), grid-shaped (N,1,..,P,d))
    lead = 1
    for sz in t.shape[:dim]:
        lead *= int(sz)
    if lead != rows:
        return t.index_select(dim, index.long())
    out = torch.empty_like(t)
    with torch.cuda.device(t.device):
        _lib.check(_lib.load().cs_carry_points(t.data_ptr(), out.data_ptr(), index.data_ptr(), rows, P, width,
                                               torch.cuda.current_stream(t.device).cuda_stream), "cs_carry_points")
    return out


def _carry(t, index, dim, cache=None):
    """t with its point axis `dim` taken in the order `index` (None stays None); `cache`: the step's list of cotangents
    already carried over -- the backward stages of one graph node chain are handed the same grad_output again and again"""
    if t is None:
        return None
    if cache is not None:
        for held, dm, r in cache:
            if dm == dim and held.same(t):
                return r
    r = _carry_points(t, index, dim)
    if cache is not None:
        cache.insert(0, (_Held(t), dim, r))
        del cache[4:]
    return r


def _one(t, N):
    return None if t is None else t.expand((N,) + tuple(t.shape[1:]))


def _shared(t, like_shape, name, device, stream=False):
    """an n-free tensor: leading extent 1"""
    if stream:
        _check_stream(t, name)
    else:
        _check(t, name)
    _shape_dev(t, (1,) + tuple(like_shape[1:]), name, device)


def forward_sum_n(input, grid, offset, padding_mode, align_corners, kernel, multicell, ctx=None):
    """-> sum over n of forward(...): (1, C, [Do,] Ho, Wo).  grid: (1, ..., dim)."""
    input, grid = _al(input, grid)
    dim, shape, P = _problem(input, grid)
    if grid.shape[0] != 1:
        raise RuntimeError("the summed op takes ONE set of points: grid must be (1, ..., %d), got %s" % (dim, tuple(grid.shape)))
    mode = sum_over_n_mode(input, grid, padding_mode, align_corners, multicell, ctx, kernel)
    if mode is None:
        return forward(input, grid, offset, padding_mode, align_corners, kernel, multicell, ctx).sum(0, keepdim=True)
    inv = None
    if mode == "sorted":           # the points as drawn: into cell order for the kernels, the result back into the caller's
        grid, _, inv, _ = _sum_n_order(ctx, grid, shape, padding_mode, align_corners, multicell)
    _offset_ok(offset, shape[0], input.device)
    output = torch.empty((1,) + out_shape(input, grid)[1:], dtype=input.dtype, device=input.device)
    _call("forward", dim, [_ptr(input), _ptr(grid), _ptr(offset), _ptr(output)], shape, P, padding_mode, align_corners,
          kernel, multicell, input.device, ctx, input, grid, offset, sum_n=True)
    return output if inv is None else _carry_points(output, inv, -1)


def backward_sum_n(grad_output, input, grid, offset, padding_mode, align_corners, input_requires_grad, kernel, multicell,
                   ctx=None):
    """grad_output: (1, C, ..., P), the one cotangent of every table -> (grad_input (N,C,H,W) | None, grad_grid (1,...,P,dim))."""
    grad_output, input, grid = _al(grad_output, input, grid)
    dim, shape, P = _problem(input, grid)
    _shared(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
    mode = sum_over_n_mode(input, grid, padding_mode, align_corners, multicell, ctx, kernel)
    if mode is None or grad_output.dtype != torch.float32:
        return backward(_one(grad_output, shape[0]), input, grid, offset, padding_mode, align_corners, input_requires_grad,
                        kernel, multicell, ctx)
    inv = None
    if mode == "sorted":
        grid, perm, inv, cache = _sum_n_order(ctx, grid, shape, padding_mode, align_corners, multicell)
        grad_output = _carry(grad_output, perm, -1, cache)
    _offset_ok(offset, shape[0], input.device)
    grad_grid = torch.empty_like(grid)
    grad_input = _call("backward", dim, [_ptr(grad_output), _ptr(input), _ptr(grid), _ptr(offset), None, _ptr(grad_grid)],
                       shape, P, padding_mode, align_corners, kernel, multicell, input.device, ctx, input, grid, offset,
                       want_grad_input=bool(input_requires_grad), go_ns=0, grad_output=grad_output, sum_n=True, gi_index=4)
    return grad_input, (grad_grid if inv is None else _carry_points(grad_grid, inv, -2))


def backward_backward_sum_n(grad_out_grid, input, grid, grad_output, offset, padding_mode, align_corners, kernel, multicell,
                            ctx=None, want_grad_input=True):
    """grad_out_input absent.  -> (grad_input (N,C,H,W) | None, grad_grid (1,...,P,dim), grad_grad_out (1,C,...,P))."""
    grad_out_grid, input, grid, grad_output = _al(grad_out_grid, input, grid, grad_output)
    dim, shape, P = _problem(input, grid)
    _shared(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
    if grad_out_grid is not None:
        _same(grad_out_grid, grid.shape, "grad_out_grid", input.device)
    mode = sum_over_n_mode(input, grid, padding_mode, align_corners, multicell, ctx, kernel)
    if mode is None or grad_output.dtype != torch.float32:
        gI, gG, ggO = backward_backward(None, grad_out_grid, input, grid, _one(grad_output, shape[0]), offset, padding_mode,
                                        align_corners, False, kernel, multicell, ctx, want_grad_input)
        return gI, gG, ggO.sum(0, keepdim=True)
    inv = None
    if mode == "sorted":
        grid, perm, inv, cache = _sum_n_order(ctx, grid, shape, padding_mode, align_corners, multicell)
        grad_output = _carry(grad_output, perm, -1, cache)
        grad_out_grid = _carry(grad_out_grid, perm, -2)
    _offset_ok(offset, shape[0], input.device)
    grad_grid = torch.empty_like(grid)
    grad_grad_out = torch.empty_like(grad_output)
    grad_input = _call("backward_backward", dim,
                       [None, _ptr(grad_out_grid), _ptr(input), _ptr(grid), _ptr(grad_output), _ptr(offset), None,
                        _ptr(grad_grid), _ptr(grad_grad_out)],
                       shape, P, padding_mode, align_corners, kernel, multicell, input.device, ctx, input, grid, offset,
                       want_grad_input=bool(want_grad_input), go_ns=0, grad_output=grad_output, sum_n=True, gi_index=6)
    if inv is not None:
        grad_grid, grad_grad_out = _carry_points(grad_grid, inv, -2), _carry_points(grad_grad_out, inv, -1)
    return grad_input, grad_grid, grad_grad_out


def bbb_fused_sum_n(input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset, padding_mode,
                    align_corners, kernel, multicell, ctx=None):
    """-> (grad_input (N,C,H,W), grad_grad_out (1,C,...,P)); grad_out_ggout: (1,C,...,P) or None."""
    input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout = _al(
        input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout)
    dim, shape, P = _problem(input, grid)
    _shared(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
    if grad_out_ggout is not None:
        _shared(grad_out_ggout, out_shape(input, grid), "grad_out_ggout", input.device, stream=True)
    for t, nm in ((grad_out_grid, "grad_out_grid"), (grad_out_ggrid, "grad_out_ggrid")):
        if t is not None:
            _same(t, grid.shape, nm, input.device)
    fp32 = grad_output.dtype == torch.float32 and (grad_out_ggout is None or grad_out_ggout.dtype == torch.float32)
    mode = sum_over_n_mode(input, grid, padding_mode, align_corners, multicell, ctx, kernel)
    if mode is None or not fp32:
        gI, ggO = bbb_fused(input, grid, _one(grad_output, shape[0]), grad_out_grid, grad_out_ggrid,
                            _one(grad_out_ggout, shape[0]), offset, padding_mode, align_corners, kernel, multicell, ctx)
        return gI, ggO.sum(0, keepdim=True)
    inv = None
    if mode == "sorted":
        grid, perm, inv, cache = _sum_n_order(ctx, grid, shape, padding_mode, align_corners, multicell)
        grad_output = _carry(grad_output, perm, -1, cache)
        grad_out_ggout = _carry(grad_out_ggout, perm, -1)
        grad_out_grid, grad_out_ggrid = _carry(grad_out_grid, perm, -2), _carry(grad_out_ggrid, perm, -2)
    _offset_ok(offset, shape[0], input.device)
    grad_grad_out = torch.empty_like(grad_output)
    grad_input = _call("bbb_fused", dim,
                       [_ptr(input), _ptr(grid), _ptr(grad_output), _ptr(grad_out_grid), _ptr(grad_out_ggrid), _ptr(grad_out_ggout),
                        _ptr(offset), None, _ptr(grad_grad_out)],
                       shape, P, padding_mode, align_corners, kernel, multicell, input.device, ctx, input, grid, offset,
                       want_grad_input=True, go_ns=0, ho_ns=0, grad_output=grad_output, sum_n=True, gi_index=7)
    return grad_input, (grad_grad_out if inv is None else _carry_points(grad_grad_out, inv, -1))


def bbb_grid(input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset, padding_mode,
             align_corners, kernel, multicell):
    """NOT in the reference (whose third backward returns no gradient for grid, modules_2d.py:111): the gradient w.r.t.
    `grid` of <grad_grid2, grad_out_ggrid> + <grad_grad_out, grad_out_ggout>, the second backward's outputs taken
    with every mixed term and grad_out_input absent (include/cosine_sampler.h, cs{2,3}d_bbb_grid).  -> like grid."""
    input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout = _al(
        input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout)
    dim, shape, P = _problem(input, grid)
    _offset_ok(offset, shape[0], input.device)
    if not isinstance(kernel, int) or (kernel & ~EXACT_MIXED) not in (0, 1, 2):
        raise TypeError("kernel enum must be 0, 1 or 2, optionally | EXACT_MIXED, got %r" % (kernel,))
    groups = _channel_groups(input, dim)
    if groups:
        return _add([bbb_grid(input[:, a:b].contiguous(), grid, _rng(grad_output, a, b), grad_out_grid, grad_out_ggrid,
                              _rng(grad_out_ggout, a, b), offset, padding_mode, align_corners, kernel, multicell)
                     for a, b in groups])
    go_ns = _same(grad_output, out_shape(input, grid), "grad_output", input.device, stream=True)
    _same(grad_out_grid, grid.shape, "grad_out_grid", input.device)
    if grad_out_ggrid is not None:
        _same(grad_out_ggrid, grid.shape, "grad_out_ggrid", input.device)
    ho_ns = None
    if grad_out_ggout is not None:
        ho_ns = _same(grad_out_ggout, grad_output.shape, "grad_out_ggout", input.device, stream=True)
    bc = grid_is_broadcast(input, grid)
    grad_grid3 = _grid_result(grid, shape[0], bc)
    if bc:
        kernel |= _lib.GRID_BROADCAST
    lib = _lib.load()
    CP = shape[1] * P
    layout = _lib.CotangentLayout(CP if go_ns is None else go_ns, CP if ho_ns is None else ho_ns)
    with torch.cuda.device(input.device):
        rc = getattr(lib, "cs%dd_bbb_grid" % dim)(
            _ptr(input), _ptr(grid), _ptr(grad_output), _ptr(grad_out_grid), _ptr(grad_out_ggrid), _ptr(grad_out_ggout),
            _ptr(offset), _ptr(grad_grid3), *shape, P, int(padding_mode), int(bool(align_corners)), int(kernel),
            int(bool(multicell)), layout, torch.cuda.current_stream(input.device).cuda_stream)
    _lib.check(rc, "cs%dd_bbb_grid" % dim)
    return _grid_reduce(grad_grid3, bc)
