"""CosineSampler2d / CosineSampler3d: the reference's public autograd surface on the HIP kernels.

    out = CosineSampler2d.apply(input, grid, padding_mode='zeros', align_corners=True,
                                kernel='cosine', multicell=True)

Same class names, positional argument order, strings and defaults as the reference
(reference cosine_sampler_2d/modules_2d.py:20-44, cosine_sampler_3d/modules_3d.py:20-45), and the
same three-level chain of Functions so that the op is differentiable to third order w.r.t.
`input` and to second order w.r.t. `grid` (modules_2d.py:47-111):

    level 1  CosineSampler{2,3}d           forward  -> ops.forward            (K1/K5)
    level 2  _SamplerBackward              forward  -> ops.backward           (K2/K6)
    level 3  _SamplerBackwardBackward      forward  -> ops.backward_backward  (K3/K7)
                                           backward -> ops.bbb_fused          (K4/K8 + second K3/K7)

What is deliberately NOT reproduced: the `.item()` host synchronisations
(modules_2d.py:87,104; modules_3d.py:41,80,94) -- absent cotangents arrive as None
(set_materialize_grads(False)) and go to the kernels as null pointers; the per-call CPU
linspace + H2D copy of `offset` (modules_2d.py:24-27) -- built once per (N, device) with the same
torch.linspace call so the bits match; the 3D 7-tuple early return (modules_3d.py:41-42).
Third-order results match the reference exactly in what they contain: grads w.r.t. `input` and
`gOut` only, no d/dgrid, `gOutgInput` ignored (modules_2d.py:111).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


def padding_mode_enum(padding_mode):
    """reference modules_2d.py:4-10: anything that is not 'zeros'/'border' means reflection."""
    if padding_mode == "zeros":
        return 0
    if padding_mode == "border":
        return 1
    return 2


_KERNELS = {"cosine": 0, "bilinear": 1, "trilinear": 1, "linear": 1, "smooth-step": 2, "smoothstep": 2}
_KERNELS.update({k + "+mixed": v | ops.EXACT_MIXED for k, v in list(_KERNELS.items())})


def kernel_enum(kernel):
    """reference modules_2d.py:12-18 / modules_3d.py:12-18.  'bilinear' (2D name) and 'trilinear'
    (3D name) are accepted by both samplers, plus the aliases 'linear' and 'smoothstep'.
    Unknown names give None, which the op layer rejects with a TypeError (the reference's pybind
    call does the same).
    Not in the reference: a '+mixed' suffix ('cosine+mixed', 'smooth-step+mixed', ...) keeps the mixed second
    derivatives the reference drops, so that u_xy and d(u_xy)/d(cells) taken through autograd are
    those of the interpolant; everything else, and every result without the suffix, is unchanged."""
    return _KERNELS.get(kernel)


class _Config(object):
    __slots__ = ("pad", "align_corners", "kernel", "multicell", "sum_n")

    def __init__(self, padding_mode, align_corners, kernel, multicell, sum_n=False):
        self.pad = padding_mode_enum(padding_mode)
        self.align_corners = bool(align_corners)
        self.kernel = kernel_enum(kernel)
        self.multicell = bool(multicell)
        self.sum_n = bool(sum_n)      # CosineSampler{2,3}dSum: one set of points / cotangents, results summed over the tables


_offset_cache = {}


def multicell_offset(N, multicell, device):
    """offset[n] = linspace(0, 1-1/N, N) (multicell) or zeros (reference modules_2d.py:24-27),
    computed by the same CPU torch.linspace call, cached on the device."""
    device = torch.device(device)
    key = (int(N), bool(multicell), device)
    t = _offset_cache.get(key)
    if t is None:
        if multicell and N > 0:
            t = torch.linspace(0, 1 - (1 / N), N)
        else:
            t = torch.zeros(N)
        t = t.to(device)
        _offset_cache[key] = t
    return t


def _c(t):
    return None if t is None else t.contiguous()


# Cotangents of `output` (gOut, gOutggOut) are NOT forced contiguous: PIXEL sums the sampled features over n
# (reference test_2d.py:51), so they arrive as `expand`ed (stride-0 along n) views of one (C,P) block and the
# reference's `.contiguous()` (modules_2d.py:42,109) would write N*C*P floats per backward call only for the kernel
# to read them back.  ops.keep_expanded leaves such views alone; the kernels take the n-stride.


# dtypes: the kernels are fp32 (include/cosine_sampler.h).  The reference dispatches double/float/half
# (2d.cu:905) but evaluates the blending weights and their derivatives in `float` whatever the tensor
# type (2d.cu:239-261, :430-431), so float64/float16/bfloat16 tensors are served by
# converting at this boundary: fp32 math, results cast back to the dtype of the tensor they are the
# gradient (or output) of.  Not a native path -- each stage pays the conversion copies.
_FLOATS = (torch.float32, torch.float64, torch.float16, torch.bfloat16)


def _f32(t):
    if t is None or t.dtype == torch.float32:
        return t
    if t.dtype not in _FLOATS:
        raise RuntimeError("CosineSampler expects floating-point tensors, got %s" % t.dtype)
    return t.float()


def _as(t, like):
    return t if t is None or t.dtype == like.dtype else t.to(like.dtype)


_HALVES = (torch.float16, torch.bfloat16)


def _stream(t, step):
    """A channel-major stream (grad_output, grad_out_ggout) as the kernels take it: float16 / bfloat16 stay as they are
    where the problem runs on a path with native 16-bit streams (step.half_ok), everything else becomes fp32."""
    if t is not None and t.dtype in _HALVES and step.half_ok:
        return t
    return _f32(t)


def _engine_wants(ctx, i):
    """Will the running backward pass use the gradient this node returns for its i-th forward input?
    `ctx.needs_input_grad` only says that the input requires grad; under
    `torch.autograd.grad(u, x, create_graph=True)` -- every derivative a PINN takes -- the gradient w.r.t.
    the cell table is computed by the reference (input_requires_grad is True, modules_2d.py:44) and dropped
    by the engine.  Asking the engine lets those calls skip the scatter half of the stage.  Whenever the
    question cannot be answered (leaf input inside autograd.grad, no graph task, ...) the answer is yes."""
    if not ctx.needs_input_grad[i]:
        return False
    try:
        node = ctx.next_functions[i][0]
        return node is None or bool(torch._C._will_engine_execute_node(node))
    except Exception:   # noqa: BLE001 -- any doubt: compute it
        return True


def _through_view(input):
    """The engine only answers `_engine_wants` for non-leaf nodes while autograd.grad() runs; a leaf table
    (the usual nn.Parameter) therefore enters the op through a no-copy autograd view."""
    if isinstance(input, torch.Tensor) and input.requires_grad and input.grad_fn is None and torch.is_grad_enabled():
        return input.view_as(input)
    return input


def _forward(ctx, dim, input, grid, padding_mode, align_corners, kernel, multicell, sum_n=False):
    if input.dim() != dim + 2:
        raise RuntimeError("CosineSampler%dd expects a %d-D input, got %s" % (dim, dim + 2, tuple(input.shape)))
    cfg = _Config(padding_mode, align_corners, kernel, multicell, sum_n)
    offset = multicell_offset(input.shape[0], multicell, input.device)
    # channels-last input copy + point plan, shared by this call's backward chain; which grad_output's sorted copy is
    # worth leaving in the plan is decided from the nodes that announce themselves (StepContext.expect)
    step = ops.StepContext(reuse_grad_output=False)
    x32, g32 = _f32(input), _f32(grid)
    # float16 / bfloat16 callers: the table and the grid (small) are converted, the big channel-major tensors are read and
    # written in the caller's type by the kernels themselves wherever a fast path applies
    step.half_ok = input.dtype in _HALVES and ops.half_streams_ok(x32, g32) and not sum_n
    if sum_n:
        output = _as(ops.forward_sum_n(x32, g32, offset, cfg.pad, cfg.align_corners, cfg.kernel, cfg.multicell, ctx=step), input)
    else:
        output = _as(ops.forward(x32, g32, offset, cfg.pad, cfg.align_corners, cfg.kernel, cfg.multicell, ctx=step,
                                 out_dtype=input.dtype if step.half_ok else None), input)
    ctx.save_for_backward(input, grid)
    ctx.offset = offset
    ctx.cfg = cfg
    ctx.step = step
    return output


def _backward(ctx, grad_out):
    input, grid = ctx.saved_tensors
    if grad_out is None:
        return None, None, None, None, None, None
    grad_out = ops.keep_expanded(grad_out)
    if torch.is_grad_enabled():        # create_graph: the node made here may run a scatter stage on grad_out later
        ctx.step.expect(grad_out)
    d_input, d_grid = _SamplerBackward.apply(input, grid, grad_out, ctx.offset, ctx.cfg,
                                             _engine_wants(ctx, 0), ctx.step)
    return d_input, d_grid, None, None, None, None


class CosineSampler2d(Function):
    @classmethod
    def apply(cls, input, *args, **kwargs):
        return super().apply(_through_view(input), *args, **kwargs)

    @staticmethod
    def forward(ctx, input, grid, padding_mode="zeros", align_corners=True, kernel="cosine", multicell=True):
        ctx.set_materialize_grads(False)
        return _forward(ctx, 2, input, grid, padding_mode, align_corners, kernel, multicell)

    @staticmethod
    def backward(ctx, grad_out):
        return _backward(ctx, grad_out)


class CosineSampler3d(Function):
    @classmethod
    def apply(cls, input, *args, **kwargs):
        return super().apply(_through_view(input), *args, **kwargs)

    @staticmethod
    def forward(ctx, input, grid, padding_mode="zeros", align_corners=True, kernel="cosine", multicell=True):
        ctx.set_materialize_grads(False)
        return _forward(ctx, 3, input, grid, padding_mode, align_corners, kernel, multicell)

    @staticmethod
    def backward(ctx, grad_out):
        return _backward(ctx, grad_out)


class CosineSampler2dSum(Function):
    """NOT in the reference: the PIXEL pattern as one op.

        feat = CosineSampler2dSum.apply(cells, points.view(1, 1, P, 2), 'zeros', True, 'cosine', True)      # (1, C, 1, P)
             = CosineSampler2d.apply(cells, points.view(1, 1, P, 2).repeat(N, 1, 1, 1), ...).sum(0, keepdim=True)

    -- what reference callers write as grid.repeat(N,1,1,1) ... .sum(0) (test/test_2d.py:38, :51) -- differentiable to the same
    orders.  Every per-point tensor of the chain (output, its cotangents, their gradients) is N times smaller; where the summing
    kernels apply (2D fast path, fp32, zeros padding with align_corners, points in cell order: ops.sum_over_n_fused) no
    (N,C,P) tensor is ever materialised, elsewhere the same values come from the plain op and torch sums."""

    @classmethod
    def apply(cls, input, *args, **kwargs):
        return super().apply(_through_view(input), *args, **kwargs)

    @staticmethod
    def forward(ctx, input, grid, padding_mode="zeros", align_corners=True, kernel="cosine", multicell=True):
        ctx.set_materialize_grads(False)
        return _forward(ctx, 2, input, grid, padding_mode, align_corners, kernel, multicell, sum_n=True)

    @staticmethod
    def backward(ctx, grad_out):
        return _backward(ctx, grad_out)


class CosineSampler3dSum(Function):
    """the 3D counterpart of CosineSampler2dSum: the channels-last point kernels walk the N tables per point (C <= 16, fp32;
    round 4), elsewhere the plain op + sums"""

    @classmethod
    def apply(cls, input, *args, **kwargs):
        return super().apply(_through_view(input), *args, **kwargs)

    @staticmethod
    def forward(ctx, input, grid, padding_mode="zeros", align_corners=True, kernel="cosine", multicell=True):
        ctx.set_materialize_grads(False)
        return _forward(ctx, 3, input, grid, padding_mode, align_corners, kernel, multicell, sum_n=True)

    @staticmethod
    def backward(ctx, grad_out):
        return _backward(ctx, grad_out)


class _SamplerBackward(Function):
    """(input, grid, gOut) -> (grad_input, grad_grid); reference CosineSamplerBackward,
    modules_2d.py:47-74."""

    @staticmethod
    def forward(ctx, input, grid, gOut, offset, cfg, input_requires_grad, step):
        ctx.set_materialize_grads(False)
        ctx.offset = offset
        ctx.cfg = cfg
        ctx.step = step
        if cfg.sum_n:
            grad_input, grad_grid = ops.backward_sum_n(_f32(gOut), _f32(input), _f32(grid), offset, cfg.pad, cfg.align_corners,
                                                       bool(input_requires_grad), cfg.kernel, cfg.multicell, ctx=step)
        else:
            grad_input, grad_grid = ops.backward(_stream(gOut, step), _f32(input), _f32(grid), offset, cfg.pad,
                                                 cfg.align_corners, bool(input_requires_grad), cfg.kernel, cfg.multicell,
                                                 ctx=step, go_owner=gOut)
        ctx.save_for_backward(input, grid, gOut)
        return _as(grad_input, input), _as(grad_grid, grid)

    @staticmethod
    def backward(ctx, gOutInput, gOutGrid):
        input, grid, gOut = ctx.saved_tensors
        if gOutInput is None and gOutGrid is None:
            return None, None, None, None, None, None, None
        if torch.is_grad_enabled():    # create_graph: a third backward through the node made here scatters with gOut again
            ctx.step.expect(gOut)
        gInput, gGrid, ggOut = _SamplerBackwardBackward.apply(input, grid, gOut, _c(gOutInput), _c(gOutGrid),
                                                              ctx.offset, ctx.cfg, ctx.step, _engine_wants(ctx, 0))
        return gInput, gGrid, ggOut, None, None, None, None


class _SamplerBackwardBackward(Function):
    """(input, grid, gOut, gOutInput, gOutGrid) -> (gInput, gGrid, ggOut); reference
    CosineSamplerBackwardBackward, modules_2d.py:76-111."""

    @staticmethod
    def forward(ctx, input, grid, gOut, gOutInput, gOutGrid, offset, cfg, step, want_grad_input=True):
        ctx.set_materialize_grads(False)
        ctx.offset = offset
        ctx.cfg = cfg
        ctx.step = step
        # ('+mixed' with gOutInput runs on kernels without native 16-bit streams: fp32 there)
        go = _f32(gOut) if (cfg.kernel & ops.EXACT_MIXED) and gOutInput is not None else _stream(gOut, step)
        if cfg.sum_n and gOutInput is None:
            gInput, gGrid, ggOut = ops.backward_backward_sum_n(_f32(gOutGrid), _f32(input), _f32(grid), _f32(gOut), offset,
                                                               cfg.pad, cfg.align_corners, cfg.kernel, cfg.multicell, ctx=step,
                                                               want_grad_input=bool(want_grad_input))
        elif cfg.sum_n:      # a cotangent of grad_input as well (not the PIXEL pattern): the plain op on the expanded gOut
            N = input.shape[0]
            gInput, gGrid, ggOut = ops.backward_backward(_f32(gOutInput), _f32(gOutGrid), _f32(input), _f32(grid),
                                                         _f32(gOut).expand((N,) + tuple(gOut.shape[1:])), offset, cfg.pad,
                                                         cfg.align_corners, True, cfg.kernel, cfg.multicell, ctx=step,
                                                         want_grad_input=bool(want_grad_input))
            ggOut = ggOut.sum(0, keepdim=True)
        else:
            gInput, gGrid, ggOut = ops.backward_backward(_f32(gOutInput), _f32(gOutGrid), _f32(input), _f32(grid),
                                                         go, offset, cfg.pad, cfg.align_corners,
                                                         gOutInput is not None, cfg.kernel, cfg.multicell, ctx=step,
                                                         want_grad_input=bool(want_grad_input), go_owner=gOut)
        gInput, gGrid, ggOut = _as(gInput, input), _as(gGrid, grid), _as(ggOut, gOut)
        ctx.has_cG = gOutGrid is not None
        if gOutGrid is None:
            ctx.save_for_backward(input, grid, gOut)
        else:
            ctx.save_for_backward(input, grid, gOut, gOutGrid)
        return gInput, gGrid, ggOut

    @staticmethod
    @once_differentiable
    def backward(ctx, gOutgInput, gOutgGrid, gOutggOut):
        # gOutgInput is ignored, exactly as in the reference (modules_2d.py:106-111).
        if ctx.has_cG:
            input, grid, gOut, gOutGrid = ctx.saved_tensors
        else:
            (input, grid, gOut), gOutGrid = ctx.saved_tensors, None
        if gOutgGrid is None and gOutggOut is None:
            return None, None, None, None, None, None, None, None, None
        cfg = ctx.cfg
        hG, hO = _f32(_c(gOutgGrid)), ops.keep_expanded(gOutggOut)
        gO = _stream(gOut, ctx.step)
        hO = _stream(hO, ctx.step)
        if hO is not None and hO.dtype != gO.dtype:      # mixed types: the streams of one call share one
            gO, hO = _f32(gO), _f32(hO)
        if cfg.sum_n:
            gInput, ggOut = ops.bbb_fused_sum_n(_f32(input), _f32(grid), _f32(gO), _f32(gOutGrid), hG, _f32(hO), ctx.offset,
                                                cfg.pad, cfg.align_corners, cfg.kernel, cfg.multicell, ctx=ctx.step)
        else:
            gInput, ggOut = ops.bbb_fused(_f32(input), _f32(grid), gO, _f32(gOutGrid), hG, hO, ctx.offset, cfg.pad,
                                          cfg.align_corners, cfg.kernel, cfg.multicell, ctx=ctx.step, go_owner=gOut)
        # '+mixed' kernels also return the gradient w.r.t. grid here (u_xxx, u_xxy): the reference has none
        # (modules_2d.py:111).  Terms through gOutInput are not propagated, as everywhere at this level.
        gGrid3 = None
        if (cfg.kernel & ops.EXACT_MIXED) and gOutGrid is not None and _engine_wants(ctx, 1):
            gO3, hO3 = _f32(gOut), _f32(hO)
            if cfg.sum_n:      # the summed op: one cotangent for every table, the (1,..) grid's gradient summed over n by ops.bbb_grid
                N = input.shape[0]
                gO3 = gO3.expand((N,) + tuple(gO3.shape[1:]))
                hO3 = None if hO3 is None else hO3.expand((N,) + tuple(hO3.shape[1:]))
            gGrid3 = _as(ops.bbb_grid(_f32(input), _f32(grid), gO3, _f32(gOutGrid), hG, hO3, ctx.offset, cfg.pad,
                                      cfg.align_corners, cfg.kernel, cfg.multicell), grid)
        return _as(gInput, input), gGrid3, _as(ggOut, gOut), None, None, None, None, None, None
