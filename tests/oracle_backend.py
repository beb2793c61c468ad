"""TEST-ONLY: swap cosinesampler_amd.ops for the CPU oracle so that the host logic (the autograd
Function chain, None handling, sharding helpers) can be exercised without a GPU.  The product
never imports this; on a GPU box the parity tests call the real ops and compare with the oracle."""
import torch

from oracle import cs_oracle


def _f(t):
    return None if t is None else t.contiguous()


def _g(t, input):
    """a grid-shaped tensor with a leading 1 = one set of points for every n (ops.grid_is_broadcast): the oracle, like the
    reference, wants it repeated"""
    if t is None:
        return None
    if t.shape[0] == 1 and input.shape[0] > 1:
        return t.expand((input.shape[0],) + tuple(t.shape[1:])).contiguous()
    return t.contiguous()


def _r(t, grid, input):
    """... and the gradient w.r.t. the shared points is the sum over n"""
    return t.sum(0, keepdim=True) if (grid.shape[0] == 1 and input.shape[0] > 1) else t


def forward(input, grid, offset, padding_mode, align_corners, kernel, multicell, ctx=None, out_dtype=None):
    if kernel not in (0, 1, 2):
        raise TypeError("kernel enum")
    return cs_oracle.forward(_f(input), _g(grid, input), offset, padding_mode, align_corners, kernel, multicell)


def backward(grad_output, input, grid, offset, padding_mode, align_corners, input_requires_grad, kernel, multicell,
             ctx=None, go_owner=None):
    gi, gg = cs_oracle.backward(_f(grad_output), _f(input), _g(grid, input), offset, padding_mode, align_corners,
                                input_requires_grad, kernel, multicell)
    return gi, _r(gg, grid, input)


def backward_backward(grad_out_input, grad_out_grid, input, grid, grad_output, offset, padding_mode, align_corners,
                      input_requires_grad, kernel, multicell, ctx=None, want_grad_input=True, go_owner=None):
    if grad_out_grid is None:
        grad_out_grid = torch.zeros_like(grid)
    gi, gg, ggo = cs_oracle.backward_backward(_f(grad_out_input), _g(grad_out_grid, input), _f(input), _g(grid, input),
                                              _f(grad_output), offset, padding_mode, align_corners,
                                              input_requires_grad, kernel, multicell)
    SKIPPED.append(not want_grad_input)
    return (gi if want_grad_input else None), _r(gg, grid, input), ggo


def bbb_fused(input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset, padding_mode,
              align_corners, kernel, multicell, ctx=None, go_owner=None):
    z = torch.zeros_like(grid)
    return cs_oracle.bbb_fused(_f(input), _g(grid, input), _f(grad_output),
                               _g(z if grad_out_grid is None else grad_out_grid, input),
                               _g(z if grad_out_ggrid is None else grad_out_ggrid, input),
                               torch.zeros_like(grad_output) if grad_out_ggout is None else _f(grad_out_ggout),
                               offset, padding_mode, align_corners, kernel, multicell)


# the summed op (ops.*_sum_n, CosineSampler{2,3}dSum): n-free cotangents expanded, per-point results summed -- the reference's
# own way of writing the PIXEL pattern
def _x(t, input):
    return None if t is None else t.expand((input.shape[0],) + tuple(t.shape[1:]))


def forward_sum_n(input, grid, offset, padding_mode, align_corners, kernel, multicell, ctx=None):
    return forward(input, grid, offset, padding_mode, align_corners, kernel, multicell).sum(0, keepdim=True)


def backward_sum_n(grad_output, input, grid, offset, padding_mode, align_corners, input_requires_grad, kernel, multicell,
                   ctx=None):
    return backward(_x(grad_output, input), input, grid, offset, padding_mode, align_corners, input_requires_grad, kernel,
                    multicell)


def backward_backward_sum_n(grad_out_grid, input, grid, grad_output, offset, padding_mode, align_corners, kernel, multicell,
                            ctx=None, want_grad_input=True):
    gi, gg, ggo = backward_backward(None, grad_out_grid, input, grid, _x(grad_output, input), offset, padding_mode,
                                    align_corners, False, kernel, multicell, want_grad_input=want_grad_input)
    return gi, gg, ggo.sum(0, keepdim=True)


def bbb_fused_sum_n(input, grid, grad_output, grad_out_grid, grad_out_ggrid, grad_out_ggout, offset, padding_mode,
                    align_corners, kernel, multicell, ctx=None):
    gi, ggo = bbb_fused(input, grid, _x(grad_output, input), grad_out_grid, grad_out_ggrid, _x(grad_out_ggout, input), offset,
                        padding_mode, align_corners, kernel, multicell)
    return gi, ggo.sum(0, keepdim=True)


SKIPPED = []   # per backward_backward call: was grad_input declared unwanted?


def install(monkeypatch):
    from cosinesampler_amd import ops
    for name in ("forward", "backward", "backward_backward", "bbb_fused", "forward_sum_n", "backward_sum_n",
                 "backward_backward_sum_n", "bbb_fused_sum_n"):
        monkeypatch.setattr(ops, name, globals()[name])
