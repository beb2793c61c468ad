"""ctypes front-end of oracle/cs_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under cosinesampler_amd/ does.  It runs the plain-C CPU
restatement of the reference kernels (see cs_oracle.c for the file:line map) on
contiguous fp32 CPU tensors and returns fresh CPU tensors.

Function names and argument order mirror the reference's pybind module
(_cosine_2d / _cosine_3d, reference cosine_sampler_2d/csrc/cosine_sampler_2d.cpp:130-135).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcs_oracle.so")
_SO_DACC = os.path.join(_HERE, "_build", "libcs_oracle_dacc.so")
_lib = None
_lib_dacc = None
# True: the input-shaped gradients come from the build that sums them in double (make dacc) -- for comparisons on crowded
# tables, where the checker's own serial fp32 rounding would otherwise be the tolerance; set through double_accumulation()
_use_dacc = False

_f = ctypes.POINTER(ctypes.c_float)
_i64 = ctypes.c_int64
_int = ctypes.c_int


def build(force=False):
    """Compile the C restatement with gcc (seconds): the plain build and the double-accumulating one."""
    src = os.path.join(_HERE, "cs_oracle.c")
    for so, target in ((_SO, "all"), (_SO_DACC, "dacc")):
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "-B", target])
    return _SO


_NAMES = ("cs2d_forward_cpu", "cs2d_backward_cpu", "cs2d_backward_backward_cpu",
          "cs2d_backward_backward_backward_cpu", "cs3d_forward_cpu", "cs3d_backward_cpu",
          "cs3d_backward_backward_cpu", "cs3d_backward_backward_backward_cpu")


def lib():
    global _lib, _lib_dacc
    if _lib is None:
        build()
        _lib, _lib_dacc = ctypes.CDLL(_SO), ctypes.CDLL(_SO_DACC)
        for l in (_lib, _lib_dacc):
            for name in _NAMES:
                getattr(l, name).restype = _int
    return _lib_dacc if _use_dacc else _lib


class double_accumulation(object):
    """with cs_oracle.double_accumulation(): ... -- the input-shaped gradients of the calls inside are summed in double
    (same fp32 terms, one rounding at the end): the checker for crowded tables."""

    def __enter__(self):
        global _use_dacc
        self._was, _use_dacc = _use_dacc, True

    def __exit__(self, *exc):
        global _use_dacc
        _use_dacc = self._was


def _p(t):
    if t is None:
        return ctypes.cast(None, _f)
    assert t.device.type == "cpu" and t.dtype == torch.float32 and t.is_contiguous(), "oracle wants contiguous fp32 CPU"
    return ctypes.cast(t.data_ptr(), _f)


def _dims(input, grid):
    """-> (dim, N, C, spatial sizes (slow..fast), P)"""
    dim = input.dim() - 2
    assert dim in (2, 3) and grid.shape[-1] == dim and grid.shape[0] == input.shape[0]
    P = 1
    for s in grid.shape[1:-1]:
        P *= int(s)
    return dim, int(input.shape[0]), int(input.shape[1]), [int(s) for s in input.shape[2:]], P


def _tail(sp, P, pad, align, kernel, multicell):
    return [_i64(s) for s in sp] + [_i64(P), _int(pad), _int(int(align)), _int(kernel), _int(int(multicell))]


def forward(input, grid, offset, padding_mode, align_corners, kernel, multicell):
    dim, N, C, sp, P = _dims(input, grid)
    out = torch.empty((N, C) + tuple(grid.shape[1:-1]), dtype=torch.float32)
    fn = getattr(lib(), "cs%dd_forward_cpu" % dim)
    rc = fn(_p(input), _p(grid), _p(offset), _p(out), _i64(N), _i64(C),
            *_tail(sp, P, padding_mode, align_corners, kernel, multicell))
    assert rc == 0
    return out


def backward(grad_output, input, grid, offset, padding_mode, align_corners, input_requires_grad, kernel, multicell):
    dim, N, C, sp, P = _dims(input, grid)
    gi = torch.empty_like(input) if input_requires_grad else None
    gg = torch.empty_like(grid)
    fn = getattr(lib(), "cs%dd_backward_cpu" % dim)
    rc = fn(_p(grad_output), _p(input), _p(grid), _p(offset), _p(gi), _p(gg), _i64(N), _i64(C),
            *_tail(sp, P, padding_mode, align_corners, kernel, multicell))
    assert rc == 0
    return gi, gg


def backward_backward(grad_out_input, grad_out_grid, input, grid, grad_output, offset, padding_mode, align_corners,
                      input_requires_grad, kernel, multicell):
    """grad_out_input is only read when input_requires_grad (reference 2d.cu:654-656)."""
    dim, N, C, sp, P = _dims(input, grid)
    gi = torch.empty_like(input)
    gg = torch.empty_like(grid)
    ggo = torch.empty_like(grad_output)
    fn = getattr(lib(), "cs%dd_backward_backward_cpu" % dim)
    rc = fn(_p(grad_out_input if input_requires_grad else None), _p(grad_out_grid), _p(input), _p(grid),
            _p(grad_output), _p(offset), _p(gi), _p(gg), _p(ggo), _i64(N), _i64(C),
            *_tail(sp, P, padding_mode, align_corners, kernel, multicell))
    assert rc == 0
    return gi, gg, ggo


def backward_backward_backward(input, grid, gOut, gOutGrid, gOutgGrid, offset, padding_mode, align_corners,
                               input_requires_grad, kernel, multicell):
    dim, N, C, sp, P = _dims(input, grid)
    gi = torch.empty_like(input)
    ggo = torch.empty_like(gOut)
    fn = getattr(lib(), "cs%dd_backward_backward_backward_cpu" % dim)
    rc = fn(_p(input), _p(grid), _p(gOut), _p(gOutGrid), _p(gOutgGrid), _p(offset), _p(gi), _p(ggo),
            _i64(N), _i64(C), *_tail(sp, P, padding_mode, align_corners, kernel, multicell))
    assert rc == 0
    return gi, ggo


def bbb_fused(input, grid, gOut, gOutGrid, gOutgGrid, gOutggOut, offset, padding_mode, align_corners, kernel,
              multicell):
    """What CosineSamplerBackwardBackward.backward returns (reference mod2d.py:98-111, mod3d.py:87-100):
    K4/K8, plus the gInput of a second K3/K7 run with gOut := gOutggOut and gOutInput := ones."""
    gi, ggo = backward_backward_backward(input, grid, gOut, gOutGrid, gOutgGrid, offset, padding_mode,
                                         align_corners, True, kernel, multicell)
    b_input, _, _ = backward_backward(torch.ones_like(input), gOutGrid, input, grid, gOutggOut, offset,
                                      padding_mode, align_corners, True, kernel, multicell)
    return gi + b_input, ggo
