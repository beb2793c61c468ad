"""cosinesampler_amd -- MI355X-native (gfx950) CosineSampler: 2D/3D grid interpolation with cosine /
linear / smoothstep blending and analytic first, second and third backward passes.

Python here is the host side only: a torch.autograd.Function chain (functions.py) that calls
hand-written HIP kernels through a C ABI (include/cosine_sampler.h, csrc/).  There is no CPU or
PyTorch fallback: importing works anywhere, running needs the built library and a GPU.
"""
from .functions import (CosineSampler2d, CosineSampler2dSum, CosineSampler3d, CosineSampler3dSum, kernel_enum, multicell_offset,
                        padding_mode_enum)

__all__ = ["CosineSampler2d", "CosineSampler3d", "CosineSampler2dSum", "CosineSampler3dSum", "padding_mode_enum", "kernel_enum",
           "multicell_offset"]
