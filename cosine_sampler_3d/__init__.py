# Drop-in import path of the reference package (`from cosine_sampler_3d import CosineSampler3d`,
# reference cosine_sampler_3d/__init__.py:1); the implementation lives in cosinesampler_amd.
from cosinesampler_amd import CosineSampler3d

__all__ = ["CosineSampler3d"]
