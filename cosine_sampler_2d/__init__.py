# Drop-in import path of the reference package (`from cosine_sampler_2d import CosineSampler2d`,
# reference cosine_sampler_2d/__init__.py:1); the implementation lives in cosinesampler_amd.
from cosinesampler_amd import CosineSampler2d

__all__ = ["CosineSampler2d"]
