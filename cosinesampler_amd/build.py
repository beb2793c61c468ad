"""Build libcosine_sampler_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m cosinesampler_amd.build [--force]

The library is built IN-TREE (cosinesampler_amd/lib/) so that it travels with the source tree;
it has no dependency on torch, only on the HIP runtime (libamdhip64).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libcosine_sampler_hip.so")
SOURCES = ["cs_abi.hip"]
ARCH = "gfx950"


def _deps():
    out = [os.path.join(HERE, "..", "include", "cosine_sampler.h")]
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".cuh", ".h")):
            out.append(os.path.join(CSRC, f))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
