"""Build libcosine_sampler_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m cosinesampler_amd.build [--force]

The library is built IN-TREE (cosinesampler_amd/lib/) so that it travels with the source tree;
it has no dependency on torch, only on the HIP runtime (libamdhip64).  Every translation unit is
compiled to its own object (in parallel, re-used while none of its sources changed) and the objects
are linked into the one shared library: each unit's device code is self-contained, only host
functions cross units (csrc/cs_units.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libcosine_sampler_hip.so")
SOURCES = ["cs_abi.hip", "cs_coherent.hip", "cs_coherent_sum.hip", "cs_sort.hip"]
ARCH = "gfx950"
# cs_coherent without the SLP vectoriser: it packs the per-sample fma chains into v_pk_fma_f32 and keeps a splatted copy
# of every multiplier alive for them (coherent forward: 92 -> 168 registers per lane and spills; MI355X_MICROARCH.md
# "packed f32 VALU").  The tiled / 3D kernels of cs_abi keep it: A/B on one box, headline step 5.43 ms with, 5.52 without
# (profiles/round3_ablation.txt).
COMMON_FLAGS = []
EXTRA_FLAGS = {"cs_coherent.hip": ["-fno-slp-vectorize"], "cs_coherent_sum.hip": ["-fno-slp-vectorize"]}


def _deps():
    out = [os.path.join(HERE, "..", "include", "cosine_sampler.h")]
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".cuh", ".h", ".map")):
            out.append(os.path.join(CSRC, f))
    return out


def _unit_deps(src):
    """Headers a unit includes, followed transitively (quoted includes only: the repo's own files)."""
    seen, todo = set(), [os.path.join(CSRC, src)]
    while todo:
        p = os.path.normpath(todo.pop())
        if p in seen or not os.path.exists(p):
            continue
        seen.add(p)
        for line in open(p, errors="replace"):
            line = line.strip()
            if line.startswith("#include \""):
                todo.append(os.path.join(os.path.dirname(p), line.split('"')[1]))
    return seen


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _compile(src, hipcc, force, verbose):
    obj = _obj(src)
    if not force and os.path.exists(obj):
        t = os.path.getmtime(obj)
        if all(os.path.getmtime(p) <= t for p in _unit_deps(src)):
            return obj
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", "-Wall", "-Wno-unused-function"]
    cmd += COMMON_FLAGS + EXTRA_FLAGS.get(src, []) + ["-o", obj, os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, hipcc, force, verbose), SOURCES))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
           "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
