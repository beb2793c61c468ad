"""Shared test helpers: fixture loading, the parity metric, stage drivers."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# SURVEY.md section 8(d) "Parity metric": max|a-b| / max|b| <= 1e-5 per output tensor in fp32
# (north_star: "cosine/smoothstep and all higher grads within 1e-5 fp32").
REL_TOL = 1e-5

PAD = {"zeros": 0, "border": 1, "reflection": 2}
KERNEL_ENUM = {"cosine": 0, "bilinear": 1, "trilinear": 1, "smooth-step": 2}


def rel_err(a, b):
    a = torch.as_tensor(a).detach().to("cpu", torch.float64).reshape(-1)
    b = torch.as_tensor(b).detach().to("cpu", torch.float64).reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    den = float(b.abs().max())
    num = float((a - b).abs().max())
    if den == 0.0:
        return num
    return num / den


def assert_close(a, b, what, tol=REL_TOL):
    e = rel_err(a, b)
    assert e <= tol, "%s: max|a-b|/max|b| = %.3e > %.1e" % (what, e, tol)


def stage_fixtures(dim=None):
    pats = sorted(glob.glob(os.path.join(GOLDEN, "stage_*.npz")))
    out = []
    for p in pats:
        name = os.path.basename(p)[:-4]
        d = int(name.split("_")[1][0])
        if dim is None or d == dim:
            out.append(name)
    return out


def parse_stage_name(name):
    """stage_2d_cosine_mc -> (2, 'cosine', True); stage_2d_config1 -> (2, 'bilinear', False)"""
    parts = name.split("_")
    d = int(parts[1][0])
    if parts[2] == "config1":
        return d, "bilinear", False
    k = {"cosine": "cosine", "smoothstep": "smooth-step", "bilinear": "bilinear", "trilinear": "trilinear"}[parts[2]]
    return d, k, parts[3] == "mc"


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def offsets(N, multicell, device="cpu"):
    """offset[n] as the reference's host code builds it (mod2d.py:24-27): torch.linspace bits."""
    if multicell:
        return torch.linspace(0, 1 - (1 / N), N).to(device)
    return torch.zeros(N).to(device)


def axis_only(t, j):
    o = torch.zeros_like(t)
    o[..., j] = t[..., j]
    return o
