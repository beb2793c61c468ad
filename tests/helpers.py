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


def pixel_pipeline(apply_fn, fx, d, device="cpu"):
    """The PIXEL-style use of the op, as the reference tests drive it (test/test_2d.py:26-240,
    test/test_3d.py:19-292): sampler -> sum over cells -> tiny MLP -> u; first / second
    derivatives w.r.t. the point coordinates; d/dcells of each; d loss / d cells.
    `apply_fn(cells, grid)` is the sampler under test.  Returns a dict with the fixture's keys."""
    cells = fx["cells"].to(device).clone().requires_grad_(True)
    N, C = cells.shape[:2]
    names = "xyz"[:d]
    coords = [fx["p_" + nm].to(device).clone().requires_grad_(True) for nm in names]
    W1, b1, W2, b2 = (fx[k].to(device) for k in ("W1", "b1", "W2", "b2"))
    P = coords[0].shape[0]
    grid = torch.cat(coords, -1).view((1,) * d + (P, d)).repeat((N,) + (1,) * (d + 1))
    val = apply_fn(cells, grid)
    feat = val.sum(0).view(C, -1).t()
    u = torch.tanh(feat @ W1.t() + b1) @ W2.t() + b2

    def grad(y, x):
        return torch.autograd.grad(y, x, torch.ones_like(y), retain_graph=True, create_graph=True)[0]

    out = dict(u=u, u_cell=grad(u, cells))
    first, second = [], []
    for j, nm in enumerate(names):
        uj = grad(u, coords[j])
        ujj = grad(uj, coords[j])
        first.append(uj)
        second.append(ujj)
        out["u_" + nm], out["u_" + nm * 2] = uj, ujj
        out["u_%s_cell" % nm], out["u_%s_cell" % (nm * 2)] = grad(uj, cells), grad(ujj, cells)
    if d == 2:
        f = first[1] * 2 + 5 * (u ** 3) - 5 * u - 0.0001 * second[0]
    else:
        f = second[0] + second[1] + second[2] + u
    loss = torch.mean(f ** 2)
    out["dloss"] = torch.autograd.grad(loss, cells)[0]
    return {k: v.detach().cpu() for k, v in out.items()}


PIXEL_KEYS_2D = ["u", "u_cell", "u_x", "u_y", "u_xx", "u_yy", "u_x_cell", "u_y_cell", "u_xx_cell", "u_yy_cell",
                 "dloss"]
PIXEL_KEYS_3D = PIXEL_KEYS_2D[:-1] + ["u_z", "u_zz", "u_z_cell", "u_zz_cell", "dloss"]
