#!/usr/bin/env python3
"""Randomised parity sweep: every stage on the GPU (through the C ABI, every execution path) against the CPU oracle on
random small problems -- shapes with ragged tails, tiny tables, all paddings / kernels / flags, out-of-range points.
    python tools/fuzz_parity.py [cases] [seed]
Prints one line per failure and a summary; exit code 1 if anything differed by more than 1e-5 (relative to the
largest reference magnitude of the tensor, as tests/helpers.py)."""
import os
import random
import sys

os.environ.setdefault("COSINESAMPLER_DEBUG", "1")      # the sweep forces execution paths (ops.force_path): a testing knob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

from cosinesampler_amd import multicell_offset, ops
from oracle import cs_oracle

DEV = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)


SCALE = {}


def rel(a, b, scale=0.0):
    a, b = a.double().cpu(), b.double().cpu()
    # inputs are O(1): an output whose largest entry is far below that (a handful of points at the edge of the table)
    # is compared on the scale of the data, not of its own cancellation (seed 7 case 394: 2 points, max|out| 3e-3)
    den = max(float(b.abs().max()), 0.05, scale)
    return float((a - b).abs().max()) / den


GRID_SHAPED = ("grid", "cG", "hG")
GRID_RESULTS = ("gG", "gG0", "bbG", "bG0")


def run(mod, t, off, pad, align, ke, mc, dev, shared, bc=False, acc=False):
    """bc: one set of points for every n.  The product (ops) is handed (1, ..., dim) tensors; the oracle, like the
    reference, the repeated ones, and its grid-shaped results are summed over n."""
    x = {k: v.to(dev) for k, v in t.items()}
    if bc:
        N = x["inp"].shape[0]
        for k in GRID_SHAPED:
            x[k] = x[k][:1].contiguous()
            if mod is cs_oracle:
                x[k] = x[k].repeat((N,) + (1,) * (x[k].dim() - 1))
    off = off.to(dev)
    kw = {}
    if shared:
        # acc (round 4): an ACCUMULATING context -- the five scatter stages below add into one step accumulator
        # (cs_cotangent_layout.accumulate_grad_input, or the context's fallback sum) and return None for grad_input
        kw["ctx"] = ops.StepContext(accumulate=acc)
    r = {}
    r["out"] = mod.forward(x["inp"], x["grid"], off, pad, align, ke, mc, **kw)
    r["gI"], r["gG"] = mod.backward(x["gOut"], x["inp"], x["grid"], off, pad, align, True, ke, mc, **kw)
    r["gG0"] = mod.backward(x["gOut"], x["inp"], x["grid"], off, pad, align, False, ke, mc, **kw)[1]
    r["bbI"], r["bbG"], r["bbO"] = mod.backward_backward(x["cI"], x["cG"], x["inp"], x["grid"], x["gOut"], off, pad, align,
                                                         True, ke, mc, **kw)
    r["bI0"], r["bG0"], r["bO0"] = mod.backward_backward(None, x["cG"], x["inp"], x["grid"], x["gOut"], off, pad, align,
                                                         False, ke, mc, **kw)
    r["k4I"], r["k4O"] = mod.backward_backward_backward(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], off, pad, align,
                                                        True, ke, mc, **kw)
    r["fI"], r["fO"] = mod.bbb_fused(x["inp"], x["grid"], x["gOut"], x["cG"], x["hG"], x["hO"], off, pad, align, ke, mc, **kw)
    if acc:
        parts = ("gI", "bbI", "bI0", "k4I", "fI")
        if mod is cs_oracle:
            r["accI"] = sum(r[k].double() for k in parts).float()
        else:
            assert all(r[k] is None for k in parts), "an accumulating context returns no grad_input"
            r["accI"] = kw["ctx"].grad_input_sum()
            for k in parts:
                del r[k]
    if bc and mod is cs_oracle:
        for k in GRID_RESULTS:
            SCALE[k] = float(r[k].abs().max())     # the summands' magnitude: a sum over n may cancel (case 544 of seed 77:
            r[k] = r[k].sum(0, keepdim=True)       # terms of +-12 summing to 0.06), its error is that of the terms
    return r


def run_sum_n(mod, t, off, pad, align, ke, mc, dev, shared):
    """the summed op (ops.*_sum_n; CS_SUM_OVER_N where it applies, else the plain op + sums): one set of points and ONE
    cotangent for every table, per-point results summed over the tables.  Oracle: the reference's way -- repeat, expand, sum."""
    x = {k: v.to(dev) for k, v in t.items()}
    N = x["inp"].shape[0]
    one = {k: x[k][:1].contiguous() for k in ("grid", "cG", "hG", "gOut", "hO")}
    off = off.to(dev)
    r = {}
    if mod is cs_oracle:
        rep = {k: v.repeat((N,) + (1,) * (v.dim() - 1)).contiguous() for k, v in one.items()}
        s0 = lambda a: a.sum(0, keepdim=True)
        r["s_out"] = s0(mod.forward(x["inp"], rep["grid"], off, pad, align, ke, mc))
        gI, gG = mod.backward(rep["gOut"], x["inp"], rep["grid"], off, pad, align, True, ke, mc)
        r["s_gI"], r["s_gG"] = gI, s0(gG)
        bI, bG, bO = mod.backward_backward(None, rep["cG"], x["inp"], rep["grid"], rep["gOut"], off, pad, align, False, ke, mc)
        r["s_bbI"], r["s_bbG"], r["s_bbO"] = bI, s0(bG), s0(bO)
        fI, fO = mod.bbb_fused(x["inp"], rep["grid"], rep["gOut"], rep["cG"], rep["hG"], rep["hO"], off, pad, align, ke, mc)
        r["s_fI"], r["s_fO"] = fI, s0(fO)
        for k, v in (("s_gG", gG), ("s_bbG", bG), ("s_bbO", bO), ("s_fO", fO), ("s_out", None)):
            if v is not None:
                SCALE[k] = float(v.abs().max())
        return r
    kw = {"ctx": ops.StepContext()} if shared else {}
    r["s_out"] = ops.forward_sum_n(x["inp"], one["grid"], off, pad, align, ke, mc, **kw)
    r["s_gI"], r["s_gG"] = ops.backward_sum_n(one["gOut"], x["inp"], one["grid"], off, pad, align, True, ke, mc, **kw)
    r["s_bbI"], r["s_bbG"], r["s_bbO"] = ops.backward_backward_sum_n(one["cG"], x["inp"], one["grid"], one["gOut"], off, pad,
                                                                    align, ke, mc, **kw)
    r["s_fI"], r["s_fO"] = ops.bbb_fused_sum_n(x["inp"], one["grid"], one["gOut"], one["cG"], one["hG"], one["hO"], off, pad,
                                               align, ke, mc, **kw)
    return r


bad = 0
for case in range(cases):
    d = rng.choice([2, 2, 3])
    C = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 31, 32, 33, 48, 70] if d == 2 else [1, 2, 3, 4, 5, 6, 8, 12, 16, 17, 24, 40])
    N = rng.choice([1, 2, 3, 5, 8, 16])       # (8, 16: the XCD-aware workgroup order of the tiled point kernels)
    sp = tuple(rng.choice([2, 3, 5, 16, 17, 18, 33, 40]) for _ in range(d)) if d == 2 else \
        tuple(rng.choice([2, 3, 5, 8, 9, 16]) for _ in range(d))
    P = rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 3001, 5000, 20000])
    pad, align, ke, mc = rng.choice([0, 1, 2]), rng.choice([True, False]), rng.choice([0, 1, 2]), rng.choice([True, False])
    force = rng.choice([0, 1, 2, 2, 2, 3, 4])
    shared = rng.choice([True, False])
    bc = N > 1 and rng.random() < 0.25
    spread = rng.choice([0.9, 1.0, 1.3])
    # round 3: the order of the points (as drawn / in cell order / blocks of the ordered set shuffled) and how the op is told
    # about it (the hint forced either way, or left to the op's own measurement); only the time may depend on either
    order = rng.choice(["drawn", "drawn", "sorted", "sorted", "half"])
    hint = rng.choice(["auto", "coherent", "random"])
    acc = shared and rng.random() < 0.4          # round 4: the scatter stages add into one step accumulator
    if os.environ.get("FUZZ_ONLY") and case != int(os.environ["FUZZ_ONLY"]):
        continue                      # (every random draw of the case is above: case k is the same problem as in a full run)
    g = torch.Generator().manual_seed(seed * 100003 + case)
    inp = torch.rand((N, C) + sp, generator=g)
    grid = torch.rand((N,) + (1,) * (d - 1) + (P, d), generator=g) * (2 * spread) - spread
    if P >= 3:
        grid.view(N, P, d)[:, 0] = -1.0
        grid.view(N, P, d)[:, 1] = 1.0
    if order != "drawn" and P >= 2:
        for n in range(N):              # every table's point set ordered by cell (table 0's cells: the multicell shift is < 1 cell)
            srt, _ = ops.sort_points(grid.view(N, P, d)[n].contiguous().to(DEV), sp, pad, align, mc)
            srt = srt.cpu()
            if order == "half":
                nb = min(P, 7)
                bounds = [P * i // nb for i in range(nb + 1)]
                srt = torch.cat([srt[bounds[b]:bounds[b + 1]] for b in torch.randperm(nb, generator=g).tolist()])
            grid.view(N, P, d)[n] = srt
    osh = (N, C) + (1,) * (d - 1) + (P,)
    t = dict(inp=inp, grid=grid, gOut=torch.randn(osh, generator=g), cI=torch.randn(inp.shape, generator=g),
             cG=torch.randn(grid.shape, generator=g), hG=torch.randn(grid.shape, generator=g), hO=torch.randn(osh, generator=g))
    off = multicell_offset(N, mc, "cpu")
    SCALE.clear()
    want = run(cs_oracle, t, off, pad, align, ke, mc, "cpu", False, bc, acc)
    if bc:
        want.update(run_sum_n(cs_oracle, t, off, pad, align, ke, mc, "cpu", False))
    ops.force_path(force)
    ops.points_order(hint)
    try:
        got = run(ops, t, off, pad, align, ke, mc, DEV, shared, bc, acc)
        if bc:
            got.update(run_sum_n(ops, t, off, pad, align, ke, mc, DEV, shared))
        torch.cuda.synchronize()
    finally:
        ops.force_path(0)
        ops.points_order("auto")
    errs = {k: rel(got[k], want[k], SCALE.get(k, 0.0)) for k in want if k in got}
    if os.environ.get("FUZZ_ONLY") or os.environ.get("FUZZ_DETAIL") == str(case):
        for k in want:
            print(k, "%.3e" % errs[k], "max|ref| %.4g" % float(want[k].abs().max()), flush=True)
        w = max(errs, key=errs.get)
        print("got ", got[w].flatten()[:12].tolist())
        print("want", want[w].flatten()[:12].tolist())
    worst = max(errs, key=errs.get)
    if not all(torch.isfinite(v).all() for v in got.values()) or errs[worst] > 1e-5:
        bad += 1
        print("FAIL case %d: d=%d N=%d C=%d sp=%s P=%d pad=%d align=%s kernel=%d mc=%s force=%d shared=%s bc=%s order=%s hint=%s acc=%s -> %s %.3e"
              % (case, d, N, C, sp, P, pad, align, ke, mc, force, shared, bc, order, hint, acc, worst, errs[worst]), flush=True)
    elif case % 25 == 0:
        print("ok   case %d (d=%d C=%d sp=%s P=%d force=%d %s/%s%s) worst %s %.1e" % (case, d, C, sp, P, force, order, hint,
                                                                                 " acc" if acc else "", worst, errs[worst]), flush=True)
print("%d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
