"""CPU, world_size 2 and 3, gloo: the sample-sharded path.  Each rank runs the op on its slice of the
points (kernels replaced by the CPU oracle), grad_input partial sums meet in ONE all-reduce, and
everything must equal the unsharded run.  The GPU job uses backend "nccl" (= RCCL) instead."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cosinesampler_amd.dist import point_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_point_range_partitions_exactly():
    for P in (0, 1, 7, 8, 1000, 1 << 20):
        for w in (1, 2, 3, 8):
            spans = [point_range(P, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == P
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, d, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_backend
        from cosinesampler_amd import CosineSampler2d, CosineSampler3d, ops
        from cosinesampler_amd.dist import all_reduce_grad_, gather_points, shard_grid
        for name in ("forward", "backward", "backward_backward", "bbb_fused"):
            setattr(ops, name, getattr(oracle_backend, name))
        Fn = CosineSampler2d if d == 2 else CosineSampler3d

        g = torch.Generator().manual_seed(123)     # same data on every rank
        N, C, S, P = 3, 4, 8, 101                  # P odd: uneven shards
        cells = torch.rand((N, C) + (S,) * d, generator=g)
        grid = torch.rand((N,) + (1,) * (d - 1) + (P, d), generator=g) * 2 - 1
        wts = torch.randn((N, C) + (1,) * (d - 1) + (P,), generator=g)

        def loss_and_grad(cells_, grid_, w_):
            cells_ = cells_.clone().requires_grad_(True)
            grid_ = grid_.clone().requires_grad_(True)
            out = Fn.apply(cells_, grid_, "zeros", True, "cosine", True)
            (gg,) = torch.autograd.grad((out * w_).sum(), grid_, create_graph=True)
            loss = (out * w_).sum() + (gg ** 2).sum()          # touches fwd, bwd, bb
            (gc,) = torch.autograd.grad(loss, cells_)
            return out.detach(), gc

        full_out, full_gc = loss_and_grad(cells, grid, wts)
        lo, hi = point_range(P, rank, world)
        loc_grid = shard_grid(grid, rank, world)
        loc_out, loc_gc = loss_and_grad(cells, loc_grid, wts[..., lo:hi].contiguous())
        # the overlapped form: three input-shaped gradients, each reduced asynchronously, one wait, their sum
        from cosinesampler_amd.dist import GradReducer
        red = GradReducer()
        for a in (0.5, 0.3, 0.2):
            red.push(loc_gc * a)
        red_total = red.finish(out=torch.empty_like(loc_gc))
        # the other schedule: summed locally, ONE all-reduce per step
        once = GradReducer(schedule="once")
        for a in (0.5, 0.3, 0.2):
            once.push(loc_gc * a)
        once_total = once.finish(out=torch.empty_like(loc_gc))
        empty = GradReducer().finish(out=torch.full_like(loc_gc, 7.0))   # nothing pushed: the sum of nothing
        all_reduce_grad_(loc_gc)                                 # the plain form: one collective
        out = gather_points(loc_out)
        ok = (torch.allclose(out, full_out, rtol=0, atol=0)
              and float((loc_gc - full_gc).abs().max()) <= 1e-5 * float(full_gc.abs().max())
              and float((red_total - full_gc).abs().max()) <= 1e-5 * float(full_gc.abs().max())
              and float((once_total - full_gc).abs().max()) <= 1e-5 * float(full_gc.abs().max())
              and float(empty.abs().max()) == 0.0)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("d,world", [(2, 2), (3, 2), (2, 3)])
def test_sharded_equals_unsharded(d, world):
    """world 2 in 2D and 3D; world 3 with 101 points: shards of 34 / 34 / 33."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, d, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(world))
    assert res == [(r, True) for r in range(world)]
