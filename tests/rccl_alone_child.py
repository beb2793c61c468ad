"""Child process of tests/test_parity_gpu.py::test_rccl_path_runs_on_one_gpu: a ONE-rank RCCL ("nccl") process group on
cuda:0 through which the collectives of the multi-GPU job really run -- the asynchronous per-stage all-reduces of
cosinesampler_amd.dist.GradReducer and the timing all-reduce of bench.py.  Prints RCCL_ALONE_OK on success."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from cosinesampler_amd import multicell_offset, ops
        from cosinesampler_amd.dist import GradReducer, all_reduce_grad_, gather_points, shard_grid
        torch.manual_seed(3)
        N, C, H, P = 4, 16, 64, 70000
        cells = torch.rand(N, C, H, H, device=dev)
        grid = shard_grid(torch.rand(N, 1, P, 2, device=dev) * 2 - 1)      # rank 0 of 1: the whole grid
        gOut, hO = torch.randn(N, C, 1, P, device=dev), torch.randn(N, C, 1, P, device=dev)
        cG, hG = torch.randn(N, 1, P, 2, device=dev), torch.randn(N, 1, P, 2, device=dev)
        off = multicell_offset(N, True, dev)
        sc = ops.StepContext()
        red = GradReducer(even_alone=True)
        ops.forward(cells, grid, off, 0, True, 0, True, ctx=sc)
        gI, _ = ops.backward(gOut, cells, grid, off, 0, True, True, 0, True, ctx=sc)
        keep = [gI.clone()]
        red.push(gI)
        bbI, _, _ = ops.backward_backward(None, cG, cells, grid, gOut, off, 0, True, False, 0, True, ctx=sc)
        keep.append(bbI.clone())
        red.push(bbI)
        tI, tO = ops.bbb_fused(cells, grid, gOut, cG, hG, hO, off, 0, True, 0, True, ctx=sc)
        keep.append(tI.clone())
        red.push(tI)
        assert all(w is not None for _, w in red._pending), "the collectives must really have been issued"
        total = red.finish(out=torch.empty_like(cells))
        torch.cuda.synchronize()
        want = keep[0] + keep[1] + keep[2]
        assert torch.equal(total, want), float((total - want).abs().max())   # one rank: the sum over ranks is the identity
        once = GradReducer(even_alone=True, schedule="once")      # the other schedule: one collective on the local sum
        for k in keep:
            once.push(k.clone())
        assert all(w is None for _, w in once._pending)
        total1 = once.finish(out=torch.empty_like(cells))
        torch.cuda.synchronize()
        assert torch.equal(total1, want)
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.25
        assert all_reduce_grad_(total.clone()) is None                         # alone and not forced: no collective
        assert gather_points(tO).shape == tO.shape
        dist.barrier()
        print("RCCL_ALONE_OK", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
