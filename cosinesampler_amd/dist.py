"""Sample-sharded multi-GPU use of the sampler (one process per GPU, torch.distributed / RCCL).

The op is embarrassingly parallel over output samples, so the point axis P shards across ranks
with no data-path exchange: `output`, `grad_grid`, `ggOut`, `gGrid` stay sharded.  The only
cross-rank dependency is every `input`-shaped gradient, for which each rank holds a partial sum:
ONE sum all-reduce per training step on the accumulated `cells.grad`.
The reference has no multi-device code at all; this is new.
"""
import torch
import torch.distributed as dist


def point_range(P, rank, world_size):
    """Contiguous [lo, hi) slice of the P points owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(int(P), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_grid(grid, rank=None, world_size=None):
    """Local shard of `grid` (N, ..., Wo, d) along its last output axis (the PIXEL point axis)."""
    if rank is None:
        rank = dist.get_rank()
    if world_size is None:
        world_size = dist.get_world_size()
    lo, hi = point_range(grid.shape[-2], rank, world_size)
    return grid[..., lo:hi, :].contiguous()


def all_reduce_grad_(grad_input, group=None, async_op=False):
    """In-place SUM of an `input`-shaped gradient over ranks: the single collective of a step.
    backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" works for CPU rehearsals."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(grad_input, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def gather_points(local, group=None):
    """Concatenate per-rank outputs (N, C, ..., P_local) along the point axis (uneven shards ok).
    Not needed by training (the loss reduces over points locally); provided for inspection/tests."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[-1]], dtype=torch.int64, device=local.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    pmax = max(sizes)
    pad = torch.zeros(local.shape[:-1] + (pmax,), dtype=local.dtype, device=local.device)
    pad[..., : local.shape[-1]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[..., :s] for o, s in zip(outs, sizes)], dim=-1)
