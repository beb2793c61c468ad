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


def all_reduce_grad_(grad_input, group=None, async_op=False, even_alone=False):
    """In-place SUM of an `input`-shaped gradient over ranks: the only kind of collective a step has.
    backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" works for CPU rehearsals.
    `even_alone`: issue the collective in a one-rank group too (tests: the RCCL call path runs on a single GPU)."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_world_size(group) == 1 and not even_alone:
        return None
    return dist.all_reduce(grad_input, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class GradReducer(object):
    """Sums the `input`-shaped gradients of one training step over the stages and over the ranks.

    A step produces up to three of them (first, second and third backward), each ready long before the step ends
    (at config 2: 1.8, 3.0 and 4.7 ms into a 5.3 ms step).  Two schedules:

    * `"per_stage"`: `push(g)` starts `g`'s sum over the ranks right away -- asynchronously, on the
      communicator's own stream, ordered after the kernels that produced `g` -- and returns at once; `finish()` waits
      for every collective started so far (the current stream waits, the host does not) and adds them up.  Three 64 MiB
      all-reduces per step, two of them hidden behind the stages that follow; three times the xGMI bytes.
    * `"once"` (bench.py's default, together with a step accumulator -- ops.StepContext(accumulate=True) -- so that
      what is pushed is already the sum): `push` only remembers `g`; `finish()` adds the gradients up locally and runs ONE all-reduce on the sum
      (what SURVEY 8e words: "one sum all-reduce per training step on the accumulated cells.grad"); a third of the
      bytes, all of it exposed at the end of the step.

    `push(g)` in the per-stage schedule reduces `g` IN PLACE and asynchronously: the caller must not read or modify a
    pushed tensor before `finish()` (nothing orders the caller's stream behind the collective until then).
    `finish(out=)` with nothing pushed zeroes `out` (the sum of nothing) and returns it.
    Without a process group (or alone in it) both schedules only sum."""

    def __init__(self, group=None, even_alone=False, enabled=True, schedule="per_stage"):
        if schedule not in ("per_stage", "once"):
            raise ValueError("schedule must be 'per_stage' or 'once', got %r" % (schedule,))
        self.group = group
        self.even_alone = even_alone
        self.enabled = enabled           # False: only sum locally (bench.py's "without the reduce" timing)
        self.schedule = schedule
        self._pending = []

    def push(self, grad):
        work = None
        if self.enabled and self.schedule == "per_stage":
            work = all_reduce_grad_(grad, self.group, async_op=True, even_alone=self.even_alone)
        self._pending.append((grad, work))
        return grad

    def finish(self, out=None):
        if not self._pending:
            return out.zero_() if out is not None else None
        for _, work in self._pending:
            if work is not None:
                work.wait()            # stream-ordered for RCCL; blocks for gloo
        grads = [g for g, _ in self._pending]
        if out is None:          # (a single gradient -- a step accumulator's sum -- is returned as it is, no copy)
            total = grads[0].clone() if len(grads) > 1 else grads[0]
            rest = grads[1:]
        elif len(grads) == 1:
            total = out.copy_(grads[0])
            rest = []
        else:                          # the first two summed straight into `out`: no copy pass over the buffer
            total = torch.add(grads[0], grads[1], out=out)
            rest = grads[2:]
        for g in rest:
            total.add_(g)
        self._pending = []
        if self.enabled and self.schedule == "once":
            all_reduce_grad_(total, self.group, async_op=False, even_alone=self.even_alone)
        return total


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
