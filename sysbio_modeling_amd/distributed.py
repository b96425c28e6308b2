"""Ensemble sharding over the GPUs of one node: one process per GPU (torch.distributed; backend
'nccl' is RCCL over xGMI on ROCm, 'gloo' in the CPU tests).

The hot path shards by parameter vector with NO data-path collective (every vector's trajectories,
scale factors, residuals and Jacobian are independent).  The only exchange is the all-gather of the
per-vector residual norms -- what a multi-start fit / multi-chain sampler needs to rank its members
(the reference's dead pypar scatter of ensemble members, project/Ensembles.py:310-327, is the
closest thing it has).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_vectors, rank, world):
    """Contiguous block of the vector axis owned by ``rank``: [lo, hi).  Blocks differ by at most one."""
    base, extra = divmod(int(n_vectors), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_norms(local_norms, n_vectors=None, group=None):
    """All-gather per-vector values (1-d tensor on this rank's device) into global vector order.

    Ranks may own blocks that differ by one vector: blocks are padded to a common length for the
    collective and trimmed afterwards.  Returns a 1-d tensor of length ``n_vectors`` on every rank.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_norms
    world = dist.get_world_size(group)
    n_local = int(local_norms.shape[0])
    if n_vectors is None:
        counts = torch.tensor([n_local], dtype=torch.int64, device=local_norms.device)
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=group)
        sizes = [int(c.item()) for c in all_counts]
    else:
        sizes = [shard_range(n_vectors, r, world)[1] - shard_range(n_vectors, r, world)[0] for r in range(world)]
    width = max(sizes)
    padded = local_norms.new_full((width,), float('nan'))
    padded[:n_local] = local_norms
    out = local_norms.new_empty((world * width,))
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * width:r * width + sizes[r]] for r in range(world)])


def evaluate_sharded(evaluate_norms, thetas, group=None, device=None):
    """Evaluate this rank's block of ``thetas`` (V, q) with ``evaluate_norms(block) -> 1-d tensor`` and
    return (global norms for all V vectors, (lo, hi) owned by this rank)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    V = int(thetas.shape[0])
    lo, hi = shard_range(V, rank, world)
    local = evaluate_norms(thetas[lo:hi])
    if not isinstance(local, torch.Tensor):
        local = torch.as_tensor(np.asarray(local), dtype=torch.float64, device=device)
    return gather_norms(local, V, group), (lo, hi)


def project_norms_evaluator(project, **integrator_overrides):
    """evaluate_norms callback for ``evaluate_sharded``: residual sum of squares on this rank's GPU."""
    import torch

    def run(block):
        if block.shape[0] == 0:
            return torch.empty((0,), dtype=torch.float64, device='cuda')
        th = block if isinstance(block, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(block))
        th = th.cuda(project._model.device_model.ctx.device)
        return project.evaluate_batch(th, **integrator_overrides)['norms']
    return run
