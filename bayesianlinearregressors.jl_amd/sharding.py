"""Sharding of independent regressors across ranks (one process per GPU) -- SURVEY.md 8(e).

The path partitions by regressor: contiguous blocks, no input exchange, no data-path collective.  The single
exchange is at the end: every rank all-gathers the per-regressor log evidences (8 bytes each; 8 KiB per rank
at B = 8192 over 8 GPUs -- latency-bound, xGMI bandwidth irrelevant) and runs the same fixed-order device sum
(`blr_logpdf_sum`) over the concatenated vector, so the total has the same bits on every rank and for every
rank count.  `torch.distributed` (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is plumbing only.
"""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of `total` regressors owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(total: int, world: int) -> list[int]:
    return [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]


def gather_logpdf(local_logpdf, total: int, group=None):
    """All-gather the per-regressor log evidences of every rank into one tensor of length `total`
    (rank order == regressor order).  `local_logpdf`: 1-D float64 tensor of this rank's block."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_logpdf
    world = dist.get_world_size(group)
    sizes = shard_sizes(total, world)
    if len(set(sizes)) == 1:
        out = torch.empty(total, dtype=local_logpdf.dtype, device=local_logpdf.device)
        dist.all_gather_into_tensor(out, local_logpdf.contiguous(), group=group)
        return out
    # uneven blocks: collectives need equal sizes -> pad to the largest block, gather, drop the padding
    nmax = max(sizes)
    padded = torch.zeros(nmax, dtype=local_logpdf.dtype, device=local_logpdf.device)
    padded[: local_logpdf.numel()] = local_logpdf
    out = torch.empty(world * nmax, dtype=local_logpdf.dtype, device=local_logpdf.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * nmax : r * nmax + sizes[r]] for r in range(world)])
