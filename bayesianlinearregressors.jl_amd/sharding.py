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


# ---- one large regressor, observations split over ranks (SURVEY.md 8e "one-exchange N-sharding") ---------------------
def stats_shape(D: int) -> tuple[int, int]:
    """(rows, cols) of the additive statistics matrix of `blr_gram_stats_*`: (DP + 128) x DP, DP = 128 ceil(D / 128)."""
    DP = (D + 127) // 128 * 128
    return DP + 128, DP


def reduce_stats(stats, scal, group=None):
    """The path's single exchange: sum the additive statistics over the ranks, in place (two all-reduces: the matrix in its
    own dtype, the two evidence scalars in float64).  No-op without an initialised process group."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(scal, op=dist.ReduceOp.SUM, group=group)
    return stats, scal


def posterior_n_sharded(handle, X_local, y_local, noise, mw, Lw, n_total: int, group=None):
    """posterior + logpdf of ONE regressor whose columns are split over the ranks (each rank passes its own block).

    Device-resident torch tensors: ``X_local`` D x N_local column-major (i.e. a contiguous ``[N_local, D]`` tensor),
    ``y_local`` [N_local], ``noise`` a 1-element tensor (isotropic) or [N_local] (diagonal), ``mw`` [D], ``Lw`` [D] (diagonal
    precision) or a contiguous symmetric [D, D].  Every rank returns the same ``(mw_post [D], T_post [D, D] (row-major view of the
    column-major upper factor, i.e. T_post.T is U), logpdf float)``.
    """
    import numpy as np
    import torch
    from . import _abi

    dt = np.float64 if X_local.dtype == torch.float64 else np.float32
    n_loc, D = X_local.shape
    rows, cols = stats_shape(D)
    dev = X_local.device
    # The library runs on the handle's stream; torch produced the inputs (and zero-fills the outputs below) on ITS current
    # stream.  Put the handle on that stream for the duration of the call so that every fill / producer is ordered before the
    # library's kernels and the collective sees finished statistics; the handle's previous stream is restored on the way out.
    with _on_torch_stream(handle, dev):
        return _posterior_n_sharded(handle, X_local, y_local, noise, mw, Lw, n_total, group, dt, n_loc, D, rows, cols, dev)


class _on_torch_stream:
    def __init__(self, handle, dev):
        self.handle, self.dev = handle, dev

    def __enter__(self):
        import torch

        self.previous = self.handle.current_stream_setting()
        self.handle.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        return self

    def __exit__(self, *exc):
        self.handle.synchronize()
        if self.previous is None:
            self.handle.reset_stream()
        else:
            self.handle.set_stream(self.previous[0])
        return False


def _posterior_n_sharded(handle, X_local, y_local, noise, mw, Lw, n_total, group, dt, n_loc, D, rows, cols, dev):
    import torch
    from . import _abi

    stats = torch.zeros((cols, rows), dtype=X_local.dtype, device=dev)  # column-major (rows x cols), lds = rows
    scal = torch.zeros(2, dtype=torch.float64, device=dev)
    diag_noise = noise.numel() > 1
    handle.gram_stats(dt, _abi.LAYOUT_COLVECS, D, n_loc, X_local.data_ptr(), D, y_local.data_ptr(),
                      _abi.NOISE_DIAGONAL if diag_noise else _abi.NOISE_ISOTROPIC, noise.data_ptr(), mw.data_ptr(),
                      stats.data_ptr(), rows, scal.data_ptr())
    handle.synchronize()
    reduce_stats(stats, scal, group)
    mw_post = torch.empty(D, dtype=X_local.dtype, device=dev)
    T_post = torch.zeros((D, D), dtype=X_local.dtype, device=dev)
    lp = torch.zeros(1, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    dense = Lw.dim() == 2
    handle.posterior_from_stats(dt, D, int(n_total), stats.data_ptr(), rows, scal.data_ptr(),
                                _abi.PRIOR_DENSE if dense else _abi.PRIOR_DIAGONAL, mw.data_ptr(), Lw.data_ptr(), D if dense else 1,
                                mw_post.data_ptr(), T_post.data_ptr(), D, None, D, lp.data_ptr(), info.data_ptr())
    handle.synchronize()
    code = int(info.item())
    if code != 0:
        raise _abi.PosDefException(code)
    return mw_post, T_post, float(lp.item())
