"""N > 1 path on CPU: world_size-2 gloo.  Checks the partition (bit-exact index work) and that the gathered
log-evidence vector -- hence the fixed-order sum every rank then runs on it -- is identical on every rank and
equal to the single-rank result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import blr_amd  # noqa: F401
from blr_amd import sharding


def fixed_order_sum(v):
    """Host emulation of blr::logpdf_sum_kernel (csrc/blr_aux_kernels.hpp): thread t sums elements
    t, t+256, ... in order, then a halving tree over the 256 partials."""
    part = np.zeros(256)
    for t in range(256):
        s = 0.0
        for x in v[t::256]:
            s += float(x)
        part[t] = s
    m = 128
    while m >= 1:
        part[:m] = part[:m] + part[m:2 * m]
        m //= 2
    return float(part[0])


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 8, 1000, 8192, 8195):
        for world in (1, 2, 3, 4, 8):
            blocks = [sharding.shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and a <= b
            sizes = sharding.shard_sizes(total, world)
            assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


TOTALS = (8192, 1001)


def _lp_all(total):
    rng = np.random.default_rng(4242 + total)
    return -1e3 * rng.random(total) - 500.0  # what the fused kernel would have produced for every regressor


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for total in TOTALS:
        lp_all = _lp_all(total)
        lo, hi = sharding.shard_range(total, rank, world)
        local = torch.from_numpy(lp_all[lo:hi].copy())
        gathered = sharding.gather_logpdf(local, total)
        q.put((rank, total, gathered.numpy().tobytes(), fixed_order_sum(gathered.numpy())))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_is_rank_count_independent():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2 * len(TOTALS))]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, total, raw, tot in res:
        lp_all = _lp_all(total)
        assert raw == lp_all.tobytes(), f"rank {rank}: gathered vector differs"
        assert tot == fixed_order_sum(lp_all)  # bitwise: same order of additions for every rank count


def _padded_worker(rank, world, port, q):
    # bench.py --config c4 (strong scaling): every rank evaluates its contiguous block, pads it with zeros to the largest
    # block (collectives want equal counts) and the ranks all-gather the padded blocks; the total is the fixed-order sum
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for total in (8192, 8191, 1001, 3):
        lp_all = _lp_all(total)
        lo, hi = sharding.shard_range(total, rank, world)
        bmax = max(sharding.shard_sizes(total, world))
        mine = torch.zeros(bmax, dtype=torch.float64)
        mine[: hi - lo] = torch.from_numpy(lp_all[lo:hi].copy())
        gathered = torch.empty(bmax * world, dtype=torch.float64)
        dist.all_gather_into_tensor(gathered, mine)
        q.put((rank, total, gathered.numpy().tobytes(), fixed_order_sum(gathered.numpy())))
    dist.barrier()
    dist.destroy_process_group()


def test_padded_blocks_of_a_fixed_batch_sum_to_the_same_total_on_every_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_padded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2 * 4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    by_total = {}
    for rank, total, raw, tot in res:
        by_total.setdefault(total, []).append((raw, tot))
    for total, items in by_total.items():
        assert len(items) == 2 and items[0] == items[1], "the ranks disagree on the gathered vector or its sum"
        g = np.frombuffer(items[0][0])
        lp_all = _lp_all(total)
        sizes = sharding.shard_sizes(total, 2)
        bmax = max(sizes)
        # every block sits at rank * bmax, the padding is zero, nothing is lost or counted twice
        off = 0
        for r_, n in enumerate(sizes):
            assert g[r_ * bmax: r_ * bmax + n].tobytes() == lp_all[off: off + n].tobytes()
            assert not g[r_ * bmax + n: (r_ + 1) * bmax].any()
            off += n
        assert items[0][1] == pytest.approx(float(np.sum(lp_all)), rel=1e-13)


def _stats_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, cols = sharding.stats_shape(130)
    g = torch.Generator().manual_seed(100 + rank)
    stats = torch.randn((cols, rows), generator=g, dtype=torch.float64)
    scal = torch.randn(2, generator=g, dtype=torch.float64)
    mine = (stats.clone(), scal.clone())
    sharding.reduce_stats(stats, scal)
    q.put((rank, mine[0].numpy().tobytes(), mine[1].numpy().tobytes(), stats.numpy().tobytes(), scal.numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_n_sharded_statistics_are_summed_once_and_identically():
    """The single exchange of the N-sharded path: both ranks end with the SAME sum of the additive statistics."""
    assert sharding.stats_shape(1) == (256, 128) and sharding.stats_shape(128) == (256, 128) and sharding.stats_shape(129) == (384, 256)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stats_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows, cols = sharding.stats_shape(130)
    own = [np.frombuffer(r[1], dtype=np.float64) for r in res]
    own_s = [np.frombuffer(r[2], dtype=np.float64) for r in res]
    for r in res:
        np.testing.assert_array_equal(np.frombuffer(r[3], dtype=np.float64), own[0] + own[1])
        np.testing.assert_array_equal(np.frombuffer(r[4], dtype=np.float64), own_s[0] + own_s[1])
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]
