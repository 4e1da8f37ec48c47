"""The N>1 path on CPU: two gloo ranks, pair tiles dealt by the engine's own
rule (ngd_shard_of_pair), each rank fills only the pairs it owns, and ONE
collective (ngsdist_amd.dist.merge_shards, the function bench.py uses over
RCCL) brings everything to rank 0 -- bit-identical to the single-rank result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_ind, n_sites, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from ngsdist_amd.dist import merge_shards, shard_of_pairs
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = O.synth_indmajor(13, n_ind, n_sites, miss_frac=0.1)
    full_s, full_c = O.all_pairs(p, pairwise_del=True)  # stand-in for the device kernels
    owner = shard_of_pairs(n_ind, world)
    mine = owner == rank
    s = torch.from_numpy(np.where(mine, full_s, 0.0))
    c = torch.from_numpy(np.where(mine, full_c, 0).astype(np.int64))
    merge_shards(s, c, dst=0)
    if rank == 0:
        q.put((np.array_equal(s.numpy(), full_s), np.array_equal(c.numpy().astype(np.uint64), full_c),
               int(mine.sum()), int((~mine).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_is_bit_exact():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_ind, n_sites = 300, 64  # 3 tile rows -> 6 tiles over 2 ranks
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_ind, n_sites, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    ok_s, ok_c, n_mine, n_other = q.get(timeout=10)
    assert ok_s and ok_c
    assert n_mine > 0 and n_other > 0  # both ranks really owned something


def _site_worker(rank, world, port, n_ind, n_sites, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from ngsdist_amd.dist import merge_shards
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # called genotypes: every term is a multiple of 0.5, so ANY split of the site axis adds up exactly
    rng = np.random.default_rng(5)
    g = rng.integers(0, 3, size=(n_ind, n_sites))
    p = np.zeros((n_ind, n_sites, 3))
    np.put_along_axis(p, g[..., None], 1.0, axis=2)
    lo, hi = n_sites * rank // world, n_sites * (rank + 1) // world
    s_r, c_r = O.all_pairs(np.ascontiguousarray(p[:, lo:hi]))  # stand-in for the device kernels
    s = torch.from_numpy(s_r.copy())
    c = torch.from_numpy(c_r.astype(np.int64))
    merge_shards(s, c, dst=0)  # site sharding: the same collective, now a true sum
    if rank == 0:
        full_s, full_c = O.all_pairs(p)
        q.put((np.array_equal(s.numpy(), full_s), np.array_equal(c.numpy().astype(np.uint64), full_c)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_site_shards_add_up_exactly():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_site_worker, args=(r, 2, port, 30, 501, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    ok_s, ok_c = q.get(timeout=10)
    assert ok_s and ok_c


def _pair_worker(rank, world, port, n_ind, n_sites, n_mat, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import ngsdist_amd as N
    from ngsdist_amd.dist import gather_cells, owned_cells, unpack_cells
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = O.synth_indmajor(17, n_ind, n_sites, miss_frac=0.1)
    maps = [None] + [N.Taus(5 + r).block_map(n_sites // 4) for r in range(n_mat - 1)]
    S, Cn = [], []
    for m in maps:  # stand-in for the device kernels: every cell of the job
        s, c = O.all_pairs(p, pairwise_del=True, site_src=None if m is None else O.boot_site_src(m, 4),
                           n_sites=n_sites if m is None else n_sites // 4 * 4)
        S.append(s), Cn.append(c)
    S, Cn = np.stack(S).reshape(-1), np.stack(Cn).reshape(-1)
    idx, cap = owned_cells(n_ind, n_mat, world)
    # this rank finishes ITS cells only (pair tiles are disjoint), then ONE all-gather of finished cells
    mine = torch.zeros(cap, dtype=torch.float64)
    with np.errstate(all="ignore"):
        mine[:len(idx[rank])] = torch.from_numpy(N.finish(S[idx[rank]], Cn[idx[rank]], 0, 2))
    everyone = torch.zeros(world * cap, dtype=torch.float64)
    gather_cells(everyone, mine)
    if rank == 0:
        out = unpack_cells(everyone.numpy().reshape(world, cap), idx, np.full(S.size, -7.0))
        with np.errstate(all="ignore"):
            want = N.finish(S, Cn, 0, 2)
        q.put((bool(np.array_equal(out.view(np.uint64), want.view(np.uint64))), [len(x) for x in idx], cap))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pair_shards_one_all_gather_of_finished_cells():
    """--shard pairs: every rank runs gen_dist()'s tail (ngsDist.cpp:372-401) on the cells of ITS pair tiles and one
    all-gather of the finished cells ends the job (ngsdist_amd.dist.owned_cells / unpack_cells, what bench.py does over
    RCCL) -- every cell of a 3-matrix job bit-identical to finishing the whole job in one process."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pair_worker, args=(r, 2, port, 300, 64, 3, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    same, lens, cap = q.get(timeout=10)
    assert same and all(n > 0 for n in lens) and cap == max(lens) and sum(lens) == 3 * (300 * 299 // 2)


def test_shard_owner_covers_every_pair_once():
    os.environ.setdefault("NGD_NO_TORCH", "1")
    from ngsdist_amd.dist import shard_of_pairs
    for world in (1, 2, 4, 8):
        o = shard_of_pairs(260, world)
        assert o.min() >= 0 and o.max() < world
        if world <= 6:
            assert len(set(o.tolist())) == min(world, 6)


def _replicate_worker(rank, world, port, n_ind, n_sites, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from ngsdist_amd.dist import gather_matrices
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py's weak-scaling mode: rank 0 the full-data matrix, rank r the r-th bootstrap replicate (block size 1)
    p = O.synth_indmajor(7, n_ind, n_sites)
    rng = O.Taus(12345)
    maps = [None] + [rng.block_map(n_sites) for _ in range(world - 1)]

    def matrix(m):  # stand-in for the device kernels + the host tail
        src = None if m is None else O.boot_site_src(m, 1)
        s, c = O.all_pairs(p, site_src=src)
        return O.finish(s, c, 0, 1)

    mine = torch.from_numpy(matrix(maps[rank]))
    all_t = torch.zeros((world, mine.numel()), dtype=torch.float64)
    gather_matrices(all_t, mine)
    if rank == 0:
        q.put(all(np.array_equal(all_t[r].numpy(), matrix(maps[r])) for r in range(world)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicate_sharding_gathers_whole_matrices():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replicate_worker, args=(r, 2, port, 12, 500, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    assert q.get(timeout=10)


def _job_worker(rank, world, port, n_ind, n_sites, n_boot, block, q):
    """bench.py's strong-scaling flow for a multi-matrix job (cfg 5's shape in small): site ranges -> per-rank partial
    sums of every matrix -> scatter_sum (reduce-scatter) -> each rank finishes its share of the cells on its host ->
    gather_cells (all-gather).  Called genotypes, so the partial sums add up exactly and every cell must equal the
    single-process result bit for bit."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import ngsdist_amd as N
    from ngsdist_amd.dist import gather_cells, scatter_sum, share_of
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    g = rng.integers(0, 3, size=(n_ind, n_sites))
    p = np.zeros((n_ind, n_sites, 3))
    np.put_along_axis(p, g[..., None], 1.0, axis=2)
    n_pairs, n_mat = N.n_pairs(n_ind), n_boot + 1
    n_eff = n_sites - n_sites % block
    t = O.Taus(12345)
    maps = [None] + [t.block_map(n_eff // block) for _ in range(n_boot)]
    # whole blocks per rank
    n_blocks = n_eff // block
    b_lo, b_hi = n_blocks * rank // world, n_blocks * (rank + 1) // world
    s_lo, s_hi = b_lo * block, (b_hi * block if rank + 1 < world else n_sites)
    total = n_mat * n_pairs
    chunk, c_lo, c_hi = share_of(total, rank, world)
    flat = torch.zeros(world * chunk, dtype=torch.float64)
    part = flat[:total].view(n_mat, n_pairs)
    for m, bm in enumerate(maps):  # stand-in for the device kernels: this rank's sites of every matrix
        if bm is None:
            src = np.arange(s_lo, s_hi, dtype=np.uint64)
        else:  # the draws that land in this rank's blocks, in this rank's positions -- multiplicities are what matter
            mult = np.bincount(bm.astype(np.int64), minlength=n_blocks)[b_lo:b_hi]
            src = np.concatenate([np.tile(np.arange(b * block, (b + 1) * block, dtype=np.uint64), k)
                                  for b, k in zip(range(b_lo, b_hi), mult)] or [np.zeros(0, dtype=np.uint64)])
        if src.size:
            part[m] = torch.from_numpy(O.all_pairs(p, site_src=src, n_sites=src.size)[0])
    mine = torch.empty(chunk, dtype=torch.float64)
    scatter_sum(flat, mine)
    cnt = np.full((n_mat, n_pairs), n_eff, dtype=np.uint64)
    cnt[0] = n_sites
    dist_mine = torch.zeros(chunk, dtype=torch.float64)
    with np.errstate(all="ignore"):
        N.finish(mine.numpy()[:c_hi - c_lo], cnt.reshape(-1)[c_lo:c_hi], 0, 1, out=dist_mine.numpy()[:c_hi - c_lo])
    every = torch.empty(world * chunk, dtype=torch.float64)
    gather_cells(every, dist_mine)
    got = every.numpy()[:total].reshape(n_mat, n_pairs)
    ok = True
    for m, bm in enumerate(maps):  # single-process result
        src = None if bm is None else O.boot_site_src(bm, block)
        s, c = O.all_pairs(p, site_src=src, n_sites=n_sites if bm is None else n_eff)
        with np.errstate(all="ignore"):
            ok = ok and np.array_equal(got[m], O.finish(s, c, 0, 1), equal_nan=True)
    q.put((rank, bool(ok), c_hi - c_lo))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_job_reduce_scatter_finish_all_gather():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, 2, port, 13, 403, 4, 10, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert all(ok for _, ok, _ in res)          # every rank ends up with every finished cell
    assert sum(n for _, _, n in res) == 5 * 78  # the shares partition the job's cells


def _pdel_worker(rank, world, port, n_ind, n_sites, q):
    """--pairwise_del under site sharding: a pair's valid-site count is a sum over the ranks' site ranges like its
    distance sum, so BOTH go through scatter_sum (one reduce-scatter each), every rank finishes its share with the
    summed counts, and gather_cells puts the cells together (bench.py --pairwise_del, ngsDist.cpp:335-338, :362)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import ngsdist_amd as N
    from ngsdist_amd.dist import gather_cells, scatter_sum, share_of
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(9)
    g = rng.integers(0, 3, size=(n_ind, n_sites))
    p = np.zeros((n_ind, n_sites, 3))
    np.put_along_axis(p, g[..., None], 1.0, axis=2)
    p[rng.random((n_ind, n_sites)) < 0.2] = 1.0 / 3  # missing data: skipped per pair, so the rest stays dyadic
    n_pairs = N.n_pairs(n_ind)
    lo, hi = n_sites * rank // world, n_sites * (rank + 1) // world
    s_r, c_r = O.all_pairs(np.ascontiguousarray(p[:, lo:hi]), pairwise_del=True)  # stand-in for the device kernels
    chunk, c_lo, c_hi = share_of(n_pairs, rank, world)
    flat = torch.zeros(world * chunk, dtype=torch.float64)
    cflat = torch.zeros(world * chunk, dtype=torch.int64)
    flat[:n_pairs] = torch.from_numpy(s_r)
    cflat[:n_pairs] = torch.from_numpy(c_r.astype(np.int64))
    mine, cmine = torch.empty(chunk, dtype=torch.float64), torch.empty(chunk, dtype=torch.int64)
    scatter_sum(flat, mine)
    scatter_sum(cflat, cmine)
    dist_mine = torch.zeros(chunk, dtype=torch.float64)
    with np.errstate(all="ignore"):
        N.finish(mine.numpy()[:c_hi - c_lo], cmine.numpy().view(np.uint64)[:c_hi - c_lo], 0, 2,
                 out=dist_mine.numpy()[:c_hi - c_lo])
    every = torch.empty(world * chunk, dtype=torch.float64)
    gather_cells(every, dist_mine)
    s, c = O.all_pairs(p, pairwise_del=True)
    with np.errstate(all="ignore"):
        want = O.finish(s, c, 0, 2)
    full_c = torch.empty(world * chunk, dtype=torch.int64)
    gather_cells(full_c, cmine)
    q.put((rank, bool(np.array_equal(every.numpy()[:n_pairs], want, equal_nan=True)),
           bool(np.array_equal(full_c.numpy()[:n_pairs].astype(np.uint64), c)), int(c.min()), int(c.max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_site_shards_with_pairwise_del_counts():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pdel_worker, args=(r, 2, port, 14, 397, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0
    res = [q.get(timeout=10) for _ in range(2)]
    assert all(ok_d and ok_c for _, ok_d, ok_c, _, _ in res)
    assert res[0][3] < res[0][4] < 397  # counts really differ from pair to pair


def test_share_of_partitions_any_job_into_equal_buffers():
    os.environ.setdefault("NGD_NO_TORCH", "1")
    from ngsdist_amd.dist import share_of
    for total in (1, 7, 499500, 65 * 124750, 1000003):
        for world in (1, 2, 3, 4, 8):
            chunks = [share_of(total, r, world) for r in range(world)]
            assert len({c for c, _, _ in chunks}) == 1 and chunks[0][0] * world >= total  # equal buffers that cover the job
            covered = 0
            for r, (chunk, lo, hi) in enumerate(chunks):
                assert lo == min(total, r * chunk) and lo <= hi <= total and hi - lo <= chunk
                assert lo == covered or lo == total
                covered = hi
            assert covered == total
