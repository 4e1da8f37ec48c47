"""Multi-GPU plumbing: one process per GPU, collectives only where a job has an exchange step.

A job = n_mat matrices of n_pairs cells (the replicate loop, ngsDist.cpp:217-289).  Three ways to split it:
  * site or pair-tile sharding (strong scaling: the same job at every N): every rank holds partial sums of every
    cell (site ranges: to be added; pair tiles: zeros outside its shard, and x + 0 is exact).  scatter_sum() adds
    them AND leaves each rank with 1/N of the cells (one RCCL reduce-scatter), so that the tail of gen_dist()
    (ngsDist.cpp:372-401: /cnt, evolutionary model, on the HOST's libm for byte-identical output) runs on every
    rank's host cores for its own share; gather_cells() then puts the finished cells together (one all-gather).
  * replicate sharding (weak scaling: one matrix per rank): gather_matrices(), one all-gather of finished matrices.
merge_shards() is the plain reduce to one rank (what a host that wants raw sums uses).
Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np

from . import _lib


def shard_of_pairs(n_ind, world):
    """owner rank of every pair, in the reference's pair order"""
    import ctypes as C
    L = _lib.load()
    out = np.empty(n_ind * (n_ind - 1) // 2, dtype=np.int32)
    L.ngd_shard_map(int(n_ind), int(world), out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def owned_cells(n_ind, n_mat, world):
    """Pair-tile sharding, ONE collective: (idx, cap) with idx[r] = the cells of the job ([n_mat][n_pairs], flat) that
    rank r owns -- its pairs, matrix after matrix -- and cap = the longest of them.  Pair tiles are disjoint, so every
    rank can finish its own cells start to finish (gen_dist()'s tail, ngsDist.cpp:372-401, on its host) and ONE
    all-gather of `cap` finished cells per rank puts the job together: unpack_cells()."""
    owner = shard_of_pairs(n_ind, world)
    n_pairs = owner.size
    idx = []
    for r in range(world):
        p = np.flatnonzero(owner == r).astype(np.int64)
        idx.append((p[None, :] + n_pairs * np.arange(n_mat, dtype=np.int64)[:, None]).reshape(-1))
    return idx, max(1, max(len(x) for x in idx))


def unpack_cells(all_cells, idx, out):
    """all_cells: [world][cap] as the all-gather left them (numpy); out[idx[r]] = rank r's finished cells"""
    for r, ix in enumerate(idx):
        out[ix] = all_cells[r, :len(ix)]
    return out


def merge_shards(sum_t, cnt_t, dst=0):
    """In-place: after the call rank `dst` holds every pair.  sum_t float64,
    cnt_t int64 torch tensors (device tensors under RCCL, CPU tensors under gloo).
    cnt_t may be None when the count is the same for every pair (no --pairwise_del):
    the caller then fills it in without a collective."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dist.reduce(sum_t, dst=dst, op=dist.ReduceOp.SUM)
    if cnt_t is not None:
        dist.reduce(cnt_t, dst=dst, op=dist.ReduceOp.SUM)


def share_of(total, rank, world):
    """(chunk, lo, hi): cells [lo, hi) of `total` belong to `rank`; every rank's buffer holds `chunk` cells
    (the last ranks' tails are padding: reduce-scatter and all-gather want equal shares)"""
    chunk = -(-total // world)
    lo = min(total, rank * chunk)
    return chunk, lo, min(total, lo + chunk)


def scatter_sum(flat_t, mine_t):
    """flat_t: [world * chunk] partial sums of every cell (padding zeroed); after the call mine_t ([chunk]) holds
    the SUM over ranks of this rank's share.  One reduce-scatter.  float64 distance sums, and with --pairwise_del
    (ngsDist.cpp:335-338, :362) the int64 valid-site counts of the ranks' site ranges the same way."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        mine_t.copy_(flat_t[:mine_t.numel()])
        return
    dist.reduce_scatter_tensor(mine_t, flat_t, op=dist.ReduceOp.SUM)


def gather_cells(all_t, mine_t):
    """all_t: [world * chunk]; after the call every rank holds every rank's share, in rank order.  One all-gather."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        all_t[:mine_t.numel()].copy_(mine_t)
        return
    dist.all_gather_into_tensor(all_t, mine_t)


def gather_matrices(all_t, mine_t):
    """Replicate sharding: rank r holds one finished matrix (mine_t, [n_pairs]); after the call every rank's
    all_t ([world][n_pairs], same device and dtype) holds all of them, row r = rank r's.  One all-gather."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        all_t[0].copy_(mine_t)
        return
    if all_t.is_contiguous():
        dist.all_gather_into_tensor(all_t.view(-1), mine_t)  # the native all-gather, rows of all_t in rank order
    else:
        dist.all_gather(list(all_t.unbind(0)), mine_t)
