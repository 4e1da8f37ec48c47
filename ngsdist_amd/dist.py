"""Multi-GPU plumbing: one process per GPU and ONE collective per job.
Replicate sharding (one finished matrix per rank): gather_matrices().
Pair-tile / site sharding: merge_shards() brings the shards to rank 0.  Because every pair is owned by exactly one rank and the others hold
exact zeros, a SUM reduce is a gather: x + 0 is exact in IEEE arithmetic (sums
are >= +0, so no -0 ambiguity).  Backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.
"""
import numpy as np

from . import _lib


def shard_of_pairs(n_ind, world):
    """owner rank of every pair, in the reference's pair order"""
    L = _lib.load()
    out = np.empty(n_ind * (n_ind - 1) // 2, dtype=np.int32)
    k = 0
    for i in range(n_ind):
        for j in range(i + 1, n_ind):
            out[k] = L.ngd_shard_of_pair(n_ind, i, j, world)
            k += 1
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


def gather_matrices(all_t, mine_t):
    """Replicate sharding: rank r holds one finished matrix (mine_t, [n_pairs]); after the call every rank's
    all_t ([world][n_pairs], same device and dtype) holds all of them, row r = rank r's.  One all-gather."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        all_t[0].copy_(mine_t)
        return
    if dist.get_backend() == "nccl" and all_t.is_contiguous():
        dist.all_gather_into_tensor(all_t, mine_t)  # RCCL's native all-gather into the [world][n_pairs] tensor
    else:
        dist.all_gather(list(all_t.unbind(0)), mine_t)
