#!/usr/bin/env python3
"""tools/rccl_preflight.py -- what the collectives of a multi-GPU job cost on THIS node, before the job runs.

No multi-GPU node was available to this build (every N > 1 figure in DESIGN.md section 7 is a prediction), so the first
hardware run has to explain its own scaling curve: `bench.py --gpus N` calls preflight() in every rank process, after
the process group is up and BEFORE the engine is created, and rank 0 prints one JSON line to stderr with
  * the RCCL version and the environment the communicator came up under,
  * all_reduce / reduce_scatter / all_gather of the job's REAL message sizes (the cells of the job: sums as float64,
    ngsDist.cpp:376's `dist` before the division; one share per rank), each timed once cold and as the best and the
    median of a few repeats (barrier + device sync on both sides, MAX over ranks),
  * the bus bandwidth those times mean (ring model: (N-1)/N x bytes / time per rank; all_reduce twice that),
so that a curve below the prediction can be split into "the collectives were slow" and "the engine was slow".

Standalone (the rehearsal the CPU suite runs, gloo):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
      tools/rccl_preflight.py --backend gloo --cells 1000000
"""
import json
import os
import sys
import time


def _timed(fn, sync, dist, t_dev, repeats):
    """fn() once cold, then `repeats` times; every sample = MAX over ranks of the wall time between two barriers"""
    import torch
    out = []
    for _ in range(repeats + 1):
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=t_dev)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        sync()
        out.append(float(dt.item()) * 1e3)
    warm = sorted(out[1:])
    return {"cold_ms": out[0], "best_ms": warm[0], "median_ms": warm[len(warm) // 2]}


def preflight(dist, dev, backend, cells_total, per_rank_gather=0, repeats=5, out=sys.stderr):
    """cells_total: float64 cells of the whole job ([n_mat][n_pairs]): the reduce-scatter's input on every rank and the
    all-gather's output; per_rank_gather: cells each rank contributes to the final all-gather when that is not
    cells_total / world (replicate sharding: one finished matrix).  Returns the report (rank 0 also prints it)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    on_gpu = backend == "nccl"
    t_dev = dev if on_gpu else "cpu"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    chunk = -(-int(cells_total) // world)
    g_chunk = int(per_rank_gather) or chunk
    flat = torch.ones(world * chunk, dtype=torch.float64, device=t_dev)
    mine = torch.empty(chunk, dtype=torch.float64, device=t_dev)
    g_mine = torch.full((g_chunk,), float(rank), dtype=torch.float64, device=t_dev)
    g_all = torch.empty(world * g_chunk, dtype=torch.float64, device=t_dev)
    rep = {"preflight": "collectives of this job's sizes, before the engine exists", "backend": backend, "world": world,
           "cells_total": int(cells_total), "reduce_scatter_bytes_per_rank_in": world * chunk * 8,
           "all_gather_bytes_per_rank_in": g_chunk * 8, "repeats": repeats}
    try:
        if on_gpu:
            v = torch.cuda.nccl.version()
            rep["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
            rep["device"] = torch.cuda.get_device_name(dev)
    except Exception as exc:  # the version is a nicety
        rep["rccl_version"] = "unknown (%r)" % (exc,)
    rep["env"] = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_P2P_DISABLE",
                                                  "NCCL_SHM_DISABLE", "NCCL_P2P_LEVEL", "NCCL_NET_GDR_LEVEL",
                                                  "RCCL_MSCCL_ENABLE", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                  "MASTER_ADDR") if os.environ.get(k) is not None}
    # HSA_ENABLE_IPC_MODE_LEGACY: "0" selects dmabuf IPC, which is what this pool's driver supports; unset or "1" and RCCL's
    # peer-to-peer set-up fails with hipIpcGetMemHandle: invalid argument (bench.py exports 0 for the ranks it launches)
    rep["hsa_enable_ipc_mode_legacy"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "(unset)")
    # did the ranks reach one another directly?  Peer access between the devices of this node, as HIP reports it: without it
    # RCCL stages every message through host memory (shared-memory transport) and the bus bandwidths below are PCIe's
    if on_gpu:
        try:
            n_dev = torch.cuda.device_count()
            me = dev.index if hasattr(dev, "index") and dev.index is not None else torch.cuda.current_device()
            peers = [bool(torch.cuda.can_device_access_peer(me, q)) for q in range(n_dev) if q != me]
            rep["peer_access"] = {"devices": n_dev, "peers_reachable": int(sum(peers)), "all": bool(all(peers)) if peers else None}
            rep["host_staging_suspected"] = bool(peers) and not all(peers)
        except Exception as exc:
            rep["peer_access"] = "unknown (%r)" % (exc,)
    ring = (world - 1) / world

    def bus(nbytes, ms, factor=1.0):  # GB/s per rank under the ring model
        return factor * ring * nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None

    r = _timed(lambda: dist.all_reduce(flat), sync, dist, t_dev, repeats)
    r["busbw_GBs"] = bus(flat.numel() * 8, r["best_ms"], 2.0)
    rep["all_reduce"] = r
    flat.fill_(1.0)
    r = _timed(lambda: dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM), sync, dist, t_dev, repeats)
    r["busbw_GBs"] = bus(flat.numel() * 8, r["best_ms"])
    rep["reduce_scatter"] = r
    ok_rs = bool(torch.all(mine == float(world)).item())
    r = _timed(lambda: dist.all_gather_into_tensor(g_all, g_mine), sync, dist, t_dev, repeats)
    r["busbw_GBs"] = bus(g_all.numel() * 8, r["best_ms"])
    rep["all_gather"] = r
    ok_ag = all(bool(torch.all(g_all[q * g_chunk:(q + 1) * g_chunk] == float(q)).item()) for q in range(world))
    rep["results_checked"] = {"reduce_scatter_sum_is_world": ok_rs, "all_gather_rows_in_rank_order": ok_ag}
    # ONE verdict for all ranks (MIN over their flags): a rank that left alone would leave the others waiting in the next
    # collective until the backend's timeout
    flag = torch.tensor([1.0 if (ok_rs and ok_ag) else 0.0], dtype=torch.float64, device=t_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    sync()
    rep["all_ranks_ok"] = bool(flag.item() == 1.0)
    if not rep["all_ranks_ok"]:
        raise SystemExit("rccl_preflight: rank %d: a collective returned wrong data on some rank (here: %s): %s"
                         % (rank, "wrong" if not (ok_rs and ok_ag) else "right", json.dumps(rep)))
    if on_gpu and flat.numel() * 8 >= (1 << 22) and rep["reduce_scatter"]["busbw_GBs"] is not None:
        # (an xGMI link moves ~50 GB/s and more; a few GB/s on warm messages of several MB means staging through the host)
        rep["busbw_suggests_host_staging"] = bool(rep["reduce_scatter"]["busbw_GBs"] < 10.0)
    if rank == 0 and out is not None:
        out.write("rccl_preflight " + json.dumps(rep) + "\n")
        out.flush()
    return rep


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--cells", type=int, default=499500, help="float64 cells of the job (n_mat x n_pairs)")
    ap.add_argument("--gather_cells", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=5)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = None
    if a.backend == "nccl":
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    preflight(dist, dev, a.backend, a.cells, a.gather_cells, a.repeats)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
