"""tools/rccl_preflight.py -- the collectives of a multi-GPU job at their real sizes, timed before the engine exists
(bench.py --gpus N calls it in every rank).  Rehearsed here over gloo with two ranks: the report must carry the three
collectives, their data checks, and the job's message sizes."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_preflight_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cells = 101 * 4950 + 1  # odd: the last rank's share is padded
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "tools", "rccl_preflight.py"), "--backend", "gloo", "--cells", str(cells),
                        "--repeats", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("rccl_preflight ")]
    assert len(lines) == 1  # rank 0 only
    rep = json.loads(lines[0][len("rccl_preflight "):])
    assert rep["world"] == 2 and rep["cells_total"] == cells
    chunk = -(-cells // 2)
    assert rep["reduce_scatter_bytes_per_rank_in"] == 2 * chunk * 8 and rep["all_gather_bytes_per_rank_in"] == chunk * 8
    for k in ("all_reduce", "reduce_scatter", "all_gather"):
        assert rep[k]["best_ms"] > 0 and rep[k]["cold_ms"] > 0 and rep[k]["busbw_GBs"] > 0
    assert rep["results_checked"] == {"reduce_scatter_sum_is_world": True, "all_gather_rows_in_rank_order": True}
