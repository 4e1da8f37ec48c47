"""`python bench.py --gpus 2` end to end on the GPU box: the parent starts the ranks itself (no outer
torch.distributed.run), both ranks run the REAL kernels on cuda:0 (--same_device), the collectives go over gloo
(one GPU here; the driver's 8-GPU run uses RCCL through the same code), and the line must carry "valid": true --
the 6-pair full-size spot check against the oracle on the gathered matrix (ngsDist.cpp:217-289 split by sites)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _bench(*extra):
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same_device",
           "--steps", "2", "--warmup", "1", "--no_cpu"] + list(extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_cfg2_by_sites_start_themselves():
    out = _bench("--workload", "cfg2")
    assert out["valid"] is True and out["n_gpus"] == 2
    assert out["spot_check"]["max_rel_err_vs_oracle"] <= 1e-9
    assert "site axis split over 2 ranks" in out["config"]["sharding"]


def test_two_ranks_cfg2_by_pair_tiles():
    out = _bench("--workload", "cfg2", "--shard", "pairs")
    assert out["valid"] is True and out["n_gpus"] == 2


def test_two_ranks_one_matrix_each_by_replicates():
    # --shard replicates, the weak-scaling split (one matrix per GPU, ONE all-gather of finished matrices: SURVEY 8e):
    # rank 0 the full-data matrix, rank 1 a bootstrap replicate at the reference's default block size
    out = _bench("--workload", "cfg2", "--shard", "replicates")
    assert out["valid"] is True and out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["matrices_per_step"] == 2


def test_two_ranks_cfg5_bootstrap_job_reduced_sites():
    # cfg 5's shape (500 individuals, 64 replicates of 1000-site blocks + the full-data matrix) on 1/10 of its sites
    out = _bench("--workload", "cfg5", "--n_sites", "50000")
    assert out["valid"] is True and out["config"]["matrices_per_step"] == 65


def test_two_ranks_em_path_reduced_sites():
    out = _bench("--workload", "cfg4", "--n_sites", "20000")
    assert out["valid"] is True


def test_two_ranks_with_the_tail_on_the_worker_thread():
    # --pipelined_tail at N > 1: a second region with the reduce-scatter, each rank's share of ngd_finish and the
    # all-gather on the worker thread beside the next job's kernels (reported apart; opt-in here until it has run over
    # RCCL).  --vary_jobs: odd and even steps compute different jobs and the region must reproduce the serial
    # region's per-step checksums of the gathered result (no mix-up of the two buffer sets).
    for w in (["--workload", "cfg3", "--n_sites", "100000"], ["--workload", "cfg5", "--n_sites", "50000"]):
        out = _bench(*w, "--pipelined_tail", "--vary_jobs")
        assert out["valid"] is True and out["pipelined"]["ms_per_step"] > 0
        assert out["pipeline_check"]["ok"] is True and out["pipeline_check"]["jobs_differ"] is True
        assert _bench(*w)["pipelined"] is None


def test_two_ranks_pairwise_del_counts_are_reduced():
    # counts differ per pair: they go through the same reduce-scatter as the sums (ngsDist.cpp:335-338, :362)
    out = _bench("--workload", "cfg2", "--pairwise_del", "--miss_frac", "0.1")
    assert out["valid"] is True
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--pairwise_del",
                          "--miss_frac", "0.1", "--steps", "1", "--warmup", "0", "--no_cpu"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    assert json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])["valid"] is True
