"""`python bench.py --gpus N` with no RANK in the environment starts its own ranks (torch.distributed.run, 127.0.0.1, a free
port) as child processes and hands back their exit code.  Without a GPU every rank must stop at once with the product's
"no GPU visible ... no CPU fallback" message -- which is what this CPU-side test can see of the launch path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks_and_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible: tests/test_gpu_multirank.py covers the launch path end to end")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same_device",
                        "--workload", "cfg2", "--steps", "1", "--warmup", "0", "--no_cpu"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no GPU visible; the product path has no CPU fallback" in r.stderr  # printed by the RANKS, i.e. they started
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # and no bench line was made up
