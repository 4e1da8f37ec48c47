#!/usr/bin/env python3
"""tools/bench_cli_em_boot.py [n_ind] [n_sites] [n_boot_rep] [block] -- the C++ host end to end on the reference's own kind
of bootstrap run: EM path (no --indep_geno), --n_boot_rep N, small blocks -- a generated binary GL file, wall time and the
host's own phase times (--verbose 2).  The whole job goes to the engine in one call: one pass of the per-site EM serves
the full-data matrix and every replicate."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_ind = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n_rep = int(sys.argv[3]) if len(sys.argv) > 3 else 100
block = int(sys.argv[4]) if len(sys.argv) > 4 else 10
path = "/tmp/ngd_emboot_%dx%d.bin" % (n_ind, n_sites)
if not os.path.exists(path):
    rng = np.random.default_rng(1)
    with open(path, "wb") as fh:
        for s0 in range(0, n_sites, 10000):
            n = min(10000, n_sites - s0)
            (rng.random((n, n_ind, 3)) ** 3 + 1e-9).tofile(fh)
print("file %s: %.2f GB" % (path, os.path.getsize(path) / 1e9), flush=True)
exe = os.path.join(ROOT, "ngsdist_amd", "bin", "ngsDist")
for reps in sorted({0, 5, n_rep}):
    for turn in range(2):
        t0 = time.time()
        r = subprocess.run([exe, "--geno", path, "--probs", "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--n_boot_rep",
                            str(reps), "--boot_block_size", str(block), "--evol_model", "2", "--out", "/tmp/ngd_emboot.dist",
                            "--verbose", "2", "--n_threads", "16"], capture_output=True, text=True)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr
    phases = "; ".join(ln.strip("> ").strip() for ln in r.stderr.splitlines()
                       if ln.startswith(">") and (" s " in ln or " s;" in ln or ln.rstrip().endswith(" s")))
    print("%3d replicates of %d-site blocks + the full data: %.2f s wall (second run); %s; output %.1f MB"
          % (reps, block, dt, phases, os.path.getsize("/tmp/ngd_emboot.dist") / 1e6), flush=True)
