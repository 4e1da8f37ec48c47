#!/usr/bin/env python3
"""tools/em_boot_job.py [n_sites] [block] [n_rep...] -- EM path + bootstrap with blocks too small for per-block partial
results (the reference's own examples: --n_boot_rep 5 --boot_block_size 10, or its default block size 1): the whole
replicate loop through ngd_run_job, per-block partials off, against one plain pass; the job's last replicate and its
full-data matrix are compared with their own ngd_run passes, all pairs, bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
block = int(sys.argv[2]) if len(sys.argv) > 2 else 10
reps = [int(x) for x in sys.argv[3:]] or [1, 2, 3, 5, 7, 15]
n_ind = 1000
for kernel in ("em_table", "em_fast"):
    with N.Engine(n_ind, n_sites, indep_geno=False, kernel=kernel) as e:
        e.synth_fill(3).set_option("boot_partials", 0)
        e.run()
        t = time.perf_counter(); s0, c0 = e.run(); plain = (time.perf_counter() - t) * 1e3
        print("%s: one plain pass %.0f ms" % (kernel, plain), flush=True)
        for n_rep in reps:
            rng = N.Taus(11)
            maps = np.stack([rng.block_map(n_sites // block) for _ in range(n_rep)])
            e.run_job(maps, block)
            t = time.perf_counter(); S, C = e.run_job(maps, block); job = (time.perf_counter() - t) * 1e3
            s1, c1 = e.run(maps[-1], block)
            print("  %2d replicates of %d-site blocks + the full-data matrix: %.0f ms (%.2f plain passes); last replicate "
                  "bits equal to its own pass: %s, matrix 0 equal to the plain pass: %s"
                  % (n_rep, block, job, job / plain, np.array_equal(S[-1], s1) and np.array_equal(C[-1], c1),
                     np.array_equal(S[0], s0)), flush=True)
