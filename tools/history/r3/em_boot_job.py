#!/usr/bin/env python3
"""tools/em_boot_job.py [n_sites] [block] [n_rep...] [--scratch_gb a,b,...] [--n_ind N] -- EM path + bootstrap with
blocks too small for per-block partial results (the reference's own examples: --n_boot_rep 5 --boot_block_size 10, or
its default block size 1): the whole replicate loop through ngd_run_job, per-block partials off, against one plain pass.
Plans: "spill" = the terms of a chunk of sites written once + one FP64 MFMA contraction with every matrix's weights
(contract_mfma.hip; the default from three matrices on), "batch" = 8 matrices per pass of the table-driven kernel.
The job's last replicate and its full-data matrix are compared with their own ngd_run passes on all pairs."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("n_sites", nargs="?", type=int, default=100000)
ap.add_argument("block", nargs="?", type=int, default=10)
ap.add_argument("reps", nargs="*", type=int, default=[2, 5, 7, 15, 100])
ap.add_argument("--n_ind", type=int, default=1000)
ap.add_argument("--scratch_gb", default="0", help="comma-separated NGD_OPT_EM_SPILL_BYTES values in GB (0 = default)")
ap.add_argument("--no_batch", action="store_true", help="skip the 8-matrices-per-pass plan")
ap.add_argument("--pairwise_del", action="store_true")
a = ap.parse_args()


def rel(x, y):
    den = np.where(y == 0, 1.0, np.abs(y))
    return float(np.max(np.abs(x - y) / den))


with N.Engine(a.n_ind, a.n_sites, indep_geno=False, kernel="em_table", pairwise_del=a.pairwise_del) as e:
    e.synth_fill(3, 0.05 if a.pairwise_del else 0.0).set_option("boot_partials", 0)
    e.run()
    t = time.perf_counter(); s0, c0 = e.run(); plain = (time.perf_counter() - t) * 1e3
    print("em_table, %d x %d: one plain pass %.0f ms" % (a.n_ind, a.n_sites, plain), flush=True)
    for n_rep in a.reps:
        rng = N.Taus(11)
        maps = np.stack([rng.block_map(a.n_sites // a.block) for _ in range(n_rep)])
        s1, c1 = e.run(maps[-1], a.block)
        plans = [("spill %s GB" % g, 1, int(float(g) * (1 << 30))) for g in a.scratch_gb.split(",")]
        if not a.no_batch:
            plans.append(("batch", 0, 0))
        for name, spill, scratch in plans:
            e.set_option("em_spill", spill).set_option("em_spill_bytes", scratch)
            e.run_job(maps, a.block)  # (allocations)
            t = time.perf_counter(); S, C = e.run_job(maps, a.block); job = (time.perf_counter() - t) * 1e3
            tm = e.timing()
            sp = e.spill_timing()
            print("  %3d replicates of %d-site blocks + the full-data matrix, %-12s: %7.0f ms (%.2f plain passes; device "
                  "%.0f ms: accumulate %.0f, scatter/reduce %.1f); last replicate vs its own pass %.1e, matrix 0 vs the "
                  "plain pass %.1e, counts equal: %s"
                  % (n_rep, a.block, name, job, job / plain, tm["ms_total"], tm["ms_accum"], tm["ms_reduce"],
                     rel(S[-1], s1), rel(S[0], s0), np.array_equal(C[-1], c1) and np.array_equal(C[0], c0)), flush=True)
            if sp["chunks"]:
                tb = sp["units"] * sp["slot_groups_live"] * 16 * 8  # bytes of terms written once and read once per 128 matrices
                fl = 2.0 * sp["units"] * sp["slot_groups"] * 16 * sp["matrix_groups"] * 16
                print("        %d chunks, units of %d sites: EM pass %.1f ms, contraction %.1f ms (%.2f TB/s of terms, %.1f TF), "
                      "weights %.1f, sanitize %.1f; %d of %d slot groups live"
                      % (sp["chunks"], sp["unit_sites"], sp["ms_terms"], sp["ms_contract"],
                         tb * ((sp["matrix_groups"] + 7) // 8) / sp["ms_contract"] / 1e9, fl / sp["ms_contract"] / 1e9,
                         sp["ms_weights"], sp["ms_sanitize"], sp["slot_groups_live"], sp["slot_groups"]), flush=True)
