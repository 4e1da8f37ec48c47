#!/usr/bin/env python3
"""tools/bench_e2e.py -- the C++ host END TO END on BASELINE's configurations (load + prepare + upload is
where a user's wall time is once the distances take milliseconds).

For every workload asked for: a binary GL file of the workload's shape is generated into --dir (tools/gen_gl_file, the
counter-based recipe of SURVEY 8d; /dev/shm or the page cache, i.e. the file is in host memory when the run starts), the
host (`ngsdist_amd/bin/ngsDist ... --verbose 2`) runs --runs times, and ONE JSON line is printed: wall time of the process
(exec to exit, the driver's clock), the host's own phases (`> phases [s]:` line: args, first HIP calls, engine creation,
load with its components, matrices, teardown), the load rate against the link's roof measured IN THE SAME LEASE by
tools/pcie_peak (pinned hipMemcpy, 128 MiB and 1 GiB), end-to-end pair-distances/s, and a check of the printed matrices
against an engine filled on the device from the same seed (every cell of every matrix, 1e-9 relative + the print's 1e-10).

usage: bench_e2e.py [--workloads cfg3,cfg4,cfg5,emboot,cfg2] [--runs 3] [--dir /dev/shm] [--n_threads 16] [--no_check]
                    [--host_args "..."]"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS  # noqa: E402  (shapes, seeds, flags: one definition)

EXE = os.path.join(ROOT, "ngsdist_amd", "bin", "ngsDist")


def build_tools(need_peak=True):
    gen = os.path.join(ROOT, "tools", "gen_gl_file")
    if not os.path.exists(gen) or os.path.getmtime(gen) < os.path.getmtime(gen + ".cpp"):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-pthread", "-o", gen, gen + ".cpp"])
    peak = os.path.join(ROOT, "tools", "pcie_peak")
    if need_peak and (not os.path.exists(peak) or os.path.getmtime(peak) < os.path.getmtime(peak + ".hip")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", peak, peak + ".hip"])
    return gen, peak


def link_roof(peak):
    r = subprocess.run([peak, "1024"], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        return {"error": r.stderr[-300:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


def parse_phases(stderr):
    m = re.search(r"^> phases \[s\]:(.*)$", stderr, re.M)
    if not m:
        return {}
    out = {}
    for tok in m.group(1).split():
        k, v = tok.split("=")
        out[k] = out.get(k, 0.0) + float(v)
    return out


def read_matrices(path, n_ind):
    """the PHYLIP-style blocks of ngsDist.cpp:282-287: an empty line, n_ind, then n_ind rows label \\t cells"""
    mats = []
    with open(path) as fh:
        toks = fh.read().split("\n")
    i = 0
    while i < len(toks):
        if toks[i].strip() == str(n_ind) and i + n_ind < len(toks):
            rows = [toks[i + 1 + r].split("\t")[1:] for r in range(n_ind)]
            mats.append(np.array(rows, dtype=np.float64))
            i += n_ind + 1
        else:
            i += 1
    return mats


def check_against_device_fill(W, dist_path, seed_rng):
    import ngsdist_amd as N
    n_ind, n_sites = W["n_ind"], W["n_sites"]
    mats = read_matrices(dist_path, n_ind)
    with N.Engine(n_ind, n_sites, indep_geno=W["indep"]) as e:
        for k, v in W.get("options", {}).items():
            e.set_option(k, v)
        e.synth_fill(W["seed"])
        if W["n_boot"]:
            nb = n_sites // W["block"]
            t = N.Taus(seed_rng)
            maps = np.stack([t.block_map(nb) for _ in range(W["n_boot"])])
            S, C = e.run_job(maps, W["block"])
        else:
            s, c = e.run()
            S, C = s[None, :], c[None, :]
    if len(mats) != S.shape[0]:
        return {"ok": False, "why": "%d matrices printed, %d expected" % (len(mats), S.shape[0])}
    iu = np.triu_indices(n_ind, 1)
    worst = 0.0
    for r in range(S.shape[0]):
        d = N.finish(S[r], C[r], 0, W["evol_model"])
        got = mats[r][iu]
        if not np.array_equal(mats[r], mats[r].T) or np.any(np.diag(mats[r]) != 0):
            return {"ok": False, "why": "matrix %d is not symmetric with a zero diagonal" % r}
        fin = np.isfinite(d)
        if not np.array_equal(fin, np.isfinite(got)):
            return {"ok": False, "why": "matrix %d: non-finite cells differ" % r}
        err = np.abs(got[fin] - d[fin]) - 0.5000001e-10  # (the print rounds to 1e-10)
        worst = max(worst, float(np.max(err / np.abs(d[fin]))))
    return {"ok": bool(worst < 1e-9), "matrices": int(S.shape[0]), "cells": int(S.size), "worst_rel_err_beyond_print_rounding": worst,
            "against": "an engine filled on the device from the same seed (k_synth), every cell of every matrix"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="cfg3")
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--dir", default="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
    ap.add_argument("--n_threads", type=int, default=16)
    ap.add_argument("--gap", type=float, default=8.0, help="seconds between runs: a process that starts within seconds of another "
                    "one's exit can wait 1-3 s in its first large device allocation while the driver clears what that one "
                    "released (gpurun_out/r6/hiptrace_cfg3_3: 984 ms in ONE hipMalloc); 0 = back to back")
    ap.add_argument("--no_check", action="store_true")
    ap.add_argument("--no_roof", action="store_true")
    ap.add_argument("--keep", action="store_true", help="leave the generated files in --dir")
    ap.add_argument("--host_args", default="", help="extra arguments for the host, e.g. '--two_images'")
    args = ap.parse_args()
    gen, peak = build_tools(need_peak=not args.no_roof)
    roof = {} if args.no_roof else link_roof(peak)
    pinned = max([v for k, v in roof.items() if k.startswith("h2d_") and "stream" in k] or [0.0])
    if roof:
        print(json.dumps({"link_roof": roof, "pinned_h2d_best_GBps": pinned}), flush=True)
    for name in args.workloads.split(","):
        W = dict(WORKLOADS[name])
        n_ind, n_sites = W["n_ind"], W["n_sites"]
        size = n_ind * n_sites * 24
        fname = "ngd_e2e_%s_%dx%d_seed%d.bin" % (name, n_ind, n_sites, W["seed"])
        # the first directory that holds the file already or has room for it
        dirs = [d for d in (args.dir, "/tmp") if os.path.isdir(d)]
        have = [d for d in dirs if os.path.exists(os.path.join(d, fname)) and os.path.getsize(os.path.join(d, fname)) == size]
        room = [d for d in dirs if shutil.disk_usage(d).free > size * 1.05]
        if not have and not room:
            print(json.dumps({"workload": name, "error": "no room for %.1f GB in %s" % (size / 1e9, dirs)}), flush=True)
            continue
        wdir = (have or room)[0]
        path = os.path.join(wdir, fname)
        if not have:
            t0 = time.time()
            subprocess.check_call([gen, path, str(n_ind), str(n_sites), str(W["seed"]), str(args.n_threads)])
            t_gen = time.time() - t0
        else:
            t_gen = 0.0
        out = os.path.join(wdir, "ngd_e2e_%s.dist" % name)
        seed_rng = 12345
        cmd = [EXE, "--geno", path, "--probs", "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--evol_model", str(W["evol_model"]),
               "--out", out, "--verbose", "2", "--n_threads", str(args.n_threads), "--seed", str(seed_rng)]
        if W["indep"]:
            cmd.append("--indep_geno")
        if W["n_boot"]:
            cmd += ["--n_boot_rep", str(W["n_boot"]), "--boot_block_size", str(W["block"])]
        cmd += args.host_args.split()
        walls, phases_all, last_err = [], [], ""
        for r in range(args.runs):
            time.sleep(args.gap)
            if os.path.exists(out):  # (truncating the previous run's output -- 1.3 GB of text for the EM bootstrap job: 0.1 s
                os.remove(out)       # in tmpfs -- is not this run's work)
            t0 = time.perf_counter()
            pr = subprocess.run(cmd, capture_output=True, text=True)
            walls.append(time.perf_counter() - t0)
            last_err = pr.stderr
            if pr.returncode != 0:
                print(json.dumps({"workload": name, "error": pr.stderr[-500:]}), flush=True)
                break
            phases_all.append(parse_phases(pr.stderr))
        else:
            best = int(np.argmin(walls))
            ph = phases_all[best]
            n_mat = W["n_boot"] + 1
            n_pairs = n_ind * (n_ind - 1) // 2
            load_s = ph.get("load", float("nan"))
            line = {
                "metric": "pair-distances/sec, END TO END through the C++ host (process start to exit, file in host memory)",
                "workload": name,
                "config": {"workload": "%s: n_ind=%d n_sites=%d %s evol_model=%d n_boot_rep=%d boot_block_size=%d" % (
                    name, n_ind, n_sites, "--indep_geno" if W["indep"] else "EM", W["evol_model"], W["n_boot"], W["block"]),
                    "file": "%s (%.2f GB, generated in %.1f s)" % (path, size / 1e9, t_gen), "host_args": args.host_args,
                    "n_threads": args.n_threads, "seconds_between_runs": args.gap},
                "wall_s": walls[best], "wall_s_runs": walls,
                "value": n_pairs * n_mat / walls[best], "unit": "pair-distances/s",
                "phases_s": ph, "phases_s_runs": phases_all,
                "outside_main_s": walls[best] - ph.get("total_since_main", float("nan")),
                "file_bytes": size,
                "load_GBps": size / 1e9 / load_s,
                "roofline_load": {"bound": "host link (pinned hipMemcpy H2D, same lease)", "achieved": size / 1e9 / load_s,
                                  "peak": pinned or None, "unit": "GB/s", "frac": (size / 1e9 / load_s / pinned) if pinned else None},
                "stderr_tail": [ln for ln in last_err.splitlines() if ln.startswith(">")][-8:],
            }
            if not args.no_check:
                line["check"] = check_against_device_fill(W, out, seed_rng)
                line["valid"] = line["check"]["ok"]
            print(json.dumps(line), flush=True)
        if not args.keep:
            for f in (path, out):
                if os.path.exists(f):
                    os.remove(f)


if __name__ == "__main__":
    main()
