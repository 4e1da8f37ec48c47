#!/usr/bin/env python3
"""tools/bench_loader.py -- end-to-end wall time of the C++ host on a generated binary GL file
(load + prepare + upload vs. distances), host-side vs. device-side preparation.
usage: bench_loader.py [n_ind] [n_sites] [n_threads]"""
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n_ind = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n_thr = int(sys.argv[3]) if len(sys.argv) > 3 else 16
path = "/tmp/ngd_loader_%dx%d.bin" % (n_ind, n_sites)
if not os.path.exists(path):
    rng = np.random.default_rng(1)
    with open(path, "wb") as fh:
        for s0 in range(0, n_sites, 10000):
            n = min(10000, n_sites - s0)
            (rng.random((n, n_ind, 3)) ** 3 + 1e-9).tofile(fh)
print("file %s: %.2f GB" % (path, os.path.getsize(path) / 1e9))
exe = os.path.join(ROOT, "ngsdist_amd", "bin", "ngsDist")
outs = {}
for prep in ("host", "device"):
    for rep in range(2):
        t0 = time.time()
        r = subprocess.run([exe, "--geno", path, "--probs", "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--indep_geno",
                            "--out", "/tmp/ngd_loader_%s.dist" % prep, "--verbose", "2", "--n_threads", str(n_thr),
                            "--prep", prep], capture_output=True, text=True)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr
    load = re.search(r"read \+ prepare \+ upload: ([0-9.]+) s", r.stderr).group(1)
    comp = re.search(r"distances: ([0-9.]+) s", r.stderr).group(1)
    outs[prep] = open("/tmp/ngd_loader_%s.dist" % prep).read()
    for l in r.stderr.splitlines():
        if "staged load" in l:
            print("   ", l.strip())
    print("prep=%-6s wall %.3f s | read+prepare+upload %s s (%.2f GB/s of file) | distances %s s" % (
        prep, dt, load, os.path.getsize(path) / 1e9 / float(load), comp))
print("outputs identical:", outs["host"] == outs["device"])
