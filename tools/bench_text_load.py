#!/usr/bin/env python3
"""tools/bench_text_load.py -- gz-text GL input through the host CLI: wall time of the whole run at
--n_threads 1 and 16, and byte-identity of the two outputs.  Run on the GPU box."""
import gzip, os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n_ind, n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
out = os.path.join(ROOT, "gpurun_out", "textload")
os.makedirs(out, exist_ok=True)
path = os.path.join(out, "gl.txt.gz")
rng = np.random.default_rng(1)
t = time.perf_counter()
with gzip.open(path, "wt", compresslevel=1) as fh:
    for s0 in range(0, n_sites, 1000):
        x = rng.random((min(1000, n_sites - s0), n_ind * 3)) ** 3
        for row in x:
            fh.write("chr1\t%d\t" % s0 + "\t".join("%.6f" % v for v in row) + "\n")
print("wrote %s (%.1f MB gz) in %.1f s" % (path, os.path.getsize(path) / 1e6, time.perf_counter() - t), flush=True)
# the same text as BGZF (what bgzip / htslib / ANGSD write): members of <= 64 KB, inflated on --n_threads threads
import struct, zlib
bpath = os.path.join(out, "gl.bgzf.gz")
data = gzip.open(path, "rb").read()
with open(bpath, "wb") as fh:
    for k in list(range(0, len(data), 0xff00)) + [None]:
        chunk = b"" if k is None else data[k:k + 0xff00]
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = c.compress(chunk) + c.flush()
        fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp
                 + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
print("wrote %s (%.1f MB, %.1f MB of text)" % (bpath, os.path.getsize(bpath) / 1e6, len(data) / 1e6), flush=True)
del data
exe = os.path.join(ROOT, "ngsdist_amd", "bin", "ngsDist")
res = []
for nt, path in ((1, path), (16, path), (1, bpath), (16, bpath)):
    o = os.path.join(out, "o%d_%d.dist" % (nt, len(res)))
    t = time.perf_counter()
    r = subprocess.run([exe, "--geno", path, "--probs", "--n_ind", str(n_ind), "--n_sites", str(n_sites), "--indep_geno",
                    "--n_threads", str(nt), "--out", o, "--verbose", "2"], check=True, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t
    res.append(open(o, "rb").read())
    load = [l for l in r.stderr.decode().splitlines() if "read + prepare" in l]
    for l in r.stderr.decode().splitlines():
        if "text load:" in l:
            print("   ", l.strip())
    print("%s n_threads=%d: %.2f s end to end; %s" % (os.path.basename(path), nt, dt, load[0].strip() if load else "?"), flush=True)
print("outputs identical:", all(r == res[0] for r in res))
for f in os.listdir(out):  # keep gpurun_out/ small
    os.remove(os.path.join(out, f))
