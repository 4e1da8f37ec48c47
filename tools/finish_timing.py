#!/usr/bin/env python3
"""tools/finish_timing.py -- what a call of ngd_finish() costs on this box's host cores: the 8.1e6 cells of a cfg 5 job in
one call, in 8 and in 65 calls (per-call overhead), and the 62 437-cell share of one rank of an 8-GPU cfg 3 step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngsdist_amd as N  # noqa: E402

n = 65 * 124750
rng = np.random.default_rng(0)
cnt = np.full(n, 500000, dtype=np.uint64)
s = rng.random(n) * 0.3 * 500000
out = np.empty(n)
for rep in range(4):
    t = time.perf_counter(); N.finish(s, cnt, 0, 1, out=out); t1 = time.perf_counter() - t
    k = n // 8
    t = time.perf_counter()
    for c in range(8):
        N.finish(s[c * k:(c + 1) * k], cnt[c * k:(c + 1) * k], 0, 1, out=out[c * k:(c + 1) * k])
    t8 = time.perf_counter() - t
    t = time.perf_counter()
    for c in range(65):
        N.finish(s[c * 124750:(c + 1) * 124750], cnt[c * 124750:(c + 1) * 124750], 0, 1, out=out[c * 124750:(c + 1) * 124750])
    t65 = time.perf_counter() - t
    m = 62437
    t = time.perf_counter(); N.finish(s[:m], cnt[:m], 0, 1, out=out[:m]); ts = time.perf_counter() - t
    print("one call %.2f ms, 8 calls %.2f ms, 65 calls %.2f ms; 62 437 cells %.3f ms" % (t1 * 1e3, t8 * 1e3, t65 * 1e3, ts * 1e3))
print("host cpus:", os.cpu_count())
