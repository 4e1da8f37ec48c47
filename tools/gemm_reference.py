#!/usr/bin/env python3
"""tools/gemm_reference.py -- what the vendor FP64 GEMM (torch.mm -> rocBLAS / hipBLASLt) does on the shape of the
indep-path contraction: C[1000 x 1000] = P[1000 x K] . Q[1000 x K]^T, K = 3e6.  A yardstick for k_accum_mfma, which
computes only the upper triangle (half the flops) of the same product.  Measurement only; nothing in the product uses it."""
import sys, time
import torch
n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000
dev = torch.device("cuda", 0)
P = torch.rand((n, K), dtype=torch.float64, device=dev)
Q = torch.rand((n, K), dtype=torch.float64, device=dev)
for layout, fn in (("P[n][K] . Q[n][K]^T (K contiguous)", lambda: torch.mm(P, Q.t())),):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ms = min(ts)
    print("%s: %.2f ms for the full %d x %d product = %.1f TFLOP/s (2 n^2 K flops); the upper triangle alone at that rate: %.2f ms"
          % (layout, ms, n, n, 2.0 * n * n * K / ms / 1e9, ms * (n - 1) / (2.0 * n)))
Pt, Qt = P.t().contiguous(), Q.t().contiguous()
del P, Q
fn = lambda: torch.mm(Pt.t(), Qt)
fn(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = min(ts)
print("P[K][n]^T . Q[K][n] (individuals contiguous): %.2f ms = %.1f TFLOP/s; triangle at that rate %.2f ms"
      % (ms, 2.0 * n * n * K / ms / 1e9, ms * (n - 1) / (2.0 * n)))
