#!/bin/bash
# tools/qb_probe.sh -- what dropping the second operand image (QB = score . P) would cost the MFMA kernel: rebuilds the
# library ON THE GPU BOX with -DNGD_QB_PROBE=<n> (n neutral FP64 FMAs per k-group on the B fragments: the in-lane
# 3 x 3 score product a PA-only kernel would have to do), runs the cfg 3 bench, then restores the product build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for n in 0 8 12; do
  touch ngsdist_amd/csrc/accum_mfma.hip
  if [ $n = 0 ]; then make -s -C ngsdist_amd/csrc; else make -s -C ngsdist_amd/csrc EXTRA=-DNGD_QB_PROBE=$n; fi
  python3 bench.py --workload cfg3 --steps 10 --warmup 3 --no_cpu 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NGD_QB_PROBE=$n: %.2f ms per launch, frac %.3f, valid %s' % (d['roofline']['ms_per_launch'], d['roofline']['frac'], d['valid']))"
done
touch ngsdist_amd/csrc/accum_mfma.hip && make -s -C ngsdist_amd/csrc
