#!/bin/bash
# tools/em_dense_sweep.sh [n_sites] -- the table-driven EM kernel's PACK_DENSE threshold (rows with at least that many
# pairs still searching are scanned the plain way in later rounds, the rest goes into packed units): rebuilds the
# library on the GPU box per value, times the cfg 4 shape, restores the product build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NS=${1:-20000}
cd "$ROOT"
for n in 8 16 24 32 48 65; do
  touch ngsdist_amd/csrc/accum_em_table.hip
  make -s -C ngsdist_amd/csrc EXTRA=-DNGD_PACK_DENSE=$n
  echo "PACK_DENSE=$n"; timeout -k 10 200 python3 tools/em_ab.py $NS 0 2>&1 | grep "^plain"
done
touch ngsdist_amd/csrc/accum_em_table.hip && make -s -C ngsdist_amd/csrc
