#!/bin/bash
# tools/em_stamps.sh [n_sites] [shapes] -- where a wavefront of the table-driven EM kernel spends its cycles: rebuilds the
# library ON THE GPU BOX with -DNGD_EMT_STAMPS (s_memtime stamps around the phases of a round; the 'sums' such a build
# returns are cycle totals, not distances), prints them per wavefront, then restores the product build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NS=${1:-20000}
SH=${2:-0}
cd "$ROOT"
touch ngsdist_amd/csrc/accum_em_table.hip
make -s -C ngsdist_amd/csrc EXTRA=-DNGD_EMT_STAMPS
for q in ${SH//,/ }; do timeout -k 10 120 python3 tools/em_stamps.py $NS $q 2>&1 | grep -v amdgpu.ids; done
touch ngsdist_amd/csrc/accum_em_table.hip && make -s -C ngsdist_amd/csrc
