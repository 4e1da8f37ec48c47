#!/bin/bash
# tools/em_ablate.sh [n_sites] [shapes] -- where the table-driven EM kernel's time goes: rebuilds the library ON THE GPU BOX
# with -DNGD_EMT_ABLATE=1 (tables built once per slice: scan cost) and =2 (no scan: build + barrier cost), times each,
# then restores the product build.  Results of the ablated builds are wrong by construction; only their times matter.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NS=${1:-20000}
SH=${2:-0,1,2,3}
cd "$ROOT"
for A in ${ABL:-0 1 3 4}; do
  touch ngsdist_amd/csrc/accum_em_table.hip
  if [ $A = 0 ]; then make -s -C ngsdist_amd/csrc; else make -s -C ngsdist_amd/csrc EXTRA=-DNGD_EMT_ABLATE=$A; fi
  echo "== ablate $A (0 = product, 1 = tables once per slice, 2 = no scan, 4 = no global loads, 5 = cycle stamps)"
  if [ $A = 5 ]; then for q in ${SH//,/ }; do timeout -k 10 120 python3 tools/em_stamps.py $NS $q 2>&1 | grep -v amdgpu.ids; done
  else timeout -k 10 120 python3 tools/em_shapes.py $NS $SH 2>&1 | grep em_table; fi
done
touch ngsdist_amd/csrc/accum_em_table.hip && make -s -C ngsdist_amd/csrc
