#!/bin/bash
# tools/exact_sync_sweep.sh [forms] -- the MFMA kernel's exact block forms at small n_ind (exact_shapes 2: 4 x 4 blocks,
# 3: 2 x 4, 4 / 5: the same with a slice's jobs in one workgroup, in step); stops at the first failure.
# NGSDIST_AMD_LIB picks an A/B build (tools/build_variant.sh).
set -e
cd "$(dirname "$0")/.."
FORMS=${1:-"2 4 5"}
for n in 24 100 200 208 250; do
  for es in $FORMS; do
    for ks in 0 256 512; do
      timeout -k 10 120 python3 tools/shape_sweep.py $n 100000 mfma exact_shapes=$es n_slices=$ks 2>&1 | grep -v amdgpu.ids
    done
  done
done
