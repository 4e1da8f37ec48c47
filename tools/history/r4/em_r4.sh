#!/bin/bash
# tools/em_r4.sh -- round 4's look at the table-driven EM kernel (cfg 4's shape, 20 000 sites): the product build against
# the build whose OLDER wavefronts build the column tables (A/B builds of tools/build_variant.sh: swap, stamps,
# swapstamps), bit-identity included, then where a wavefront's cycles go in both
set -e
rm -f /tmp/em_r4_ref.npz
cd "$(dirname "$0")/../.."
for v in "" .swap; do
  echo "== libngsdist_amd.so$v"
  NGSDIST_AMD_LIB=$PWD/ngsdist_amd/libngsdist_amd.so$v timeout -k 10 300 python3 tools/em_ab.py 20000 4 0 --ref /tmp/em_r4_ref.npz 2>&1 | grep -v amdgpu.ids
done
for v in .stamps .swapstamps; do
  echo "== libngsdist_amd.so$v"
  NGSDIST_AMD_LIB=$PWD/ngsdist_amd/libngsdist_amd.so$v timeout -k 10 120 python3 tools/em_stamps.py 20000 0 2>&1 | grep -v amdgpu.ids
done
