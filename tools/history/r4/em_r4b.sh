#!/bin/bash
# tools/em_r4b.sh -- wavefront priorities in the table-driven EM kernel (A/B builds fair, young of tools/build_variant.sh)
set -e
cd "$(dirname "$0")/../.."
rm -f /tmp/em_r4_ref.npz
for v in "" .fair .young ""; do
  echo "== libngsdist_amd.so$v"
  NGSDIST_AMD_LIB=$PWD/ngsdist_amd/libngsdist_amd.so$v timeout -k 10 300 python3 tools/em_ab.py 20000 0 --ref /tmp/em_r4_ref.npz 2>&1 | grep -v amdgpu.ids | grep -v pairwise
done
