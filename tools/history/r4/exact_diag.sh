#!/bin/bash
# tools/exact_diag.sh -- where a k-group's time goes in the in-step 2 x 4 form of the MFMA kernel (cfg 2's shape):
# timing-only A/B builds (tools/build_variant.sh diag_X -DNGD_DIAG_X: no barrier / no operand loads / no MFMAs;
# their results are meaningless) beside the product build
set -e
cd "$(dirname "$0")/.."
for v in "" .diag_NOBARRIER .diag_NOLOAD .diag_NOMFMA; do
  echo "== libngsdist_amd.so$v"
  NGSDIST_AMD_LIB=$PWD/ngsdist_amd/libngsdist_amd.so$v timeout -k 10 120 python3 tools/shape_sweep.py 200 100000 mfma exact_shapes=5 2>&1 | grep -v amdgpu.ids
done
