#!/bin/bash
# tools/exact_depth_sweep.sh -- the 2 x 4 block form of the MFMA kernel with 1, 2 and 4 k-groups of operands in flight
# per wavefront (A/B builds: tools/build_variant.sh x2d2 -DNGD_EXACT2_DEPTH=2, x2d4 ...=4), at cfg 2's and cfg 5's shapes
cd "$(dirname "$0")/.."
for lib in "" .x2d2 .x2d4; do
  echo "== libngsdist_amd.so$lib"
  export NGSDIST_AMD_LIB=$PWD/ngsdist_amd/libngsdist_amd.so$lib
  for ks in 160 320 400 640; do python3 tools/shape_sweep.py 200 100000 mfma exact_shapes=3 n_slices=$ks; done
  for n in 100 300 384; do python3 tools/shape_sweep.py $n 100000 mfma exact_shapes=3 n_slices=320; done
  for ks in 256 504 1000; do python3 tools/shape_sweep.py 500 500000 mfma exact_shapes=3 n_slices=$ks; done
done
unset NGSDIST_AMD_LIB
for ks in 0 504; do python3 tools/shape_sweep.py 500 500000 mfma n_slices=$ks; python3 tools/shape_sweep.py 500 500000 mfma exact_shapes=2 n_slices=$ks; done
