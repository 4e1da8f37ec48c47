#!/bin/bash
# tools/exact_sweep.sh -- the MFMA kernel's exact block forms (exact_shapes 2: blocks of 4 x 4 tiles, 3: 2 x 4) at a few
# hundred individuals, over slice counts (0 = the engine's choice); run on the GPU box
cd "$(dirname "$0")/../.."
for n in 24 100 200 300 384; do
  for es in 2 3; do
    for ks in 0 160 240 320 480 640; do
      python3 tools/shape_sweep.py $n 100000 mfma exact_shapes=$es n_slices=$ks
    done
  done
done
