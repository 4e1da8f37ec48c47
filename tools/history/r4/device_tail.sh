#!/bin/bash
# tools/device_tail.sh [cells] -- builds and runs tools/device_tail.hip (is the tail of gen_dist() on the device
# bit-equal to the product's host tail?) on the GPU box
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-value tools/device_tail.hip -o tools/device_tail \
  -Lngsdist_amd -lngsdist_amd -Wl,-rpath,"$PWD/ngsdist_amd"
timeout -k 10 600 tools/device_tail ${1:-100000000}
