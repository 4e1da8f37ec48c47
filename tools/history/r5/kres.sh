#!/bin/bash
# tools/kres.sh <file.hip> [extra hipcc flags]: registers / spills / LDS / occupancy of every kernel in a source file
# (hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel; runs without a GPU)
f=$1; shift
cd "$(dirname "$0")/../../ngsdist_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | sed -e 's/ \[-Rpass[^]]*\]//g' |
  awk '/Function Name:/ {name=$NF}
       / VGPRs:/ {v=$NF} / AGPRs:/ {a=$NF} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$NF}
       /SGPRs Spill/ {ss=$NF} /VGPRs Spill/ {vs=$NF}
       /LDS Size/ {l=$NF; printf "%s\tvgpr %s agpr %s scratch %s occ %s sspill %s vspill %s lds %s\n", name, v, a, sc, o, ss, vs, l}' |
  while IFS=$'\t' read -r n rest; do
    d=$(echo "$n" | c++filt | sed -e 's/(anonymous namespace):://' -e 's/>(.*$/>/' -e 's/^void //')
    printf "%-60s %s\n" "$d" "$rest"
  done
