#!/bin/bash
# A/B libraries (tags) on cfg 3 with --single_image: ms per step (timing-only builds print valid=False)
set -e
mkdir -p gpurun_out
for tag in "$@"; do
  NGSDIST_AMD_LIB=ngsdist_amd/libngsdist_amd.so.$tag timeout -k 10 300 python3 bench.py --workload cfg3 --single_image --no_cpu --steps 10 --warmup 3 > gpurun_out/r4_ab_$tag.json 2> gpurun_out/r4_ab_$tag.err || { tail -5 gpurun_out/r4_ab_$tag.err; continue; }
  python3 - "$tag" <<'PY'
import json, sys
t = sys.argv[1]
j = json.loads(open("gpurun_out/r4_ab_%s.json" % t).read().strip().splitlines()[-1])
print(t, "%.2f ms/step" % j["ms_per_step"], "kernel median %.2f" % j["roofline"]["ms_per_launch_median"], j.get("valid"))
PY
done
