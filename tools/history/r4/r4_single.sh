#!/bin/bash
# round 4: the single-image engines -- parity tests, then cfg 3 with both images, with one image and the other formed in
# ranges (mode 1; other scratch sizes, part of the second image resident), and with one image in congruent coordinates (2)
set -e
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_abi.py -x -q -m gpu -k "single_image or abi" > gpurun_out/r4_single_tests.log 2>&1 || { tail -30 gpurun_out/r4_single_tests.log; exit 1; }
tail -2 gpurun_out/r4_single_tests.log
line() { python3 - "$1" <<'PY'
import json, sys
f = sys.argv[1]
j = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
print(f, "%.2f ms/step" % j["ms_per_step"], "frac %.3f" % j["roofline"]["frac"], "%.1f GB" % (j["config"]["device_bytes"] / 1e9), j.get("valid"),
      "spot check %.2e" % j["spot_check"]["max_rel_err_vs_oracle"])
PY
}
B="timeout -k 10 300 python3 bench.py --workload cfg3 --no_cpu --steps 10 --warmup 3"
$B > gpurun_out/r4_two_cfg3.json 2> gpurun_out/r4_two_cfg3.err
line r4_two_cfg3
$B --single_image 2 > gpurun_out/r4_single2_cfg3.json 2> gpurun_out/r4_single2_cfg3.err
line r4_single2_cfg3
$B --single_image 1 > gpurun_out/r4_single_cfg3.json 2> gpurun_out/r4_single_cfg3.err
line r4_single_cfg3
for gb in ${SCRATCH_GB:-}; do
  $B --single_image 1 --single_image_gb $gb > gpurun_out/r4_single_cfg3_${gb}gb.json 2> gpurun_out/r4_single_cfg3_${gb}gb.err
  line r4_single_cfg3_${gb}gb
done
for gb in ${RESIDENT_GB:-}; do
  $B --single_image 1 --second_image_gb $gb > gpurun_out/r4_single_cfg3_res${gb}.json 2> gpurun_out/r4_single_cfg3_res${gb}.err
  line r4_single_cfg3_res${gb}
done
