#!/bin/bash
# tools/r6_load_trace.sh [tag] -- the C++ host on cfg 3 under rocprofv3: kernel trace + memory-copy trace with stats (what
# do K0 `k_prep_layout` and the H2D copies of the load phase take?), then FETCH_SIZE / WRITE_SIZE of K0 in two separate
# counter passes (--pmc with --kernel-trace only).  Output under gpurun_out/r6/load_trace_<tag>/.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-0}
F=/dev/shm/ngd_e2e_cfg3_1000x1000000_seed3.bin
[ -f $F ] || $ROOT/tools/gen_gl_file $F 1000 1000000 3 16
OUT=$ROOT/gpurun_out/r6/load_trace_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--geno $F --probs --n_ind 1000 --n_sites 1000000 --evol_model 1 --indep_geno --out /tmp/x.dist --verbose 2 --n_threads 16 $HOST_ARGS"
$ROOT/ngsdist_amd/bin/ngsDist $ARGS > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/kt -- $ROOT/ngsdist_amd/bin/ngsDist $ARGS > $OUT/kt.log 2>&1
if [ -z "$NO_PMC" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $ROOT/ngsdist_amd/bin/ngsDist $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $ROOT/ngsdist_amd/bin/ngsDist $ARGS > $OUT/write.log 2>&1
fi
grep -h phases $OUT/plain.log $OUT/kt.log
find $OUT -name "*_kernel_stats.csv" -exec head -5 {} \;
find $OUT -name "*memory_copy_stats.csv" -exec head -5 {} \;
