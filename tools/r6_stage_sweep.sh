#!/bin/bash
# tools/r6_stage_sweep.sh -- cfg 3 through the C++ host with the load pipeline's geometry varied (--stage piece,ring,share,drop):
# the host's own `> phases` line of the second of two runs each, + wall / user / sys of the process.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
F=/dev/shm/ngd_e2e_cfg3_1000x1000000_seed3.bin
[ -f $F ] || $ROOT/tools/gen_gl_file $F 1000 1000000 3 16
SWEEP=${SWEEP:-32,6,2,1 32,6,2,0 32,8,2,1 64,4,2,1 64,6,2,1 128,4,2,1 16,8,2,1 32,6,1,1 32,6,4,1 32,6,8,1}
TIMEFORMAT="%R s wall %U user %S sys"
for st in $SWEEP; do
  for thr in ${THREADS:-16}; do
    for i in 1 2; do
      { time $ROOT/ngsdist_amd/bin/ngsDist --geno $F --probs --n_ind 1000 --n_sites 1000000 --evol_model 1 --indep_geno \
        --out /tmp/x.dist --verbose 2 --n_threads $thr --stage $st > /tmp/sweep.log 2>&1 ; } 2> /tmp/sweep.time
    done
    echo "stage=$st threads=$thr: $(grep phases /tmp/sweep.log) | $(cat /tmp/sweep.time)"
    sleep 2
  done
done
