#!/bin/bash
# tools/rehearse_multi.sh -- the N > 1 flows of bench.py on a ONE-GPU box: gloo ranks that share cuda:0
# (--backend gloo --same_device).  Flow and full-size spot checks only; the timings mean nothing.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
run() { # n, args...
  local n=$1; shift
  echo "== N=$n $*"
  timeout -k 10 280 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
    bench.py --gpus $n --backend gloo --same_device --steps 2 --warmup 1 --no_cpu "$@" 2>&1 | grep -E '^\{|Error|error|Traceback' | \
    python3 -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l)
        print('  valid=%s value=%.4g ms_per_step=%.2f scaling=%s spot=%s' % (d['valid'], d['value'], d['ms_per_step'], d['scaling'], d['spot_check']))
        print('  sharding: ' + d['config']['sharding'])
    except Exception:
        print('  ' + l.rstrip())
"
}
run 2 --workload cfg3
run 4 --workload cfg3
run 4 --workload cfg5
run 2 --workload cfg2 --shard pairs
run 2 --workload cfg3 --shard replicates
run 2 --workload cfg4 --n_sites 50000
