#!/bin/bash
# usage (GPU box): tools/ab_bench.sh ENV_VAR v1 v2 [...]  -- default bench (no kNN / CPU legs), 80 steps, each value twice, interleaved
VAR=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = default ]; then unset $VAR; else export $VAR=$v; fi
  python3 $R/bench.py --no-cpu-baseline --knn-n 0 --no-cfg3 --no-cfg4 --no-cfg5 --steps 80 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step'],4))"
done; done
