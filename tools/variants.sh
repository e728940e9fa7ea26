#!/bin/bash
# usage (on the GPU box): tools/variants.sh ENV_VAR value [value ...]
# Detect-stage kernel trace per value of one environment switch; prints the first rows of tools/conv_table.py and its total.
VAR=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = default ]; then unset $VAR; else export $VAR=$v; fi
  rm -rf /tmp/kt_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -- python3 $R/bench.py --steps 4 --warmup 2 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 > /tmp/kt_$v.log 2>&1 || { tail -5 /tmp/kt_$v.log; exit 1; }
  echo "== $VAR=$v"
  python3 $R/tools/conv_table.py /tmp/kt_$v n 64 | grep -E "${ROWS:-^model\.[0-4]\.|TOTAL}" | cut -c1-160
done
