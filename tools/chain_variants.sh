#!/bin/bash
# per-layer table of the detect stage for the EIOKU_CHAIN_DB variants (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in default 0 1; do
  if [ "$v" = default ]; then unset EIOKU_CHAIN_DB; else export EIOKU_CHAIN_DB=$v; fi
  rm -rf /tmp/kt_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$v -- python3 $R/bench.py --steps 4 --warmup 2 --stages detect --knn-n 0 --no-cpu-baseline --overlap 0 --depth 1 > /tmp/kt_$v.log 2>&1 || exit 1
  echo "== EIOKU_CHAIN_DB=$v"
  python3 $R/tools/conv_table.py /tmp/kt_$v n 64 | grep -E "chain|TOTAL|fused"
done
