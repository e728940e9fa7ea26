#!/bin/bash
# GPU box: kernel time of the list-major IVF-PQ scan under the measurement switches of k_lscan
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export IVFPQ_LISTS_ONLY=1
for nw in 8 4; do
for ab in ${ABL:-0 1 2 3}; do
  rm -rf /tmp/prof
  EIOKU_LSCAN_NW=$nw EIOKU_LSCAN_ABLATE=$ab timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $R/tools/ivfpq_bench.py ${1:-10000000} 1024 > /tmp/ab.json 2>/tmp/err.log
  echo "nw=$nw ablate=$ab $(python3 $R/tools/kstats.py /tmp/prof 22 60 | grep -E 'k_lscan')"
  python3 -c "import json;d=json.load(open('/tmp/ab.json'));print('   search_ms',d['search_ms'],'cand/q',d['candidates_per_query'],'items',d['work_items'],'ovf',d['overflow_flag'])"
done
done
