#!/bin/bash
# usage: tools/pmc.sh <kernel-substring> <outfile> -- <python script args...>
# Runs rocprofv3 --pmc in separate passes (SQ 8 slots, TCC 4) and prints per-dispatch averages.
KSUB=$1; OUT=$2; shift 3
cd /tmp && export TMPDIR=/tmp
: > $OUT
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
         "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pm
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pm -- python3 "$@" > /tmp/pm.log 2>&1
  F=$(ls /tmp/pm/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$F" ]; then echo "no counters for: $c" >> $OUT; tail -3 /tmp/pm.log >> $OUT; continue; fi
  python3 - "$F" "$KSUB" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k, v in acc.items():
    print(f"{k} avg_per_dispatch {v / n[k]:.6g} dispatches {n[k]}")
PY
done
cat $OUT
