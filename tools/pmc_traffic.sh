#!/bin/bash
# usage: tools/pmc_traffic.sh <kernel-substring> <forwards> <out.json> -- <python script + args>
# HBM traffic of a kernel family from PMC counters, per MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (TCC slots), units KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream
# (TCC_EA0_RDREQ x 64 B with 128-B requests) -> doubled.
KSUB=$1; FWD=$2; OUT=$3; shift 4
cd /tmp && export TMPDIR=/tmp
declare -A TOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d /tmp/pm -- python3 "$@" > /tmp/pm.log 2>&1
  F=$(ls /tmp/pm/*/*counter_collection.csv 2>/dev/null | head -1)
  TOT[$c]=$(python3 - "$F" "$KSUB" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        tot += float(r["Counter_Value"])
print(tot)
PY
)
done
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 - "$OUT" "$KSUB" "$FWD" "${TOT[FETCH_SIZE]}" "${TOT[WRITE_SIZE]}" "$R" <<'PY'
import json, sys
sys.path.insert(0, sys.argv[6])
import bench
out, ksub, fwd, fetch_kib, write_kib = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5])
d = {"kernel_family": ksub, "forwards": fwd, "csrc_sha16": bench.kernel_source_hash(),
     "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
     "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE as is; KiB -> bytes",
     "hbm_bytes_per_forward": (2.0 * fetch_kib + write_kib) * 1024.0 / fwd}
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
PY
