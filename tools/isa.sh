#!/bin/bash
# usage: tools/isa.sh <file.hip> <mangled-substring> [extra hipcc flags]
# Compiles one translation unit for gfx950 (device only), disassembles the first kernel whose symbol
# contains the substring and prints a condensed stream: runs of the same instruction class collapsed.
SRC=$1; SUB=$2; shift 2
D=$(dirname "$SRC")
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off --cuda-device-only "$@" -c "$SRC" -o /tmp/isa_dev.o || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --input=/tmp/isa_dev.o --type=o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=/tmp/isa_gfx950.o || exit 1
SYM=$(/opt/rocm/lib/llvm/bin/llvm-readelf -s /tmp/isa_gfx950.o | grep FUNC | awk '{print $8}' | grep -- "$SUB" | head -1)
echo "symbol: $SYM  bytes: $(/opt/rocm/lib/llvm/bin/llvm-readelf -s /tmp/isa_gfx950.o | grep FUNC | grep -- "$SYM" | head -1 | awk '{print $3}')"
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn /tmp/isa_gfx950.o --disassemble-symbols=$SYM > /tmp/isa.s
python3 - <<'PY'
import re
out = []
def cls(op, rest):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "GLOAD"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "GSTORE"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "DSR"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "DSW"
    if op == "s_waitcnt": return "wait " + rest.strip()
    if op == "s_barrier": return "BARRIER"
    if op.startswith("s_cbranch") or op == "s_branch": return "br"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("scratch_"): return "SCRATCH"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    return op
prev, n = None, 0
ln = 0
for line in open("/tmp/isa.s"):
    m = re.match(r"\s+(\S+)\s*(.*?)(//.*)?$", line)
    if not m or line.strip().endswith(":"): continue
    ln += 1
    c = cls(m.group(1), m.group(2))
    if c == prev: n += 1
    else:
        if prev: out.append(f"{prev}x{n}" if n > 1 else prev)
        prev, n = c, 1
out.append(f"{prev}x{n}")
print(f"{ln} instructions")
print(" ".join(out))
PY
