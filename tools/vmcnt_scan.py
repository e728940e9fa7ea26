"""Flags kernels whose ISA has `s_waitcnt vmcnt(0)` between two global stores that are close together (a store round
trip per store: the pattern r3 found in the K8 GEMM epilogue and in k_conv1x1) or inside an MFMA loop.
usage: tools/vmcnt_scan.py <file.hip> [extra hipcc flags]"""
import re, subprocess, sys, os
src = sys.argv[1]; extra = sys.argv[2:]
subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only",
                       "-Iinclude", "-Ieioku_amd/csrc", *extra, "-c", src, "-o", "/tmp/scan_dev.o"])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--unbundle", "--input=/tmp/scan_dev.o", "--type=o",
                       "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=/tmp/scan_gfx950.o"])
asm = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", "/tmp/scan_gfx950.o"], text=True)
demangle = lambda s: subprocess.check_output(["c++filt", s], text=True).strip()
cur, ops = None, []
kernels = {}
for line in asm.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
        cur = m.group(1); kernels[cur] = []; continue
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*(//.*)?$", line)
    if m and cur: kernels[cur].append((m.group(1), m.group(2)))
for name, ins in kernels.items():
    st = [i for i, (op, _) in enumerate(ins) if op.startswith("global_store") or op.startswith("buffer_store")]
    w0 = [i for i, (op, a) in enumerate(ins) if op == "s_waitcnt" and "vmcnt(0)" in a]
    between = 0
    for a, b in zip(st, st[1:]):
        if b - a < 80 and any(a < w < b for w in w0): between += 1
    if between >= 3:
        print(f"{between:4d} store..vmcnt(0)..store  of {len(st):4d} stores   {demangle(name)[:150]}")
