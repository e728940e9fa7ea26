"""Idle time between kernels from a rocprofv3 kernel trace: python tools/gaps.py <dir> [skip_first_n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[skip:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
pos = [g for g in gaps if g > 0]
print(f"{len(rows)} kernels, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms ({100*busy/span:.1f}%), "
      f"gaps {sum(pos)/1e6:.3f} ms, median gap {sorted(pos)[len(pos)//2]/1e3:.2f} us, overlapped {sum(1 for g in gaps if g < 0)}")
big = sorted(((g, a["Kernel_Name"][:50], b["Kernel_Name"][:50]) for g, a, b in zip(gaps, rows, rows[1:])), reverse=True)[:8]
for g, a, b in big:
    print(f"  {g/1e3:8.1f} us after {a} before {b}")
