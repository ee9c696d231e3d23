"""Per-kernel duration and idle gap from a rocprofv3 kernel_trace.csv: steady-state steps of bench.py (single stream).
A step is delimited by the stem kernel (conv_stem_lds_kernel).  Prints the median step's kernel list."""
import csv, sys, statistics
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "conv_stem" in r[2]]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = [s for s in steps if 60 <= len(s) <= 90]          # plain enqueue steps (not profile/parity passes)
if not steps:
    print("no steps found", len(rows), len(starts)); sys.exit(0)
def short(n):
    n = n.replace("void ", "").replace("trtx::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]
spans = [s[-1][1] - s[0][0] for s in steps]
med = sorted(range(len(steps)), key=lambda i: spans[i])[len(steps) // 2]
s = steps[med]
busy = sum(e - b for b, e, _ in s)
gaps = [max(0, s[i][0] - s[i - 1][1]) for i in range(1, len(s))]
print(f"# {len(steps)} steps, median-step span {spans[med]/1e3:.1f} us, kernels {len(s)}, busy {busy/1e3:.1f} us, gaps {sum(gaps)/1e3:.1f} us "
      f"(median gap {statistics.median(gaps)/1e3:.2f} us)")
for i, (b, e, n) in enumerate(s):
    g = 0 if i == 0 else b - s[i - 1][1]
    print(f"{i:3d} dur {(e-b)/1e3:7.2f} gap {g/1e3:6.2f}  {short(n)}")
allgaps = [max(0, st[i][0] - st[i - 1][1]) for st in steps for i in range(1, len(st))]
allbusy = [sum(e - b for b, e, _ in st) for st in steps]
print(f"# all steps: mean span {statistics.mean(spans)/1e3:.1f} us, mean busy {statistics.mean(allbusy)/1e3:.1f} us, mean gap {statistics.mean(allgaps)/1e3:.2f} us x {len(s)-1}")
