"""Concurrency summary of a rocprofv3 --kernel-trace CSV: per queue kernel counts, sum of kernel durations vs union busy
time over the last `bench.py` step (window = last N kernels).  usage: python tools/trace_overlap.py trace.csv [n_last]"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 75
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
win = rows[-n_last:]
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win]
total = sum(e - s for s, e in iv)
busy, cur_s, cur_e = 0, None, None
for s, e in sorted(iv):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e in iv) - min(s for s, _ in iv)
print("kernels", len(win), "queues", collections.Counter(r["Queue_Id"] for r in win))
print(f"sum of durations {total / 1e3:.1f} us, union busy {busy / 1e3:.1f} us, span {span / 1e3:.1f} us")
t0 = min(s for s, _ in iv)
for r in win:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"  q{r['Queue_Id']:>2} {(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f} us  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}  {r['Kernel_Name'][:70]}")
