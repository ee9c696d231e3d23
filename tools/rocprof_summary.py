"""Summarise a rocprofv3 `--kernel-trace --stats` run (sqlite .db or *_kernel_stats.csv) as a per-kernel table."""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    return [(r[0], r[1], r[2], r[3], r[4]) for r in c.execute(q)]


def from_csv(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["MinNs"]), float(r["MaxNs"])))
    return rows


def main():
    src = sys.argv[1]
    if os.path.isdir(src):
        cand = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True) + glob.glob(
            os.path.join(src, "**", "*.db"), recursive=True)
        src = cand[0]
    rows = from_db(src) if src.endswith(".db") else from_csv(src)
    tot = sum(r[2] for r in rows)
    print(f"# source: {os.path.basename(src)}   total kernel time {tot / 1e6:.3f} ms")
    print(f"{'kernel':100s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for n, calls, t, mn, mx in rows:
        print(f"{n[:100]:100s} {calls:6d} {t / calls / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {t / 1e6:10.3f} {100 * t / tot:6.2f}")


if __name__ == "__main__":
    main()
