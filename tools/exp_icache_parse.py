import csv, sys, statistics, collections
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
names = ["A only", "B only", "C only", "A,B alternating", "A,B,C rotating", "A, 256 MB fill"]
ph = -1; acc = collections.defaultdict(list)
for b, e, n in rows:
    if "add" in n.lower() and "conv" not in n: ph += 1; continue
    if ph < 0 or "conv" not in n: continue
    key = "igemm<" + n.split("conv_igemm_f16_kernel<")[1].split(",")[0] + ">" if "conv_igemm_f16_kernel<" in n else n[:40]
    acc[(ph, key)].append((e - b) / 1e3)
for (p, k), v in sorted(acc.items()):
    v = v[5:]
    print(f"{names[p] if p < len(names) else p:20s} {k:12s} n {len(v):4d} median {statistics.median(v):6.2f} us  mean {statistics.mean(v):6.2f}")
