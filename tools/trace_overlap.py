#!/usr/bin/env python
"""How much do kernels of different streams overlap?  Reads a rocprofv3 --kernel-trace csv and prints, for the
training kernels, total kernel time, the union of busy intervals, and a sample of the timeline."""
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:28], r.get("Queue_Id", "?")))
rows.sort()
train = [r for r in rows if r[2].startswith(("k_w1_update", "k_mid", "k_reduce"))]
tot = sum(e - s for s, e, _, _ in train)
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in train:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
span = train[-1][1] - train[0][0]
print("training kernels: %d, sum of durations %.1f ms, union of busy time %.1f ms, first-to-last span %.1f ms" % (len(train), tot / 1e6, union / 1e6, span / 1e6))
t0 = train[len(train) // 2][0]
for s, e, n, q in train[len(train) // 2: len(train) // 2 + 16]:
    print("  +%8.1f us  %6.1f us  q=%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
