#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 run (rocpd sqlite .db or *_kernel_trace.csv)."""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(r[0], r[1]) for r in cur.execute("select name, end-start from kernels")]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main(root):
    rows = []
    for p in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
        rows += from_db(p)
    for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        rows += from_csv(p)
    agg = defaultdict(list)
    for name, ns in rows:
        agg[name.split("(")[0]].append(ns)
    tot = sum(sum(v) for v in agg.values())
    print("%-44s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-44s %8d %12.3f %10.2f %10.2f %10.2f %6.1f" % (name[-44:], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3,
                                                                min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
