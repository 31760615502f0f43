#!/usr/bin/env python3
"""Compact view of a rocprofv3 kernel_stats csv: our kernels only, calls, average, total.  usage: show_stats.py file.csv [calls-per-sequence divisor]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else None
tot = 0.0
for r in rows:
    n = re.sub(r'\(.*', '', r['Name'].replace('(anonymous namespace)::', '')).replace('void ', '')
    if 'at::' in n or 'rocclr' in n or 'rccl' in n.lower():
        continue
    t = float(r['TotalDurationNs']) / 1e6
    tot += t
    extra = "  per-seq %8.1f us" % (float(r['TotalDurationNs']) / 1e3 / div) if div else ""
    print("%-58s calls %6s avg %9.1f us  tot %9.2f ms%s" % (n[:58], r['Calls'], float(r['AverageNs']) / 1e3, t, extra))
print("sum of listed kernels: %.2f ms%s" % (tot, "  per-seq %.1f us" % (tot * 1e3 / div) if div else ""))
