"""Aggregate a rocprofv3 --pmc run (…_counter_collection.csv) into per-kernel (and per-grid), per-launch averages.

usage: python tools/pmc_summary.py <counter_collection.csv> [kernel-name substring] [--by-grid]
"""
import csv, sys, collections, json
args = [a for a in sys.argv[1:] if not a.startswith("--")]
by_grid = "--by-grid" in sys.argv
rows = csv.DictReader(open(args[0]))
flt = args[1] if len(args) > 1 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if flt not in k: continue
    if by_grid: k += " grid=%s" % r.get("Grid_Size", "?")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
out = {k: {c: v / len(disp[k]) for c, v in cs.items()} | {"launches": len(disp[k])} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
