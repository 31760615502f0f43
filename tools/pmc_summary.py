"""Aggregate a rocprofv3 --pmc run (…_counter_collection.csv) into per-kernel, per-launch averages.

usage: python tools/pmc_summary.py <counter_collection.csv> [substring filter]
"""
import csv, sys, collections, json
rows = csv.DictReader(open(sys.argv[1]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if flt not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
out = {k: {c: v / len(disp[k]) for c, v in cs.items()} | {"launches": len(disp[k])} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
