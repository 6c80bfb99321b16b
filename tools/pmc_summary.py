"""Dev tool: average the counters of a rocprofv3 --pmc CSV per kernel.  usage: pmc_summary.py <counter_collection.csv> [substr]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"][:90]
    if sub in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
