"""Dev tool: print a rocprofv3 kernel_stats.csv compactly (name truncated, calls, avg/min/max us, %)."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void ", "").replace("attwarp::", "")
    print(f"{n[:70]:70s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f} us  min={float(r['MinNs'])/1e3:9.1f}  max={float(r['MaxNs'])/1e3:9.1f}  {float(r['Percentage']):5.1f}%")
