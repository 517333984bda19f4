"""Reduce a rocprofv3 --kernel-trace --stats directory to a short per-kernel table (calls, avg us, total ms, share)."""
import csv, glob, sys, re
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not f: print("no kernel_stats.csv under", d); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    name = re.sub(r"\(.*", "", r["Name"])[:70]
    print(f"{name:70s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.1f} ms {100*float(r['TotalDurationNs'])/tot:5.1f} %")
