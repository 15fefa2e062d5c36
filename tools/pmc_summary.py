"""Per-kernel average of a rocprofv3 --pmc counter (csv output dir given as argv[1], counter name argv[2])."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
name = sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != name:
        continue
    a = acc[r["Kernel_Name"][:70]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{k:72s} launches {n:5d}  avg {s / n:14.1f}  total {s:16.1f}")
