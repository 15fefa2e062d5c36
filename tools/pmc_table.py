"""Per-kernel averages of every counter in one or more rocprofv3 --pmc output dirs -> JSON on stdout.
usage: pmc_table.py <dir> [<dir> ...]   (each dir = one pass; counters of all passes are merged per kernel)"""
import collections, csv, glob, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
out = {}
for k, cs in acc.items():
    tot = max(v[0] for v in cs.values())
    out[k] = {"launches": max(v[1] for v in cs.values()), **{c: v[0] / v[1] for c, v in cs.items()}}
keep = sorted(out.items(), key=lambda kv: -kv[1]["launches"] * max(v for c, v in kv[1].items() if c != "launches"))[:16]
print(json.dumps(dict(keep), indent=1))
