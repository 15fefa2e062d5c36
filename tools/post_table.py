"""Per-iteration kernel table of tools/r02_post_profile.sh's kernel_stats.csv (run_post.py runs 12 iterations)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02post/kernel_stats.csv")))
tot = 0.0
for r in rows:
    per = float(r["TotalDurationNs"]) / 12 / 1e3
    tot += per
    if per >= 2.0:
        print(f"{r['Name'][:44]:46s} calls/iter {int(r['Calls']) / 12:5.1f} avg {float(r['AverageNs']) / 1e3:7.1f} us  per-iter {per:7.1f} us")
print(f"sum of kernel time per 8-tile batch: {tot:.1f} us")
