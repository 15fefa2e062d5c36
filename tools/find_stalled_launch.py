"""Given a rocprofv3 output directory (--kernel-trace [--memory-copy-trace], csv), list every launch of a layer kernel that took more than 3 x the
median of its kernel name, and everything (kernels of other queues, memory copies) that overlapped it in time."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
mc = sorted(glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True))
rows = []
for f in kt:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(("K", r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
for f in mc:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append(("C", r.get("Direction", r.get("Name", "copy")), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "-", r.get("Stream_Id", "?")))
print(f"{len(rows)} activities from {len(kt)} kernel-trace and {len(mc)} copy-trace files")
by = defaultdict(list)
for r in rows:
    if r[0] == "K":
        by[r[1]].append(r[3] - r[2])
med = {k: sorted(v)[len(v) // 2] for k, v in by.items()}
t0 = min(r[2] for r in rows)
layer = [k for k in med if any(t in k for t in ("k_gemm256p", "k_gemm4w", "k_attention"))]
n = 0
for r in sorted(rows, key=lambda r: r[2]):
    if r[0] != "K" or r[1] not in layer or (r[3] - r[2]) <= 3 * med[r[1]]:
        continue
    n += 1
    print(f"\nSTALLED: {r[1][:60]}  queue {r[4]} stream {r[5]}  start +{(r[2] - t0) / 1e6:.3f} ms  duration {(r[3] - r[2]) / 1e3:.1f} us (median {med[r[1]] / 1e3:.1f})")
    for o in sorted(rows, key=lambda o: o[2]):
        if o is r or o[3] < r[2] - 200000 or o[2] > r[3] + 200000:
            continue
        if o[0] == "K" and o[4] == r[4] and o[5] == r[5]:
            continue                                    # same queue: its neighbours in the stream
        print(f"    {'kernel' if o[0] == 'K' else 'copy  '} {o[1][:70]:70s} queue {o[4]} stream {o[5]}  start {(o[2] - r[2]) / 1e3:+9.1f} us  end {(o[3] - r[2]) / 1e3:+9.1f} us  ({(o[3] - o[2]) / 1e3:.1f} us)")
print(f"\n{n} stalled launches")
# gaps INSIDE a forward: between two consecutive layer kernels of the same queue (a HIP-event pair around the second one would count the gap as its duration)
ks = sorted((r for r in rows if r[0] == "K"), key=lambda r: r[2])
last = {}
g = 0
gaps = []
for r in ks:
    key = (r[4], r[5])
    p = last.get(key)
    if p is not None and r[1] in layer and p[1] in layer:
        gaps.append(r[2] - p[3])
        if r[2] - p[3] > 100000:
            g += 1
            print(f"GAP of {(r[2] - p[3]) / 1e3:.1f} us on queue {r[4]} at +{(p[3] - t0) / 1e6:.3f} ms between {p[1][:40]} and {r[1][:40]}")
            for o in sorted(rows, key=lambda o: o[2]):
                if o[3] < p[3] - 100000 or o[2] > r[2] + 100000 or (o[0] == "K" and (o[4], o[5]) == key):
                    continue
                print(f"    {'kernel' if o[0] == 'K' else 'copy  '} {o[1][:70]:70s} queue {o[4]} stream {o[5]}  start {(o[2] - p[3]) / 1e3:+9.1f} us  end {(o[3] - p[3]) / 1e3:+9.1f} us")
    last[key] = r
gaps.sort()
print(f"{g} gaps > 100 us between consecutive layer kernels; all {len(gaps)} such gaps: median {gaps[len(gaps) // 2] / 1e3:.2f} us, p99 {gaps[int(len(gaps) * 0.99)] / 1e3:.2f} us, max {gaps[-1] / 1e3:.1f} us")
