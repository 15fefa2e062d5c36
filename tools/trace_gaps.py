"""Idle time between consecutive kernels of the network stream, from a rocprofv3 --kernel-trace csv."""
import csv, glob, sys, collections
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
print(path, len(rows), list(rows[0].keys()))
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
by = collections.defaultdict(list)
for r in rows:
    by[r[key]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, ks in by.items():
    ks.sort()
    busy = sum(e - s for s, e, _ in ks)
    span = ks[-1][1] - ks[0][0]
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 100000]
    print(f"{key} {q}: {len(ks)} kernels, busy {busy/1e6:.2f} ms, span {span/1e6:.2f} ms, "
          f"gaps<100us: n={len(small)} sum={sum(small)/1e6:.3f} ms median={sorted(small)[len(small)//2] if small else 0} ns")
    names = collections.Counter(n[:40] for _, _, n in ks).most_common(4)
    print("   ", names)
# network stream: the one with k_gemm256; per-step gap budget over the last steps
net = max(by.values(), key=lambda ks: sum("k_gemm256" in n for _, _, n in ks))
g = collections.defaultdict(list)
for i in range(len(net) - 1):
    gap = net[i + 1][0] - net[i][1]
    if 0 <= gap < 100000:
        g[(net[i][2][:28], net[i + 1][2][:28])].append(gap)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k[0]:30s} -> {k[1]:30s} n={len(v):5d} mean={sum(v)/len(v)/1e3:7.2f} us total={sum(v)/1e6:7.3f} ms")
