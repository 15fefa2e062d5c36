"""Print the headline and the stage table of a bench.py JSON line (file argument or stdin)."""
import json, sys
src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
d = json.loads([l for l in src if l.startswith('{"metric"')][-1])
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  n_gpus {d['n_gpus']}  steps {d['steps']}  build {d.get('build_id')}")
r = d["roofline"]
print(f"roofline: {r['kernel'][:60]}  achieved {r['achieved']:.0f} {r['unit']}  frac {r['frac']:.3f}  avg launch {r['avg_launch_ms'] * 1e3:.1f} us  kernel-time sum {r.get('kernel_time_sum_ms_per_step')} ms/step")
for k, v in (r.get("stages") or {}).items():
    if "launch_ms" in v:
        m = v["launch_ms"]
        print(f"  {k:16s} min {m['min'] * 1e3:7.1f}  median {m['median'] * 1e3:7.1f}  max {m['max'] * 1e3:7.1f} us   frac {v['frac']:.3f}{'  OUTLIER' if v.get('outlier') else ''}")
    else:
        print(f"  {k:16s} {v.get('ms_per_batch')} ms per batch, {v.get('launches_per_batch')} launches, frac {v.get('frac')}")
