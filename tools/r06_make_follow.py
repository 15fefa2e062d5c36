"""profiles/r06_pmc_follow.json from gpurun_out/r06follow/table.json (tools/r06_follow_pmc.sh: --pmc passes over the post-processing chain alone):
the counters of the SHIPPED Euler-loop kernel (k_follow<true, 1, true>: 32 x 32-cell segments, LDS window, grouped orbit test) with a note computed
from the numbers -- round 5's file profiled the kernel that round REPLACED and carried an inference its own DESIGN retracted."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06follow", "table.json")
t = json.load(open(src))
keep = {k: v for k, v in t.items() if any(n in k for n in ("k_follow", "k_diffuse", "k_flow_err_label", "k_prep_flow"))}
fk = next(k for k in keep if "k_follow" in k)
f = keep[fk]
gui = f["GRBM_GUI_ACTIVE"] / 8
note = (f"rocprofv3 --pmc passes over tools/run_post.py (the post-processing chain of one 8-tile batch, 12 iterations; tools/r06_follow_pmc.sh), per-launch averages.  "
        f"{fk.split('(')[0]} is the kernel that ships (round 5's file profiled k_follow<true>, the form that round replaced).  GRBM_GUI_ACTIVE is summed over the 8 XCDs: "
        f"{gui / 1e3:.0f} k cycles per launch.  Texture addresser busy {f['TA_BUSY_avr'] / 1e3:.1f} k cycles = {100 * f['TA_BUSY_avr'] / gui:.0f} % of that; "
        f"L1 -> L2 read requests {f['TCP_TCC_READ_REQ_sum'] / 1e3:.0f} k per launch; vector-memory read instructions {f['SQ_INSTS_VMEM_RD'] / 1e3:.0f} k "
        f"(the taps of a step come from the segment's LDS window unless a lane has left it); SQ_WAIT_ANY / SQ_WAVE_CYCLES = {f['SQ_WAIT_ANY'] / f['SQ_WAVE_CYCLES']:.2f}, "
        f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {f['SQ_WAIT_INST_ANY'] / f['SQ_WAVE_CYCLES']:.2f}, {f['SQ_INSTS_VALU'] / 1e6:.1f} M vector instructions per launch.  "
        "What the counters do NOT show is a memory-bound kernel: round 5's reading of the old kernel's waits as 'each wave's own chain takes ~1 550 cycles per step' was wrong -- its "
        "waves sat eight to a SIMD on two XCDs waiting for an issue slot (DESIGN section 0, round-5 ledger item 6, erratum); the shipped kernel spreads them over the chip.")
json.dump({"note": note, "kernels": keep}, open(os.path.join(ROOT, "profiles", "r06_pmc_follow.json"), "w"), indent=1)
print(note)
