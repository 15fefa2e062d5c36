"""Per-instruction issue budget of the SHIPPED attention loop (round-5 review item 2): disassembles libclasspose_hip.so, takes the production
instantiation of k_attention4p (bf16), finds its steady-state loop (the back edge with the largest body: four key tiles per iteration), and
counts what one wave issues per 32-key tile by pipe.  Issue costs are the measured ones of MI355X_MICROARCH.md / profiles/r03_coexec_valu_beside_mfma.txt
(cycles of the SIMD's issue port a 64-lane instruction occupies).  Also prints the register / LDS footprint from the code object's metadata and
the occupancy it allows, i.e. why no 4-waves-per-SIMD form exists.

    python tools/attn_issue_budget.py [library.so]        (CPU only)"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lint_isa

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "classpose_amd", "libclasspose_hip.so")
kern = [(s, b) for s, b in lint_isa.disassemble(lib) if re.match(r"^_Z\d+k_attention4pILb0ELb0ELb0ELb0ELb0ELb0ELb0E", s)]
assert len(kern) == 1, [s for s, _ in kern]
sym, body = kern[0]
# basic blocks: split at LABEL; the loop = the block sequence between a label and the s_cbranch that jumps back to it.  With --symbolize-operands
# branch targets read "<L12>"; lint_isa keeps only "LABEL" markers, so redo a light disassembly pass here to keep the names.
import shutil, tempfile
with tempfile.TemporaryDirectory() as tmp:
    local = os.path.join(tmp, os.path.basename(lib)); shutil.copy(lib, local)
    subprocess.run([lint_isa.OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
    txt = ""
    meta = ""
    for name in sorted(os.listdir(tmp)):
        if "amdgcn" in name:
            t = subprocess.run([lint_isa.OBJDUMP, "-d", "--no-show-raw-insn", "--symbolize-operands", os.path.join(tmp, name)], check=True, capture_output=True, text=True).stdout
            if f"<{sym}>:" in t:
                txt = t
                meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", os.path.join(tmp, name)], capture_output=True, text=True).stdout
lines, on = [], False
for line in txt.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
    if m and not m.group(1).startswith("L"):
        on = m.group(1) == sym
        continue
    if on:
        m2 = re.match(r"^<(L\d+)>:", line) or (m and re.match(r"(L\d+)", m.group(1)))
        if m2:
            lines.append(("L", m2.group(1)))
        elif line.startswith("\t"):
            lines.append(("I", line.split("//")[0].strip()))
labels = {v: i for i, (k, v) in enumerate(lines) if k == "L"}
best = None
for i, (k, v) in enumerate(lines):
    if k == "I" and v.startswith("s_cbranch"):
        t = re.search(r"<?(L\d+)>?", v.split()[-1])
        if t and t.group(1) in labels and labels[t.group(1)] < i:
            n = sum(1 for kk, _ in lines[labels[t.group(1)]:i] if kk == "I")
            if best is None or n > best[0]:
                best = (n, labels[t.group(1)], i)
assert best, "no loop found"
# the rescale path (exact maximum, alpha, second exp pass) sits INSIDE the loop behind a forward `s_cbranch_vccz` that skips it when no lane voted
# (every tile but the first and genuine outliers): instructions between such a branch and its target are counted apart, as the cold path
loop, cold, skip_to = [], [], None
for k, v in lines[best[1]:best[2] + 1]:
    if k == "L":
        if skip_to == v:
            skip_to = None
        continue
    if skip_to is not None:
        cold.append(v)
        continue
    loop.append(v)
    if v.startswith(("s_cbranch_vccz", "s_cbranch_execz", "s_cbranch_scc")):
        t = re.search(r"(L\d+)", v.split()[-1])
        if t and t.group(1) in labels and best[1] < labels[t.group(1)] <= best[2] + 1 and labels[t.group(1)] > lines.index(("I", v), best[1]):
            skip_to = t.group(1)
TILES = 4
CLASSES = [("matrix: v_mfma_f32_32x32x16", lambda s: s.startswith("v_mfma"), 32.0),
           ("vector transcendental: v_exp_f32", lambda s: s.startswith(("v_exp", "v_rcp", "v_rsq", "v_log")), 8.0),
           ("vector conversion: v_cvt_pk_*", lambda s: s.startswith("v_cvt"), 4.0),
           ("vector, other (fma / add / or / max / mov / cndmask / lane ops)", lambda s: s.startswith("v_"), 4.0),
           ("LDS reads: ds_read_b128 / ds_read_u16", lambda s: s.startswith(("ds_read", "ds_load")), 0.0),
           ("LDS-DMA requests: global_load_lds_dwordx4", lambda s: s.startswith(("global_load", "buffer_load")), 0.0),
           ("waits: s_waitcnt", lambda s: s.startswith("s_waitcnt"), 0.0),
           ("barrier: s_barrier", lambda s: s.startswith("s_barrier"), 0.0),
           ("scalar / branch / nop", lambda s: True, 0.0)]
count = [0] * len(CLASSES)
detail = {}
for ins in loop:
    for ci, (_, pred, _) in enumerate(CLASSES):
        if pred(ins):
            count[ci] += 1
            detail.setdefault(ci, {}).setdefault(ins.split()[0], 0)
            detail[ci][ins.split()[0]] += 1
            break
print(f"{sym}\nsteady-state loop: {len(loop)} instructions per iteration on the hot path = {TILES} key tiles of 32 keys (one wave: 32 queries); {len(cold)} more on the rescale path "
      f"(behind a forward branch, taken for the first tile and for outliers only: {sum(1 for c in cold if c.startswith('v_exp')) / TILES:g} v_exp, {sum(1 for c in cold if c.startswith('v_')) / TILES:g} vector instructions per tile there)\n")
print(f"{'per 32-key tile and wave':70s} {'count':>6s} {'issue cycles each':>18s} {'cycles':>8s}")
tot_v = tot_m = 0.0
for ci, (name, _, cost) in enumerate(CLASSES):
    n = count[ci] / TILES
    cyc = n * cost
    if ci == 0:
        tot_m += cyc
    elif 1 <= ci <= 3:
        tot_v += cyc
    print(f"{name:70s} {n:6.1f} {cost:18.1f} {cyc:8.0f}    " + ", ".join(f"{k} x{v / TILES:g}" for k, v in sorted(detail.get(ci, {}).items(), key=lambda kv: -kv[1])[:8]))
print(f"\nmatrix pipe per wave-tile: {tot_m:.0f} cycles; vector issue per wave-tile: {tot_v:.0f} cycles (a lone wave: 6.5 per plain instruction, 9.75 per v_exp; two or more waves of a SIMD share one "
      f"port at ~3.3 / ~8 -- profiles/r03_coexec_valu_beside_mfma.txt)")
print("measured (rocprofv3, 32 sub-tiles): 178 - 186 us per launch = 524 288 wave-tiles on 1 024 SIMDs -> ~700 - 760 SIMD cycles per wave-tile at the ~2.05 GHz the part holds:")
print(f"   matrix pipe {tot_m:.0f} / ~730 = {tot_m / 730:.2f} busy, vector port {tot_v:.0f} / ~730 = {tot_v / 730:.2f} busy -- neither pipe is the limiter; each wave's own in-order timeline is")
print("   (LDS latency -> 4 dependent QK^T MFMAs -> 16 exp behind 16 fma -> pack -> vote -> LDS latency -> 4 P.V MFMAs -> barrier), profiles/r03_attn4p_stamps.txt: 1 655 - 2 094 cycles per")
print("   tile and wave with three waves on the SIMD, 1 118 - 1 411 for a wave nearly alone; 3 waves x 256 matrix cycles / 2 094 = 0.37 = the MfmaUtil the counters report.")
for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size", ".vgpr_spill_count"):
    pass
m = re.search(re.escape(sym) + r".*?(?=\.name:|\Z)", meta, re.S)
blk = None
for part in meta.split(".name:"):
    if sym in part.split("\n")[0]:
        blk = part
if blk:
    g = lambda k: int(re.search(re.escape(k) + r":\s*(\d+)", blk).group(1)) if re.search(re.escape(k) + r":\s*(\d+)", blk) else None
    v, a, l = g(".vgpr_count"), g(".agpr_count"), g(".group_segment_fixed_size")
    print(f"\nfootprint: {v} VGPRs (+ {a} AGPRs) per lane, static LDS {l} B + 49 664 B dynamic (4-slot K / V^T ring 32 KB + G scratch 16.5 KB) per 4-wave workgroup")
    print(f"   waves per SIMD by registers: floor(512 / {v}) = {512 // max(v, 1)} (the file is allocated in blocks of 8: 3 waves need <= 168, 4 waves <= 128)")
    print("   workgroups per CU by LDS: floor(160 KB / 48.5 KB) = 3  -> 3 waves per SIMD either way")
    print("   what 4 waves per SIMD would need: <= 128 VGPRs.  Live across the loop: S and the next S (2 x 16), O (2 x 16), Q fragments (16), Gw (16), probabilities before packing (16),")
    print("   K / V fragments of the tile (2 x 16), packed P (8), addresses / running max / sum (~12) = 164 - 16 (p reuses S's registers once exp has run) = ~150: the second score tile (the")
    print("   software pipeline that takes the QK^T chain out of the wave's critical path, -6 % when it was introduced) and the register-resident Gw are what 128 cannot hold; without them the")
    print("   kernel is k_attention (round 1: 0.26 of peak at 4 waves per SIMD).")
