"""ISA lint of the shipped code objects.  Rule 1: no instruction may read the destination of an LDS load before an `s_waitcnt lgkmcnt(..)`
that covers it (rule 4, vmem_findings: the same for inline-asm vector-memory loads and vmcnt in the one-wave-per-SIMD GEMM).  Rule 2 (permlane_findings): two wait states between a VALU write and a v_permlane16/32_swap that reads it.  Both are things
the compiler guarantees for code it can see and cannot guarantee around inline asm.  Rule 3: no register spill in the persistent GEMM and the
production attention kernel.

Why: the hot kernels read their LDS fragments through inline asm (`ds_read_b128` inside `asm volatile`; an ordinary LDS load would make
hipcc drain every LDS-DMA in flight with vmcnt(0)) and wait for them with an inline-asm `s_waitcnt`.  For the compiler the asm's output is
ready the moment the asm statement has executed, so nothing but the statement order -- `asm volatile` against `asm volatile`, a
`sched_barrier` behind the wait, or the loaded register carried through the wait as a "+v" operand -- keeps a consumer behind the wait.
Round 4 found a wait with none of the three (the last key tile of the attention kernel): harmless while that tile shared its code with the
other 31, wrong by 6 % on average the moment the loop tail was peeled and the compiler hoisted `v_cvt_f32_f16 gh` above the wait.  The rule
checked here holds for compiler-generated LDS loads as well, so every finding is a real bug in the build it was found in.

Model: per kernel, in program order, a queue of the LDS operations in flight (lgkmcnt counts them in order; scalar memory loads share the
counter and are queued as well); `s_waitcnt lgkmcnt(N)` retires all but the N youngest; a label or a branch clears the queue (findings
across basic blocks are not looked for).  Any VGPR / AGPR source operand that overlaps the destination of a queued load is a finding.

    python tools/lint_isa.py [library.so ...]        (default: classpose_amd/libclasspose_hip.so and libclasspose_hip_debug.so)
Exit status 1 and one line per finding when there are any.  tests/test_host_logic.py runs it on both libraries (CPU only, ~5 s)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
RE_REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+)\b)")
RE_LGKM = re.compile(r"lgkmcnt\((\d+)\)")
LDS_LOAD = ("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append", "ds_ordered")
DEST_FIRST = ("v_", "ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle", "global_load", "buffer_load", "scratch_load", "flat_load",
              "image_load", "image_sample")
DEST_ALSO_SRC = ("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c", "v_dot4c", "v_dot8c", "v_fmaak", "v_movrel")


def disassemble(so_path: str) -> list[tuple[str, list[str]]]:
    """[(kernel symbol, [instruction or 'LABEL' ...])] of every gfx950 code object bundled in the library"""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(so_path))
        shutil.copy(so_path, local)                      # --offloading extracts next to its input
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--symbolize-operands", os.path.join(tmp, name)], check=True,
                                 capture_output=True, text=True).stdout
            cur, body = None, []
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m and not m.group(1).startswith("L"):
                    if cur is not None:
                        out.append((cur, body))
                    cur, body = m.group(1), []
                elif cur is not None and (re.match(r"^<L\d+>:", line) or (m and m.group(1).startswith("L"))):
                    body.append("LABEL")
                elif cur is not None and line.startswith("\t"):
                    body.append(line.split("//")[0].strip())
            if cur is not None:
                out.append((cur, body))
    return out


def _regs(text: str) -> list[tuple[str, int, int]]:
    return [(m.group(1), int(m.group(2)), int(m.group(3))) if m.group(2) is not None else (m.group(1), int(m.group(4)), int(m.group(4)))
            for m in RE_REG.finditer(text)]


# timing-only ablation instantiations whose loads go to dead registers by design (results are garbage, the debug build says so)
ALLOW = re.compile(r"k_attention2qILb[01]ELb1ELi[1-9]")
# debug-build experiments measured once and not shipped (the no-vote attention variant of round 5 spills two registers): rule 3 is about what ships
EXPERIMENT = re.compile(r"k_attention4pILb0ELb0ELb0ELb0ELb1E")
NO_SPILL = re.compile(r"^_Z\d+(k_gemm256pI|k_attention4pI|k_gemm4wI)")


def permlane_findings(sym: str, body: list[str], lib: str) -> list[str]:
    """Rule 2 (gfx950 hazard "VALU write vdst -> v_permlane*_swap read"): two wait states between a VALU write of either swap operand and the
    swap.  hipcc pads the builtin with `s_nop 1` itself -- unless the write hides in inline asm (round 4: the GEMM's `pack2` conversion was
    inline asm; one launch in ~1 500 of the light bf16 direct-store epilogue stored a stale half-row)."""
    res = []
    for i, ins in enumerate(body):
        if not ins.startswith(("v_permlane16_swap", "v_permlane32_swap")):
            continue
        ops = _regs(ins.partition(" ")[2])
        states, j = 0, i - 1
        while j >= 0 and states < 2:
            prev = body[j]
            if prev == "LABEL" or prev.startswith(("s_branch", "s_cbranch")):
                break
            mnem, _, pops = prev.partition(" ")
            if mnem == "s_nop":
                states += int(pops.strip() or 0) + 1
            else:
                if mnem.startswith("v_") and not mnem.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    pr = _regs(pops)
                    if pr and any(pr[0][0] == b and not (pr[0][2] < lo or pr[0][1] > hi) for b, lo, hi in ops):
                        res.append(f"{lib}: {sym[:70]}: `{prev}` writes an operand of `{ins}` {states} wait state(s) before it (2 required)")
                states += 1
            j -= 1
    return res


RE_VM = re.compile(r"vmcnt\((\d+)\)")
VMEM_ASM_LOADS = re.compile(r"^_Z\d+k_gemm4wI")          # kernels that load registers through inline asm and wait by hand


def vmem_findings(sym: str, body: list[str], lib: str) -> list[str]:
    """Rule 4 (kernels of VMEM_ASM_LOADS): no instruction reads -- or overwrites -- the destination of a vector-memory load to REGISTERS before
    an `s_waitcnt vmcnt(N)` that covers it.  The one-wave-per-SIMD GEMM requests its residual rows with inline-asm buffer loads (an ordinary
    load makes hipcc drain the LDS-DMA queue) and waits with a counted vmcnt that carries the registers as "+v" operands; what the operands
    cannot forbid is a register COPY the allocator inserts between the load and the wait (seen in round 4's hoist experiment).  Model: every
    vector-memory instruction (loads, stores, LDS-DMA) enters an in-order queue; vmcnt(N) retires all but the N youngest."""
    res, queue = [], []
    for ins in body:
        if ins == "LABEL" or ins.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc")):
            queue = []
            continue
        mnem, _, ops = ins.partition(" ")
        if mnem == "s_waitcnt":
            m = RE_VM.search(ins)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n else []
            continue
        regs = _regs(ops)
        for bank, lo, hi in regs:                         # any touch of a pending destination, as source or as destination
            for dest, text in queue:
                if dest is not None and dest[0] == bank and not (hi < dest[1] or lo > dest[2]) and text != ins:
                    res.append(f"{lib}: {sym[:70]}: `{ins}` touches the destination of `{text}` before a covering s_waitcnt vmcnt")
        if mnem.startswith(("buffer_", "global_", "flat_", "scratch_")):
            is_load_to_regs = "_load" in mnem and " lds" not in ins and ops.lstrip().startswith("v")
            queue.append((regs[0] if is_load_to_regs and regs else None, ins))
    return res


def findings(so_path: str) -> list[str]:
    res = []
    for sym, body in disassemble(so_path):
        if ALLOW.search(sym):
            continue
        res += permlane_findings(sym, body, os.path.basename(so_path))
        if VMEM_ASM_LOADS.search(sym):
            res += vmem_findings(sym, body, os.path.basename(so_path))
        # Rule 3: the persistent GEMM and the production attention kernel pace their LDS-DMA with counted vmcnt waits and live at the edge of
        # the register file; a spill there is legal (extra vector-memory operations only make a counted wait stricter) but costs a drained
        # queue per reload, and the one combination that ever returned wrong numbers (round 4) was one that spilled -- keep them spill-free
        if NO_SPILL.search(sym) and not EXPERIMENT.search(sym):
            n = sum(1 for ins in body if ins.startswith("scratch_"))
            if n:
                res.append(f"{os.path.basename(so_path)}: {sym[:70]}: {n} scratch instruction(s) (register spill) in a kernel that must not spill")
        queue = []                                            # [(dest or None, text)] oldest first
        for ins in body:
            if ins == "LABEL" or ins.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc", "s_swappc", "s_barrier")):
                if ins.startswith("s_barrier") is False:
                    queue = []
                if ins != "LABEL" and not ins.startswith("s_barrier"):
                    continue
                if ins == "LABEL":
                    continue
            mnem, _, ops = ins.partition(" ")
            if mnem == "s_waitcnt":
                m = RE_LGKM.search(ins)
                if m:
                    n = int(m.group(1))
                    queue = queue[len(queue) - n:] if n else []
                continue
            regs = _regs(ops)
            dest_first = mnem.startswith(DEST_FIRST) and not mnem.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and \
                ops.lstrip().startswith(("v", "a"))
            srcs = regs[1:] if dest_first and regs else regs
            if dest_first and regs and mnem.startswith(DEST_ALSO_SRC):
                srcs = regs
            for bank, lo, hi in srcs:
                for dest, text in queue:
                    if dest is not None and dest[0] == bank and not (hi < dest[1] or lo > dest[2]):
                        res.append(f"{os.path.basename(so_path)}: {sym[:70]}: `{ins}` reads the destination of `{text}` with no covering s_waitcnt lgkmcnt in between")
            if mnem.startswith(LDS_LOAD) or (mnem.startswith("ds_") and mnem.endswith("_rtn")) or "_rtn_" in mnem and mnem.startswith("ds_"):
                queue.append((regs[0] if regs else None, ins))
            elif mnem.startswith("ds_") or mnem.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_sendmsg")):
                queue.append((None, ins))                     # counted by lgkmcnt, no vector destination
    return res


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = sys.argv[1:] or [os.path.join(root, "classpose_amd", n) for n in ("libclasspose_hip.so", "libclasspose_hip_debug.so")]
    bad = [f for p in libs for f in findings(p)]
    for f in bad[:40]:
        print(f)
    print(f"{len(bad)} finding(s) in {len(libs)} librar{'y' if len(libs) == 1 else 'ies'}")
    sys.exit(1 if bad else 0)
