#!/bin/bash
# What bounds the Euler loop (k_follow)?  PMC passes over the post-processing chain alone (tools/run_post.py): texture-addresser busy, L1 tag
# requests / reads to the L2, vector-memory instruction cycles, wave cycles and issue stalls.  Kernel-trace + counters only (no other trace domain).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r06follow; rm -rf $O; mkdir -p $O
# (one or two counters of a block per pass: a request the hardware cannot serve makes rocprofv3 abort and HANG -- every pass under its own timeout)
P() { local n=$1; shift; timeout 150 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -o run -- python3 $R/tools/run_post.py > $O/$n.log 2>&1 || echo "pass $n failed / timed out"; }
P p1 TA_BUSY_avr GRBM_GUI_ACTIVE
P p2 TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
P p3 TCP_TAGRAM0_REQ_sum TCP_TCC_READ_REQ_sum
P p4 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
P p5 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY
cd $R
python3 tools/pmc_table.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 > $O/table.json 2> $O/table.err
python3 - <<PY
import json
t = json.load(open("$O/table.json"))
for k, v in t.items():
    if "k_follow" in k or "k_diffuse" in k:
        print(k[:60]); print("   ", {c: round(x, 1) for c, x in v.items()})
PY
find $O -type f \( -name "*counter_collection.csv" -o -name "*.db" -o -name "*agent_info.csv" \) -delete
