#!/bin/bash
# kernel trace of a short bench run -> idle gaps between consecutive kernels of the network stream (tools/trace_gaps.py)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)     # the repo root, from the script's own location (works outside the harness)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02gaps
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o run -- python3 $R/bench.py --gpus 1 --steps 12 --warmup 3 --no-cpu-baseline --no-stages > $O/run.log 2>&1
python3 $R/tools/trace_gaps.py $O/t > $O/gaps.txt 2>&1
find $O -type f \( -name "*kernel_trace.csv" -o -name "*.db" -o -name "*agent_info.csv" \) -delete
