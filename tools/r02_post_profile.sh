#!/bin/bash
# post-processing alone under rocprofv3 --kernel-trace --stats (run on the GPU box from the repo root)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)     # the repo root, from the script's own location (works outside the harness)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r02post
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/tools/run_post.py > $O/stats.log 2>&1
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -type f \( -name "*kernel_trace.csv" -o -name "*.db" -o -name "*agent_info.csv" \) -delete
