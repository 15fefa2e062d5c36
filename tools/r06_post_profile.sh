#!/bin/bash
# Post-processing chain alone under rocprofv3 --kernel-trace --stats (tools/run_post.py: 12 iterations of cpx_compute_masks on one 8-tile
# batch of the bench's fields); prints the per-kernel table.  Run on the GPU box from the repo root.  $1 = output tag (default r06post).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/${1:-r06post}
rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o run -- python3 $R/tools/run_post.py > $P/stats.log 2>&1
find $P/stats -name "*kernel_stats.csv" -exec cp {} $P/kernel_stats.csv \;
find $P -type f \( -name "*kernel_trace.csv" -o -name "*.db" -o -name "*agent_info.csv" \) -delete
cd $R
python3 tools/post_table.py $P/kernel_stats.csv
