#!/bin/bash
# Round-6 profile collection (run on the GPU box from the repo root): kernel-trace stats of the bench command and
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA utilisation).  The program itself follows `--` (no wrapper).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)     # the repo root, from the script's own location (works outside the harness)
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r06prof
mkdir -p $O
B="python3 $R/bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-stages --no-side-lines --no-live-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-side-lines --no-live-traffic > $O/stats.log 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -o run -- $B > $O/pmc_$C.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_mfma1 -o run -- $B > $O/pmc_mfma1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_mfma2 -o run -- $B > $O/pmc_mfma2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_mfma3 -o run -- $B > $O/pmc_mfma3.log 2>&1
cd $R
python3 tools/pmc_table.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma1 $O/pmc_mfma2 $O/pmc_mfma3 > $O/pmc_table.json 2> $O/pmc_table.err
ls $O/stats/*/* 2>/dev/null | head; find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep -i "mfma" $O/counters.txt | head -40 > $O/mfma_counters.txt
# keep the merge small: drop the raw traces
find $O -type f \( -name "*kernel_trace.csv" -o -name "*.db" -o -name "*counter_collection.csv" -o -name "*agent_info.csv" \) -delete
find $O -type f -size +1M -delete
du -sh $O; du -sh $R/gpurun_out
# post-processing chain alone (tools/run_post.py: 12 iterations of compute_masks on one 8-tile batch) under the same tracer
cd /tmp
P=$R/gpurun_out/r06post
rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o run -- python3 $R/tools/run_post.py > $P/stats.log 2>&1
find $P/stats -name "*kernel_stats.csv" -exec cp {} $P/kernel_stats.csv \;
find $P -type f \( -name "*kernel_trace.csv" -o -name "*.db" -o -name "*agent_info.csv" \) -delete
cd $R
python3 tools/r06_make_profiles.py > $O/make_profiles.log 2>&1; cat $O/make_profiles.log
