#!/bin/bash
# BASELINE configs[4] geometry on ONE GPU (the config names 8): 512-px tiles, fp16, semantic head, synthetic 40 000^2 slide
set -u
R=$(cd "$(dirname "$0")/.." && pwd)     # the repo root, from the script's own location (works outside the harness)
export CLASSPOSE_SYNTHETIC_WEIGHTS=1 CLASSPOSE_MODEL_DIR=/tmp/cpx_models
mkdir -p $R/gpurun_out/r06; [ -f /tmp/cpx_models/conic.pt ] || python $R/tools/make_synthetic_checkpoint.py conic > /dev/null
rm -rf /tmp/out5; mkdir -p /tmp/out5
T0=$(date +%s)
python -m classpose_amd.entrypoints.predict_wsi --model_config conic --slide_path "synthetic://40000x40000?mpp=0.5&seed=1234" \
    --output_folder /tmp/out5 --tile_size 512 --overlap 64 --precision fp16 --device cuda:0 > $R/gpurun_out/r06/r06_cli_config5.log 2>&1
echo "wall seconds: $(( $(date +%s) - T0 ))" >> $R/gpurun_out/r06/r06_cli_config5.log
grep -v "Predicted tiles" $R/gpurun_out/r06/r06_cli_config5.log > /tmp/short5.log; grep "Predicted tiles" $R/gpurun_out/r06/r06_cli_config5.log | awk 'NR % 10 == 0' >> /tmp/short5.log; cp /tmp/short5.log $R/gpurun_out/r06/r06_cli_config5.log
