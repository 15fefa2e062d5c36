#!/bin/bash
# BASELINE configs[3] geometry on ONE GPU (the config itself names 8): synthetic 80 000^2 slide, conic, --tta, random weights
set -u
R=$(cd "$(dirname "$0")/.." && pwd)     # the repo root, from the script's own location (works outside the harness)
export CLASSPOSE_SYNTHETIC_WEIGHTS=1 CLASSPOSE_MODEL_DIR=/tmp/cpx_models
mkdir -p $R/gpurun_out/r06; [ -f /tmp/cpx_models/conic.pt ] || python $R/tools/make_synthetic_checkpoint.py conic > /dev/null
rm -rf /tmp/out80; mkdir -p /tmp/out80
T0=$(date +%s)
python -m classpose_amd.entrypoints.predict_wsi --model_config conic --slide_path "synthetic://80000x80000?mpp=0.5&seed=1234" \
    --output_folder /tmp/out80 --tile_size 256 --overlap 32 --tta --device cuda:0 > $R/gpurun_out/r06/r06_cli_80k_tta.log 2>&1
echo "wall seconds: $(( $(date +%s) - T0 ))" >> $R/gpurun_out/r06/r06_cli_80k_tta.log
ls -la /tmp/out80 >> $R/gpurun_out/r06/r06_cli_80k_tta.log
grep -v "Predicted tiles" $R/gpurun_out/r06/r06_cli_80k_tta.log > /tmp/short.log; grep "Predicted tiles" $R/gpurun_out/r06/r06_cli_80k_tta.log | awk 'NR % 25 == 0' >> /tmp/short.log; cp /tmp/short.log $R/gpurun_out/r06/r06_cli_80k_tta.log
