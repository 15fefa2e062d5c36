#!/bin/bash
# the CLI end to end on the 40 000 x 40 000 synthetic slide (configs[2] geometry, 1 GPU): plain (random weights, few cells)
# and with the synth plug-in (2.5 M cells: exercises records, device polygons, exact (scipy-order) de-duplication, GeoJSON)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)     # the repo root, from the script's own location (works outside the harness)
O=$R/gpurun_out
export CLASSPOSE_SYNTHETIC_WEIGHTS=1 CLASSPOSE_MODEL_DIR=/tmp/nomodels
for MODE in plain plugin; do
  if [ $MODE = plugin ]; then export CLASSPOSE_AMD_PLUGINS=classpose_amd.synth:flow; fi
  rm -rf /tmp/out40_$MODE; mkdir -p /tmp/out40_$MODE
  T0=$(date +%s)
  python -m classpose_amd.entrypoints.predict_wsi --model_config conic --slide_path "synthetic://40000x40000?mpp=0.5&seed=1234" \
      --output_folder /tmp/out40_$MODE --tile_size 256 --overlap 32 --device cuda:0 > $O/r04_cli_40k_$MODE.log 2>&1
  echo "mode=$MODE wall seconds: $(( $(date +%s) - T0 ))" >> $O/r04_cli_40k_$MODE.log
  ls -la /tmp/out40_$MODE >> $O/r04_cli_40k_$MODE.log
done
