#!/bin/bash
# the CLI end to end on the 40 000 x 40 000 synthetic slide (north-star geometry, 1 GPU): plain (random weights, few cells)
# and with the synth plug-in (2.5 M cells: exercises records, device polygons, exact (scipy-order) de-duplication, GeoJSON).
# Round 5 on: real checkpoint file (written once, untimed) instead of generating the synthetic weights inside the timed run.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r06; mkdir -p $O
export CLASSPOSE_MODEL_DIR=/tmp/cpx_models CLASSPOSE_SYNTHETIC_WEIGHTS=1 LOG_LEVEL_NON_MAIN=INFO   # (every rank logs its stage table)
[ -f /tmp/cpx_models/conic.pt ] || python $R/tools/make_synthetic_checkpoint.py conic > /dev/null
for MODE in ${MODES:-plain plugin}; do
  if [ $MODE = plugin ]; then export CLASSPOSE_AMD_PLUGINS=classpose_amd.synth:flow; fi
  rm -rf /tmp/out40_$MODE; mkdir -p /tmp/out40_$MODE
  T0=$(date +%s.%N)
  python -m classpose_amd.entrypoints.predict_wsi --model_config conic --slide_path "synthetic://${SLIDE:-40000x40000}?mpp=0.5&seed=1234" \
      --output_folder /tmp/out40_$MODE --tile_size 256 --overlap 32 --device ${DEVICES:-cuda:0} > $O/cli_40k_${MODE}${TAG:-}.log 2>&1
  echo "mode=$MODE wall seconds: $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $T0)" >> $O/cli_40k_${MODE}${TAG:-}.log
  ls -la /tmp/out40_$MODE >> $O/cli_40k_${MODE}${TAG:-}.log
done
