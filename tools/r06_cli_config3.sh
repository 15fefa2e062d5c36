#!/bin/bash
# BASELINE configs[2] through the CLI on one GPU: synthetic 40 000^2 slide at 0.22 um/px, puma, default 1024 / 64 tiles,
# GrandQC tissue + artefact detection (class maps from the synth plug-in: the GrandQC weights are random), artefact filter, csv.
# Round 5: the synthetic ViT-L weights are written ONCE as a real checkpoint (untimed) and the timed run loads it like a user's run would.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)     # the repo root, from the script's own location (works outside the harness)
export CLASSPOSE_MODEL_DIR=/tmp/cpx_models CLASSPOSE_AMD_PLUGINS=classpose_amd.synth:flow+qc CLASSPOSE_SYNTHETIC_WEIGHTS=1
mkdir -p $R/gpurun_out/r06
[ -f /tmp/cpx_models/puma.pt ] || python $R/tools/make_synthetic_checkpoint.py puma > /dev/null
rm -rf /tmp/out3; mkdir -p /tmp/out3
T0=$(date +%s.%N)
python -m classpose_amd.entrypoints.predict_wsi --model_config puma --slide_path "synthetic://40000x40000?mpp=0.22&seed=1234" \
    --output_folder /tmp/out3 --device cuda:0 --tissue_detection_model_path /tmp/td.pth --artefact_detection_model_path /tmp/art.pth \
    --filter_artefacts --output_type csv > $R/gpurun_out/r06/config3_cli.log 2>&1
echo "wall seconds: $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1))" $T0); cpus: $(nproc)" >> $R/gpurun_out/r06/config3_cli.log
ls -la /tmp/out3 >> $R/gpurun_out/r06/config3_cli.log
