#!/bin/bash
# BASELINE configs[2] through the CLI on one GPU: synthetic 40 000^2 slide at 0.22 um/px, puma, default 1024 / 64 tiles,
# GrandQC tissue + artefact detection (class maps from the synth plug-in: the GrandQC weights are random), artefact filter, csv
set -u
R=$(cd "$(dirname "$0")/.." && pwd)     # the repo root, from the script's own location (works outside the harness)
export CLASSPOSE_SYNTHETIC_WEIGHTS=1 CLASSPOSE_MODEL_DIR=/tmp/nomodels CLASSPOSE_AMD_PLUGINS=classpose_amd.synth:flow+qc
rm -rf /tmp/out3; mkdir -p /tmp/out3
T0=$(date +%s)
python -m classpose_amd.entrypoints.predict_wsi --model_config puma --slide_path "synthetic://40000x40000?mpp=0.22&seed=1234" \
    --output_folder /tmp/out3 --device cuda:0 --tissue_detection_model_path /tmp/td.pth --artefact_detection_model_path /tmp/art.pth \
    --filter_artefacts --output_type csv > $R/gpurun_out/r04_config3_cli.log 2>&1
echo "wall seconds: $(( $(date +%s) - T0 )); cpus: $(nproc)" >> $R/gpurun_out/r04_config3_cli.log
ls -la /tmp/out3 >> $R/gpurun_out/r04_config3_cli.log
