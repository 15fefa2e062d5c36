"""Write the seeded synthetic weights of a model config as a REAL checkpoint file, so that CLI runs load it the way a user's run loads a
downloaded one (torch.load, memory-mapped) instead of generating 300 M random parameters at every start (5 s of `synth.make_state_dict`).

    python tools/make_synthetic_checkpoint.py conic [depth]      -> $CLASSPOSE_MODEL_DIR/conic.pt   (default ~/.classpose_models)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from classpose_amd import model_configs, synth
name = sys.argv[1]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = model_configs.ModelConfig(**model_configs.DEFAULT_MODEL_CONFIGS[name])
os.makedirs(os.path.dirname(cfg.path), exist_ok=True)
sd = synth.make_state_dict(len(cfg.cell_types) + 1, None, depth=depth, seed=0)
torch.save(sd, cfg.path)
print(f"wrote {cfg.path}: {sum(v.numel() for v in sd.values()) / 1e6:.1f} M parameters, depth {depth}")
