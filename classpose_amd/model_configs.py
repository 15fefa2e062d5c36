"""``--model_config`` surface: names, YAML schema and cell types of the reference
(/root/reference/src/classpose/model_configs.py:20-148).  Weights cannot be downloaded
here (no network; HF repo ``classpose/classpose``): a config whose ``path`` does not exist can
be pointed at seeded synthetic weights with ``CLASSPOSE_SYNTHETIC_WEIGHTS=1`` (bench / tests)."""
from __future__ import annotations

import os
from pathlib import Path

import yaml
from pydantic import BaseModel

from .log import get_logger

logger = get_logger("classpose.model_configs")

ROOT_MODEL_DIR = Path(os.getenv("CLASSPOSE_MODEL_DIR", Path.home() / ".classpose_models"))
REPO_ID = "classpose/classpose"


def _cfg(name, mpp, cell_types):
    return {"path": str(ROOT_MODEL_DIR / f"{name}.pt"), "mpp": mpp, "url": None,
            "hf": {"repo_id": REPO_ID, "filename": f"{name}.pt"}, "cell_types": cell_types}


DEFAULT_MODEL_CONFIGS = {
    "conic": _cfg("conic", 0.5, ["Neutrophil", "Epithelial", "Lymphocyte", "Plasma cell", "Eosinophil", "Connective"]),
    "consep": _cfg("consep", 0.25, ["Other", "Inflammatory", "Healthy epithelial", "Malignant epithelial", "Stroma", "Muscle"]),
    "glysac": _cfg("glysac", 0.25, ["Other", "Lymphocyte", "Epithelial", "Ambiguous"]),
    "monusac": _cfg("monusac", 0.25, ["Epithelial", "Lymphocyte", "Macrophage", "Neutrophil"]),
    "nucls": _cfg("nucls", 0.2, ["Tumor", "Stroma", "Lymphocyte", "Plasma cell", "Macrophage", "Other"]),
    "puma": _cfg("puma", 0.22, ["Apoptosis", "Tumor", "Endothelial", "Stroma", "Lymphocyte", "Histocyte",
                                "Epithelial", "Melanophage", "Other"]),
}


class HuggingFaceConfig(BaseModel):
    repo_id: str
    filename: str


class ModelConfig(BaseModel):
    path: str
    mpp: float
    url: str | None = None
    hf: HuggingFaceConfig | None = None
    cell_types: list[str]

    @staticmethod
    def load_from_yaml(path: str) -> "ModelConfig":
        logger.info(f"Loading model config from {path}")
        with open(path) as o:
            config = yaml.safe_load(o)
        if "hf" in config and config["hf"] is not None:
            config["hf"] = HuggingFaceConfig(**config["hf"])
        return ModelConfig(**config)

    def download_if_necessary(self) -> None:
        if Path(self.path).exists():
            logger.info("Model weights already in %s", self.path)
            return
        if os.getenv("CLASSPOSE_SYNTHETIC_WEIGHTS", "0").lower() in ("1", "true"):
            logger.warning("No weights at %s: using seeded synthetic weights (CLASSPOSE_SYNTHETIC_WEIGHTS)", self.path)
            return
        raise FileNotFoundError(
            f"model weights {self.path} not found and downloads are unavailable in this build "
            "(no network). Place the checkpoint there or set CLASSPOSE_SYNTHETIC_WEIGHTS=1.")

    def load_state_dict(self, depth: int | None = None):
        """torch.load of the checkpoint, or the synthetic stand-in with the same key layout."""
        import torch
        if Path(self.path).exists():
            # mmap: the tensors are paged in as NetWeights.from_state_dict converts them (on a background thread of the CLI), not read
            # up front -- a ViT-L checkpoint is 1.2 GB of float32
            try:
                return torch.load(self.path, map_location="cpu", weights_only=True, mmap=True)
            except (RuntimeError, ValueError):          # legacy (non-zipfile) checkpoints cannot be mapped
                return torch.load(self.path, map_location="cpu", weights_only=True)
        from .synth import make_state_dict
        d = int(os.getenv("CLASSPOSE_SYNTHETIC_DEPTH", depth or 24))
        return make_state_dict(len(self.cell_types) + 1, None, depth=d, seed=0)
