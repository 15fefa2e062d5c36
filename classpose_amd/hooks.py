"""Instrumentation points of the WSI command line, filled by plug-ins.

The CLI itself knows nothing about what a plug-in does.  ``CLASSPOSE_AMD_PLUGINS`` is a comma-separated list of
``module[:argument]`` entries; every rank imports each module once and calls its ``register(hooks, argument)``.
A plug-in may set

``field_provider(slide, plan, n_classes) -> None | callable(tile_index, R, W, H) -> (dP, cellprob, logits, ...)``
    extra per-tile tensors that replace the network's flow / probability / class fields in the dynamics (the network
    still runs on the pixels).  ``classpose_amd.synth`` uses it for slides whose nuclei are known analytically, which
    is how the parity tests get meaningful cells out of randomly initialised weights.

``qc_provider(kind) -> None | callable(image) -> class map``
    replaces the GrandQC arg-max map (``kind`` is "tissue" or "artefact"); the QC network still runs.

An environment variable (rather than an argument) because ranks are fresh processes (``mp.spawn`` / ``torchrun``).
"""
from __future__ import annotations

import importlib
import os

field_provider = None
qc_provider = None
_loaded: set[str] = set()


def load_plugins() -> None:
    import sys
    me = sys.modules[__name__]
    for entry in filter(None, (e.strip() for e in os.getenv("CLASSPOSE_AMD_PLUGINS", "").split(","))):
        if entry in _loaded:
            continue
        mod, _, arg = entry.partition(":")
        importlib.import_module(mod).register(me, arg)
        _loaded.add(entry)


def reset() -> None:
    """Forget every plug-in (tests that change CLASSPOSE_AMD_PLUGINS between in-process CLI runs)."""
    global field_provider, qc_provider
    field_provider = None
    qc_provider = None
    _loaded.clear()
