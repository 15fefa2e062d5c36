"""Plug-in mechanism of the CLI (classpose_amd/hooks.py): nothing is installed unless CLASSPOSE_AMD_PLUGINS names it."""
import numpy as np
import pytest

from classpose_amd import hooks, synth


@pytest.fixture(autouse=True)
def _fresh():
    hooks.reset()
    yield
    hooks.reset()


def test_no_plugins_by_default(monkeypatch):
    monkeypatch.delenv("CLASSPOSE_AMD_PLUGINS", raising=False)
    hooks.load_plugins()
    assert hooks.field_provider is None and hooks.qc_provider is None


def test_synth_plugin_flow_only(monkeypatch):
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow")
    hooks.load_plugins()
    assert hooks.field_provider is not None and hooks.qc_provider is None

    class Plan:
        coords = [((0, 0), 64)]
    slide = synth.SyntheticSlide.from_uri("synthetic://300x300?mpp=0.5&seed=5")
    f = hooks.field_provider(slide, Plan, 7)(0, 64, 64, 64)
    ref = synth.analytic_fields(5, 0, 0, 64, 64, 7, 64, 64)
    assert all(np.array_equal(a, b) for a, b in zip(f[:3], ref[:3]))
    assert hooks.field_provider(object(), Plan, 7) is None          # a real reader is left alone


def test_synth_plugin_qc_and_bad_option(monkeypatch):
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:flow+qc")
    hooks.load_plugins()
    m = hooks.qc_provider("artefact")(np.zeros((50, 70, 3), np.uint8))
    assert m.shape == (50, 70) and np.array_equal(m, synth.analytic_qc_map("artefact", 50, 70))
    hooks.reset()
    monkeypatch.setenv("CLASSPOSE_AMD_PLUGINS", "classpose_amd.synth:nonsense")
    with pytest.raises(ValueError):
        hooks.load_plugins()
