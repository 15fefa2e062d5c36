"""Freezes a few small outputs of the oracle's cellpose-derived half: the restatements in oracle/dynamics.py
and oracle/tiling.py that NO reference fixture can pin (the cellpose / fastremap / fill-voids wheels are absent) are at least
pinned against THEMSELVES, so that an edit to the oracle that changes any id map, flow error or normalised pixel shows up as
a test failure instead of silently moving the target the HIP kernels are compared with.
    python tests/golden/make_oracle_self_golden.py        -> tests/golden/oracle_self_golden.npz
Inputs are regenerated from seeds by tests/test_oracle_hardening.py (classpose_amd.synth is deterministic)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from classpose_amd import synth          # noqa: E402
from oracle import classmask, dynamics, tiling   # noqa: E402

CASES = [(1234, 0, 0, 96, 128), (77, 300, 40, 128, 96), (5, 1000, 2000, 112, 112)]


def case_inputs(seed, x0, y0, h, w, ncls=7):
    dP, cp, lg, _ = synth.analytic_fields(seed, x0, y0, w, h, ncls)
    return dP.astype(np.float32), cp.astype(np.float32), lg.astype(np.float32)


def main():
    out = {}
    for k, (seed, x0, y0, h, w) in enumerate(CASES):
        dP, cp, lg = case_inputs(seed, x0, y0, h, w)
        masks, st = dynamics.compute_masks(dP, cp, return_stages=True)
        out[f"c{k}_masks"] = masks
        out[f"c{k}_seeded"] = st["masks_seeded"].astype(np.uint16)
        out[f"c{k}_flowfiltered"] = st["masks_flowfiltered"].astype(np.uint16)
        out[f"c{k}_flow_errors"] = st["flow_errors"]
        out[f"c{k}_p_final"] = st["p_final"][:, ::17].copy()
        cm, _ = classmask.compute_class_masks(masks, lg)
        out[f"c{k}_class"] = cm.astype(np.uint8)
        tile = synth.render_region(seed, x0, y0, w, h)
        out[f"c{k}_norm"] = tiling.normalize_img(tile[None])[0][::9, ::9].copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_self_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
