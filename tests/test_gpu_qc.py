"""GPU parity: GrandQC UNet++/EfficientNet-B0 forward (cpx_qc_forward) vs the torch-CPU oracle."""
import numpy as np
import pytest
import torch

from classpose_amd import grandqc, synth
from oracle import grandqc as og

pytestmark = pytest.mark.gpu


def _oracle_logits(sd, patches):
    with torch.no_grad():
        x = torch.cat([og.preprocess(p) for p in patches])
        return og.forward(sd, x).permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize("n_classes,nB,H,W,seed", [(2, 1, 512, 512, 1), (8, 2, 256, 320, 2), (2, 3, 64, 96, 3)])
def test_qc_forward_matches_oracle(cuda, n_classes, nB, H, W, seed):
    """float32 logits: rel-L2 <= 2e-5 of the oracle (BatchNorm folding + summation order are the only
    differences); class maps identical wherever the oracle's top-2 margin exceeds 1e-3"""
    sd = synth.make_grandqc_state_dict(n_classes, seed)
    patches = np.stack([synth.render_region(40 + seed, 700 * i, 33 * i, W, H) for i in range(nB)])
    patches[0, : H // 2, : W // 3] = 245                      # a flat background region
    net = grandqc.QcNet.from_state_dict(sd, cuda)
    cls, logits = net.forward(torch.from_numpy(patches).to(cuda), return_logits=True)
    ref = _oracle_logits(sd, patches)
    got = logits.cpu().numpy()
    err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    assert err < 2e-5, err
    assert np.abs(got - ref).max() < 1e-3 * max(1.0, np.abs(ref).max())
    ref_cls = ref.argmax(-1)
    srt = np.sort(ref, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 1e-3
    assert decided.mean() > 0.99
    assert np.array_equal(cls.cpu().numpy()[decided], ref_cls[decided].astype(np.int8))
    # np.argmax tie rule on the device's own logits: bit-exact
    assert np.array_equal(cls.cpu().numpy(), got.argmax(-1).astype(np.int8))
    # deterministic
    cls2 = net.forward(torch.from_numpy(patches).to(cuda))
    assert torch.equal(cls, cls2)
