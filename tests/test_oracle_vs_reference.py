"""Live cross-check of the oracle against the imported reference.  Runs only in
the build container (skipped wherever /root/reference is absent, e.g. the GPU box)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import as R  # noqa: E402

from msclip_amd import synth  # noqa: E402
from oracle import msclip_oracle as O  # noqa: E402

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="reference tree not present")


def test_b32_other_seed_and_batch():
    model, _ = R.build_reference_model("b32-yfcc-msclips")
    sd = synth.synth_state_dict(synth.schema_of(model), seed=7)
    model.load_state_dict(sd, strict=True)
    img = synth.synth_images(2, seed=11)
    tok = synth.synth_tokens(3, seed=12)
    with torch.no_grad():
        ri, rt = model.encode_image(img), model.encode_text(tok)
        oi, ot = O.encode_image(img, sd, O.arch_b32()), O.encode_text(tok, sd, O.arch_b32())
    assert (ri - oi).abs().max() < 1e-5
    assert (rt - ot).abs().max() < 1e-5
