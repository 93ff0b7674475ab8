"""Drop-in check of the boundary (this container only: needs /root/reference; skipped on the GPU box).

The reference's own model files are imported with ``sys.modules['MinkowskiEngine']`` aliased to
``gcl_amd.MinkowskiEngine`` and its ``ResUNetBN2C`` is constructed on our operator surface: every
call the reference's constructors make must be accepted, and the resulting parameters must match our own model's
names and shapes (so checkpoints are interchangeable).  No compute is run here (no GPU)."""
import codecs
import importlib
import os
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def test_reference_model_builds_on_our_operator_surface():
    import gcl_amd.MinkowskiEngine as ME_amd
    codecs.register(lambda n: codecs.lookup("utf-8") if n in ("future_fstrings", "future-fstrings") else None)
    saved = {k: sys.modules.get(k) for k in ("MinkowskiEngine", "MinkowskiEngine.MinkowskiFunctional", "model",
                                             "model.resunet", "model.common", "model.residual_block")}
    sys.modules["MinkowskiEngine"] = ME_amd
    sys.modules["MinkowskiEngine.MinkowskiFunctional"] = ME_amd.MinkowskiFunctional
    sys.path.insert(0, REF)
    try:
        for m in ("model", "model.common", "model.residual_block", "model.resunet"):
            sys.modules.pop(m, None)
        ref_resunet = importlib.import_module("model.resunet")
        ref = ref_resunet.ResUNetBN2C(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
        from gcl_amd.model import load_model
        ours = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
        a, b = ref.state_dict(), ours.state_dict()
        assert list(a.keys()) != [] and set(a.keys()) == set(b.keys())
        assert all(tuple(a[k].shape) == tuple(b[k].shape) for k in a)
        ours.load_state_dict(a)                       # checkpoints are interchangeable
        fat = ref_resunet.ResUNetFatBN(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3)
        assert fat.conv1_tr.kernel.shape == (160, 128)
    finally:
        sys.path.remove(REF)
        for m in ("model", "model.common", "model.residual_block", "model.resunet"):
            sys.modules.pop(m, None)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
