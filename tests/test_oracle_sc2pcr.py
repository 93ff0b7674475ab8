"""oracle/sc2pcr_oracle.py against outputs of the reference's own Matcher.SC2_PCR (tests/golden/sc2pcr_*.npz)."""
import glob
import os

import numpy as np
import pytest

from oracle.sc2pcr_oracle import sc2_pcr

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("inlier_threshold", "d_thre", "num_iterations", "ratio", "nms_radius", "max_points", "k1", "k2")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "sc2pcr_*.npz"))))
def test_sc2pcr_oracle_matches_reference_output(path):
    z = np.load(path)
    cfg = {k: (int(z[k]) if k in ("num_iterations", "max_points", "k1", "k2") else float(z[k])) for k in KEYS}
    T = sc2_pcr(z["src"], z["tgt"], **cfg).numpy()
    # tie order of argsort / argmax is unspecified in the reference; the refined transformation agrees anyway
    assert np.abs(T - z["T_ref"]).max() < 2e-3
    assert np.abs(T - z["T_true"]).max() < 2e-2
