"""Validation step's host arithmetic: the oracle restatement AND the product's host functions against vectors captured
from the reference's own code (tests/golden/validation_*.npz <- tests/golden/make_golden.py::validation_golden)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import transform_oracle as TO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PATHS = sorted(glob.glob(os.path.join(G, "validation_s*.npz")))


def test_fixtures_present():
    assert len(PATHS) == 3


@pytest.mark.parametrize("path", PATHS)
def test_transform_oracle_matches_reference_output(path):
    d = np.load(path)
    assert np.abs(TO.est_quad_linear_robust(d["src"], d["tgt"]) - d["T_ref"]).max() < 5e-6
    assert np.abs(TO.est_quad_linear_robust(d["src"], d["tgt"], d["weight"]) - d["T_ref_weighted"]).max() < 5e-6
    assert abs(TO.corr_dist(d["T_ref"], d["T_gt"], d["src"]) - float(d["corr_dist"])) < 1e-7
    assert abs(TO.hit_ratio(d["src"], d["tgt"], d["T_gt"], float(d["hit_thresh"])) - float(d["hit_ratio"])) < 1e-6


@pytest.mark.parametrize("path", PATHS)
def test_product_host_functions_match_reference_output(path):
    """gcl_amd.util.transform_estimation / lib.metrics.corr_dist / the trainer's evaluate_hit_ratio are host code (as in
    the reference): checked here without a GPU."""
    from gcl_amd.lib.colocation_trainer import FinestContrastiveLossTrainer
    from gcl_amd.lib.metrics import corr_dist
    from gcl_amd.util import transform_estimation as te
    d = np.load(path)
    src, tgt, T_gt = torch.from_numpy(d["src"]), torch.from_numpy(d["tgt"]), torch.from_numpy(d["T_gt"])
    T = te.est_quad_linear_robust(src, tgt)
    assert T.dtype == torch.float32 and (T - torch.from_numpy(d["T_ref"])).abs().max() < 5e-6
    Tw = te.est_quad_linear_robust(src, tgt, torch.from_numpy(d["weight"]))
    assert (Tw - torch.from_numpy(d["T_ref_weighted"])).abs().max() < 5e-6
    assert abs(float(corr_dist(torch.from_numpy(d["T_ref"]), T_gt, src, tgt)) - float(d["corr_dist"])) < 1e-6
    tr = FinestContrastiveLossTrainer.__new__(FinestContrastiveLossTrainer)
    assert abs(tr.evaluate_hit_ratio(src, tgt, T_gt, thresh=float(d["hit_thresh"])) - float(d["hit_ratio"])) < 1e-6
    # the estimate recovers the true motion on these problems (outliers included)
    assert (T - T_gt).abs().max() < 5e-3
