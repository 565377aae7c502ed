"""Pins the NAML CPU oracle (oracle/naml_oracle.py) against golden vectors produced by the reference's own
components (tests/golden/make_golden_naml.py).  CPU-only."""
import numpy as np
import pytest

from oracle import naml_oracle as NO
from tests.helpers import (NAML_CASES, check_lstur_grads, load_golden, lstur_golden_batch, naml_golden_cfg,
                           naml_golden_params)


@pytest.mark.parametrize("name", NAML_CASES)
def test_naml_forward_and_grads_match_reference(name):
    g = load_golden(name)
    cfg = naml_golden_cfg(g)
    batch = lstur_golden_batch(g)
    params = naml_golden_params(cfg)
    out, grads = NO.naml_loss_and_grads(batch, params, text_order=cfg["text_order"], p_drop=cfg["p_drop"], seed=cfg["seed"])
    rs = int(g["cfg_row_stride"])
    for k in ("user_vec", "scores", "y_true"):
        assert np.abs(out[k].detach().numpy() - g["out_" + k]).max() <= 2e-5, k
    for k in ("hist_vec", "cand_vec"):
        assert np.abs(out[k].detach().numpy()[::rs] - g["out_" + k]).max() <= 2e-5, k
    assert abs(float(out["loss"].detach()) - float(g["out_loss"])) <= 1e-5
    check_lstur_grads(g, grads)
