"""Pins the TANR CPU oracle (oracle/tanr_oracle.py) against golden vectors produced by the reference's own
components (tests/golden/make_golden_tanr.py).  CPU-only."""
import numpy as np
import pytest

from oracle import tanr_oracle as TO
from tests.helpers import TANR_CASES, check_lstur_grads, load_golden, tanr_golden_batch, tanr_golden_cfg, tanr_golden_params


@pytest.mark.parametrize("name", TANR_CASES)
def test_tanr_forward_and_grads_match_reference(name):
    g = load_golden(name)
    cfg = tanr_golden_cfg(g)
    out, grads = TO.tanr_loss_and_grads(tanr_golden_batch(g), tanr_golden_params(cfg), coef=cfg["coef"],
                                        p_drop=cfg["p_drop"], seed=cfg["seed"])
    rs = int(g["cfg_row_stride"])
    for k in ("user_vec", "scores", "y_true"):
        assert np.abs(out[k].detach().numpy() - g["out_" + k]).max() <= 2e-5, k
    for k in ("hist_vec", "cand_vec", "topic_scores"):
        assert np.abs(out[k].detach().numpy()[::rs] - g["out_" + k]).max() <= 2e-5, k
    assert abs(float(out["loss"].detach()) - float(g["out_loss"])) <= 1e-5
    check_lstur_grads(g, grads)
