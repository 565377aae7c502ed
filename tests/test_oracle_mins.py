"""Pins the MINS CPU oracle (oracle/mins_oracle.py) against golden vectors produced by the reference's own
components (tests/golden/make_golden_mins.py).  CPU-only."""
import numpy as np
import pytest

from oracle import mins_oracle as MO
from tests.helpers import MINS_CASES, check_lstur_grads, load_golden, lstur_golden_batch, mins_golden_cfg, mins_golden_params


@pytest.mark.parametrize("name", MINS_CASES)
def test_mins_forward_and_grads_match_reference(name):
    g = load_golden(name)
    cfg = mins_golden_cfg(g)
    out, grads = MO.mins_loss_and_grads(lstur_golden_batch(g), mins_golden_params(cfg), text_order=cfg["text_order"],
                                        num_heads=cfg["heads"], channels=cfg["channels"], p_drop=cfg["p_drop"],
                                        seed=cfg["seed"])
    rs = int(g["cfg_row_stride"])
    for k in ("user_vec", "scores", "y_true"):
        assert np.abs(out[k].detach().numpy() - g["out_" + k]).max() <= 2e-5, k
    for k in ("hist_vec", "cand_vec"):
        assert np.abs(out[k].detach().numpy()[::rs] - g["out_" + k]).max() <= 2e-5, k
    assert abs(float(out["loss"].detach()) - float(g["out_loss"])) <= 1e-5
    check_lstur_grads(g, grads)
    # the closing additive attention pools one element: its parameters get exactly no gradient
    for k in MO.ATT_KEYS:
        assert float(g["gnorm/" + MO.USER + k]) == 0.0
