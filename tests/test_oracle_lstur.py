"""Pins the LSTUR CPU oracle (oracle/lstur_oracle.py) against golden vectors produced by the
reference's own components (tests/golden/make_golden_lstur.py).  CPU-only."""
import numpy as np
import pytest
import torch

from oracle import lstur_oracle as LO
from tests.helpers import (LSTUR_CASES, check_lstur_grads, load_golden, lstur_golden_batch, lstur_golden_cfg,
                           lstur_golden_params)


@pytest.mark.parametrize("name", LSTUR_CASES)
def test_lstur_forward_and_grads_match_reference(name):
    g = load_golden(name)
    cfg = lstur_golden_cfg(g)
    batch = lstur_golden_batch(g)
    params = lstur_golden_params(cfg)
    out, grads = LO.lstur_loss_and_grads(batch, params, text_order=cfg["text_order"], method=cfg["method"],
                                         p_drop=cfg["p_drop"], p_mask=cfg["p_mask"], seed=cfg["seed"])
    rs = int(g["cfg_row_stride"])
    for k in ("user_vec", "scores", "y_true"):
        assert np.abs(out[k].detach().numpy() - g["out_" + k]).max() <= 2e-5, k
    for k in ("hist_vec", "cand_vec"):
        assert np.abs(out[k].detach().numpy()[::rs] - g["out_" + k]).max() <= 2e-5, k
    assert abs(float(out["loss"]) - float(g["out_loss"])) <= 1e-5
    check_lstur_grads(g, grads)


def test_gru_rejects_empty_history():
    """pack_padded_sequence (user/lstur.py:74-79) refuses zero-length sequences; so does the restatement."""
    with pytest.raises(ValueError):
        LO.gru_last_hidden(torch.zeros(2, 3, 4), torch.tensor([2, 0]), torch.zeros(2, 4), torch.zeros(12, 4),
                           torch.zeros(12, 4), torch.zeros(12), torch.zeros(12))


def test_user_mask_drops_whole_users():
    m = LO.dropout_multiplier(7, LO.USER_MASK_STREAM, 0.5, (64,))
    assert set(np.unique(m.numpy()).tolist()) <= {0.0, 2.0} and 10 < int((m == 0).sum()) < 54
