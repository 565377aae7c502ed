"""Pins the CenNewsRec CPU oracle (oracle/cen_news_rec_oracle.py) against golden vectors produced by the
reference's own components (tests/golden/make_golden_cen_news_rec.py).  CPU-only."""
import numpy as np
import pytest

from oracle import cen_news_rec_oracle as CO
from tests.helpers import CEN_CASES, cen_golden_batch, cen_golden_cfg, cen_golden_params, check_lstur_grads, load_golden


@pytest.mark.parametrize("name", CEN_CASES)
def test_cen_news_rec_forward_and_grads_match_reference(name):
    g = load_golden(name)
    cfg = cen_golden_cfg(g)
    out, grads = CO.cen_news_rec_loss_and_grads(cen_golden_batch(g), cen_golden_params(cfg), num_heads=cfg["heads"],
                                                num_recent_news=cfg["recent"], late_fusion=cfg["late_fusion"],
                                                p_drop=cfg["p_drop"], seed=cfg["seed"])
    rs = int(g["cfg_row_stride"])
    for k in ("user_vec", "scores", "y_true"):
        assert np.abs(out[k].detach().numpy() - g["out_" + k]).max() <= 2e-5, k
    for k in ("hist_vec", "cand_vec"):
        assert np.abs(out[k].detach().numpy()[::rs] - g["out_" + k]).max() <= 2e-5, k
    assert abs(float(out["loss"].detach()) - float(g["out_loss"])) <= 1e-5
    check_lstur_grads(g, grads)


def test_recent_slice_takes_trailing_padded_slots():
    """cen_news_rec.py:75 slices the DENSE history: users with short histories feed zero rows to the GRU."""
    import torch
    p = CO.make_cen_news_rec_params(16, 8, 12, 3, 6, seed=0)
    hist = torch.zeros(2, 5, 12)
    hist[0, :5] = torch.randn(5, 12)
    hist[1, :1] = torch.randn(1, 12)
    full = CO.cen_news_rec_user_encoder_fwd(hist, p, 3, 2)
    changed = hist.clone()
    changed[1, 0] += 1.0                       # outside the trailing 2 slots: short-term branch must not see it
    gru = [p[CO.USER + k] for k in CO.GRU_KEYS]
    from oracle.lstur_oracle import gru_last_hidden
    h = lambda x: gru_last_hidden(x[:, -2:], torch.full((2,), 2), torch.zeros(2, 12), *gru)  # noqa: E731
    assert torch.equal(h(hist)[1], h(changed)[1])
    assert full.shape == (2, 12)
