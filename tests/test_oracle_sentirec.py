"""Pins the SentiRec CPU oracle (oracle/sentirec_oracle.py) against golden vectors produced with the reference's
own NRMS components and its restated extra loss terms (tests/golden/make_golden_sentirec.py).  CPU-only."""
import numpy as np
import pytest

from oracle import sentirec_oracle as SO
from tests.helpers import SENTIREC_CASES, check_grads_against_golden, load_golden, sentirec_golden_batch


@pytest.mark.parametrize("name", SENTIREC_CASES)
def test_sentirec_forward_and_grads_match_reference(name):
    g = load_golden(name)
    params = SO.make_sentirec_params(int(g["cfg_vocab"]), int(g["cfg_n_sent"]), seed=int(g["cfg_param_seed"]))
    out, grads = SO.sentirec_loss_and_grads(sentirec_golden_batch(g), params, pred_coef=float(g["cfg_pred_coef"]),
                                            div_coef=float(g["cfg_div_coef"]), p_drop=float(g["cfg_p_drop"]),
                                            seed=int(g["cfg_seed"]))
    rs = int(g["cfg_row_stride"])
    assert np.abs(out["scores"].detach().numpy() - g["out_scores"]).max() <= 2e-5
    assert np.abs(out["sent_scores"].detach().numpy()[::rs] - g["out_sent_scores"]).max() <= 2e-5
    assert abs(float(out["sent_div_loss"].detach()) - float(g["out_sent_div_loss"])) <= 1e-6
    assert abs(float(out["loss"].detach()) - float(g["out_loss"])) <= 2e-5
    check_grads_against_golden(g, grads)
    # the predictor's output never reaches the loss (sentirec_module.py:348-352): no gradient at all
    assert float(g["gnorm/sent_predictor.weight"]) == 0.0 and float(grads["sent_predictor.weight"].abs().max()) == 0.0
