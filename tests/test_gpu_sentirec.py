"""GPU parity of the SentiRec module mirror against the golden vectors (reference NRMS components + its extra
sentiment loss terms)."""
import numpy as np
import pytest
import torch

from oracle import sentirec_oracle as SO
from tests.helpers import (SENTIREC_CASES, batch_to, build_sentirec_module, check_grads_against_golden, load_golden,
                           module_grads, sentirec_golden_batch)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f32", "bf16x3"])
def engine(request):
    from newsreclib_amd import _lib
    prev = _lib.get_gemm_engine()
    _lib.set_gemm_engine(request.param)
    yield request.param
    _lib.set_gemm_engine(prev)


@pytest.mark.parametrize("name", SENTIREC_CASES)
def test_sentirec_module_matches_reference_golden(name, engine):
    from newsreclib_amd.nrms_module import prepare_batch
    g = load_golden(name)
    params = SO.make_sentirec_params(int(g["cfg_vocab"]), int(g["cfg_n_sent"]), seed=int(g["cfg_param_seed"]))
    mod = build_sentirec_module(g, params)
    p_drop = float(g["cfg_p_drop"])
    mod.train() if p_drop > 0 else mod.eval()
    pb = prepare_batch(batch_to(sentirec_golden_batch(g), "cuda"))
    if p_drop > 0:      # pin the dropout draw of the fixture
        import newsreclib_amd.news_encoder as NE
        orig, NE._draw_seed = NE._draw_seed, (lambda: int(g["cfg_seed"]))
    try:
        out = mod.model_step(pb)
    finally:
        if p_drop > 0:
            NE._draw_seed = orig
    loss = out[0]
    scores, (sent_scores, _) = mod.forward(pb) if p_drop == 0 else (None, (None, None))
    if scores is not None:
        assert float(np.abs(scores.detach().cpu().numpy() - g["out_scores"]).max()) <= 1e-4
        rs = int(g["cfg_row_stride"])
        assert float(np.abs(sent_scores.detach().cpu().numpy()[::rs] - g["out_sent_scores"]).max()) <= 2e-4
    assert abs(float(loss.detach()) - float(g["out_loss"])) <= 2e-4 * max(1.0, abs(float(g["out_loss"])))
    loss.backward()
    grads = module_grads(mod)
    tol = 2e-4 if engine == "f32" else 6e-4
    check_grads_against_golden(g, grads, rtol=tol)
    assert float(grads["sent_predictor.weight"].abs().max()) == 0.0
