"""GPU parity of the TANR module against the golden vectors made from the reference's own components."""
import numpy as np
import pytest
import torch

from tests.helpers import (TANR_CASES, batch_to, build_tanr_module, check_lstur_grads, load_golden, module_grads,
                           tanr_golden_batch, tanr_golden_cfg, tanr_golden_params)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f32", "bf16x3"])
def engine(request):
    from newsreclib_amd import _lib
    prev = _lib.get_gemm_engine()
    _lib.set_gemm_engine(request.param)
    yield request.param
    _lib.set_gemm_engine(prev)


@pytest.mark.parametrize("name", TANR_CASES)
def test_tanr_module_matches_reference_golden(name, engine):
    g = load_golden(name)
    cfg = tanr_golden_cfg(g)
    mod = build_tanr_module(cfg, tanr_golden_params(cfg))
    mod.train() if cfg["p_drop"] > 0 else mod.eval()
    batch = batch_to(tanr_golden_batch(g), "cuda")
    orig = mod.forward
    mod.forward = lambda b, seed=None: orig(b, seed=cfg["seed"])
    loss, preds, targets, cand_news_size, *_ = mod.model_step(batch)
    ftol, gtol = (2e-5, 2e-4) if engine == "f32" else (1e-4, 5e-4)
    dense = np.zeros_like(g["out_scores"])
    sizes = cand_news_size.cpu().numpy()
    p, o = preds.cpu().numpy(), 0
    for b, n in enumerate(sizes):
        dense[b, :n] = p[o:o + n]
        o += n
    assert float(np.abs(dense - g["out_scores"]).max()) <= max(ftol * 5, 1e-4)       # contract 1e-3
    assert abs(float(loss) - float(g["out_loss"])) <= 2e-4
    loss.backward()
    check_lstur_grads(g, module_grads(mod), tol=gtol, rtol=5e-4)


def test_tanr_trainer_updates_every_parameter():
    """The topic predictor's gradient arrives through ordinary autograd (``.grad``), not through a kernel writing
    ``main_grad``: the flat-buffer trainer must fold it in (Adam moves every parameter)."""
    from newsreclib_amd.trainer import NRMSTrainer
    g = load_golden("tanr_tiny_train")
    cfg = tanr_golden_cfg(g)
    mod = build_tanr_module(cfg, tanr_golden_params(cfg))
    before = {k: p.detach().clone() for k, p in mod.named_parameters()}
    NRMSTrainer(mod, lr=1e-3).step(batch_to(tanr_golden_batch(g), "cuda"))
    for k, p in mod.named_parameters():
        assert float((p.detach() - before[k]).abs().max()) > 0.0, k
