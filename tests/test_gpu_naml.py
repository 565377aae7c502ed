"""GPU parity of the generic blocks (standalone additive attention, linear + activation) and of the NAML
module against the CPU oracle and the golden vectors made from the reference's own components."""
import numpy as np
import pytest
import torch

from tests.helpers import (NAML_CASES, batch_to, build_naml_module, check_lstur_grads, load_golden, lstur_golden_batch,
                           module_grads, naml_golden_cfg, naml_golden_params)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f32", "bf16x3"])
def engine(request):
    from newsreclib_amd import _lib
    prev = _lib.get_gemm_engine()
    _lib.set_gemm_engine(request.param)
    yield request.param
    _lib.set_gemm_engine(prev)


def _tols(engine):
    return (2e-5, 2e-4) if engine == "f32" else (1e-4, 5e-4)


@pytest.mark.parametrize("shape", [(7, 3, 64, 32), (5, 50, 400, 200), (130, 4, 16, 8), (1, 1, 32, 16)])
def test_additive_attention_matches_oracle(shape, engine):
    from newsreclib_amd.attention import AdditiveAttention
    from oracle.nrms_oracle import additive_attention
    G, S, D, Q = shape
    rng = np.random.default_rng(G + S)
    t = lambda *s, scale=1.0: torch.from_numpy((rng.standard_normal(s) * scale).astype(np.float32))  # noqa: E731
    y, w, b, q, d_out = t(G, S, D, scale=0.5), t(Q, D, scale=D ** -0.5), t(Q, scale=0.05), t(Q, scale=0.1), t(G, D)
    leaves = [x.clone().requires_grad_(True) for x in (y, w, b, q)]
    ref = additive_attention(*leaves)
    ref.backward(d_out)
    att = AdditiveAttention(D, Q).cuda()
    with torch.no_grad():
        att.linear.weight.copy_(w); att.linear.bias.copy_(b); att.query.copy_(q)
    yd = y.cuda().requires_grad_(True)
    out = att(yd)
    out.backward(d_out.cuda())
    ftol, gtol = _tols(engine)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= ftol * 5
    for name, got, want in (("y", yd.grad, leaves[0].grad), ("weight", att.linear.weight.grad, leaves[1].grad),
                            ("bias", att.linear.bias.grad, leaves[2].grad), ("query", att.query.grad, leaves[3].grad)):
        assert float((got.cpu() - want).abs().max()) <= gtol * max(1.0, float(want.abs().max())), name


@pytest.mark.parametrize("act", ["none", "tanh", "relu"])
@pytest.mark.parametrize("shape", [(37, 64, 16), (1000, 400, 100), (3, 8, 4)])
def test_linear_act_matches_torch(shape, act, engine):
    from newsreclib_amd.ops_blocks import LinearActFn
    M, N, K = shape
    rng = np.random.default_rng(M + N)
    t = lambda *s, scale=1.0: torch.from_numpy((rng.standard_normal(s) * scale).astype(np.float32))  # noqa: E731
    a, w, b, d_c = t(M, K), t(N, K, scale=K ** -0.5), t(N, scale=0.1), t(M, N)
    leaves = [x.clone().requires_grad_(True) for x in (a, w, b)]
    pre = leaves[0] @ leaves[1].t() + leaves[2]
    ref = {"none": pre, "tanh": torch.tanh(pre), "relu": torch.relu(pre)}[act]
    ref.backward(d_c)
    dev = [x.cuda().requires_grad_(True) for x in (a, w, b)]
    out = LinearActFn.apply(*dev, act, None)
    out.backward(d_c.cuda())
    ftol, gtol = _tols(engine)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= ftol * 5
    if act == "relu":
        # a pre-activation within rounding of 0 may gate differently under bf16x3 (~2^-16 relative error per
        # product), and one flipped gate moves a whole row of dW: compare the gradients under the GATES THE GPU
        # TOOK (the forward comparison above already bounds the outputs), and bound the number of flips
        gate = (out.detach().cpu() > 0)
        assert int((gate != (ref.detach() > 0)).sum()) <= 1e-4 * gate.numel() + 1
        for x in leaves:
            x.grad = None
        ((leaves[0] @ leaves[1].t() + leaves[2]) * gate).backward(d_c)
    for name, got, want in zip(("a", "w", "b"), dev, leaves):
        assert float((got.grad.cpu() - want.grad).abs().max()) <= gtol * max(1.0, float(want.grad.abs().max())), name


@pytest.mark.parametrize("name", NAML_CASES)
def test_naml_module_matches_reference_golden(name, engine):
    from newsreclib_amd.dense_batch import to_dense_batch
    from newsreclib_amd.nrms_module import prepare_batch
    g = load_golden(name)
    cfg = naml_golden_cfg(g)
    mod = build_naml_module(cfg, naml_golden_params(cfg))
    mod.train() if cfg["p_drop"] > 0 else mod.eval()
    pb = prepare_batch(batch_to(lstur_golden_batch(g), "cuda"))
    scores = mod.forward(pb, seed=cfg["seed"])
    ftol, gtol = _tols(engine)
    assert float(np.abs(scores.detach().cpu().numpy() - g["out_scores"]).max()) <= max(ftol * 5, 1e-4)   # contract 1e-3
    y_true, _ = to_dense_batch(pb["labels"], pb["batch_cand"], pb["batch_size"], pb["max_cand"], pb["cand_offsets"],
                               pb["cand_flat_idx"])
    loss = mod.criterion(scores, y_true.float())
    assert abs(float(loss) - float(g["out_loss"])) <= 1e-4
    loss.backward()
    check_lstur_grads(g, module_grads(mod), tol=gtol, rtol=5e-4)


def test_naml_module_with_plm_text_encoder_matches_reference_golden(tmp_path, engine):
    """``use_plm=True`` in a sibling mirror against the REFERENCE's components wired the reference's way
    (tests/golden/make_golden_naml_plm.py: reference ``PLM`` shared by title and abstract, ``LinearEncoder`` with the linear
    transform to the text width, ``NewsEncoder`` add_att, NAML ``UserEncoder``, ``DotProduct``; naml_module.py:149-207, 261-286):
    news vectors, user vectors, scores, loss and every gradient outside the body element by element, the body's by norm."""
    from functools import partial

    from newsreclib_amd.dense_batch import to_dense_batch
    from newsreclib_amd.naml_module import NAMLModule
    from newsreclib_amd.nrms_module import prepare_batch
    from tests.helpers import PLM_HEADS, PLM_Q, make_plm_tail_params, make_tiny_roberta
    g = load_golden("naml_plm_tiny")
    path = make_tiny_roberta(str(tmp_path))
    mod = NAMLModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title", "abstract", "category"],
        outputs={"train": ["preds", "targets", "cand_news_size"], "val": [], "test": []}, dual_loss_training=False,
        dual_loss_coef=None, loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=True,
        pretrained_embeddings_path=None, plm_model=path, frozen_layers=[0], text_embed_dim=int(g["cfg_dim"]), num_heads=PLM_HEADS,
        num_filters=None, window_size=None, query_dim=PLM_Q, categ_embed_dim=int(g["cfg_categ_dim"]), dropout_probability=0.2,
        top_k_list=[5, 10], num_categ_classes=int(g["cfg_n_categ"]) - 1, num_sent_classes=3, save_recs=False, recs_fpath=None,
        optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None)
    params = {k[len("param/"):]: torch.from_numpy(g[k]) for k in g if k.startswith("param/")}
    for attr in ("title", "abstract"):
        for k, v in make_plm_tail_params().items():
            params[f"news_encoder.text_encoders.{attr}.{k}"] = v
    res = mod.load_state_dict(params, strict=False)
    assert not res.unexpected_keys and all(".plm_model." in k for k in res.missing_keys), res
    mod.news_encoder.set_text_order([str(a) for a in g["cfg_text_order"]])
    mod = mod.cuda().eval()

    def side(part):
        x = {a: {"input_ids": torch.from_numpy(g[f"in_{a}_{part}_input_ids"]),
                 "attention_mask": torch.from_numpy(g[f"in_{a}_{part}_attention_mask"])} for a in ("title", "abstract")}
        x["category"] = torch.from_numpy(g[f"in_category_{part}"])
        return x

    B = int(g["in_batch_hist"].max()) + 1
    batch = batch_to({"x_hist": side("hist"), "x_cand": side("cand"), "batch_hist": torch.from_numpy(g["in_batch_hist"]),
                      "batch_cand": torch.from_numpy(g["in_batch_cand"]), "labels": torch.from_numpy(g["in_labels"]),
                      "user_ids": torch.arange(B) + 1}, "cuda")
    pb = prepare_batch(batch)
    ftol, gtol = _tols(engine)
    with torch.no_grad():
        hv, cv = mod.news_encoder(pb["x_hist"]), mod.news_encoder(pb["x_cand"])
    assert float(np.abs(hv.cpu().numpy() - g["out_hist_vec"]).max()) <= 5 * ftol
    assert float(np.abs(cv.cpu().numpy() - g["out_cand_vec"]).max()) <= 5 * ftol
    out = mod.forward(pb)
    scores = out[0] if isinstance(out, tuple) else out
    assert scores.shape == g["out_scores"].shape
    assert float(np.abs(scores.detach().cpu().numpy() - g["out_scores"]).max()) <= max(5 * ftol, 1e-4)
    y_true, _ = to_dense_batch(pb["labels"], pb["batch_cand"], pb["batch_size"], pb["max_cand"], pb["cand_offsets"],
                               pb["cand_flat_idx"])
    assert np.array_equal(y_true.cpu().numpy(), g["out_y_true"])
    loss = mod.criterion(scores, y_true.float())
    assert abs(float(loss.detach()) - float(g["out_loss"])) <= 1e-4
    loss.backward()
    grads = module_grads(mod)
    n_full = n_norm = 0
    for k in g:
        if k.startswith("gfull/"):
            ref = torch.from_numpy(g[k]).double()
            got = grads[k[len("gfull/"):]].detach().cpu().double()
            assert float((got - ref).abs().max()) <= gtol * max(1.0, float(ref.abs().max())), k
            n_full += 1
        elif k.startswith("gnorm/"):
            ref = float(g[k])
            got = float(grads[k[len("gnorm/"):]].detach().cpu().double().norm())
            assert abs(got - ref) <= 2e-3 * ref + 1e-6, (k, got, ref)
            n_norm += 1
    assert n_full >= 14 and n_norm >= 20
