"""CPU restatement of the reference TANR forward / loss (TEST INFRASTRUCTURE -- see oracle/__init__.py).

``TANRModule`` (tanr_module.py:258-286,361-367): CNN + additive-attention title encoder (text.py:112-176),
NAML's additive-attention user encoder (user/naml.py:26-33), dot-product scorer, CE loss, plus the topic
prediction task: ``nn.Linear`` over [candidate; history] news vectors, CE against the one-hot category, added
with weight ``topic_pred_loss_coef``.  Pinned by tests/golden/make_golden_tanr.py."""
from __future__ import annotations

from typing import Dict

import torch

from .lstur_oracle import TEXT_PREFIX, cnn_text_encoder_fwd
from .nrms_oracle import additive_attention, ce_loss, click_scores, dropout_multiplier, to_dense_batch

USER_PREFIX = "user_encoder.additive_attention."


def make_tanr_params(vocab: int, n_categ: int, embed_dim: int = 300, num_filters: int = 400, window: int = 3,
                     query_dim: int = 200, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale):
        return (torch.randn(*shape, generator=g) * scale).float()

    D, F, W, Q = embed_dim, num_filters, window, query_dim
    pre = TEXT_PREFIX.format("title")
    return {
        pre + "embedding_layer.weight": rnd(vocab, D, scale=0.3),
        pre + "cnn.weight": rnd(F, 1, W, D, scale=(W * D) ** -0.5),
        pre + "cnn.bias": rnd(F, scale=0.05),
        pre + "additive_attention.linear.weight": rnd(Q, F, scale=F ** -0.5),
        pre + "additive_attention.linear.bias": rnd(Q, scale=0.05),
        pre + "additive_attention.query": rnd(Q, scale=0.1),
        "topic_predictor.weight": rnd(n_categ, F, scale=F ** -0.5),
        "topic_predictor.bias": rnd(n_categ, scale=0.05),
        USER_PREFIX + "linear.weight": rnd(Q, F, scale=F ** -0.5),
        USER_PREFIX + "linear.bias": rnd(Q, scale=0.05),
        USER_PREFIX + "query": rnd(Q, scale=0.1),
    }


def tanr_forward(batch, params, coef: float = 0.2, p_drop: float = 0.0, seed: int = 0) -> dict:
    B = int(batch.get("batch_size", int(batch["batch_hist"].max()) + 1))
    pre = TEXT_PREFIX.format("title")
    ids_h, ids_c = batch["x_hist"]["title"], batch["x_cand"]["title"]
    nh, nc, L = ids_h.shape[0], ids_c.shape[0], ids_h.shape[1]
    m1 = m2 = None
    if p_drop > 0.0:
        D = params[pre + "embedding_layer.weight"].shape[1]
        F_ = params[pre + "cnn.weight"].shape[0]
        m1 = dropout_multiplier(seed, 0, p_drop, (nh + nc, L, D))
        m2 = dropout_multiplier(seed, 1, p_drop, (nh + nc, L, F_))
    news = cnn_text_encoder_fwd(torch.cat([ids_h, ids_c]), params, pre, m1, m2)
    hist_vec, cand_vec = news[:nh], news[nh:]
    hist_dense, _ = to_dense_batch(hist_vec, batch["batch_hist"], B)
    cand_dense, _ = to_dense_batch(cand_vec, batch["batch_cand"], B)
    user = additive_attention(hist_dense, params[USER_PREFIX + "linear.weight"], params[USER_PREFIX + "linear.bias"],
                              params[USER_PREFIX + "query"])
    scores = click_scores(user, cand_dense)
    topic_scores = torch.cat((cand_vec, hist_vec)) @ params["topic_predictor.weight"].t() + params["topic_predictor.bias"]
    y_true, _ = to_dense_batch(batch["labels"], batch["batch_cand"], B)
    topics = torch.cat((batch["x_cand"]["category"], batch["x_hist"]["category"]))
    topic_prob = torch.nn.functional.one_hot(topics, num_classes=params["topic_predictor.weight"].shape[0]).float()
    rec_loss, topic_loss = ce_loss(scores, y_true), ce_loss(topic_scores, topic_prob)
    return dict(hist_vec=hist_vec, cand_vec=cand_vec, user_vec=user, scores=scores, topic_scores=topic_scores,
                y_true=y_true, loss=rec_loss + coef * topic_loss)


def tanr_loss_and_grads(batch, params, **kw):
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = tanr_forward(batch, leaves, **kw)
    grads = torch.autograd.grad(out["loss"], list(leaves.values()), allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(leaves, grads)}
    g[TEXT_PREFIX.format("title") + "embedding_layer.weight"][0] = 0.0
    return out, g
