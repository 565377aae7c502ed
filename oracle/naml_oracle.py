"""CPU restatement of the reference NAML forward / loss (TEST INFRASTRUCTURE -- see oracle/__init__.py).

``NAMLModule`` (naml_module.py:261-286): one CNN + additive-attention text encoder shared by title and
abstract (``text.py:112-176``), the category view ``relu(Linear(embedding))`` (``category.py:72-82``), the view
vectors stacked and combined by additive attention (``news.py:118-121,164-165``), additive-attention user
encoder over the clicked news (``user/naml.py:26-33``), dot-product scorer, CE loss.  Elementary fp32 torch ops
on CPU; pinned by tests/golden/make_golden_naml.py against the imported reference components.

Dropout streams as in lstur_oracle (title (0, 1), abstract (2, 3))."""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from .lstur_oracle import TEXT_PREFIX, TEXT_STREAMS, cnn_text_encoder_fwd, unique_params
from .nrms_oracle import additive_attention, ce_loss, click_scores, dropout_multiplier, to_dense_batch

CATEG_PREFIX = "news_encoder.category_encoders.category."
COMBINE_PREFIX = "news_encoder.combine_layer."
USER_PREFIX = "user_encoder.additive_attention."


def make_naml_params(vocab: int, n_categ: int, embed_dim: int = 300, num_filters: int = 400, window: int = 3,
                     query_dim: int = 200, categ_dim: int = 100, text_attrs: Sequence[str] = ("title", "abstract"),
                     seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale):
        return (torch.randn(*shape, generator=g) * scale).float()

    D, F, W, Q = embed_dim, num_filters, window, query_dim
    shared = {
        "embedding_layer.weight": rnd(vocab, D, scale=0.3),
        "cnn.weight": rnd(F, 1, W, D, scale=(W * D) ** -0.5),
        "cnn.bias": rnd(F, scale=0.05),
        "additive_attention.linear.weight": rnd(Q, F, scale=F ** -0.5),
        "additive_attention.linear.bias": rnd(Q, scale=0.05),
        "additive_attention.query": rnd(Q, scale=0.1),
    }
    params = {}
    for a in text_attrs:
        for k, v in shared.items():
            params[TEXT_PREFIX.format(a) + k] = v
    params[CATEG_PREFIX + "embedding_layer.weight"] = rnd(n_categ, categ_dim, scale=0.3)
    params[CATEG_PREFIX + "linear.weight"] = rnd(F, categ_dim, scale=categ_dim ** -0.5)
    params[CATEG_PREFIX + "linear.bias"] = rnd(F, scale=0.05)
    for pre in (COMBINE_PREFIX, USER_PREFIX):
        params[pre + "linear.weight"] = rnd(Q, F, scale=F ** -0.5)
        params[pre + "linear.bias"] = rnd(Q, scale=0.05)
        params[pre + "query"] = rnd(Q, scale=0.1)
    return params


def naml_news_encoder_fwd(x, params, text_order, p_drop, seed, row_offset=0, total_rows=None):
    vecs = []
    for a in text_order:
        ids = x[a]
        pre = TEXT_PREFIX.format(a)
        m1 = m2 = None
        if p_drop > 0.0:
            n, L = ids.shape
            tot = total_rows if total_rows is not None else n
            D = params[pre + "embedding_layer.weight"].shape[1]
            F_ = params[pre + "cnn.weight"].shape[0]
            s1, s2 = TEXT_STREAMS[a]
            m1 = dropout_multiplier(seed, s1, p_drop, (tot, L, D))[row_offset:row_offset + n]
            m2 = dropout_multiplier(seed, s2, p_drop, (tot, L, F_))[row_offset:row_offset + n]
        vecs.append(cnn_text_encoder_fwd(ids, params, pre, m1, m2))
    c = params[CATEG_PREFIX + "embedding_layer.weight"][x["category"]]              # category.py:73
    vecs.append(torch.relu(c @ params[CATEG_PREFIX + "linear.weight"].t() + params[CATEG_PREFIX + "linear.bias"]))
    stacked = torch.stack(vecs, dim=1)                                              # news.py:165
    return additive_attention(stacked, params[COMBINE_PREFIX + "linear.weight"], params[COMBINE_PREFIX + "linear.bias"],
                              params[COMBINE_PREFIX + "query"])


def naml_forward(batch, params, text_order=("title", "abstract"), p_drop=0.0, seed=0) -> dict:
    B = int(batch.get("batch_size", int(batch["batch_hist"].max()) + 1))
    nh = batch["x_hist"][text_order[0]].shape[0]
    nc = batch["x_cand"][text_order[0]].shape[0]
    hist_vec = naml_news_encoder_fwd(batch["x_hist"], params, text_order, p_drop, seed, 0, nh + nc)
    cand_vec = naml_news_encoder_fwd(batch["x_cand"], params, text_order, p_drop, seed, nh, nh + nc)
    hist_dense, _ = to_dense_batch(hist_vec, batch["batch_hist"], B)
    cand_dense, _ = to_dense_batch(cand_vec, batch["batch_cand"], B)
    user = additive_attention(hist_dense, params[USER_PREFIX + "linear.weight"], params[USER_PREFIX + "linear.bias"],
                              params[USER_PREFIX + "query"])                        # user/naml.py:31
    scores = click_scores(user, cand_dense)
    y_true, _ = to_dense_batch(batch["labels"], batch["batch_cand"], B)
    return dict(hist_vec=hist_vec, cand_vec=cand_vec, user_vec=user, scores=scores, y_true=y_true,
                loss=ce_loss(scores, y_true))


def naml_loss_and_grads(batch, params, **kw):
    keys = unique_params(params)
    leaves = {k: params[k].clone().requires_grad_(True) for k in keys}
    ptr2key = {params[k].data_ptr(): k for k in keys}
    full = {k: leaves[ptr2key[v.data_ptr()]] for k, v in params.items()}
    out = naml_forward(batch, full, **kw)
    grads = torch.autograd.grad(out["loss"], [leaves[k] for k in keys], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(keys, grads)}
    for k in g:
        if k.endswith("embedding_layer.weight"):
            g[k][0] = 0.0                                                           # padding_idx = 0
    return out, g
