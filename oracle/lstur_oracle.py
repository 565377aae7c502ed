"""CPU restatement of the reference LSTUR train step (TEST INFRASTRUCTURE -- see oracle/__init__.py).

BASELINE.json config 5 / SURVEY.md section 8 row a16: ``LSTURModule`` = CNN + additive-attention text
encoder shared by title and abstract (``text.py:112-176``), category embedding
(``category.py:9-80``), concatenation (``news.py:134-183``, ``combine_type="concat"``), GRU user
encoder initialised with a masked long-term user embedding (``user/lstur.py:6-87``), dot-product
scorer and CE loss (``lstur_module.py:278-303,326-360``).

Arithmetic is written with elementary fp32 torch ops on CPU (matmul / sigmoid / tanh), NOT with
``nn.Conv2d`` / ``nn.GRU`` / ``pack_padded_sequence``, so it is an independent statement;
``tests/golden/make_golden.py`` pins it against the imported reference components.

Dropout (product-defined, see nrms_oracle.dropout_keep_mask): every text attribute has its own
stream pair, fixed BY NAME so the result does not depend on the reference's set-iteration order of
``text_encoders``: title -> (0, 1), abstract -> (2, 3); first stream = post-embedding dropout over
the (N, L, D) activation, second = post-ReLU dropout over the (N, L, F) activation (the reference
applies it to the (N, F, L) tensor, ``text.py:171-172``; elementwise, so only the flat index
convention differs).  The long-term user vector is masked by ``nn.Dropout2d`` on a (1, B, D) tensor
(``user/lstur.py:58,71``), which drops WHOLE USERS: stream 8, flat index = position in the batch.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .nrms_oracle import (additive_attention, ce_loss, click_scores, dropout_multiplier,
                          to_dense_batch)

TEXT_STREAMS = {"title": (0, 1), "abstract": (2, 3)}
USER_MASK_STREAM = 8

TEXT_PREFIX = "news_encoder.text_encoders.{}."
CATEG_KEY = "news_encoder.category_encoders.category.embedding_layer.weight"
USER_PREFIX = "user_encoder."
CNN_KEYS = ("embedding_layer.weight", "cnn.weight", "cnn.bias", "additive_attention.linear.weight",
            "additive_attention.linear.bias", "additive_attention.query")
GRU_KEYS = ("gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0")


def make_lstur_params(vocab: int, n_categ: int, n_users: int, embed_dim: int = 300, num_filters: int = 300,
                      window: int = 3, query_dim: int = 200, categ_dim: int = 100,
                      text_attrs: Sequence[str] = ("title", "abstract"), method: str = "ini",
                      seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded parameters under the reference's state_dict keys.  The text encoder is ONE module
    registered under every text attribute (``news.py:69-79``), so its tensors appear once per
    attribute prefix and are the same storage."""
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale):
        return (torch.randn(*shape, generator=g) * scale).float()

    D, F, W, Q = embed_dim, num_filters, window, query_dim
    emb = rnd(vocab, D, scale=0.3)
    shared = {
        "embedding_layer.weight": emb,
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
    params[CATEG_KEY] = rnd(n_categ, categ_dim, scale=0.3)
    din = F * len(text_attrs) + categ_dim
    hd = din if method == "ini" else din // 2
    params[USER_PREFIX + "long_term_user_embedding.weight"] = rnd(n_users, hd, scale=0.3)
    params[USER_PREFIX + "gru.weight_ih_l0"] = rnd(3 * hd, din, scale=din ** -0.5)
    params[USER_PREFIX + "gru.weight_hh_l0"] = rnd(3 * hd, hd, scale=hd ** -0.5)
    params[USER_PREFIX + "gru.bias_ih_l0"] = rnd(3 * hd, scale=0.05)
    params[USER_PREFIX + "gru.bias_hh_l0"] = rnd(3 * hd, scale=0.05)
    return params


# --------------------------------------------------------------------------------------
# CNNAddAtt.forward (text.py:163-176)
# --------------------------------------------------------------------------------------
def cnn_text_encoder_fwd(ids: torch.Tensor, params: Dict[str, torch.Tensor], prefix: str,
                         mult1: Optional[torch.Tensor] = None, mult2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ids (N, L) int64 -> (N, F).  ``nn.Conv2d(1, F, (W, D), padding=((W-1)//2, 0))`` over the
    (L, D) "image" of a news = for every token l the dot product of the window rows
    [l - pad, l - pad + W) (zero rows outside the news) with the filter, + bias."""
    x = params[prefix + "embedding_layer.weight"][ids]           # text.py:165 (id 0 = ordinary row)
    if mult1 is not None:
        x = x * mult1                                              # text.py:166
    w = params[prefix + "cnn.weight"]                              # (F, 1, W, D)
    F_, _, W, D = w.shape
    pad = (W - 1) // 2                                             # text.py:152  int((W-1)/2)
    N, L, _ = x.shape
    xp = torch.zeros(N, L + 2 * pad, D, dtype=x.dtype)
    xp[:, pad:pad + L] = x
    l_out = L + 2 * pad - W + 1                                    # == L for odd W
    win = torch.cat([xp[:, t:t + l_out] for t in range(W)], dim=2)  # (N, l_out, W*D), k = t*D + d
    c = win @ w.reshape(F_, W * D).t() + params[prefix + "cnn.bias"]  # text.py:169
    c = torch.relu(c)                                              # text.py:170
    if mult2 is not None:
        c = c * mult2                                              # text.py:171
    return additive_attention(c, params[prefix + "additive_attention.linear.weight"],
                              params[prefix + "additive_attention.linear.bias"],
                              params[prefix + "additive_attention.query"])  # text.py:174


# --------------------------------------------------------------------------------------
# LSTUR UserEncoder.forward (user/lstur.py:66-87); nn.GRU cell equations (torch docs, gate order r|z|n)
# --------------------------------------------------------------------------------------
def gru_last_hidden(x: torch.Tensor, lengths: torch.Tensor, h0: torch.Tensor, w_ih, w_hh, b_ih, b_hh) -> torch.Tensor:
    """x (B, T, Din) batch-first, lengths (B,) >= 1, h0 (B, Hd) -> hidden state of every sequence
    after ITS OWN last valid step (what ``nn.GRU`` returns as h_n for a packed sequence)."""
    if int(lengths.min()) < 1:
        raise ValueError("Length of all samples has to be greater than 0")   # pack_padded_sequence's check
    Hd = h0.shape[1]
    h = h0
    for t in range(int(lengths.max())):
        gi = x[:, t] @ w_ih.t() + b_ih
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :Hd] + gh[:, :Hd])
        z = torch.sigmoid(gi[:, Hd:2 * Hd] + gh[:, Hd:2 * Hd])
        n = torch.tanh(gi[:, 2 * Hd:] + r * gh[:, 2 * Hd:])
        h_new = (1.0 - z) * n + z * h
        active = (t < lengths).unsqueeze(1)
        h = torch.where(active, h_new, h)
    return h


def lstur_user_encoder_fwd(user_idx: torch.Tensor, hist: torch.Tensor, hist_size: torch.Tensor,
                           params: Dict[str, torch.Tensor], method: str = "ini",
                           user_mult: Optional[torch.Tensor] = None) -> torch.Tensor:
    """user_idx (B,), hist (B, H, Din), hist_size (B,) -> (B, Hd) ("ini") or (B, 2*Hd) ("con").
    user_mult (B,) in {0, 1/(1-p)}: the Dropout2d channel mask (whole users)."""
    P = USER_PREFIX
    u = params[P + "long_term_user_embedding.weight"][user_idx]    # lstur.py:70
    if user_mult is not None:
        u = u * user_mult.unsqueeze(1)                             # lstur.py:71
    gru = [params[P + k] for k in GRU_KEYS]
    if method == "ini":
        return gru_last_hidden(hist, hist_size, u, *gru)           # lstur.py:81-83
    h0 = torch.zeros(hist.shape[0], gru[1].shape[1], dtype=hist.dtype)
    return torch.cat([gru_last_hidden(hist, hist_size, h0, *gru), u], dim=1)   # lstur.py:85-87


# --------------------------------------------------------------------------------------
# LSTURModule.forward (lstur_module.py:278-303) + loss (lstur_module.py:326-360)
# --------------------------------------------------------------------------------------
def lstur_news_encoder_fwd(x: Dict[str, torch.Tensor], params, text_order: Sequence[str], p_drop: float, seed: int,
                           row_offset: int = 0, total_rows: Optional[int] = None) -> torch.Tensor:
    """``NewsEncoder.forward`` (news.py:134-183): text vectors in ``text_order`` (the reference iterates
    a ModuleDict built from a Python set, so the order is an input here), then the category vector,
    concatenated.  Dropout multipliers are indexed over the concatenated [hist; cand] rows: this call
    covers rows [row_offset, row_offset + n) of ``total_rows``."""
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
    vecs.append(params[CATEG_KEY][x["category"]])                 # category.py:73 (no dropout / linear)
    return torch.cat(vecs, dim=1)                                 # news.py:128 lambda


def lstur_forward(batch: dict, params: Dict[str, torch.Tensor], text_order: Sequence[str] = ("title", "abstract"),
                  method: str = "ini", p_drop: float = 0.0, p_mask: float = 0.0, seed: int = 0) -> dict:
    B = int(batch.get("batch_size", int(batch["batch_hist"].max()) + 1))
    nh = batch["x_hist"][text_order[0]].shape[0]
    nc = batch["x_cand"][text_order[0]].shape[0]
    hist_vec = lstur_news_encoder_fwd(batch["x_hist"], params, text_order, p_drop, seed, 0, nh + nc)
    cand_vec = lstur_news_encoder_fwd(batch["x_cand"], params, text_order, p_drop, seed, nh, nh + nc)
    hist_dense, mask_hist = to_dense_batch(hist_vec, batch["batch_hist"], B)
    cand_dense, mask_cand = to_dense_batch(cand_vec, batch["batch_cand"], B)
    hist_size = mask_hist.sum(dim=1)                              # lstur_module.py:287-290
    um = dropout_multiplier(seed, USER_MASK_STREAM, p_mask, (B,)) if p_mask > 0.0 else None
    user = lstur_user_encoder_fwd(batch["user_idx"], hist_dense, hist_size, params, method, um)
    scores = click_scores(user, cand_dense)
    y_true, _ = to_dense_batch(batch["labels"], batch["batch_cand"], B)
    loss = ce_loss(scores, y_true)
    return dict(hist_vec=hist_vec, cand_vec=cand_vec, hist_dense=hist_dense, cand_dense=cand_dense,
                user_vec=user, scores=scores, y_true=y_true, loss=loss, mask_hist=mask_hist,
                mask_cand=mask_cand, hist_size=hist_size)


def unique_params(params: Dict[str, torch.Tensor]) -> List[str]:
    """Keys with one representative per shared storage (the text encoder appears once per attribute)."""
    seen, keys = set(), []
    for k, v in params.items():
        if v.data_ptr() not in seen:
            seen.add(v.data_ptr())
            keys.append(k)
    return keys


def lstur_loss_and_grads(batch, params, **kw):
    keys = unique_params(params)
    leaves = {k: params[k].clone().requires_grad_(True) for k in keys}
    ptr2key = {params[k].data_ptr(): k for k in keys}
    full = {k: leaves[ptr2key[v.data_ptr()]] for k, v in params.items()}
    out = lstur_forward(batch, full, **kw)
    grads = torch.autograd.grad(out["loss"], [leaves[k] for k in keys], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(keys, grads)}
    # padding_idx=0 rows never receive a gradient (nn.Embedding semantics; text.py:147, category.py:58,
    # user/lstur.py:55)
    for k in g:
        if k.endswith("embedding_layer.weight") or k.endswith("long_term_user_embedding.weight"):
            g[k][0] = 0.0
    return out, g
