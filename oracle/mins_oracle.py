"""CPU restatement of the reference MINS forward / loss (TEST INFRASTRUCTURE -- see oracle/__init__.py).

``MINSModule`` (mins_module.py:250-279): ``MHSAAddAtt`` text encoder shared by title and abstract
(``text.py:179-236``), the category view ``relu(Linear(embedding))`` (``category.py:72-82``), additive-attention
view combination (``news.py:118-121,164-165``), the multi-channel GRU user encoder (``user/mins.py:53-86``),
dot-product scorer, CE loss.  Elementary fp32 torch ops on CPU; pinned by tests/golden/make_golden_mins.py
against the imported reference components.

Dropout streams by attribute name as everywhere (title (0, 1), abstract (2, 3)): first = post-embedding over
(N, L, D), second = post-attention over (N, L, D) (the reference holds (L, N, D) there, ``text.py:229-230``)."""
from __future__ import annotations

from typing import Sequence

import torch

from .lstur_oracle import TEXT_PREFIX, TEXT_STREAMS, gru_last_hidden, unique_params
from .naml_oracle import CATEG_PREFIX, COMBINE_PREFIX
from .nrms_oracle import (_mhsa_seq_first, additive_attention, ce_loss, click_scores, dropout_multiplier,
                          to_dense_batch)

USER = "user_encoder."
MHA_KEYS = ("multihead_attention.in_proj_weight", "multihead_attention.in_proj_bias",
            "multihead_attention.out_proj.weight", "multihead_attention.out_proj.bias")
ATT_KEYS = ("additive_attention.linear.weight", "additive_attention.linear.bias", "additive_attention.query")
GRU_KEYS = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")


def make_mins_params(vocab: int, n_categ: int, embed_dim: int = 300, query_dim: int = 200, categ_dim: int = 100,
                     channels: int = 6, text_attrs: Sequence[str] = ("title", "abstract"), seed: int = 0):
    """Seeded parameters under the reference's state_dict keys.  The user encoder's ONE GRU is registered both as
    ``gru`` and as every entry of ``multi_channel_gru`` (mins.py:48-49): same storage under all those keys."""
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale):
        return (torch.randn(*shape, generator=g) * scale).float()

    D, Q = embed_dim, query_dim

    def block(dim):
        return {MHA_KEYS[0]: rnd(3 * dim, dim, scale=dim ** -0.5), MHA_KEYS[1]: rnd(3 * dim, scale=0.05),
                MHA_KEYS[2]: rnd(dim, dim, scale=dim ** -0.5), MHA_KEYS[3]: rnd(dim, scale=0.05),
                ATT_KEYS[0]: rnd(Q, dim, scale=dim ** -0.5), ATT_KEYS[1]: rnd(Q, scale=0.05), ATT_KEYS[2]: rnd(Q, scale=0.1)}

    shared = {"embedding_layer.weight": rnd(vocab, D, scale=0.3), **block(D)}
    params = {}
    for a in text_attrs:
        for k, v in shared.items():
            params[TEXT_PREFIX.format(a) + k] = v
    params[CATEG_PREFIX + "embedding_layer.weight"] = rnd(n_categ, categ_dim, scale=0.3)
    params[CATEG_PREFIX + "linear.weight"] = rnd(D, categ_dim, scale=categ_dim ** -0.5)
    params[CATEG_PREFIX + "linear.bias"] = rnd(D, scale=0.05)
    params[COMBINE_PREFIX + "linear.weight"] = rnd(Q, D, scale=D ** -0.5)
    params[COMBINE_PREFIX + "linear.bias"] = rnd(Q, scale=0.05)
    params[COMBINE_PREFIX + "query"] = rnd(Q, scale=0.1)
    for k, v in block(D).items():
        params[USER + k] = v
    dc = D // channels
    gru = {"weight_ih_l0": rnd(3 * dc, dc, scale=dc ** -0.5), "weight_hh_l0": rnd(3 * dc, dc, scale=dc ** -0.5),
           "bias_ih_l0": rnd(3 * dc, scale=0.05), "bias_hh_l0": rnd(3 * dc, scale=0.05)}
    for pre in ["gru."] + [f"multi_channel_gru.{i}." for i in range(channels)]:
        for k, v in gru.items():
            params[USER + pre + k] = v
    return params


def mhsa_text_encoder_fwd(ids, params, prefix, num_heads, m1=None, m2=None):
    """``MHSAAddAtt.forward`` (text.py:222-236) with an explicit key prefix."""
    x = params[prefix + "embedding_layer.weight"][ids]
    if m1 is not None:
        x = x * m1
    y = _mhsa_seq_first(x.permute(1, 0, 2), *[params[prefix + k] for k in MHA_KEYS], num_heads).permute(1, 0, 2)
    if m2 is not None:
        y = y * m2
    return additive_attention(y, *[params[prefix + k] for k in ATT_KEYS])


def mins_news_encoder_fwd(x, params, text_order, num_heads, p_drop, seed, row_offset=0, total_rows=None):
    vecs = []
    for a in text_order:
        ids = x[a]
        pre = TEXT_PREFIX.format(a)
        m1 = m2 = None
        if p_drop > 0.0:
            n, L = ids.shape
            tot = total_rows if total_rows is not None else n
            D = params[pre + "embedding_layer.weight"].shape[1]
            s1, s2 = TEXT_STREAMS[a]
            m1 = dropout_multiplier(seed, s1, p_drop, (tot, L, D))[row_offset:row_offset + n]
            m2 = dropout_multiplier(seed, s2, p_drop, (tot, L, D))[row_offset:row_offset + n]
        vecs.append(mhsa_text_encoder_fwd(ids, params, pre, num_heads, m1, m2))
    c = params[CATEG_PREFIX + "embedding_layer.weight"][x["category"]]              # category.py:73
    vecs.append(torch.relu(c @ params[CATEG_PREFIX + "linear.weight"].t() + params[CATEG_PREFIX + "linear.bias"]))
    stacked = torch.stack(vecs, dim=1)                                              # news.py:165
    return additive_attention(stacked, params[COMBINE_PREFIX + "linear.weight"], params[COMBINE_PREFIX + "linear.bias"],
                              params[COMBINE_PREFIX + "query"])


def mins_user_encoder_fwd(hist: torch.Tensor, hist_size: torch.Tensor, params, channels: int) -> torch.Tensor:
    """``UserEncoder.forward`` (user/mins.py:53-86): hist (B, H, D) -> (B, D).  Seq-first MHA with ``channels``
    heads over (B, H, D) = attention ACROSS USERS per history slot; the output's feature axis is cut into
    ``channels`` chunks, each run through the SAME GRU over the user's valid history (packed), the last hidden
    states concatenated; the closing additive attention pools ONE element, i.e. multiplies it by softmax(.) = 1."""
    y = _mhsa_seq_first(hist, *[params[USER + k] for k in MHA_KEYS], channels)     # mins.py:55-57
    B, H, D = y.shape
    dc = D // channels
    gru = [params[USER + "gru." + k] for k in GRU_KEYS]
    h0 = torch.zeros(B, dc, dtype=hist.dtype)
    last = [gru_last_hidden(y[:, :, c * dc:(c + 1) * dc], hist_size, h0, *gru) for c in range(channels)]   # :60-76
    multi = torch.cat(last, dim=1).unsqueeze(1)                                      # :79  (B, 1, D)
    return additive_attention(multi, *[params[USER + k] for k in ATT_KEYS])          # :82


def mins_forward(batch, params, text_order=("title", "abstract"), num_heads: int = 15, channels: int = 6,
                 p_drop: float = 0.0, seed: int = 0) -> dict:
    B = int(batch.get("batch_size", int(batch["batch_hist"].max()) + 1))
    nh = batch["x_hist"][text_order[0]].shape[0]
    nc = batch["x_cand"][text_order[0]].shape[0]
    hist_vec = mins_news_encoder_fwd(batch["x_hist"], params, text_order, num_heads, p_drop, seed, 0, nh + nc)
    cand_vec = mins_news_encoder_fwd(batch["x_cand"], params, text_order, num_heads, p_drop, seed, nh, nh + nc)
    hist_dense, mask_hist = to_dense_batch(hist_vec, batch["batch_hist"], B)
    cand_dense, _ = to_dense_batch(cand_vec, batch["batch_cand"], B)
    user = mins_user_encoder_fwd(hist_dense, mask_hist.sum(dim=1), params, channels)   # mins_module.py:261-267
    scores = click_scores(user, cand_dense)
    y_true, _ = to_dense_batch(batch["labels"], batch["batch_cand"], B)
    return dict(hist_vec=hist_vec, cand_vec=cand_vec, user_vec=user, scores=scores, y_true=y_true,
                loss=ce_loss(scores, y_true))


def mins_loss_and_grads(batch, params, **kw):
    keys = unique_params(params)
    leaves = {k: params[k].clone().requires_grad_(True) for k in keys}
    ptr2key = {params[k].data_ptr(): k for k in keys}
    full = {k: leaves[ptr2key[v.data_ptr()]] for k, v in params.items()}
    out = mins_forward(batch, full, **kw)
    grads = torch.autograd.grad(out["loss"], [leaves[k] for k in keys], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(keys, grads)}
    for k in g:
        if k.endswith("embedding_layer.weight"):
            g[k][0] = 0.0                                                           # padding_idx = 0
    return out, g
