"""CPU restatement of the reference CenNewsRec train step (TEST INFRASTRUCTURE -- see oracle/__init__.py).

SURVEY.md section 8f row 4: ``CenNewsRecModule`` = ``CNNMHSAAddAtt`` title encoder (``text.py:239-309``: embedding
-> dropout -> Conv1d over tokens -> ReLU -> dropout -> seq-first MHA over the tokens of each news -> dropout ->
additive attention), the long/short-term user encoder (``user/cen_news_rec.py:62-88``), dot-product scorer and CE
loss (``cen_news_rec_module.py:236-267``).

Written with elementary fp32 torch ops on CPU (matmul / softmax / sigmoid / tanh), not ``nn.Conv1d`` /
``nn.MultiheadAttention`` / ``nn.GRU``; ``tests/golden/make_golden_cen_news_rec.py`` pins it against the imported
reference components.

Dropout (product-defined counter-based masks, nrms_oracle.dropout_keep_mask): the title encoder uses streams
0 (post-embedding, flat index over (N, L, D)), 1 (post-ReLU, over (N, L, F); the reference holds (N, F, L) at
that point, ``text.py:297-299``, elementwise so only the index convention differs) and 2 (post-attention, over
(N, L, F); the reference holds (L, N, F), ``text.py:303-304``).  The user encoder drops only the attention
output (``cen_news_rec.py:66-67``): stream 9, flat index over (B, H, F).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .lstur_oracle import gru_last_hidden
from .nrms_oracle import (_mhsa_seq_first, additive_attention, ce_loss, click_scores, dropout_multiplier,
                          to_dense_batch)

TEXT_STREAMS = (0, 1, 2)
USER_STREAM = 9
TEXT = "news_encoder.text_encoders.title."
USER = "user_encoder."
MHA_KEYS = ("multihead_attention.in_proj_weight", "multihead_attention.in_proj_bias",
            "multihead_attention.out_proj.weight", "multihead_attention.out_proj.bias")
ATT_KEYS = ("linear.weight", "linear.bias", "query")
GRU_KEYS = ("gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0")


def make_cen_news_rec_params(vocab: int, embed_dim: int = 300, num_filters: int = 400, window: int = 3,
                             query_dim: int = 200, late_fusion: bool = False, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded parameters under the reference's state_dict keys (``cnn.weight`` in ``nn.Conv1d``'s (F, D, W))."""
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, scale):
        return (torch.randn(*shape, generator=g) * scale).float()

    D, F, W, Q = embed_dim, num_filters, window, query_dim

    def block(prefix, dim):
        return {
            prefix + "multihead_attention.in_proj_weight": rnd(3 * dim, dim, scale=dim ** -0.5),
            prefix + "multihead_attention.in_proj_bias": rnd(3 * dim, scale=0.05),
            prefix + "multihead_attention.out_proj.weight": rnd(dim, dim, scale=dim ** -0.5),
            prefix + "multihead_attention.out_proj.bias": rnd(dim, scale=0.05),
            prefix + "additive_attention.linear.weight": rnd(Q, dim, scale=dim ** -0.5),
            prefix + "additive_attention.linear.bias": rnd(Q, scale=0.05),
            prefix + "additive_attention.query": rnd(Q, scale=0.1),
        }

    p = {TEXT + "embedding_layer.weight": rnd(vocab, D, scale=0.3),
         TEXT + "cnn.weight": rnd(F, D, W, scale=(W * D) ** -0.5),
         TEXT + "cnn.bias": rnd(F, scale=0.05)}
    p.update(block(TEXT, F))
    if not late_fusion:
        p.update(block(USER, F))
        p[USER + "gru.weight_ih_l0"] = rnd(3 * F, F, scale=F ** -0.5)
        p[USER + "gru.weight_hh_l0"] = rnd(3 * F, F, scale=F ** -0.5)
        p[USER + "gru.bias_ih_l0"] = rnd(3 * F, scale=0.05)
        p[USER + "gru.bias_hh_l0"] = rnd(3 * F, scale=0.05)
        p[USER + "final_additive_attention.linear.weight"] = rnd(Q, F, scale=F ** -0.5)
        p[USER + "final_additive_attention.linear.bias"] = rnd(Q, scale=0.05)
        p[USER + "final_additive_attention.query"] = rnd(Q, scale=0.1)
    return p


# --------------------------------------------------------------------------------------
# CNNMHSAAddAtt.forward (text.py:291-309)
# --------------------------------------------------------------------------------------
def conv1d_tokens(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``nn.Conv1d(D, F, W, padding=1)`` over the token axis, channels-last: x (N, L, D), w (F, D, W) -> the
    PRE-activation (N, L_out, F).  For every output token l the dot product of input tokens [l - 1, l - 1 + W)
    (zero outside the news) with the filter taps, + bias; L_out = L + 2 - W + 1 (== L for the reference's W = 3)."""
    F_, D, W = w.shape
    N, L, _ = x.shape
    xp = torch.zeros(N, L + 2, D, dtype=x.dtype)                    # padding=1 whatever W is (text.py:283)
    xp[:, 1:1 + L] = x
    l_out = L + 2 - W + 1
    win = torch.cat([xp[:, t:t + l_out] for t in range(W)], dim=2)  # (N, l_out, W*D), k = t*D + d
    return win @ w.permute(0, 2, 1).reshape(F_, W * D).t() + b


def cnn_mhsa_text_encoder_fwd(ids: torch.Tensor, params: Dict[str, torch.Tensor], num_heads: int,
                              mults=(None, None, None), prefix: str = TEXT) -> torch.Tensor:
    """ids (N, L) -> (N, F)."""
    m1, m2, m3 = mults
    x = params[prefix + "embedding_layer.weight"][ids]            # text.py:293
    if m1 is not None:
        x = x * m1                                                  # text.py:294
    c = conv1d_tokens(x, params[prefix + "cnn.weight"], params[prefix + "cnn.bias"])    # text.py:297
    c = torch.relu(c)                                               # text.py:298
    if m2 is not None:
        c = c * m2                                                  # text.py:299
    y = _mhsa_seq_first(c.permute(1, 0, 2), *[params[prefix + k] for k in MHA_KEYS], num_heads).permute(1, 0, 2)
    if m3 is not None:
        y = y * m3                                                  # text.py:304
    return additive_attention(y, *[params[prefix + "additive_attention." + k] for k in ATT_KEYS])   # text.py:307


# --------------------------------------------------------------------------------------
# UserEncoder.forward (user/cen_news_rec.py:62-88)
# --------------------------------------------------------------------------------------
def cen_news_rec_user_encoder_fwd(hist: torch.Tensor, params: Dict[str, torch.Tensor], num_heads: int,
                                  num_recent_news: int, mult: Optional[torch.Tensor] = None) -> torch.Tensor:
    """hist (B, H, F) -> (B, F).  The MHA is seq-first and sees (B, H, F): attention ACROSS THE B USERS per
    history slot (same quirk as NRMS).  The recent-news slice is the TRAILING ``num_recent_news`` slots of the
    zero-padded dense history; the GRU runs over all of them from a zero state."""
    P = USER
    y = _mhsa_seq_first(hist, *[params[P + k] for k in MHA_KEYS], num_heads)      # cen_news_rec.py:65-67
    if mult is not None:
        y = y * mult                                                                 # :68
    longterm = additive_attention(y, *[params[P + "additive_attention." + k] for k in ATT_KEYS])   # :71
    recent = hist[:, -num_recent_news:, :]                                           # :75
    B, R, _ = recent.shape
    gru = [params[P + k] for k in GRU_KEYS]
    h0 = torch.zeros(B, gru[1].shape[1], dtype=hist.dtype)
    shortterm = gru_last_hidden(recent, torch.full((B,), R, dtype=torch.int64), h0, *gru)   # :78-81
    stacked = torch.stack([shortterm, longterm], dim=1)                              # :85
    return additive_attention(stacked, *[params[P + "final_additive_attention." + k] for k in ATT_KEYS])   # :88


# --------------------------------------------------------------------------------------
# CenNewsRecModule.forward (cen_news_rec_module.py:236-267) + loss (:297-301)
# --------------------------------------------------------------------------------------
def cen_news_rec_forward(batch: dict, params: Dict[str, torch.Tensor], num_heads: int, num_recent_news: int,
                         late_fusion: bool = False, p_drop: float = 0.0, seed: int = 0) -> dict:
    B = int(batch.get("batch_size", int(batch["batch_hist"].max()) + 1))
    ids_h, ids_c = batch["x_hist"]["title"], batch["x_cand"]["title"]
    nh, nc = ids_h.shape[0], ids_c.shape[0]
    L = ids_h.shape[1]
    D = params[TEXT + "embedding_layer.weight"].shape[1]
    F_ = params[TEXT + "cnn.weight"].shape[0]

    def mults(lo, hi):
        if p_drop <= 0.0:
            return (None, None, None)
        return tuple(dropout_multiplier(seed, s, p_drop, (nh + nc, L, d))[lo:hi]
                     for s, d in zip(TEXT_STREAMS, (D, F_, F_)))

    hist_vec = cnn_mhsa_text_encoder_fwd(ids_h, params, num_heads, mults(0, nh))
    cand_vec = cnn_mhsa_text_encoder_fwd(ids_c, params, num_heads, mults(nh, nh + nc))
    hist_dense, mask_hist = to_dense_batch(hist_vec, batch["batch_hist"], B)
    cand_dense, mask_cand = to_dense_batch(cand_vec, batch["batch_cand"], B)
    if late_fusion:
        user = hist_dense.sum(dim=1) / mask_hist.sum(dim=1).unsqueeze(-1)             # :258-263
    else:
        um = dropout_multiplier(seed, USER_STREAM, p_drop, tuple(hist_dense.shape)) if p_drop > 0.0 else None
        user = cen_news_rec_user_encoder_fwd(hist_dense, params, num_heads, num_recent_news, um)
    scores = click_scores(user, cand_dense)
    y_true, _ = to_dense_batch(batch["labels"], batch["batch_cand"], B)
    loss = ce_loss(scores, y_true)
    return dict(hist_vec=hist_vec, cand_vec=cand_vec, hist_dense=hist_dense, cand_dense=cand_dense, user_vec=user,
                scores=scores, y_true=y_true, loss=loss, mask_hist=mask_hist, mask_cand=mask_cand)


def cen_news_rec_loss_and_grads(batch, params, **kw):
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = cen_news_rec_forward(batch, leaves, **kw)
    keys = list(leaves)
    grads = torch.autograd.grad(out["loss"], [leaves[k] for k in keys], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(keys, grads)}
    g[TEXT + "embedding_layer.weight"][0] = 0.0          # padding_idx=0 (text.py:279)
    return out, g
