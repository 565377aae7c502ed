"""CenNewsRec user encoder with the reference's interface (encoders/user/cen_news_rec.py) on HIP kernels."""
from typing import Optional

import torch
import torch.nn as nn

from . import ops, ops_lstur
from .attention import AdditiveAttention
from .news_encoder import _draw_seed, _grad_bufs


class UserEncoder(nn.Module):
    """Long-term vector: ``nn.MultiheadAttention`` over the clicked-news vectors -> dropout -> additive attention;
    short-term vector: last hidden state of a GRU over the ``num_recent_news`` trailing history slots; the two are
    stacked and combined by a final additive attention (cen_news_rec.py:62-88).

    Reproduced as they are in the reference: the multi-head attention is seq-first and is fed (B, H, F), so it
    runs ACROSS THE USERS of the batch for each history slot (the NRMS quirk); ``hist[:, -num_recent_news:]``
    takes the TRAILING slots of the zero-padded dense history, i.e. padding for users with short histories, and
    the GRU runs over all of them (no packing)."""

    def __init__(self, num_filters: int, num_heads: int, query_dim: int, gru_hidden_dim: int, num_recent_news: int,
                 dropout_probability: float) -> None:
        super().__init__()
        if not isinstance(num_recent_news, int):
            raise ValueError(f"Expected keyword argument `num_recent_news` to be an `int` but got {num_recent_news}")
        if not isinstance(dropout_probability, float):
            raise ValueError(
                f"Expected keyword argument `dropout_probability` to be a `float` but got {dropout_probability}")
        self.num_recent_news = num_recent_news
        self.multihead_attention = nn.MultiheadAttention(embed_dim=num_filters, num_heads=num_heads)
        self.additive_attention = AdditiveAttention(input_dim=num_filters, query_dim=query_dim)
        self.gru = nn.GRU(input_size=num_filters, hidden_size=gru_hidden_dim, batch_first=True)
        self.final_additive_attention = AdditiveAttention(input_dim=gru_hidden_dim, query_dim=query_dim)
        self.dropout = nn.Dropout(p=dropout_probability)
        self.num_heads = num_heads

    def forward(self, hist_news_vector: torch.Tensor, seed: Optional[int] = None) -> torch.Tensor:
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 and seed is None:
            seed = _draw_seed()
        mha, att = self.multihead_attention, self.additive_attention
        block = (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, att.linear.weight,
                 att.linear.bias, att.query)
        # dropout only AFTER the attention (stream 9), none on the input
        longterm = ops.UserEncoderFn.apply(hist_news_vector, *block, self.num_heads, _grad_bufs(block), p, seed or 0,
                                           False, 8)
        recent = hist_news_vector[:, -self.num_recent_news:, :].contiguous()
        B, R, _ = recent.shape
        g = self.gru
        gp = (g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        lengths = torch.full((B,), R, dtype=torch.int64, device=recent.device)
        shortterm = ops_lstur.GruFn.apply(recent, lengths, None, *gp, _grad_bufs(gp))
        return self.final_additive_attention(torch.stack([shortterm, longterm], dim=1))
