"""NRMS user encoder with the reference's interface (encoders/user/nrms.py:7-41) on HIP kernels."""
import torch
import torch.nn as nn

from . import ops
from .attention import AdditiveAttention
from .news_encoder import _grad_bufs


class UserEncoder(nn.Module):
    """Multi-head self-attention + additive attention over clicked-news vectors.

    Reproduces the reference's call of a seq-first ``nn.MultiheadAttention`` on a (B, H, D) tensor
    (user/nrms.py:34-36): attention runs ACROSS THE USERS of the batch for each history slot, so a
    user's vector depends on the batch composition (SURVEY.md headline fact 3).  Zero-padded history
    slots take part in every softmax (no masks anywhere on the reference path)."""

    def __init__(self, news_embed_dim: int, num_heads: int, query_dim: int) -> None:
        super().__init__()
        self.multihead_attention = nn.MultiheadAttention(embed_dim=news_embed_dim, num_heads=num_heads)
        self.additive_attention = AdditiveAttention(input_dim=news_embed_dim, query_dim=query_dim)
        self.num_heads = num_heads

    def forward(self, hist_news_vector: torch.Tensor) -> torch.Tensor:
        mha, att = self.multihead_attention, self.additive_attention
        params = (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                  att.linear.weight, att.linear.bias, att.query)
        return ops.UserEncoderFn.apply(hist_news_vector, *params, self.num_heads, _grad_bufs(params))
