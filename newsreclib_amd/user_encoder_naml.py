"""NAML user encoder with the reference's interface (encoders/user/naml.py:7-34) on HIP kernels."""
import torch
import torch.nn as nn

from .attention import AdditiveAttention


class UserEncoder(nn.Module):
    """Additive attention over the clicked-news vectors: (B, H, D) -> (B, D).  As everywhere on the reference
    path there is no mask: zero-padded history slots take part in the softmax."""

    def __init__(self, news_embed_dim: int, query_dim: int) -> None:
        super().__init__()
        self.additive_attention = AdditiveAttention(input_dim=news_embed_dim, query_dim=query_dim)

    def forward(self, hist_news_vector: torch.Tensor) -> torch.Tensor:
        return self.additive_attention(hist_news_vector)
