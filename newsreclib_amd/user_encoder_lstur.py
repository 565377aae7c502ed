"""LSTUR user encoder with the reference's interface (encoders/user/lstur.py:6-87) on HIP kernels."""
from typing import Optional

import torch
import torch.nn as nn

from . import ops_lstur
from .news_encoder import _draw_seed, _grad_bufs

USER_MASK_STREAM = 8


class UserEncoder(nn.Module):
    """Long-term user embedding (randomly masked per user) + GRU over the clicked-news vectors.

    ``long_short_term_method="ini"``: the GRU is initialised with the long-term vector and its last hidden
    state is the user vector (lstur.py:81-83); ``"con"``: GRU from zeros, concatenated with the long-term
    vector (lstur.py:85-87).  ``nn.Dropout2d`` on the (1, B, D) long-term tensor (lstur.py:58,71) zeroes WHOLE
    users with probability ``user_masking_probability``; the kernels draw that mask (stream 8, one draw per
    batch position).  Like ``pack_padded_sequence`` (lstur.py:74-79), empty histories are rejected."""

    def __init__(self, num_users: int, input_dim: int, user_masking_probability: float,
                 long_short_term_method: str) -> None:
        super().__init__()
        if not isinstance(num_users, int):
            raise ValueError(f"Expected keyword argument `num_users` to be an `int` but got {num_users}")
        if not isinstance(input_dim, int):
            raise ValueError(f"Expected keyword argument `input_dim` to be an `int` but got {input_dim}")
        if not isinstance(user_masking_probability, float):
            raise ValueError("Expected keyword argument `user_masking_probability` to be a `float` but got "
                             f"{user_masking_probability}")
        if not isinstance(long_short_term_method, str):
            raise ValueError("Expected keyword argument `long_short_term_method` to be a `str` but got "
                             f"{long_short_term_method}")
        assert long_short_term_method in ["ini", "con"]
        self.long_short_term_method = long_short_term_method
        hidden = input_dim if long_short_term_method == "ini" else int(input_dim * 0.5)
        self.long_term_user_embedding = nn.Embedding(num_embeddings=num_users, embedding_dim=hidden, padding_idx=0)
        self.dropout = nn.Dropout2d(p=user_masking_probability)   # holds p; the kernels draw the mask
        self.gru = nn.GRU(input_dim, hidden)                      # parameter container (names / init)

    def forward(self, user: torch.Tensor, hist_news_vector: torch.Tensor, hist_size: torch.Tensor,
                seed: Optional[int] = None, min_hist_size: Optional[int] = None) -> torch.Tensor:
        if min_hist_size is None:
            min_hist_size = int(hist_size.min())      # one sync; prepare_batch supplies it for free
        if min_hist_size < 1:
            raise RuntimeError("Length of all samples has to be greater than 0, but found an element in "
                               "'lengths' that is <= 0")
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 and seed is None:
            seed = _draw_seed()
        table = self.long_term_user_embedding.weight
        user_vector = ops_lstur.EmbeddingRowsFn.apply(user, table, p, seed or 0, USER_MASK_STREAM,
                                                      _grad_bufs((table,)))
        g = self.gru
        params = (g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        lengths = hist_size.to(torch.int64)
        if self.long_short_term_method == "ini":
            return ops_lstur.GruFn.apply(hist_news_vector, lengths, user_vector, *params, _grad_bufs(params))
        last_hidden = ops_lstur.GruFn.apply(hist_news_vector, lengths, None, *params, _grad_bufs(params))
        return torch.cat((last_hidden, user_vector), dim=1)
