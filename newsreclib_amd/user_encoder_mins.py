"""MINS user encoder with the reference's interface (encoders/user/mins.py) on HIP kernels."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops_blocks, ops_lstur
from .attention import AdditiveAttention

_HEAD_DIMS = (16, 20, 32, 48, 64)          # what the attention kernels are instantiated for


class UserEncoder(nn.Module):
    """``nn.MultiheadAttention`` (``num_gru_channels`` heads) over the clicked-news vectors, the output chunked
    into ``num_gru_channels`` channels, ONE shared ``nn.GRU`` run over every channel's packed history, the last
    hidden states concatenated, additive attention over the resulting length-1 sequence (user/mins.py:53-86).

    As in the reference: the attention is seq-first and sees (B, H, D), so it runs ACROSS THE USERS of the batch
    for each history slot; all channels share one GRU (``multi_channel_gru`` repeats the same module), so the
    channel loop is one GRU call over B * channels sequences; the final additive attention pools a single
    element (softmax over one logit = 1), which returns the concatenation unchanged and leaves its own
    parameters without gradient.

    Shapes the kernels are not built for are zero-padded on the way in (plumbing, exact): the reference config
    (300 / 6 channels) has head dim 50, padded to 64 per head with zero weight rows/columns -- q k^T, the
    softmax and p v are unchanged by zero features, and the q scale stays 1/sqrt(50); the 50-wide GRU is padded
    to 52 units with zero weights, whose extra units stay exactly 0."""

    def __init__(self, news_embed_dim: int, query_dim: int, num_filters: int, num_gru_channels: int) -> None:
        super().__init__()
        if not isinstance(num_gru_channels, int):
            raise ValueError(
                f"Expected keyword argument `num_gru_channels` to be an `int` but got {num_gru_channels}")
        assert num_filters % num_gru_channels == 0
        self.num_gru_channels = num_gru_channels
        self.multihead_attention = nn.MultiheadAttention(embed_dim=news_embed_dim, num_heads=num_gru_channels)
        self.additive_attention = AdditiveAttention(input_dim=news_embed_dim, query_dim=query_dim)
        self.gru = nn.GRU(int(num_filters / num_gru_channels), int(num_filters / num_gru_channels))
        self.multi_channel_gru = nn.ModuleList([self.gru for _ in range(num_gru_channels)])

    def _padded_attention_params(self):
        mha, C = self.multihead_attention, self.num_gru_channels
        D = mha.embed_dim
        dh = D // C
        dhp = next((d for d in _HEAD_DIMS if d >= dh), None)
        if dhp is None:
            raise NotImplementedError(f"head dim {dh} > 64 is not built")
        w_in, b_in, w_o, b_o = mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias
        if dhp == dh:
            return w_in, b_in, w_o, b_o, D, None
        Dp = C * dhp
        col = (torch.arange(C, device=w_in.device).repeat_interleave(dh) * dhp
               + torch.arange(dh, device=w_in.device).repeat(C))                      # feature -> padded slot
        rows = torch.cat([col, col + Dp, col + 2 * Dp])
        # the attention runs in the padded feature space: in-projection (3 Dp, Dp) over inputs padded to Dp
        # columns (extra columns zero), out-projection (Dp, Dp) whose first D rows are the real ones
        w_in_p = w_in.new_zeros(3 * Dp, Dp).index_put((rows[:, None], torch.arange(D, device=w_in.device)[None, :]), w_in)
        b_in_p = b_in.new_zeros(3 * Dp).index_copy(0, rows, b_in)
        w_o_p = w_o.new_zeros(Dp, Dp).index_put((torch.arange(D, device=w_o.device)[:, None], col[None, :]), w_o)
        b_o_p = F.pad(b_o, (0, Dp - D))
        return w_in_p, b_in_p, w_o_p, b_o_p, Dp, 1.0 / math.sqrt(dh)

    def _padded_gru_params(self):
        g = self.gru
        hd = g.hidden_size
        hp = (hd + 3) // 4 * 4
        params = (g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        if hp == hd:
            return (*params, hd)
        dev = params[0].device
        rows = (torch.arange(3, device=dev).repeat_interleave(hd) * hp + torch.arange(hd, device=dev).repeat(3))
        cols = torch.arange(hd, device=dev)
        pad_w = lambda w: w.new_zeros(3 * hp, hp).index_put((rows[:, None], cols[None, :]), w)  # noqa: E731
        pad_b = lambda b: b.new_zeros(3 * hp).index_copy(0, rows, b)  # noqa: E731
        return pad_w(params[0]), pad_w(params[1]), pad_b(params[2]), pad_b(params[3]), hp

    def forward(self, hist_news_vector: torch.Tensor, hist_size: torch.Tensor) -> torch.Tensor:
        B, H, D = hist_news_vector.shape
        C = self.num_gru_channels
        w_in, b_in, w_o, b_o, Dp, scale = self._padded_attention_params()
        x = hist_news_vector if Dp == D else F.pad(hist_news_vector, (0, Dp - D))
        y = ops_blocks.MhaFn.apply(x.contiguous(), w_in, b_in, w_o, b_o, C, scale, None)[..., :D]   # mins.py:55-57
        dc = D // C
        w_ih, w_hh, b_ih, b_hh, hp = self._padded_gru_params()
        # (B, H, C, dc) -> (B * C, H, dc): every channel of every user is one sequence of the shared GRU
        ch = y.reshape(B, H, C, dc).permute(0, 2, 1, 3).reshape(B * C, H, dc)
        if hp != dc:
            ch = F.pad(ch, (0, hp - dc))
        lengths = hist_size.to(device=ch.device, dtype=torch.int64).repeat_interleave(C)
        last = ops_lstur.GruFn.apply(ch.contiguous(), lengths, None, w_ih, w_hh, b_ih, b_hh, None)   # mins.py:63-76
        multi_channel_vector = last[:, :dc].reshape(B, 1, D)                                          # mins.py:79
        return self.additive_attention(multi_channel_vector)                                          # mins.py:82
