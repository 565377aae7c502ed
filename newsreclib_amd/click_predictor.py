"""Dot-product click predictor with the reference's interface (layers/click_predictor.py:5-11)."""
import torch
import torch.nn as nn

from . import ops


class DotProduct(nn.Module):
    def __init__(self) -> None:
        super().__init__()

    def forward(self, user_vec: torch.Tensor, cand_news_vector: torch.Tensor) -> torch.Tensor:
        """user_vec (B, 1, D), cand_news_vector (B, D, C) [the reference passes a permuted view of
        (B, C, D), nrms_module.py:251-253] -> (B, C)."""
        cand = cand_news_vector.permute(0, 2, 1)  # back to (B, C, D); contiguous for the usual caller
        return ops.DotScoresFn.apply(user_vec.squeeze(1), cand)


class CrossEntropyLoss(nn.Module):
    """``torch.nn.CrossEntropyLoss()`` for float (probability) targets as used at
    nrms_module.py:287-288, fused with its gradient."""

    def forward(self, scores: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
        return ops.CrossEntropyFn.apply(scores, y_true, 1.0)
