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


class SupConLoss(nn.Module):
    """The reference's ``SupConLoss`` (components/losses.py:6-40) applied to the score matrix as at
    nrms_module.py:289-318, fused with its gradient.  ``temperature`` defaults to 0.1 as in the reference, which
    builds the loss without arguments (abstract_recommender.py:117-120)."""

    def __init__(self, temperature: float = 0.1) -> None:
        super().__init__()
        self.temperature = temperature

    def forward(self, scores: torch.Tensor, y_true: torch.Tensor, cand_sizes: torch.Tensor) -> torch.Tensor:
        return ops.SupConFn.apply(scores, y_true, cand_sizes, self.temperature)
