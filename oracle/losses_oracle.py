"""CPU restatement of the reference's supervised-contrastive / dual loss (TEST INFRASTRUCTURE -- see
oracle/__init__.py).

The reference's ``SupConLoss`` (``models/components/losses.py:6-40``) subclasses
``pytorch_metric_learning.losses.SupConLoss`` -- third-party, pinned ``pytorch-metric-learning==2.2.0``
(``setup.py:26``), NOT installed here and not vendored, so neither it nor the reference's subclass can be imported:
**parity unpinned** for the library half.  Restated from the library's published source for 2.2.0:

* ``BaseMetricLossFunction.forward`` -> ``compute_loss`` (the reference's override, losses.py:12-18: zero loss when
  every index list has <= 1 entries, else ``loss_method(mat, indices_tuple)`` with mat = the score matrix);
* ``GenericPairLoss.mat_based_loss``: ``pos_mask[a1, p] = 1``, ``neg_mask[a2, n] = 1`` -> ``_compute_loss`` (the
  reference's override, losses.py:20-40, restated line by line below);
* ``loss_and_miner_utils.logsumexp(x, keep_mask, add_one=False, dim=1)``: masked entries filled with the dtype's
  most negative value, ``torch.logsumexp``, rows with nothing kept -> 0;
* ``common_functions.small_val(dtype)`` = ``torch.finfo(dtype).tiny``;
* the default reducer of ``SupConLoss`` is ``AvgNonZeroReducer``: mean over the per-row losses that are > 0
  (0 when there is none).

The index lists are built exactly as ``nrms_module.py:289-304`` does, with the per-user Python loops.
"""
from __future__ import annotations

import torch


def indices_tuple(y_true: torch.Tensor, mask_cand: torch.Tensor):
    """nrms_module.py:290-306."""
    B = mask_cand.shape[0]
    pos_idx = [torch.where(y_true[i])[0] for i in range(B)]
    pos_repeats = torch.tensor([len(pos_idx[i]) for i in range(len(pos_idx))])
    q_p = torch.repeat_interleave(torch.arange(B), pos_repeats)
    p = torch.cat(pos_idx)
    neg_idx = [torch.where(~y_true[i].bool())[0][: len(torch.where(mask_cand[i])[0]) - pos_repeats[i]] for i in range(B)]
    neg_repeats = torch.tensor([len(t) for t in neg_idx])
    q_n = torch.repeat_interleave(torch.arange(B), neg_repeats)
    n = torch.cat(neg_idx)
    return q_p, p, q_n, n


def masked_logsumexp(x: torch.Tensor, keep_mask: torch.Tensor) -> torch.Tensor:
    x = x.masked_fill(~keep_mask, torch.finfo(x.dtype).min)
    out = torch.logsumexp(x, dim=1, keepdim=True)
    return out.masked_fill(~torch.any(keep_mask, dim=1, keepdim=True), 0)


def sup_con_loss(scores: torch.Tensor, y_true: torch.Tensor, mask_cand: torch.Tensor, temperature: float = 0.1):
    """``SupConLoss()(embeddings=scores, labels=None, indices_tuple=..., ref_emb=None, ref_labels=None)``."""
    idx = indices_tuple(y_true, mask_cand)
    zero = scores.sum() * 0
    if all(len(x) <= 1 for x in idx):                                       # losses.py:14-15
        return zero
    a1, p, a2, n = idx
    pos_mask, neg_mask = torch.zeros_like(scores), torch.zeros_like(scores)  # GenericPairLoss.mat_based_loss
    pos_mask[a1, p] = 1
    neg_mask[a2, n] = 1
    if not (pos_mask.bool().any() and neg_mask.bool().any()):               # losses.py:21
        return zero
    mat = scores / temperature                                              # :22
    mat_max, _ = mat.max(dim=1, keepdim=True)
    mat = mat - mat_max.detach()                                            # :23-24
    denominator = masked_logsumexp(mat, (pos_mask + neg_mask).bool())       # :26-28
    log_prob = mat - denominator
    mean_log_prob_pos = (pos_mask * log_prob).sum(dim=1) / (pos_mask.sum(dim=1) + torch.finfo(mat.dtype).tiny)  # :30-32
    losses = -mean_log_prob_pos                                             # :36, reduction "element"
    keep = losses > 0                                                       # AvgNonZeroReducer
    return losses[keep].mean() if int(keep.sum()) >= 1 else zero


def dual_loss(scores, y_true, mask_cand, coef: float, temperature: float = 0.1):
    """nrms_module.py:316-328."""
    ce = -(y_true * torch.log_softmax(scores, dim=1)).sum(dim=1).mean()
    return (1 - coef) * ce + coef * sup_con_loss(scores, y_true, mask_cand, temperature)
