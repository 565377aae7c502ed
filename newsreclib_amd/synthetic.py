"""Synthetic MIND-shaped ``RecommendationBatch`` generator.

Reproduces the *layout* of the reference's collated batch (``data/components/batch.py:6-32``,
``rec_dataset.py:148-168,289-293``): ragged, concatenated news rows plus sorted assignment vectors.
There is no network, so MIND itself is unavailable; shapes and distributions follow the reference
configs (``configs/data/mind_rec.yaml:55-58``: title 30 tokens, history <= 50, 4 negatives per
positive).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch


def _zipf_ids(rng: np.random.Generator, n: int, vocab: int, a: float = 1.1) -> np.ndarray:
    """n token ids in [1, vocab) with P(k) ~ k**-a (bounded Zipf; id 0 is the pad token)."""
    ranks = np.arange(1, vocab, dtype=np.float64)
    cdf = np.cumsum(ranks ** (-a))
    cdf /= cdf[-1]
    return (np.searchsorted(cdf, rng.random(n), side="left") + 1).astype(np.int64)


def _titles(rng: np.random.Generator, n_news: int, vocab: int, L: int) -> np.ndarray:
    """(n_news, L) int64: ``len ~ clip(round(N(11.5, 3.5)), 3, L)`` real tokens, then 0-padding
    (right padding as ``rec_dataset.py:170-178``)."""
    lens = np.clip(np.rint(rng.normal(11.5, 3.5, n_news)), 3, L).astype(np.int64)
    ids = _zipf_ids(rng, n_news * L, vocab).reshape(n_news, L)
    ids[np.arange(L)[None, :] >= lens[:, None]] = 0
    return ids


def make_batch(batch_size: int, vocab: int = 70_000, mode: str = "fixed", seed: int = 1234,
               L: int = 30, H: int = 50, neg_ratio: int = 4, device="cpu") -> Dict:
    """One synthetic training batch.

    mode="fixed":  every user has H clicks and (1 + neg_ratio) candidates, one positive at a
                   random slot (the headline shape the FLOP/byte math in DESIGN.md uses).
    mode="ragged": hist_i ~ clip(round(lognormal(3.0, 0.8)), 1, H); npos_i ~ 1 + Poisson(0.35);
                   C_i = (1 + neg_ratio) * npos_i with shuffled labels (``rec_dataset.py:60-95``).
    """
    rng = np.random.default_rng(seed)
    B = batch_size
    if mode == "fixed":
        hist_sizes = np.full(B, H, dtype=np.int64)
        npos = np.ones(B, dtype=np.int64)
    elif mode == "ragged":
        hist_sizes = np.clip(np.rint(rng.lognormal(3.0, 0.8, B)), 1, H).astype(np.int64)
        npos = 1 + rng.poisson(0.35, B).astype(np.int64)
    else:
        raise ValueError(f"unknown mode {mode!r}")
    cand_sizes = npos * (1 + neg_ratio)
    labels = []
    for b in range(B):
        lab = np.zeros(cand_sizes[b], dtype=np.float32)
        lab[: npos[b]] = 1.0
        rng.shuffle(lab)
        labels.append(lab)
    n_hist, n_cand = int(hist_sizes.sum()), int(cand_sizes.sum())
    batch = {
        "batch_hist": np.repeat(np.arange(B, dtype=np.int64), hist_sizes),
        "batch_cand": np.repeat(np.arange(B, dtype=np.int64), cand_sizes),
        "x_hist": {"title": _titles(rng, n_hist, vocab, L)},
        "x_cand": {"title": _titles(rng, n_cand, vocab, L)},
        "labels": np.concatenate(labels),
        "user_ids": np.arange(B, dtype=np.int64) + 1,
        "user_idx": np.arange(B, dtype=np.int64),
    }
    return batch_to_torch(batch, B, device)


def batch_to_torch(batch: Dict, batch_size: int, device="cpu") -> Dict:
    def cv(a):
        return torch.as_tensor(a).to(device)

    out = {}
    for k, v in batch.items():
        out[k] = {kk: cv(vv) for kk, vv in v.items()} if isinstance(v, dict) else cv(v)
    out["batch_size"] = int(batch_size)
    return out


def batch_from_sizes(hist_sizes, cand_sizes, labels, vocab: int, seed: int, L: int = 30) -> Dict:
    """Explicit ragged batch (used for the hand-made golden cases)."""
    rng = np.random.default_rng(seed)
    hist_sizes = np.asarray(hist_sizes, dtype=np.int64)
    cand_sizes = np.asarray(cand_sizes, dtype=np.int64)
    B = len(hist_sizes)
    batch = {
        "batch_hist": np.repeat(np.arange(B, dtype=np.int64), hist_sizes),
        "batch_cand": np.repeat(np.arange(B, dtype=np.int64), cand_sizes),
        "x_hist": {"title": _titles(rng, int(hist_sizes.sum()), vocab, L)},
        "x_cand": {"title": _titles(rng, int(cand_sizes.sum()), vocab, L)},
        "labels": np.asarray(labels, dtype=np.float32),
        "user_ids": np.arange(B, dtype=np.int64) + 1,
        "user_idx": np.arange(B, dtype=np.int64),
    }
    return batch_to_torch(batch, B)


def add_lstur_fields(batch: Dict, vocab: int, n_categ: int = 19, n_users: int = 1000, L_abstract: int = 50,
                     seed: int = 4321) -> Dict:
    """Adds what ``LSTURModule.forward`` reads on top of the NRMS batch (``lstur_module.py:278-303``):
    abstract token ids and category ids per news row (``rec_dataset.py:148-168``), and ``user_idx``
    drawn from [0, n_users) -- 0 is the reference's "unknown user" / padding row."""
    rng = np.random.default_rng(seed)
    dev = batch["labels"].device
    for part in ("x_hist", "x_cand"):
        n = batch[part]["title"].shape[0]
        lens = np.clip(np.rint(rng.normal(32.0, 12.0, n)), 5, L_abstract).astype(np.int64)
        ids = _zipf_ids(rng, n * L_abstract, vocab).reshape(n, L_abstract)
        ids[np.arange(L_abstract)[None, :] >= lens[:, None]] = 0
        batch[part]["abstract"] = torch.as_tensor(ids).to(dev)
        batch[part]["category"] = torch.as_tensor(rng.integers(1, n_categ, n)).to(dev)
    B = batch["batch_size"]
    uidx = rng.integers(1, n_users, B)
    uidx[rng.random(B) < 0.15] = 0
    batch["user_idx"] = torch.as_tensor(uidx).to(dev)
    return batch
