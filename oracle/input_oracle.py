"""CPU restatement of the reference's host input pipeline (TEST INFRASTRUCTURE -- see oracle/__init__.py).

SURVEY.md section 8 rows a1 / f.2: ``RecommendationDatasetTrain.__getitem__`` + ``_sample_candidates``
(``data/components/rec_dataset.py:39-95``), ``RecommendationDatasetTest.__getitem__`` (``:104-118``) and
``DatasetCollate.__call__`` (``:148-293``), written the way the reference does it: pandas ``.loc`` per impression,
``pd.concat``, one padded row per news in a Python loop.

PARITY UNPINNED at this boundary: ``rec_dataset.py`` imports ``mind_dataframe.py`` -> ``omegaconf`` / ``newsreclib.utils``
(lightning, hydra), none of which is installed, so the reference's own classes cannot be run here to produce
golden batches, and the reference holds no test for them.  The restatement is pinned by hand-made known-answer
cases (``tests/test_input_pipeline.py``) instead; the sampling RNG (numpy's global MT19937 in the reference) is a
distributional contract only.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd
import torch


def pad_tokens(text: Sequence[Sequence[int]], max_len: Optional[int]) -> torch.Tensor:
    """``_tokenize_embeddings`` (rec_dataset.py:170-178): right-pad with 0 to ``max_len``; a row LONGER than
    ``max_len`` is truncated (``F.pad`` with a negative amount removes from the end); ``max_len=None`` = the
    longest row of this call."""
    if max_len is None:
        max_len = max(len(item) for item in text)
    rows = []
    for item in text:
        row = list(item)[:max_len]
        rows.append(row + [0] * (max_len - len(row)))
    return torch.tensor(rows, dtype=torch.int64).reshape(len(rows), max_len)


def sample_candidates(labels: np.ndarray, neg_sampling_ratio: int, rng: np.random.Generator) -> np.ndarray:
    """``_sample_candidates`` (rec_dataset.py:60-95): every positive + ``ratio * npos`` negatives drawn uniformly
    (a k-subset when the impression has enough negatives, i.i.d. with replacement when it does not), then one
    uniform shuffle of the selection.  Returns indices into the impression's candidate list."""
    labels = np.asarray(labels)
    pos = np.where(labels == 1)[0]
    neg = np.where(labels == 0)[0]
    k = neg_sampling_ratio * len(pos)
    replace = k > len(labels) - len(pos)                                  # :76-80
    picks = rng.choice(rng.permutation(neg), k, replace=replace)         # :83-87 (raises on an empty pool, k > 0)
    return rng.permutation(np.concatenate((pos, picks)).astype(np.int64))   # :89-90


def get_item(news: pd.DataFrame, behaviors: pd.DataFrame, index: int, max_history_len: int,
             cand_picks: Optional[np.ndarray] = None):
    """``__getitem__`` (rec_dataset.py:39-55 train / :104-118 test); ``cand_picks`` = the sampled candidate
    positions (train) or None (test: all candidates in order)."""
    bhv = behaviors.iloc[index]
    user_id = np.array([int(bhv["uid"].split("U")[-1])])
    user_idx = np.array([int(bhv["user"])])
    history = np.array(bhv["history"])[:max_history_len]
    candidates = np.array(bhv["candidates"])
    labels = np.array(bhv["labels"])
    if cand_picks is not None:
        candidates, labels = candidates[cand_picks], labels[cand_picks]
    return user_id, user_idx, news.loc[history], news.loc[candidates], labels


def tokenize_df(df: pd.DataFrame, dataset_attributes: Sequence[str], max_title_len: int,
                max_abstract_len: Optional[int], concatenate_inputs: bool) -> Dict[str, torch.Tensor]:
    """``_tokenize_df`` (rec_dataset.py:184-287), ``use_plm=False``."""
    out = {"news_ids": torch.from_numpy(np.array([int(nid.split("N")[-1]) for nid in df.index.values],
                                                 dtype=np.int64).reshape(-1))}
    if not concatenate_inputs:
        out["title"] = pad_tokens(df["tokenized_title"].values.tolist(), max_title_len)
        if "abstract" in dataset_attributes:
            out["abstract"] = pad_tokens(df["tokenized_abstract"].values.tolist(), max_abstract_len)
        if "title_entities" in dataset_attributes:
            out["title_entities"] = pad_tokens(df["title_entities"].values.tolist(), max_title_len)
        if "abstract_entities" in dataset_attributes:
            out["abstract_entities"] = pad_tokens(df["abstract_entities"].values.tolist(), max_abstract_len)
    else:
        if "abstract" in dataset_attributes:
            text = [[*a, *b] for a, b in zip(df["tokenized_title"].values.tolist(),
                                             df["tokenized_abstract"].values.tolist())]
            out["text"] = pad_tokens(text, max_title_len + max_abstract_len)
        else:
            out["text"] = pad_tokens(df["tokenized_title"].values.tolist(), max_title_len)
    out["category"] = torch.from_numpy(df["category_class"].values.astype(np.int64))
    out["subcategory"] = torch.from_numpy(df["subcategory_class"].values.astype(np.int64))
    if "sentiment_class" in dataset_attributes or "sentiment_score" in dataset_attributes:
        out["sentiment"] = torch.from_numpy(df["sentiment_class"].values.astype(np.int64))
        out["sentiment_score"] = torch.from_numpy(df["sentiment_score"].values.astype(np.float32))
    return out


def collate(items: List[Tuple], dataset_attributes: Sequence[str], max_title_len: int,
            max_abstract_len: Optional[int] = None, concatenate_inputs: bool = False) -> Dict:
    """``DatasetCollate.__call__`` (rec_dataset.py:148-168)."""
    user_ids, user_idx, histories, candidates, labels = zip(*items)

    def assignees(frames):                                                 # :289-293
        sizes = torch.tensor([len(x) for x in frames])
        return torch.repeat_interleave(torch.arange(len(frames)), sizes)

    return {
        "batch_hist": assignees(histories), "batch_cand": assignees(candidates),
        "x_hist": tokenize_df(pd.concat(histories), dataset_attributes, max_title_len, max_abstract_len,
                              concatenate_inputs),
        "x_cand": tokenize_df(pd.concat(candidates), dataset_attributes, max_title_len, max_abstract_len,
                              concatenate_inputs),
        "labels": torch.from_numpy(np.concatenate(labels)).float(),
        "user_ids": torch.from_numpy(np.concatenate(user_ids)).long(),
        "user_idx": torch.from_numpy(np.concatenate(user_idx)).long(),
    }
