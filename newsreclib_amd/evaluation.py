"""Evaluation path: device-resident news table, batches built by index, and
scoring of full impressions with every unique news encoded ONCE.

The reference re-tokenises and re-encodes every history and candidate news of every impression in every
validation / test batch (``rec_dataset.py:98-121,148-293`` -> ``nrms_module.py:230-255``): with ~37 candidates
and up to 50 clicks per impression over a pool of ~65k news that is ~50x redundant work, and the collate does
pandas ``.loc`` + per-row ``F.pad`` on the host.  Here

* ``DeviceNewsTable`` keeps the pre-tokenised attributes of all news in HBM (one row per unique news) and
  builds a ``RecommendationBatch`` from per-impression news-index lists with index gathers on the device --
  the same layout ``DatasetCollate`` produces (concatenated rows + sorted assignment vectors);
* ``NewsVectorCache`` runs the module's news encoder over the table once (eval mode: no dropout, rows are
  independent, so the cached vector of a news is BIT-IDENTICAL to what any batch would compute for it) and
  then scores impressions from gathered vectors through the module's own user encoder and click predictor
  (``score_news_vectors``).  The seq-first user-attention quirk still couples the users of a batch, exactly
  as in the uncached forward, so scores match the uncached path for the same batch composition.

Everything runs through the same C-ABI kernels; nothing here is a second implementation of the model.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import ops
from .dense_batch import dense_rows
from .nrms_module import prepare_batch

TEXT_ATTRS = ("title", "abstract")
ASPECT_ATTRS = ("category", "subcategory", "sentiment")


class DeviceNewsTable:
    """attrs: name -> (num_news, ...) tensor (token ids (num_news, L) int64 for text attributes, (num_news,)
    int64 for category / sentiment ...).  Row 0 may be a padding news; indices are plain row numbers."""

    def __init__(self, attrs: Dict[str, torch.Tensor], device="cuda"):
        if not attrs:
            raise ValueError("DeviceNewsTable needs at least one news attribute")
        n = {int(v.shape[0]) for v in attrs.values()}
        if len(n) != 1:
            raise ValueError(f"all news attributes must have the same number of rows, got {sorted(n)}")
        self.num_news = n.pop()
        self.attrs = {k: v.to(device).contiguous() for k, v in attrs.items()}
        self.device = torch.device(device)

    def gather(self, idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {k: v.index_select(0, idx) for k, v in self.attrs.items()}

    def build_batch(self, hist_idx: torch.Tensor, hist_sizes: torch.Tensor, cand_idx: torch.Tensor,
                    cand_sizes: torch.Tensor, labels: torch.Tensor, user_idx: Optional[torch.Tensor] = None,
                    user_ids: Optional[torch.Tensor] = None) -> Dict:
        """The collated batch of ``rec_dataset.py:289-293`` from index lists: ``x_hist`` / ``x_cand`` hold the
        rows of the clicked / candidate news (concatenated over impressions), ``batch_hist`` / ``batch_cand``
        the sorted impression number of every row."""
        dev = self.device
        hist_idx, cand_idx = hist_idx.to(dev), cand_idx.to(dev)
        hist_sizes, cand_sizes = hist_sizes.to(dev).long(), cand_sizes.to(dev).long()
        B = int(hist_sizes.numel())
        if int(cand_sizes.numel()) != B:
            raise ValueError("hist_sizes and cand_sizes must have one entry per impression")
        ar = torch.arange(B, device=dev)
        batch = {
            "x_hist": self.gather(hist_idx), "x_cand": self.gather(cand_idx),
            "batch_hist": torch.repeat_interleave(ar, hist_sizes), "batch_cand": torch.repeat_interleave(ar, cand_sizes),
            "labels": labels.to(dev).float(),
            "user_idx": user_idx.to(dev) if user_idx is not None else ar.clone(),
            "user_ids": user_ids.to(dev) if user_ids is not None else ar + 1,
            "batch_size": B,
        }
        if "news_ids" not in self.attrs:            # no id column in the table: the row number stands in
            batch["x_cand"]["news_ids"] = cand_idx
        return batch


class NewsVectorCache:
    """Encode-once evaluation of a drop-in recommender (any of the module mirrors: NRMS, LSTUR, NAML, TANR,
    CenNewsRec, MINS -- whatever exposes ``news_encoder`` and ``score_news_vectors``)."""

    def __init__(self, module, table: DeviceNewsTable, chunk: int = 16384):
        self.module, self.table, self.chunk = module, table, int(chunk)
        self.vectors: Optional[torch.Tensor] = None

    @torch.no_grad()
    def build(self) -> torch.Tensor:
        """news vectors (num_news, D) of the whole table, in chunks (the encoder workspace is O(rows))."""
        if getattr(getattr(self.module, "hparams", None), "use_plm", False):
            raise NotImplementedError("the PLM text encoder attends across the news of one call (text.py:92-96): "
                                      "a news vector is not a function of the news alone and cannot be cached")
        was_training = self.module.training
        self.module.eval()
        enc = self.module.news_encoder
        names = [k for k in self.table.attrs if k in TEXT_ATTRS or k in ("category", "subcategory")]
        out = []
        import contextlib
        with contextlib.ExitStack() as stack:
            # one pass over the whole corpus under frozen weights: MHSAAddAtt text encoders run their in-projection once per
            # VOCABULARY id instead of once per token position (news_encoder.MHSAAddAtt.token_table; same bits)
            for te in {id(t): t for t in (getattr(enc, "text_encoders", {}) or {}).values()}.values():
                if hasattr(te, "token_table"):
                    stack.enter_context(te.token_table())
            for lo in range(0, self.table.num_news, self.chunk):
                hi = min(lo + self.chunk, self.table.num_news)
                out.append(enc({k: self.table.attrs[k][lo:hi] for k in names}))
        self.vectors = torch.cat(out, dim=0)
        self.module.train(was_training)
        return self.vectors

    def _meta(self, hist_sizes, cand_sizes, labels, user_idx, user_ids) -> Dict:
        dev = self.table.device
        hist_sizes, cand_sizes = hist_sizes.to(dev).long(), cand_sizes.to(dev).long()
        B = int(hist_sizes.numel())
        ar = torch.arange(B, device=dev)
        meta = {
            "x_hist": {}, "x_cand": {},
            "batch_hist": torch.repeat_interleave(ar, hist_sizes), "batch_cand": torch.repeat_interleave(ar, cand_sizes),
            "labels": labels.to(dev).float() if labels is not None else None,
            "user_idx": user_idx.to(dev) if user_idx is not None else ar.clone(),
            "user_ids": user_ids.to(dev) if user_ids is not None else ar + 1,
            "batch_size": B,
        }
        return prepare_batch(meta)

    @torch.no_grad()
    def scores(self, hist_idx: torch.Tensor, hist_sizes: torch.Tensor, cand_idx: torch.Tensor,
               cand_sizes: torch.Tensor, user_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(B, max_cand) click scores of a batch of impressions given by news-index lists."""
        if self.vectors is None:
            self.build()
        meta = self._meta(hist_sizes, cand_sizes, None, user_idx, None)
        dev = self.table.device
        hv = ops.embedding_gather(self.vectors, hist_idx.to(dev).reshape(-1, 1)).reshape(-1, self.vectors.shape[1])
        cv = ops.embedding_gather(self.vectors, cand_idx.to(dev).reshape(-1, 1)).reshape(-1, self.vectors.shape[1])
        was_training = self.module.training
        self.module.eval()
        out = self.module.score_news_vectors(hv, cv, meta)
        self.module.train(was_training)
        return out

    @torch.no_grad()
    def model_step(self, hist_idx, hist_sizes, cand_idx, cand_sizes, labels, user_idx=None, user_ids=None):
        """The tuple ``model_step`` returns (loss, preds, targets, cand_news_size, hist_news_size, aspects ...)
        for a batch given by index lists; feeds ``test_step`` / the epoch-end metrics unchanged."""
        if self.vectors is None:
            self.build()
        dev = self.table.device
        meta = self._meta(hist_sizes, cand_sizes, labels, user_idx, user_ids)
        scores = self.scores(hist_idx, hist_sizes, cand_idx, cand_sizes, user_idx)
        y_true = dense_rows(meta["labels"], meta["batch_cand"], meta["batch_size"], meta["max_cand"],
                                   meta["cand_offsets"], meta["cand_flat_idx"])
        loss = self.module._loss(scores, y_true.float(), meta)
        preds = scores.reshape(-1)[meta["cand_flat_idx"]]
        empty = torch.empty(0, dtype=torch.int64, device=dev)

        def attr(idx, name):
            return self.table.attrs[name].index_select(0, idx.to(dev)) if name in self.table.attrs else empty

        return (loss, preds, meta["labels"], meta["cand_sizes"], meta["hist_sizes"], attr(cand_idx, "category"),
                attr(cand_idx, "sentiment"), attr(hist_idx, "category"), attr(hist_idx, "sentiment"),
                meta["user_ids"], cand_idx.to(dev))


def evaluate_impressions(cache: NewsVectorCache, impressions: Sequence[Dict], batch_size: int = 512,
                         top_k_list: Sequence[int] = (5, 10), num_categ_classes: Optional[int] = None,
                         num_sent_classes: Optional[int] = None) -> Dict[str, float]:
    """Scores a list of impressions ({"hist": idx tensor, "cand": idx tensor, "labels": tensor[, "user_idx"]})
    in batches and returns the epoch-end metrics of ``on_test_epoch_end`` (nrms_module.py:456-493)."""
    from .metrics import aspect_metrics, ranking_metrics
    outs = []
    for lo in range(0, len(impressions), batch_size):
        chunk = impressions[lo:lo + batch_size]
        hs = torch.tensor([len(i["hist"]) for i in chunk])
        cs = torch.tensor([len(i["cand"]) for i in chunk])
        uidx = torch.stack([torch.as_tensor(i["user_idx"]) for i in chunk]) if "user_idx" in chunk[0] else None
        outs.append(cache.model_step(torch.cat([torch.as_tensor(i["hist"]) for i in chunk]), hs,
                                     torch.cat([torch.as_tensor(i["cand"]) for i in chunk]), cs,
                                     torch.cat([torch.as_tensor(i["labels"]).float() for i in chunk]), uidx))
    cat = lambda j: torch.cat([o[j] for o in outs])  # noqa: E731
    logs = {"loss": float(sum(float(o[0]) for o in outs) / max(1, len(outs)))}
    logs.update(ranking_metrics(cat(1), cat(2), cat(3), top_k_list))
    for name, tj, hj, ncls in (("categ", 5, 7, num_categ_classes), ("sent", 6, 8, num_sent_classes)):
        if ncls and cat(tj).numel() and cat(hj).numel():
            logs.update(aspect_metrics(cat(1), cat(tj), cat(hj), cat(3), cat(4), ncls, top_k_list, prefix=name))
    return logs
