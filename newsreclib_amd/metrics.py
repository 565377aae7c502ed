"""Epoch-end ranking metrics (AUC, MRR, nDCG@k) over concatenated per-impression scores.

The reference wires torchmetrics ``AUROC``/``RetrievalMRR``/``RetrievalNormalizedDCG`` grouped by an
``indexes`` vector (nrms_module.py:182-195,380-396).  These run once per epoch, outside the timed
step, on small vectors; they are host-side bookkeeping in plain torch, vectorised over queries."""
from typing import Dict, Sequence

import torch


def _dense(preds, targets, sizes):
    B, C = sizes.numel(), int(sizes.max()) if sizes.numel() else 0
    mask = torch.arange(C, device=preds.device)[None, :] < sizes[:, None]
    p = preds.new_full((B, C), float("-inf"))
    t = targets.new_zeros((B, C))
    p[mask], t[mask] = preds, targets
    return p, t, mask


def ranking_metrics(preds: torch.Tensor, targets: torch.Tensor, cand_news_size: torch.Tensor,
                    top_k_list: Sequence[int] = (5, 10)) -> Dict[str, float]:
    sizes = cand_news_size.to(preds.device)
    p, t, mask = _dense(preds.float(), targets.float(), sizes)
    order = torch.argsort(p, dim=1, descending=True, stable=True)
    t_sorted = torch.gather(t, 1, order)
    ranks = torch.arange(1, p.shape[1] + 1, device=p.device, dtype=torch.float32)[None, :]
    has_pos = t.sum(1) > 0
    # MRR: reciprocal rank of the first relevant item.  torchmetrics' retrieval metrics default to
    # empty_target_action="neg": an impression WITHOUT a positive counts as 0 and stays in the mean
    # (RetrievalMRR() / RetrievalNormalizedDCG(top_k=k) at nrms_module.py:186,190 take that default).
    first = torch.where(t_sorted > 0, ranks, torch.full_like(ranks, float("inf"))).min(1).values
    rr = torch.where(has_pos, 1.0 / first, torch.zeros_like(first))
    mrr = rr.mean() if rr.numel() else preds.new_tensor(0.0)
    out = {"mrr": float(mrr)}
    disc = 1.0 / torch.log2(ranks + 1.0)
    ideal = torch.sort(t, dim=1, descending=True).values
    for k in top_k_list:
        dcg = (t_sorted[:, :k] * disc[:, :k]).sum(1)
        idcg = (ideal[:, :k] * disc[:, :k]).sum(1)
        ndcg = torch.where(idcg > 0, dcg / idcg.clamp_min(1e-12), torch.zeros_like(dcg))
        out[f"ndcg@{k}"] = float(ndcg.mean()) if ndcg.numel() else 0.0      # 0 for impressions without a positive
    # global AUROC over all (score, label) pairs, as torchmetrics' binary AUROC without indexes
    pos, neg = preds[targets > 0], preds[targets <= 0]
    if pos.numel() and neg.numel():
        allv = torch.cat([pos, neg])
        r = torch.empty(allv.shape, dtype=torch.float64, device=allv.device)
        srt, idx = torch.sort(allv)
        # average ranks for ties; rank sums in float64 (an epoch holds millions of pairs: float32 ranks stop being integers at 2^24)
        uniq, inv, cnt = torch.unique_consecutive(srt, return_inverse=True, return_counts=True)
        ends = torch.cumsum(cnt, 0).double()
        avg_rank = ends - (cnt.double() - 1) / 2
        r[idx] = avg_rank[inv]
        auc = (r[: pos.numel()].sum() - pos.numel() * (pos.numel() + 1) / 2) / (float(pos.numel()) * neg.numel())
        out["auc"] = float(auc)
    else:
        out["auc"] = 0.0
    return out


def aspect_metrics(preds: torch.Tensor, cand_aspects: torch.Tensor, hist_aspects: torch.Tensor,
                   cand_news_size: torch.Tensor, hist_news_size: torch.Tensor, num_classes: int,
                   top_k_list: Sequence[int] = (5, 10), prefix: str = "categ") -> Dict[str, float]:
    """Aspect-based diversity and personalization of the top-k recommendations, mean over impressions --
    the reference's ``Diversity`` / ``Personalization`` (metrics/diversity.py, metrics/personalization.py,
    metrics/base.py:137-182 on top of metrics/functional.py:8-127), vectorised over impressions instead of
    a Python loop per impression:

    * diversity@k: entropy of the aspect distribution of the k highest-scored candidates / log(num_classes)
      (``functional.py:36-46``: the counts are divided by num_classes and re-normalised by ``Categorical``);
    * personalization@k: generalised Jaccard sum(min) / sum(max) between the aspect counts of those k
      candidates and the aspect counts of the clicked history (``functional.py:85-127``);
    * an impression whose candidate aspects sum to 0 scores 0 (``empty_target_action="neg"``).
    Flat inputs in impression order (as ``model_step`` returns them) + the per-impression sizes."""
    dev = preds.device
    csz, hsz = cand_news_size.to(dev).long(), hist_news_size.to(dev).long()
    B = csz.numel()
    C = int(csz.max()) if B else 0
    cmask = torch.arange(C, device=dev)[None, :] < csz[:, None]
    p = preds.new_full((B, C), float("-inf")).float()
    a = torch.zeros((B, C), dtype=torch.long, device=dev)
    p[cmask], a[cmask] = preds.float(), cand_aspects.to(dev).long()
    order = torch.argsort(p, dim=1, descending=True, stable=True)
    a_sorted = torch.gather(a, 1, order)
    valid_sorted = torch.gather(cmask, 1, order)
    nonempty = (a * cmask).sum(1) > 0
    hist_q = torch.repeat_interleave(torch.arange(B, device=dev), hsz)
    hist_cnt = torch.zeros((B, num_classes), device=dev)
    hist_cnt.index_put_((hist_q, hist_aspects.to(dev).long()), torch.ones(hist_q.numel(), device=dev), accumulate=True)
    out = {}
    log_nc = float(torch.log(torch.tensor(float(num_classes))))
    for k in top_k_list:
        take = valid_sorted & (torch.arange(C, device=dev)[None, :] < k)
        cnt = torch.zeros((B, num_classes), device=dev)
        cnt.scatter_add_(1, a_sorted, take.float())
        prob = cnt / cnt.sum(1, keepdim=True).clamp_min(1.0)
        ent = -(torch.where(prob > 0, prob * torch.log(prob.clamp_min(1e-38)), torch.zeros_like(prob))).sum(1)
        div = torch.where(nonempty, ent / log_nc, torch.zeros_like(ent))
        jac = torch.minimum(cnt, hist_cnt).sum(1) / torch.maximum(cnt, hist_cnt).sum(1).clamp_min(1e-38)
        pers = torch.where(nonempty, jac, torch.zeros_like(jac))
        out[f"{prefix}_div@{k}"] = float(div.mean()) if B else 0.0
        out[f"{prefix}_pers@{k}"] = float(pers.mean()) if B else 0.0
    return out
