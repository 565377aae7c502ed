"""Per-impression restatement of the reference's aspect metrics (TEST INFRASTRUCTURE).

Follows ``newsreclib/metrics/functional.py:8-127`` (``diversity``, ``personalization``,
``generalized_jaccard``) and the grouping / empty-target rule of ``metrics/base.py:137-182`` and
torchmetrics' ``RetrievalMetric.compute`` (third-party, torchmetrics==1.x: split by ``indexes``, a query
whose target sums to 0 scores 0 under ``empty_target_action="neg"``, mean over queries).  Plain Python loops
over impressions.  torchmetrics is not installed in the build container, so the reference classes cannot be
executed here: parity of these metrics is against this restatement only ("parity unpinned")."""
import math

import torch


def diversity(preds, target, num_classes, top_k=None):
    top_k = preds.shape[-1] if top_k is None else top_k
    sorted_target = target[torch.argsort(preds, dim=-1, descending=True, stable=True)][:top_k]
    count = torch.bincount(sorted_target, minlength=num_classes).double()
    prob = count / num_classes                       # functional.py:39-40
    prob = prob / prob.sum()                         # torch.distributions.Categorical normalises
    ent = -sum(float(p) * math.log(float(p)) for p in prob if p > 0)
    return ent / math.log(num_classes)


def personalization(preds, predicted_aspects, target_aspects, num_classes, top_k=None):
    top_k = preds.shape[-1] if top_k is None else top_k
    top = predicted_aspects[torch.argsort(preds, dim=-1, descending=True, stable=True)][:top_k]
    a = torch.bincount(top, minlength=num_classes).double()
    b = torch.bincount(target_aspects, minlength=num_classes).double()
    return float(torch.minimum(a, b).sum() / torch.maximum(a, b).sum())


def aspect_metrics(preds, cand_aspects, hist_aspects, cand_sizes, hist_sizes, num_classes, top_k_list, prefix="categ"):
    out = {}
    for k in top_k_list:
        div, pers, c0, h0 = [], [], 0, 0
        for cs, hs in zip(cand_sizes.tolist(), hist_sizes.tolist()):
            p, a, h = preds[c0:c0 + cs], cand_aspects[c0:c0 + cs], hist_aspects[h0:h0 + hs]
            c0, h0 = c0 + cs, h0 + hs
            if not int(a.sum()):                      # empty_target_action="neg"
                div.append(0.0)
                pers.append(0.0)
            else:
                div.append(diversity(p, a, num_classes, k))
                pers.append(personalization(p, a, h, num_classes, k))
        out[f"{prefix}_div@{k}"] = sum(div) / len(div)
        out[f"{prefix}_pers@{k}"] = sum(pers) / len(pers)
    return out
