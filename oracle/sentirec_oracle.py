"""CPU restatement of the reference SentiRec forward / loss (TEST INFRASTRUCTURE -- see oracle/__init__.py).

``SentiRecModule`` (fair_rec/sentirec_module.py:236-273, 347-364) = the NRMS path (nrms_oracle) + a sentiment
predictor ``nn.Linear(embed_dim, num_sent_classes)`` over [candidate; history] news vectors + two loss terms:
``sent_pred_loss_coef * L1Loss(labels.flatten(), labels)`` -- the predictor's output is overwritten by the label
tensor before the loss (:348-352), so the term is exactly 0 and the predictor gets no gradient -- and
``sent_div_loss_coef * relu(mean_hist_sentiment[:, None] * cand_sentiment * scores).mean()`` over the DENSE,
zero-padded (B, C_max) matrices (:355-364).  Pinned by tests/golden/make_golden_sentirec.py."""
from __future__ import annotations

from typing import Dict

import torch

from .nrms_oracle import make_params, nrms_forward, to_dense_batch


def make_sentirec_params(vocab: int, n_sent: int, embed_dim: int = 300, query_dim: int = 200, seed: int = 0) -> Dict:
    p = make_params(vocab, embed_dim=embed_dim, query_dim=query_dim, seed=seed)
    g = torch.Generator().manual_seed(seed + 1000)
    p["sent_predictor.weight"] = (torch.randn(n_sent, embed_dim, generator=g) * embed_dim ** -0.5).float()
    p["sent_predictor.bias"] = (torch.randn(n_sent, generator=g) * 0.05).float()
    return p


def sentirec_forward(batch, params, num_heads: int = 15, pred_coef: float = 0.4, div_coef: float = 10.0,
                     p_drop: float = 0.0, seed: int = 0) -> dict:
    out = nrms_forward(batch, params, num_heads=num_heads, p_drop=p_drop, seed=seed)
    B = out["scores"].shape[0]
    rows = torch.cat((out["cand_vec"], out["hist_vec"]), dim=0)                                  # :271
    sent_scores = rows @ params["sent_predictor.weight"].t() + params["sent_predictor.bias"]
    labels = torch.cat((batch["x_cand"]["sentiment_score"], batch["x_hist"]["sentiment_score"]))  # :348-350
    sent_pred_loss = (labels.flatten() - labels).abs().mean()                                     # :352 (== 0)
    sent_hist, mask_hist = to_dense_batch(batch["x_hist"]["sentiment_score"], batch["batch_hist"], B)
    sent_cand, _ = to_dense_batch(batch["x_cand"]["sentiment_score"], batch["batch_cand"], B)
    user_mean = sent_hist.sum(dim=1) / mask_hist.sum(dim=1)                                       # :360-362
    sent_div_loss = torch.relu(user_mean.unsqueeze(-1) * sent_cand * out["scores"]).mean()         # :363
    out = dict(out)
    out.update(sent_scores=sent_scores, rec_loss=out["loss"], sent_div_loss=sent_div_loss,
               loss=out["loss"] + pred_coef * sent_pred_loss + div_coef * sent_div_loss)
    return out


def sentirec_loss_and_grads(batch, params, **kw):
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = sentirec_forward(batch, leaves, **kw)
    grads = torch.autograd.grad(out["loss"], list(leaves.values()), allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(leaves[k])) for k, gr in zip(leaves, grads)}
    g["news_encoder.text_encoders.title.embedding_layer.weight"][0] = 0.0
    return out, g
