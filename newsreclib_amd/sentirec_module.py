"""Drop-in for ``newsreclib.models.fair_rec.sentirec_module.SentiRecModule`` on MI355X HIP kernels::

    model._target_: newsreclib_amd.sentirec_module.SentiRecModule        # configs/model/sentirec.yaml:1

The NRMS path (same encoders, same 23 + 2 constructor keyword arguments, sentirec_module.py:84-110) with a
sentiment predictor ``nn.Linear(embed_dim, num_sent_classes)`` over every encoded news and two extra loss terms
(sentirec_module.py:347-364).  Reproduced as they are in the reference:

* the sentiment-prediction term is ``L1Loss(labels.flatten(), labels)`` -- the variable holding the predictor's
  output is overwritten by the label tensor two lines earlier (:348-352), so the term is exactly 0 and the
  predictor never receives a gradient; it is still evaluated and returned by ``forward``;
* the sentiment-diversity regulariser ``relu(mean_hist_sentiment * cand_sentiment * scores).mean()`` runs over the
  DENSE (B, C_max) matrices, padded slots included (they contribute 0 to the sum and count in the mean).

The regulariser is a few hundred elementwise operations on the score matrix and is written with torch ops on the
device; everything with arithmetic weight goes through the C ABI as in ``NRMSModule``."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops, ops_blocks
from .dense_batch import dense_rows
from .nrms_module import NRMSModule, prepare_batch, text_vocab


class SentiRecModule(NRMSModule):
    def __init__(
        self,
        dataset_attributes: List[str],
        attributes2encode: List[str],
        outputs: Dict[str, List[str]],
        dual_loss_training: bool,
        dual_loss_coef: Optional[float],
        loss: str,
        late_fusion: bool,
        temperature: Optional[float],
        use_plm: bool,
        pretrained_embeddings_path: Optional[str],
        plm_model: Optional[str],
        frozen_layers: Optional[List[int]],
        embed_dim: int,
        num_heads: int,
        query_dim: int,
        dropout_probability: float,
        sent_pred_loss_coef: float,
        sent_div_loss_coef: float,
        top_k_list: List[int],
        num_categ_classes: int,
        num_sent_classes: int,
        save_recs: bool,
        recs_fpath: Optional[str],
        optimizer: Any,
        scheduler: Any,
        pretrained_embeddings: Optional[torch.Tensor] = None,
    ) -> None:
        super().__init__(
            dataset_attributes=dataset_attributes, attributes2encode=attributes2encode, outputs=outputs,
            dual_loss_training=dual_loss_training, dual_loss_coef=dual_loss_coef, loss=loss, late_fusion=late_fusion,
            temperature=temperature, use_plm=use_plm, pretrained_embeddings_path=pretrained_embeddings_path,
            plm_model=plm_model, frozen_layers=frozen_layers, embed_dim=embed_dim, num_heads=num_heads,
            query_dim=query_dim, dropout_probability=dropout_probability, top_k_list=top_k_list,
            num_categ_classes=num_categ_classes, num_sent_classes=num_sent_classes, save_recs=save_recs,
            recs_fpath=recs_fpath, optimizer=optimizer, scheduler=scheduler, pretrained_embeddings=pretrained_embeddings)
        # (sent_pred_loss_coef / sent_div_loss_coef are in self.hparams: save_hyperparameters walks every __init__ frame)
        self.sent_pred_loss = nn.L1Loss()                                               # sentirec_module.py:129
        self.sent_predictor = nn.Linear(in_features=embed_dim, out_features=self.num_sent_classes)   # :186-188

    # -- reference: sentirec_module.py:236-273 ---------------------------------------------------------
    def forward(self, batch: Dict):
        batch = prepare_batch(batch, text_vocab(self))
        if self.hparams.use_plm:
            hist_vec = self.news_encoder(batch["x_hist"])
            cand_vec = self.news_encoder(batch["x_cand"])
        else:
            n_hist = batch["batch_hist"].shape[0]
            news_vector = self.news_encoder(batch["x_all"])
            hist_vec, cand_vec = ops.split_rows(news_vector, n_hist)
        scores = self.score_news_vectors(hist_vec, cand_vec, batch)
        w, b = self.sent_predictor.weight, self.sent_predictor.bias
        n_cls = w.shape[0]
        pad = (-n_cls) % 4                                                 # the GEMM wants 4-column multiples
        wp = torch.cat([w, w.new_zeros(pad, w.shape[1])]) if pad else w
        bp = torch.cat([b, b.new_zeros(pad)]) if pad else b
        rows = torch.cat((cand_vec, hist_vec), dim=0)                                   # :271
        sent_scores = ops_blocks.LinearActFn.apply(rows.contiguous(), wp, bp, "none", None)[:, :n_cls]
        return scores, (sent_scores, scores)

    # -- reference: sentirec_module.py:347-364 ---------------------------------------------------------
    def _aux_loss(self, batch: Dict, aux) -> torch.Tensor:
        _, scores = aux
        B = batch["batch_size"]
        # (:348-352) the predictor's scores are replaced by the labels before the L1 loss: exactly 0
        sent_labels = torch.cat((batch["x_cand"]["sentiment_score"], batch["x_hist"]["sentiment_score"]))
        loss = self.hparams.sent_pred_loss_coef * self.sent_pred_loss(sent_labels.flatten(), sent_labels)
        sent_hist = dense_rows(batch["x_hist"]["sentiment_score"], batch["batch_hist"], B, batch["max_hist"],
                                      batch["hist_offsets"])
        sent_cand = dense_rows(batch["x_cand"]["sentiment_score"], batch["batch_cand"], B, batch["max_cand"],
                                      batch["cand_offsets"], batch["cand_flat_idx"])
        user_mean_sent_score = sent_hist.sum(dim=1) / batch["hist_sizes"].to(sent_hist.dtype)          # :360-362
        sent_div_loss = torch.relu(user_mean_sent_score.unsqueeze(dim=-1) * sent_cand * scores).mean()  # :363
        return loss + self.hparams.sent_div_loss_coef * sent_div_loss
