"""Drop-in for ``newsreclib.models.general_rec.tanr_module.TANRModule`` on MI355X HIP kernels::

    model._target_: newsreclib_amd.tanr_module.TANRModule        # configs/model/tanr.yaml:1

Same constructor keyword arguments (tanr_module.py:84-112), sub-module attributes and ``state_dict`` keys:
CNN + additive-attention title encoder, NAML's additive-attention user encoder (tanr_module.py:16,193-200), a
topic predictor ``nn.Linear(num_filters, num_categ_classes)`` over all encoded news whose cross entropy against
the one-hot category is added with weight ``topic_pred_loss_coef`` (tanr_module.py:284-286,361-367)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops, ops_blocks
from .abstract_recommender import AbstractRecommender
from .click_predictor import CrossEntropyLoss, DotProduct
from .dense_batch import dense_rows
from .news_encoder import CNNAddAtt, NewsEncoder, _draw_seed
from .nrms_module import prepare_batch, text_vocab
from .user_encoder_naml import UserEncoder


class TANRModule(AbstractRecommender):
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
        num_filters: Optional[int],
        window_size: Optional[int],
        query_dim: int,
        dropout_probability: float,
        topic_pred_loss_coef: float,
        top_k_list: List[int],
        num_categ_classes: int,
        num_sent_classes: int,
        save_recs: bool,
        recs_fpath: Optional[str],
        optimizer: Any,
        scheduler: Any,
        pretrained_embeddings: Optional[torch.Tensor] = None,
    ) -> None:
        super().__init__()
        self.save_hyperparameters(logger=False, ignore=["pretrained_embeddings"])
        self.num_categ_classes = num_categ_classes + 1
        self.num_sent_classes = num_sent_classes + 1
        if save_recs:
            assert isinstance(recs_fpath, str)
        self._init_loss(loss, dual_loss_training, dual_loss_coef)      # CE / SupCon / dual
        self.topic_pred_loss = CrossEntropyLoss()
        if use_plm:                                             # tanr_module.py:154-166; widths :179,187,197
            text_encoder = self._plm_text_encoder(plm_model, frozen_layers, embed_dim, num_heads, query_dim,
                                                  dropout_probability)
            num_filters = embed_dim
        else:
            assert isinstance(num_filters, int) and isinstance(window_size, int)
            if pretrained_embeddings is None:
                assert isinstance(pretrained_embeddings_path, str)
                pretrained_embeddings = self._init_embedding(pretrained_embeddings_path)
            text_encoder = CNNAddAtt(pretrained_embeddings=pretrained_embeddings, embed_dim=embed_dim,
                                     num_filters=num_filters, window_size=window_size, query_dim=query_dim,
                                     dropout_probability=dropout_probability)
        self.news_encoder = NewsEncoder(
            dataset_attributes=dataset_attributes, attributes2encode=attributes2encode, concatenate_inputs=False,
            text_encoder=text_encoder, category_encoder=None, entity_encoder=None, combine_vectors=False,
            combine_type=None, input_dim=num_filters, query_dim=query_dim, output_dim=None)
        self.topic_predictor = nn.Linear(in_features=num_filters, out_features=self.num_categ_classes)
        if not late_fusion:
            self.user_encoder = UserEncoder(news_embed_dim=num_filters, query_dim=query_dim)
        self.click_predictor = DotProduct()
        self._init_step_outputs(outputs)

    def _prepare(self, batch: Dict) -> Dict:
        return prepare_batch(batch, text_vocab(self))

    # -- reference: tanr_module.py:258-286 -------------------------------------------------------------
    def forward(self, batch: Dict, seed: Optional[int] = None):
        batch = prepare_batch(batch, text_vocab(self))
        if self.training and seed is None:
            seed = _draw_seed()
        hist_vec, cand_vec = self._encode_news(batch, seed)
        n_hist = hist_vec.shape[0]
        news_vector = torch.cat((hist_vec, cand_vec), dim=0)              # rows: [history; candidates]
        scores = self.score_news_vectors(hist_vec, cand_vec, batch)
        # topic scores of every encoded news.  The reference orders the rows [candidates; history]
        # (tanr_module.py:284); the loss is a mean over rows, so the order only matters for the returned tensor.
        w, b = self.topic_predictor.weight, self.topic_predictor.bias
        n_cls = w.shape[0]
        pad = (-n_cls) % 4                                                 # the GEMM wants 4-column multiples
        wp = torch.cat([w, w.new_zeros(pad, w.shape[1])]) if pad else w
        bp = torch.cat([b, b.new_zeros(pad)]) if pad else b
        topic_all = ops_blocks.LinearActFn.apply(news_vector, wp, bp, "none", None)[:, :n_cls]
        topic_scores = torch.cat((topic_all[n_hist:], topic_all[:n_hist]), dim=0)
        return scores, topic_scores

    def score_news_vectors(self, hist_news_vector: torch.Tensor, cand_news_vector: torch.Tensor,
                           batch: Dict) -> torch.Tensor:
        """User encoding + click scores from already-encoded news (tanr_module.py:262-282; also the entry of the
        encode-once evaluation path, evaluation.NewsVectorCache)."""
        B = batch["batch_size"]
        hist_news_vector_agg = dense_rows(hist_news_vector, batch["batch_hist"], B,
                                                 batch["max_hist"], batch["hist_offsets"])
        cand_news_vector_agg = dense_rows(cand_news_vector, batch["batch_cand"], B,
                                                 batch["max_cand"], batch["cand_offsets"])
        if not self.hparams.late_fusion:
            user_vector = self.user_encoder(hist_news_vector_agg)
        else:
            user_vector = ops.HistMeanFn.apply(hist_news_vector_agg, batch["hist_offsets"])
        return self.click_predictor(user_vector.unsqueeze(dim=1), cand_news_vector_agg.permute(0, 2, 1))

    # -- reference: tanr_module.py:361-367 -------------------------------------------------------------
    def _aux_loss(self, batch: Dict, topic_scores: torch.Tensor) -> torch.Tensor:
        topics = torch.cat((batch["x_cand"]["category"], batch["x_hist"]["category"]))
        topic_prob = torch.nn.functional.one_hot(topics, num_classes=self.num_categ_classes).to(topic_scores.dtype)
        return self.hparams.topic_pred_loss_coef * self.topic_pred_loss(topic_scores.contiguous(), topic_prob)
