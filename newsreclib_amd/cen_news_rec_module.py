"""Drop-in for ``newsreclib.models.general_rec.cen_news_rec_module.CenNewsRecModule`` on MI355X HIP kernels::

    model._target_: newsreclib_amd.cen_news_rec_module.CenNewsRecModule     # configs/model/cen_news_rec.yaml:1

Same constructor keyword arguments, sub-module attributes and ``state_dict`` keys: ``CNNMHSAAddAtt`` title
encoder, the long/short-term user encoder (MHA + additive attention, GRU over the recent clicks, final additive
attention), dot-product scorer."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from . import ops
from .abstract_recommender import AbstractRecommender
from .click_predictor import DotProduct
from .dense_batch import dense_rows
from .news_encoder import CNNMHSAAddAtt, NewsEncoder, _draw_seed
from .nrms_module import prepare_batch, text_vocab
from .user_encoder_cen_news_rec import UserEncoder


class CenNewsRecModule(AbstractRecommender):
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
        gru_hidden_dim: int,
        num_recent_news: int,
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
        if use_plm:                                             # cen_news_rec_module.py:149-161; width :182
            text_encoder = self._plm_text_encoder(plm_model, frozen_layers, embed_dim, num_heads, query_dim,
                                                  dropout_probability)
            num_filters = embed_dim
        else:
            assert isinstance(num_filters, int) and isinstance(window_size, int)
            if pretrained_embeddings is None:
                assert isinstance(pretrained_embeddings_path, str)
                pretrained_embeddings = self._init_embedding(pretrained_embeddings_path)
            text_encoder = CNNMHSAAddAtt(pretrained_embeddings=pretrained_embeddings, embed_dim=embed_dim,
                                         num_filters=num_filters, window_size=window_size, num_heads=num_heads,
                                         query_dim=query_dim, dropout_probability=dropout_probability)
        self.news_encoder = NewsEncoder(
            dataset_attributes=dataset_attributes, attributes2encode=attributes2encode, concatenate_inputs=False,
            text_encoder=text_encoder, category_encoder=None, entity_encoder=None, combine_vectors=False,
            combine_type=None, input_dim=None, query_dim=None, output_dim=None)
        if not late_fusion:
            self.user_encoder = UserEncoder(num_filters=num_filters, num_heads=num_heads, query_dim=query_dim,
                                            gru_hidden_dim=gru_hidden_dim, num_recent_news=num_recent_news,
                                            dropout_probability=dropout_probability)
        self.click_predictor = DotProduct()
        self._init_step_outputs(outputs)

    def _prepare(self, batch: Dict) -> Dict:
        return prepare_batch(batch, text_vocab(self))

    def forward(self, batch: Dict, seed: Optional[int] = None) -> torch.Tensor:
        batch = prepare_batch(batch, text_vocab(self))
        if self.training and seed is None:
            seed = _draw_seed()
        hist_vec, cand_vec = self._encode_news(batch, seed)
        return self.score_news_vectors(hist_vec, cand_vec, batch, seed=seed)

    def score_news_vectors(self, hist_news_vector: torch.Tensor, cand_news_vector: torch.Tensor, batch: Dict,
                           seed: Optional[int] = None) -> torch.Tensor:
        B = batch["batch_size"]
        hist_news_vector_agg = dense_rows(hist_news_vector, batch["batch_hist"], B,
                                                 batch["max_hist"], batch["hist_offsets"])
        cand_news_vector_agg = dense_rows(cand_news_vector, batch["batch_cand"], B,
                                                 batch["max_cand"], batch["cand_offsets"])
        if not self.hparams.late_fusion:
            user_vector = self.user_encoder(hist_news_vector_agg, seed=seed)
        else:
            user_vector = ops.HistMeanFn.apply(hist_news_vector_agg, batch["hist_offsets"])
        return self.click_predictor(user_vector.unsqueeze(dim=1), cand_news_vector_agg.permute(0, 2, 1))
