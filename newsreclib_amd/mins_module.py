"""Drop-in for ``newsreclib.models.general_rec.mins_module.MINSModule`` on MI355X HIP kernels::

    model._target_: newsreclib_amd.mins_module.MINSModule        # configs/model/mins.yaml:1

Same 26 constructor keyword arguments (mins_module.py:84-111), sub-module attributes and ``state_dict`` keys:
``MHSAAddAtt`` text encoder shared by title and abstract, category view ``relu(Linear(embedding))``,
additive-attention view combination, the multi-channel GRU user encoder, dot-product scorer."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from . import ops
from .abstract_recommender import AbstractRecommender
from .click_predictor import DotProduct
from .dense_batch import dense_rows
from .news_encoder import LinearEncoder, MHSAAddAtt, NewsEncoder, _draw_seed
from .nrms_module import prepare_batch, text_vocab
from .user_encoder_mins import UserEncoder


class MINSModule(AbstractRecommender):
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
        text_embed_dim: int,
        categ_embed_dim: int,
        num_heads: int,
        query_dim: int,
        dropout_probability: float,
        num_filters: int,
        num_gru_channels: int,
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
        if use_plm:                                             # mins_module.py:145-157
            text_encoder = self._plm_text_encoder(plm_model, frozen_layers, text_embed_dim, num_heads, query_dim,
                                                  dropout_probability)
        else:
            if pretrained_embeddings is None:
                assert isinstance(pretrained_embeddings_path, str)
                pretrained_embeddings = self._init_embedding(pretrained_embeddings_path)
            text_encoder = MHSAAddAtt(pretrained_embeddings=pretrained_embeddings, embed_dim=text_embed_dim,
                                      num_heads=num_heads, query_dim=query_dim,
                                      dropout_probability=dropout_probability)
        category_encoder = LinearEncoder(pretrained_embeddings=None, from_pretrained=False,
                                         freeze_pretrained_emb=False, num_categories=self.num_categ_classes,
                                         embed_dim=categ_embed_dim, use_dropout=False, dropout_probability=None,
                                         linear_transform=True, output_dim=text_embed_dim)
        self.news_encoder = NewsEncoder(
            dataset_attributes=dataset_attributes, attributes2encode=attributes2encode, concatenate_inputs=False,
            text_encoder=text_encoder, category_encoder=category_encoder, entity_encoder=None,
            combine_vectors=True, combine_type="add_att", input_dim=text_embed_dim, query_dim=query_dim,
            output_dim=None)
        if not late_fusion:
            self.user_encoder = UserEncoder(news_embed_dim=text_embed_dim, query_dim=query_dim,
                                            num_filters=num_filters, num_gru_channels=num_gru_channels)
        self.click_predictor = DotProduct()
        self._init_step_outputs(outputs)

    def _prepare(self, batch: Dict) -> Dict:
        return prepare_batch(batch, text_vocab(self))

    # -- reference: mins_module.py:250-279 -------------------------------------------------------------
    def forward(self, batch: Dict, seed: Optional[int] = None) -> torch.Tensor:
        batch = prepare_batch(batch, text_vocab(self))
        if self.training and seed is None:
            seed = _draw_seed()
        hist_vec, cand_vec = self._encode_news(batch, seed)
        return self.score_news_vectors(hist_vec, cand_vec, batch)

    def score_news_vectors(self, hist_news_vector: torch.Tensor, cand_news_vector: torch.Tensor,
                           batch: Dict) -> torch.Tensor:
        B = batch["batch_size"]
        hist_news_vector_agg = dense_rows(hist_news_vector, batch["batch_hist"], B,
                                                 batch["max_hist"], batch["hist_offsets"])
        cand_news_vector_agg = dense_rows(cand_news_vector, batch["batch_cand"], B,
                                                 batch["max_cand"], batch["cand_offsets"])
        if not self.hparams.late_fusion:
            if batch["min_hist"] < 1:
                raise RuntimeError("Length of all samples has to be greater than 0")   # pack_padded_sequence's check
            user_vector = self.user_encoder(hist_news_vector_agg, batch["hist_sizes"])
        else:
            user_vector = ops.HistMeanFn.apply(hist_news_vector_agg, batch["hist_offsets"])
        return self.click_predictor(user_vector.unsqueeze(dim=1), cand_news_vector_agg.permute(0, 2, 1))
