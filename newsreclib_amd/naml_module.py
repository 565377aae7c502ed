"""Drop-in for ``newsreclib.models.general_rec.naml_module.NAMLModule`` on MI355X HIP kernels::

    model._target_: newsreclib_amd.naml_module.NAMLModule        # configs/model/naml.yaml:1

Same 26 constructor keyword arguments (naml_module.py:82-110), sub-module attributes and ``state_dict`` keys.
News encoder: ONE ``CNNAddAtt`` shared by title and abstract (num_filters wide), the category view
``relu(Linear(embedding))`` (naml_module.py:165-178), the views combined by additive attention
(``combine_type="add_att"``, naml_module.py:181-195); user encoder: additive attention over the clicked news
(user/naml.py:26-33); dot-product scorer."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from . import ops
from .abstract_recommender import AbstractRecommender
from .click_predictor import DotProduct
from .dense_batch import dense_rows
from .news_encoder import CNNAddAtt, LinearEncoder, NewsEncoder, _draw_seed
from .nrms_module import prepare_batch, text_vocab
from .user_encoder_naml import UserEncoder


class NAMLModule(AbstractRecommender):
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
        num_heads: int,
        num_filters: Optional[int],
        window_size: Optional[int],
        query_dim: int,
        categ_embed_dim: int,
        dropout_probability: float,
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
        if use_plm:                                             # naml_module.py:149-161; widths :174,189,199
            text_encoder = self._plm_text_encoder(plm_model, frozen_layers, text_embed_dim, num_heads, query_dim,
                                                  dropout_probability)
            num_filters = text_embed_dim                        # every downstream width follows the text vector
        else:
            assert isinstance(num_filters, int) and isinstance(window_size, int)
            if pretrained_embeddings is None:
                assert isinstance(pretrained_embeddings_path, str)
                pretrained_embeddings = self._init_embedding(pretrained_embeddings_path)
            text_encoder = CNNAddAtt(pretrained_embeddings=pretrained_embeddings, embed_dim=text_embed_dim,
                                     num_filters=num_filters, window_size=window_size, query_dim=query_dim,
                                     dropout_probability=dropout_probability)
        category_encoder = LinearEncoder(pretrained_embeddings=None, from_pretrained=False,
                                         freeze_pretrained_emb=False, num_categories=self.num_categ_classes,
                                         embed_dim=categ_embed_dim, use_dropout=False, dropout_probability=None,
                                         linear_transform=True, output_dim=num_filters)
        self.news_encoder = NewsEncoder(
            dataset_attributes=dataset_attributes, attributes2encode=attributes2encode, concatenate_inputs=False,
            text_encoder=text_encoder, category_encoder=category_encoder, entity_encoder=None,
            combine_vectors=True, combine_type="add_att", input_dim=num_filters, query_dim=query_dim, output_dim=None)
        if not late_fusion:
            self.user_encoder = UserEncoder(news_embed_dim=num_filters, query_dim=query_dim)
        self.click_predictor = DotProduct()
        self._init_step_outputs(outputs)

    def _prepare(self, batch: Dict) -> Dict:
        return prepare_batch(batch, text_vocab(self))

    # -- reference: naml_module.py:261-286 -------------------------------------------------------------
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
            user_vector = self.user_encoder(hist_news_vector_agg)
        else:
            user_vector = ops.HistMeanFn.apply(hist_news_vector_agg, batch["hist_offsets"])
        return self.click_predictor(user_vector.unsqueeze(dim=1), cand_news_vector_agg.permute(0, 2, 1))
