"""Drop-in for ``newsreclib.models.general_rec.lstur_module.LSTURModule`` on MI355X HIP kernels
(BASELINE config 5).  Select it from the reference's Hydra configs with::

    model._target_: newsreclib_amd.lstur_module.LSTURModule      # configs/model/lstur.yaml:1

Same 29 constructor keyword arguments (lstur_module.py:88-119), same sub-module attributes and
``state_dict`` keys; the text encoder is ONE ``CNNAddAtt`` shared by title and abstract
(lstur_module.py:146-156, news.py:69-79).  History and candidate news are encoded in one call per
attribute (row-independent => identical results)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from . import ops
from .abstract_recommender import AbstractRecommender
from .click_predictor import DotProduct
from .dense_batch import dense_rows
from .news_encoder import CNNAddAtt, LinearEncoder, NewsEncoder, _draw_seed
from .nrms_module import prepare_batch, text_vocab
from .user_encoder_lstur import UserEncoder


class LSTURModule(AbstractRecommender):
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
        num_users: int,
        user_masking_probability: float,
        long_short_term_method: str,
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
        self.num_categ_classes = num_categ_classes + 1          # lstur_module.py:126-128
        self.num_sent_classes = num_sent_classes + 1
        self.num_users = num_users + 1
        if save_recs:
            assert isinstance(recs_fpath, str)
        self._init_loss(loss, dual_loss_training, dual_loss_coef)      # CE / SupCon / dual

        if use_plm:                                             # lstur_module.py:158-170
            text_encoder = self._plm_text_encoder(plm_model, frozen_layers, text_embed_dim, num_heads, query_dim,
                                                  dropout_probability)
        else:
            if pretrained_embeddings is None:
                assert isinstance(pretrained_embeddings_path, str)
                pretrained_embeddings = self._init_embedding(pretrained_embeddings_path)
            text_encoder = CNNAddAtt(pretrained_embeddings=pretrained_embeddings, embed_dim=text_embed_dim,
                                     num_filters=num_filters, window_size=window_size, query_dim=query_dim,
                                     dropout_probability=dropout_probability)
        category_encoder = LinearEncoder(pretrained_embeddings=None, from_pretrained=False,
                                         freeze_pretrained_emb=False, num_categories=self.num_categ_classes,
                                         embed_dim=categ_embed_dim, use_dropout=False, dropout_probability=None,
                                         linear_transform=False, output_dim=None)
        self.news_encoder = NewsEncoder(
            dataset_attributes=dataset_attributes, attributes2encode=attributes2encode, concatenate_inputs=False,
            text_encoder=text_encoder, category_encoder=category_encoder, entity_encoder=None,
            combine_vectors=True, combine_type="concat", input_dim=None, query_dim=None, output_dim=None)
        # lstur_module.py:201-212 (the reference sizes the GRU with text_embed_dim; the text vectors are
        # num_filters wide, so the config must have num_filters == text_embed_dim as lstur.yaml does)
        text_dim = text_embed_dim * 2 if "title" in attributes2encode and "abstract" in attributes2encode \
            else text_embed_dim
        categ_dim = categ_embed_dim * 2 if "category" in attributes2encode and "subcategory" in attributes2encode \
            else categ_embed_dim
        if not late_fusion:                                     # lstur_module.py:213-218
            self.user_encoder = UserEncoder(num_users=self.num_users, input_dim=text_dim + categ_dim,
                                            user_masking_probability=user_masking_probability,
                                            long_short_term_method=long_short_term_method)
        self.click_predictor = DotProduct()
        self._init_step_outputs(outputs)

    def _prepare(self, batch: Dict) -> Dict:
        return prepare_batch(batch, text_vocab(self))

    # -- reference: lstur_module.py:278-303 -----------------------------------------------------------
    def forward(self, batch: Dict, seed: Optional[int] = None) -> torch.Tensor:
        batch = prepare_batch(batch, text_vocab(self))
        B = batch["batch_size"]
        if self.training and seed is None:
            seed = _draw_seed()                       # one draw per step; streams separate the dropouts
        hist_vec, cand_vec = self._encode_news(batch, seed)
        return self.score_news_vectors(hist_vec, cand_vec, batch, seed=seed)

    def score_news_vectors(self, hist_news_vector: torch.Tensor, cand_news_vector: torch.Tensor, batch: Dict,
                           seed: Optional[int] = None) -> torch.Tensor:
        """lstur_module.py:280-303 from already-encoded news rows (see ``evaluation.NewsVectorCache``)."""
        B = batch["batch_size"]
        hist_news_vector_agg = dense_rows(hist_news_vector, batch["batch_hist"], B,
                                                 batch["max_hist"], batch["hist_offsets"])
        cand_news_vector_agg = dense_rows(cand_news_vector, batch["batch_cand"], B,
                                                 batch["max_cand"], batch["cand_offsets"])
        hist_size = batch["hist_sizes"]               # == mask_hist row sums (lstur_module.py:287-290)
        if not self.hparams.late_fusion:
            user_vector = self.user_encoder(batch["user_idx"], hist_news_vector_agg, hist_size, seed=seed,
                                            min_hist_size=batch["min_hist"])
        else:                                         # lstur_module.py:295-296
            user_vector = ops.HistMeanFn.apply(hist_news_vector_agg, batch["hist_offsets"])
        scores = self.click_predictor(user_vector.unsqueeze(dim=1), cand_news_vector_agg.permute(0, 2, 1))
        return scores
