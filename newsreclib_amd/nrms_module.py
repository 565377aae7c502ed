"""Drop-in for ``newsreclib.models.general_rec.nrms_module.NRMSModule`` on MI355X HIP kernels.

Select it from the reference's Hydra configs by overriding one key::

    model._target_: newsreclib_amd.nrms_module.NRMSModule        # configs/model/nrms.yaml:1

Same 23 constructor keyword arguments (nrms_module.py:75-100), same sub-module attributes
(``news_encoder`` / ``user_encoder`` / ``click_predictor``) and ``state_dict`` keys, same
``forward`` / ``model_step`` / ``training_step`` / ``configure_optimizers`` contracts.  What
differs is underneath: the encoders and the scorer call the C-ABI HIP library, history and
candidate news are encoded in ONE encoder call (row-independent, so identical results), and
``model_step`` builds its outputs without per-user Python loops or device->host syncs
(the reference pays ~2B syncs + 6B tiny kernels per step, nrms_module.py:331-345).
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import torch

from . import ops
from .abstract_recommender import AbstractRecommender
from .click_predictor import DotProduct
from .dense_batch import dense_rows, dense_slot_index
from .news_encoder import PLM, MHSAAddAtt, NewsEncoder
from .user_encoder import UserEncoder


def attach_layout(batch: Dict) -> Dict:
    """Attach the ragged-layout metadata the forward needs (offsets, max lengths, batch size).

    A collate function has these on the host for free (it builds ``batch_hist`` from the per-user
    list lengths, rec_dataset.py:289-293; ``input_pipeline.build_batch`` supplies them); computing
    them from the device vectors costs two syncs, so a loader does it once per batch, outside the step."""
    if "cand_flat_idx" in batch:
        return batch
    B = int(batch["batch_size"]) if "batch_size" in batch else (
        int(batch["user_idx"].shape[0]) if "user_idx" in batch else int(batch["batch_hist"].max()) + 1)
    out = dict(batch)
    out["batch_size"] = B
    for key in ("hist", "cand"):
        if key + "_offsets" in batch:       # a loader that knows the row lengths on the host supplies these
            continue                        # (input_pipeline.build_batch): no device read-back at all
        off = ops.offsets_from_sorted_batch(batch["batch_" + key], B)
        out[key + "_offsets"] = off
        sizes = off[1:] - off[:-1]
        out["max_" + key] = int(sizes.max())
        out[key + "_sizes"] = sizes
    if "min_hist" not in out:
        out["min_hist"] = int(out["hist_sizes"].min())
    out["cand_flat_idx"] = dense_slot_index(batch["batch_cand"], out["cand_offsets"], out["max_cand"])
    return out


def text_vocab(module) -> Optional[int]:
    """Rows of the word-embedding table the module's text encoders share (None: no embedding table, e.g. a PLM): the
    exclusive upper bound of the token ids, which lets ``prepare_batch`` group them with the counting sort."""
    enc = getattr(module, "news_encoder", None)
    for te in getattr(enc, "text_encoders", {}).values() if enc is not None else ():
        emb = getattr(te, "embedding_layer", None)
        if emb is not None:
            return int(emb.weight.shape[0])
    return None


def prepare_batch(batch: Dict, vocab: Optional[int] = None, need_order: Optional[bool] = None) -> Dict:
    """``attach_layout`` + the per-step device work on the token ids: history and candidate ids as the single
    encoder call sees them, and their id-sorted visiting order for the embedding gradient (the sort the reference
    pays inside ``embedding_dense_backward``).  No host sync; part of the train step (bench.py times it).
    ``need_order`` (default: whether gradients are enabled): the visiting order serves the backward only, so a forward
    under ``torch.no_grad()`` does not sort."""
    if need_order is None:
        need_order = torch.is_grad_enabled()
    out = attach_layout(batch)
    if "x_all" in out:
        return out
    out = dict(out)
    for attr in ("title", "abstract"):
        if attr in batch["x_hist"] and attr in batch["x_cand"]:
            h, c = batch["x_hist"][attr], batch["x_cand"][attr]
            if torch.is_tensor(h):
                ids = torch.cat([h, c], dim=0)
                out.setdefault("x_all", {})[attr] = ids
                # counting sort over the vocabulary on a side stream: the order is needed by the backward only, so its four
                # small launches (~50 us at B = 128) run beside the fused forward instead of in front of it.  Measured slower
                # mid-round (3.30 vs 3.26 ms: the sort was 80 us of atomics then); with the current sort, two alternating pairs
                # of 100 steps: 2.98 / 3.00 vs 3.03 / 3.02 ms.  NRL_SORT_ASYNC=0 keeps it on the launch stream.
                # NRL_SORT_ASYNC=2: not here at all -- the news encoder's forward issues it on the side stream AFTER its own launches,
                # so it runs beside the user encoder's few-row launches instead of beside the fused forward
                mode = os.environ.get("NRL_SORT_ASYNC", "1")
                if mode != "2" and need_order:
                    sort = ops.sort_positions_async if mode == "1" else ops.sort_positions
                    out["x_all"][attr + "_order"] = sort(ids, vocab)
            # (PLM tokenizer output -- a dict of (N, L) tensors, rec_dataset.py:180-190 -- is NOT merged: the
            #  two sides are padded to their own longest text and the PLM encoder must see them in separate calls)
    for attr in ("category", "subcategory"):
        if attr in batch["x_hist"] and attr in batch["x_cand"]:
            out.setdefault("x_all", {})[attr] = torch.cat([batch["x_hist"][attr], batch["x_cand"][attr]], dim=0)
    return out


class NRMSModule(AbstractRecommender):
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
        hp = self.hparams
        self.num_categ_classes = num_categ_classes + 1
        self.num_sent_classes = num_sent_classes + 1
        if save_recs:
            assert isinstance(recs_fpath, str)
        self._init_loss(loss, dual_loss_training, dual_loss_coef)      # CE / SupCon / dual

        if not use_plm:
            # pretrained embeddings + contextualisation (nrms_module.py:122-135)
            if pretrained_embeddings is None:
                assert isinstance(pretrained_embeddings_path, str)
                pretrained_embeddings = self._init_embedding(pretrained_embeddings_path)
            text_encoder = MHSAAddAtt(pretrained_embeddings=pretrained_embeddings, embed_dim=embed_dim,
                                      num_heads=num_heads, query_dim=query_dim,
                                      dropout_probability=dropout_probability)
        else:
            # PLM news encoder (nrms_module.py:136-149)
            assert isinstance(plm_model, str)
            text_encoder = PLM(plm_model=plm_model, frozen_layers=frozen_layers, embed_dim=embed_dim,
                               use_mhsa=True, apply_reduce_dim=False, reduced_embed_dim=None,
                               num_heads=num_heads, query_dim=query_dim, dropout_probability=dropout_probability)
        self.news_encoder = NewsEncoder(
            dataset_attributes=dataset_attributes, attributes2encode=attributes2encode,
            concatenate_inputs=False, text_encoder=text_encoder, category_encoder=None,
            entity_encoder=None, combine_vectors=False, combine_type=None, input_dim=None,
            query_dim=None, output_dim=None)
        if not late_fusion:                                     # nrms_module.py:165-171
            self.user_encoder = UserEncoder(news_embed_dim=embed_dim, num_heads=num_heads, query_dim=query_dim)
        self.click_predictor = DotProduct()
        self._text_attr = next(iter(self.news_encoder.text_encoders.keys()))

        self._init_step_outputs(outputs)
        assert hp is not None

    def _prepare(self, batch: Dict) -> Dict:
        te = self.news_encoder.text_encoders[self._text_attr]
        emb = getattr(te, "embedding_layer", None)
        return prepare_batch(batch, emb.weight.shape[0] if emb is not None else None)

    # -- reference: nrms_module.py:230-255 ---------------------------------------------------------
    def forward(self, batch: Dict) -> torch.Tensor:
        batch = self._prepare(batch)
        B = batch["batch_size"]
        hist_text = batch["x_hist"][self._text_attr]
        n_hist = (hist_text if torch.is_tensor(hist_text) else next(iter(hist_text.values()))).shape[0]
        if self.hparams.use_plm:
            # the PLM encoder's seq-first attention runs ACROSS THE NEWS OF ONE CALL (text.py:92-96), so the two
            # calls of the reference (:232,236) are NOT interchangeable with one call over [history; candidates]
            # (the transformer BODY treats every news on its own: it runs once over both calls' news, news_encoder.PLM.share_body)
            self.news_encoder.share_plm_bodies(batch["x_hist"], batch["x_cand"])
            hist_vec = self.news_encoder(batch["x_hist"])
            cand_vec = self.news_encoder(batch["x_cand"])
            return self.score_news_vectors(hist_vec, cand_vec, batch)
        # rows of the MHSAAddAtt encoder are independent: one call for history + candidate news gives the same
        # vectors as the reference's two (:232,236)
        news_vector = self.news_encoder(batch["x_all"])
        return self.score_news_vectors(*ops.split_rows(news_vector, n_hist), batch)

    def score_news_vectors(self, hist_news_vector: torch.Tensor, cand_news_vector: torch.Tensor,
                           batch: Dict) -> torch.Tensor:
        """nrms_module.py:233-253 from already-encoded news rows (also the entry of the evaluation path that
        encodes every unique news once, ``evaluation.NewsVectorCache``)."""
        B = batch["batch_size"]
        hist_news_vector_agg = dense_rows(hist_news_vector, batch["batch_hist"], B,
                                          batch["max_hist"], batch["hist_offsets"], max_is_exact=True)
        cand_news_vector_agg = dense_rows(cand_news_vector, batch["batch_cand"], B,
                                          batch["max_cand"], batch["cand_offsets"], max_is_exact=True)
        if not self.hparams.late_fusion:
            user_vector = self.user_encoder(hist_news_vector_agg)
        else:  # aggregate embeddings of clicked news (nrms_module.py:243-248)
            user_vector = ops.HistMeanFn.apply(hist_news_vector_agg, batch["hist_offsets"])
        scores = self.click_predictor(user_vector.unsqueeze(dim=1), cand_news_vector_agg.permute(0, 2, 1))
        return scores
