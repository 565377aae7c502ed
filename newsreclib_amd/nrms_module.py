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

from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops
from ._lightning import LightningModuleBase
from .click_predictor import CrossEntropyLoss, DotProduct
from .dense_batch import dense_slot_index, to_dense_batch
from .metrics import ranking_metrics
from .news_encoder import PLM, MHSAAddAtt, NewsEncoder
from .user_encoder import UserEncoder


def prepare_batch(batch: Dict) -> Dict:
    """Attach the ragged-layout metadata the forward needs (offsets, max lengths, batch size).

    A collate function has these on the host for free (it builds ``batch_hist`` from the per-user
    list lengths, rec_dataset.py:289-293); computing them from the device vectors costs two syncs,
    so do it once per batch, outside the timed step."""
    if "hist_offsets" in batch:
        return batch
    B = int(batch["user_idx"].shape[0]) if "user_idx" in batch else int(batch["batch_hist"].max()) + 1
    out = dict(batch)
    out["batch_size"] = B
    for key in ("hist", "cand"):
        off = ops.offsets_from_sorted_batch(batch["batch_" + key], B)
        out[key + "_offsets"] = off
        out["max_" + key] = int((off[1:] - off[:-1]).max())
    out["cand_flat_idx"] = dense_slot_index(batch["batch_cand"], out["cand_offsets"], out["max_cand"])
    # history + candidate token ids as the single encoder call sees them, and their id-sorted
    # visiting order for the embedding gradient (pure index bookkeeping, like the offsets above)
    for attr in ("title", "abstract"):
        if attr in batch["x_hist"] and attr in batch["x_cand"]:
            h, c = batch["x_hist"][attr], batch["x_cand"][attr]
            if torch.is_tensor(h):
                ids = torch.cat([h, c], dim=0)
                out.setdefault("x_all", {})[attr] = ids
                out["x_all"][attr + "_order"] = torch.argsort(ids.reshape(-1))
            else:   # PLM tokenizer output: dict of (N, L) tensors (rec_dataset.py:180-190)
                out.setdefault("x_all", {})[attr] = {k: torch.cat([h[k], c[k]], dim=0) for k in h.keys()}
    return out


class NRMSModule(LightningModuleBase):
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
        if dual_loss_training or loss != "cross_entropy_loss":
            raise NotImplementedError("newsreclib_amd.NRMSModule implements loss='cross_entropy_loss' "
                                      "(configs/model/nrms.yaml:6); sup_con / dual loss are out of scope")
        if late_fusion:
            raise NotImplementedError("late_fusion=True is not built yet")
        self.criterion = CrossEntropyLoss()

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
        self.user_encoder = UserEncoder(news_embed_dim=embed_dim, num_heads=num_heads, query_dim=query_dim)
        self.click_predictor = DotProduct()
        self._text_attr = next(iter(self.news_encoder.text_encoders.keys()))

        self.step_outputs = {stage: {key: [] for key in keys} for stage, keys in outputs.items()}
        self.training_step_outputs = {key: [] for key in self.step_outputs.get("train", {})}
        self.val_step_outputs = {key: [] for key in self.step_outputs.get("val", {})}
        self.test_step_outputs = {key: [] for key in self.step_outputs.get("test", {})}
        self._loss_sums = {"train": [0.0, 0], "val": [0.0, 0], "test": [0.0, 0]}
        self.val_loss_best = float("inf")
        assert hp is not None

    # -- reference: abstract_recommender.py:110-111 ------------------------------------------------
    def _init_embedding(self, filepath: str) -> torch.Tensor:
        return torch.from_numpy(np.load(filepath)).float()

    # -- reference: nrms_module.py:230-255 ---------------------------------------------------------
    def forward(self, batch: Dict) -> torch.Tensor:
        batch = prepare_batch(batch)
        B = batch["batch_size"]
        hist_text = batch["x_hist"][self._text_attr]
        n_hist = (hist_text if torch.is_tensor(hist_text) else next(iter(hist_text.values()))).shape[0]
        # one encoder call for history + candidate news (the reference makes two, :232,236)
        news_vector = self.news_encoder(batch["x_all"])
        hist_news_vector_agg, _ = to_dense_batch(news_vector[:n_hist], batch["batch_hist"], B,
                                                 batch["max_hist"], batch["hist_offsets"])
        cand_news_vector_agg, _ = to_dense_batch(news_vector[n_hist:], batch["batch_cand"], B,
                                                 batch["max_cand"], batch["cand_offsets"])
        user_vector = self.user_encoder(hist_news_vector_agg)
        scores = self.click_predictor(user_vector.unsqueeze(dim=1), cand_news_vector_agg.permute(0, 2, 1))
        return scores

    # -- reference: nrms_module.py:260-362 ---------------------------------------------------------
    def model_step(self, batch: Dict) -> Tuple:
        batch = prepare_batch(batch)
        B = batch["batch_size"]
        scores = self.forward(batch)
        y_true, _ = to_dense_batch(batch["labels"], batch["batch_cand"], B, batch["max_cand"],
                                   batch["cand_offsets"], batch["cand_flat_idx"])
        loss = self.criterion(scores, y_true.float())

        # outputs for metric computation: gathering the valid slots in row-major order == the
        # reference's per-user concatenation (abstract_recommender.py:126-130), no loops, no syncs
        preds = scores.detach().reshape(-1)[batch["cand_flat_idx"]]
        targets = batch["labels"]
        cand_news_size = batch["cand_offsets"][1:] - batch["cand_offsets"][:-1]
        hist_news_size = batch["hist_offsets"][1:] - batch["hist_offsets"][:-1]

        def attr(side, name):
            v = batch["x_" + side].get(name)
            return v if v is not None else torch.empty(0, dtype=torch.int64, device=scores.device)

        return (loss, preds, targets, cand_news_size, hist_news_size, attr("cand", "category"),
                attr("cand", "sentiment"), attr("hist", "category"), attr("hist", "sentiment"),
                batch["user_ids"], attr("cand", "news_ids"))

    def _collect_step_outputs(self, outputs_dict, local_vars):
        for key in outputs_dict.keys():
            outputs_dict[key].append(local_vars.get(key, []))
        return outputs_dict

    def _track(self, stage, loss):
        s = self._loss_sums[stage]
        s[0] = s[0] + loss.detach()
        s[1] += 1

    def training_step(self, batch: Dict, batch_idx: int):
        loss, preds, targets, cand_news_size, *_ = self.model_step(batch)
        self._track("train", loss)
        self.training_step_outputs = self._collect_step_outputs(self.training_step_outputs, locals())
        return loss

    def validation_step(self, batch: Dict, batch_idx: int):
        loss, preds, targets, cand_news_size, *_ = self.model_step(batch)
        self._track("val", loss)
        self.val_step_outputs = self._collect_step_outputs(self.val_step_outputs, locals())

    def test_step(self, batch: Dict, batch_idx: int):
        (loss, preds, targets, cand_news_size, hist_news_size, target_categories, target_sentiments,
         hist_categories, hist_sentiments, user_ids, cand_news_ids) = self.model_step(batch)
        self._track("test", loss)
        self.test_step_outputs = self._collect_step_outputs(self.test_step_outputs, locals())

    def _epoch_end(self, stage: str, outputs: Dict[str, list]) -> Dict[str, float]:
        s = self._loss_sums[stage]
        logs = {}
        if s[1]:
            logs[f"{stage}/loss"] = float(s[0]) / s[1]
        if outputs.get("preds"):
            m = ranking_metrics(torch.cat(outputs["preds"]), torch.cat(outputs["targets"]),
                                torch.cat(outputs["cand_news_size"]), self.hparams.top_k_list)
            logs.update({f"{stage}/{k}": v for k, v in m.items()})
        for v in outputs.values():
            v.clear()
        self._loss_sums[stage] = [0.0, 0]
        self.log_dict(logs, on_step=False, on_epoch=True, prog_bar=True, logger=True)
        return logs

    def on_train_epoch_end(self) -> None:
        self._epoch_end("train", self.training_step_outputs)

    def on_validation_epoch_end(self) -> None:
        logs = self._epoch_end("val", self.val_step_outputs)
        if "val/loss" in logs:
            self.val_loss_best = min(self.val_loss_best, logs["val/loss"])
            self.log("val/loss_best", self.val_loss_best, prog_bar=True, logger=True, sync_dist=True)

    def on_test_epoch_end(self) -> None:
        self._epoch_end("test", self.test_step_outputs)

    # -- reference: abstract_recommender.py:89-108 ---------------------------------------------------
    def configure_optimizers(self) -> Dict[str, Any]:
        optimizer = self.hparams.optimizer(params=self.parameters())
        if self.hparams.scheduler is not None:
            scheduler = self.hparams.scheduler(optimizer=optimizer)
            return {"optimizer": optimizer,
                    "lr_scheduler": {"scheduler": scheduler, "monitor": "valid/loss", "interval": "epoch",
                                     "frequency": 1}}
        return {"optimizer": optimizer}
