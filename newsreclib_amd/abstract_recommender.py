"""Shared step / epoch / optimizer logic of the drop-in recommenders, mirroring the reference's
``AbstractRecommneder`` (abstract_recommender.py:17-130) plus the ``model_step`` / ``*_step`` /
``on_*_epoch_end`` bodies that the reference repeats in every module (nrms_module.py:260-362,
lstur_module.py:308-410 -- identical apart from the class name)."""
from __future__ import annotations

import json
from typing import Any, Dict, List, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ._lightning import LightningModuleBase
from . import ops
from .dense_batch import dense_rows
from .metrics import aspect_metrics, ranking_metrics


class AbstractRecommender(LightningModuleBase):
    def _init_step_outputs(self, outputs: Dict) -> None:
        self.step_outputs = {stage: {key: [] for key in keys} for stage, keys in outputs.items()}
        self.training_step_outputs = {key: [] for key in self.step_outputs.get("train", {})}
        self.val_step_outputs = {key: [] for key in self.step_outputs.get("val", {})}
        self.test_step_outputs = {key: [] for key in self.step_outputs.get("test", {})}
        self._loss_sums = {"train": [0.0, 0], "val": [0.0, 0], "test": [0.0, 0]}
        self.val_loss_best = float("inf")

    # -- reference: abstract_recommender.py:113-124 and the constructors' loss block (nrms_module.py:110-118) ----
    def _get_loss(self, criterion: str):
        from .click_predictor import CrossEntropyLoss, SupConLoss
        if criterion == "cross_entropy_loss":
            return CrossEntropyLoss()
        if criterion == "sup_con_loss":
            return SupConLoss()
        if criterion == "dual_loss":
            return CrossEntropyLoss(), SupConLoss()
        raise ValueError(f"Loss not defined: {criterion}")

    def _init_loss(self, loss: str, dual_loss_training: bool, dual_loss_coef) -> None:
        if not dual_loss_training:
            self.criterion = self._get_loss(loss)
            if isinstance(self.criterion, tuple):
                raise ValueError("loss='dual_loss' needs dual_loss_training=True")
        else:
            assert isinstance(dual_loss_coef, float)
            self.ce_criterion, self.scl_criterion = self._get_loss(loss)

    def _loss(self, scores: torch.Tensor, y_true: torch.Tensor, batch: Dict) -> torch.Tensor:
        """nrms_module.py:286-328.  The positive / negative index lists the reference builds with per-user Python
        loops (:290-304) are what ``SupConLoss`` derives from y_true and the candidate counts on the device."""
        hp = self.hparams
        if hp.loss == "cross_entropy_loss":
            return self.criterion(scores, y_true)
        sizes = batch["cand_sizes"]
        if not hp.dual_loss_training:
            return self.criterion(scores, y_true, sizes)
        ce_loss = self.ce_criterion(scores, y_true)
        scl_loss = self.scl_criterion(scores, y_true, sizes)
        return (1 - hp.dual_loss_coef) * ce_loss + hp.dual_loss_coef * scl_loss

    # -- the two news-encoder calls of every recommender's forward (e.g. nrms_module.py:232,236) ----
    def _plm_text_encoder(self, plm_model, frozen_layers, embed_dim, num_heads, query_dim, dropout_probability):
        """``PLM(use_mhsa=True, apply_reduce_dim=False)`` -- the configuration every recommender of the reference
        builds under ``use_plm`` (e.g. lstur_module.py:158-170, naml_module.py:149-161)."""
        from .news_encoder import PLM
        assert isinstance(plm_model, str)
        return PLM(plm_model=plm_model, frozen_layers=frozen_layers, embed_dim=embed_dim, use_mhsa=True,
                   apply_reduce_dim=False, reduced_embed_dim=None, num_heads=num_heads, query_dim=query_dim,
                   dropout_probability=dropout_probability)

    def _encode_news(self, batch: Dict, seed=None):
        """-> (history news vectors, candidate news vectors).  Row-independent text encoders see history and
        candidates in ONE call (identical vectors, half the launches); the PLM encoder's seq-first attention
        couples the news of a call (text.py:92-96), so there the reference's two calls are kept."""
        if self.hparams.use_plm:
            # (the transformer BODY is per news: one pass over both calls' texts, news_encoder.PLM.share_body; the tails stay two)
            self.news_encoder.share_plm_bodies(batch["x_hist"], batch["x_cand"])
            return (self.news_encoder(batch["x_hist"], seed=seed),
                    self.news_encoder(batch["x_cand"], seed=seed, stream_base=4))
        n_hist = batch["batch_hist"].shape[0]
        news_vector = self.news_encoder(batch["x_all"], seed=seed)
        return ops.split_rows(news_vector, n_hist)

    # -- reference: abstract_recommender.py:110-111 ------------------------------------------------
    def _init_embedding(self, filepath: str) -> torch.Tensor:
        return torch.from_numpy(np.load(filepath)).float()

    # -- reference: nrms_module.py:260-362 ---------------------------------------------------------
    def model_step(self, batch: Dict) -> Tuple:
        batch = self._prepare(batch)
        B = batch["batch_size"]
        out = self.forward(batch)
        scores, aux = out if isinstance(out, tuple) else (out, None)
        y_true = dense_rows(batch["labels"], batch["batch_cand"], B, batch["max_cand"],
                            batch["cand_offsets"], batch["cand_flat_idx"], max_is_exact=True)
        loss = self._loss(scores, y_true.float(), batch)
        if aux is not None:          # recommenders with an auxiliary task (TANR topic prediction)
            loss = loss + self._aux_loss(batch, aux)

        # outputs for metric computation: gathering the valid slots in row-major order == the
        # reference's per-user concatenation (abstract_recommender.py:126-130), no loops, no syncs
        if batch["labels"].shape[0] == scores.numel():      # every row full: the valid slots are all slots, in order
            preds = scores.detach().reshape(-1)
        else:
            preds = scores.detach().reshape(-1)[batch["cand_flat_idx"]]
        targets = batch["labels"]
        cand_news_size, hist_news_size = batch["cand_sizes"], batch["hist_sizes"]    # == the mask row sums (:331-345)

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

    # -- data-parallel epoch ends: every rank sees every rank's step outputs -------------------------
    @staticmethod
    def _world() -> int:
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _gather_outputs(self, outputs: Dict[str, list]) -> Dict[str, list]:
        """All-gather the per-rank step outputs (ragged, so as objects; epoch-end only) and concatenate them in
        rank order.  The reference's torchmetrics objects synchronise their states across DDP ranks at compute()
        (SURVEY.md section 5); with plain floats logged here the gather is explicit.  Impressions keep their own
        group: the reference restarts its `indexes` at 0 on every rank (nrms_module.py:385,429,487), which would
        merge unrelated impressions under torchmetrics' sync -- that collision is not reproduced."""
        if self._world() == 1:
            return outputs
        local = {k: [t.detach().cpu() for t in v if torch.is_tensor(t)] for k, v in outputs.items()}
        gathered: List[Dict[str, list]] = [None] * self._world()
        dist.all_gather_object(gathered, local)
        dev = self.device
        return {k: [t.to(dev) for part in gathered for t in part.get(k, [])] for k in outputs}

    def _epoch_end(self, stage: str, outputs: Dict[str, list]) -> Dict[str, float]:
        s = self._loss_sums[stage]
        logs = {}
        if self._world() > 1:
            tot = torch.tensor([float(s[0]), float(s[1])], dtype=torch.float64, device=self.device)
            dist.all_reduce(tot)                       # mean over the steps of ALL ranks (MeanMetric sync)
            s = [float(tot[0]), int(tot[1])]
        if s[1]:
            logs[f"{stage}/loss"] = float(s[0]) / s[1]
        local_outputs = outputs
        outputs = self._gather_outputs(outputs)
        if outputs.get("preds") and outputs.get("targets") and outputs.get("cand_news_size"):
            m = ranking_metrics(torch.cat(outputs["preds"]), torch.cat(outputs["targets"]),
                                torch.cat(outputs["cand_news_size"]), self.hparams.top_k_list)
            logs.update({f"{stage}/{k}": v for k, v in m.items()})
            # test stage: aspect-based diversity / personalization of the recommendations
            # (nrms_module.py:475-493: categories and sentiments of candidates vs clicked history)
            for asp, ncls in (("categories", getattr(self, "num_categ_classes", None)),
                              ("sentiments", getattr(self, "num_sent_classes", None))):
                tk, hk = f"target_{asp}", f"hist_{asp}"
                if ncls and outputs.get(tk) and outputs.get(hk) and outputs.get("hist_news_size") \
                        and all(t.numel() for t in outputs[tk]) and all(t.numel() for t in outputs[hk]):
                    a = aspect_metrics(torch.cat(outputs["preds"]), torch.cat(outputs[tk]), torch.cat(outputs[hk]),
                                       torch.cat(outputs["cand_news_size"]), torch.cat(outputs["hist_news_size"]),
                                       ncls, self.hparams.top_k_list, prefix="categ" if asp == "categories" else "sent")
                    logs.update({f"{stage}/{k}": v for k, v in a.items()})
        if stage == "test" and getattr(self.hparams, "save_recs", False) and outputs.get("user_ids") \
                and outputs.get("cand_news_ids") and outputs.get("preds"):
            # nrms_module.py:520-531
            recs = self._get_recommendations(user_ids=torch.cat(outputs["user_ids"]),
                                             news_ids=torch.cat(outputs["cand_news_ids"]),
                                             scores=torch.cat(outputs["preds"]),
                                             cand_news_size=torch.cat(outputs["cand_news_size"]))
            if self._world() == 1 or dist.get_rank() == 0:
                self._save_recommendations(recommendations=recs, fpath=self.hparams.recs_fpath)
        elif stage == "test" and getattr(self.hparams, "save_recs", False):
            raise RuntimeError("save_recs=True needs `user_ids`, `cand_news_ids`, `preds` and `cand_news_size` among "
                               "outputs.test (configs/model/nrms.yaml lists them)")
        for v in local_outputs.values():
            v.clear()
        self._loss_sums[stage] = [0.0, 0]
        self.log_dict(logs, on_step=False, on_epoch=True, prog_bar=True, logger=True)
        return logs

    # -- reference: abstract_recommender.py:159-193 ------------------------------------------------
    def _get_recommendations(self, user_ids: torch.Tensor, news_ids: torch.Tensor, scores: torch.Tensor,
                             cand_news_size: torch.Tensor) -> Dict[str, Dict[str, float]]:
        """{"U<user id>": {"N<news id>": score, ...}, ...}; a user seen in several impressions keeps ONE entry per
        news, the later score overwriting the earlier, as the reference's dictionary fill does."""
        sizes = cand_news_size.detach().cpu()
        users = torch.repeat_interleave(user_ids.detach().cpu(), sizes).tolist()
        news = news_ids.detach().cpu().tolist()
        vals = scores.detach().cpu().tolist()
        if not (len(users) == len(news) == len(vals)):
            raise ValueError("recommendations: user / news / score vectors disagree in length")
        recs: Dict[str, Dict[str, float]] = {}
        for u, n, v in zip(users, news, vals):
            recs.setdefault(f"U{u}", {})[f"N{n}"] = v
        return recs

    def _save_recommendations(self, recommendations: Dict[str, Dict[str, float]], fpath: str) -> None:
        with open(fpath, "w") as f:
            json.dump(recommendations, f)

    # -- evaluation epochs (nrms_module.py:398-535) run under frozen weights: every ``MHSAAddAtt`` text encoder serves its forwards
    #    from ONE per-token q|k|v table for the length of the epoch (news_encoder.MHSAAddAtt.token_table; results torch.equal) ----
    def _token_tables(self, enter: bool) -> None:
        import contextlib
        stack = getattr(self, "_tt_stack", None)
        if stack is not None:
            stack.close()
            self._tt_stack = None
        if not enter:
            return
        stack = contextlib.ExitStack()
        seen = set()
        enc = getattr(self, "news_encoder", None)
        for te in (getattr(enc, "text_encoders", {}) or {}).values():
            if hasattr(te, "token_table") and id(te) not in seen:
                seen.add(id(te))
                stack.enter_context(te.token_table())
        self._tt_stack = stack

    def on_validation_epoch_start(self) -> None:
        self._token_tables(True)

    def on_test_epoch_start(self) -> None:
        self._token_tables(True)

    def on_train_epoch_end(self) -> None:
        self._epoch_end("train", self.training_step_outputs)

    def on_validation_epoch_end(self) -> None:
        self._token_tables(False)
        logs = self._epoch_end("val", self.val_step_outputs)
        if "val/loss" in logs:
            self.val_loss_best = min(self.val_loss_best, logs["val/loss"])
            self.log("val/loss_best", self.val_loss_best, prog_bar=True, logger=True, sync_dist=True)

    def on_test_epoch_end(self) -> None:
        self._token_tables(False)
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
