"""News-encoder modules with the reference's interfaces, running on the HIP kernels.

``MHSAAddAtt`` mirrors ``newsreclib.models.components.encoders.news.text.MHSAAddAtt``
(text.py:179-236) and ``NewsEncoder`` mirrors ``...encoders.news.news.NewsEncoder``
(news.py:9-183) for the NRMS configuration (one text attribute, no category/entity encoders).
Constructor signatures, attribute names and ``state_dict`` keys are the reference's.
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .attention import AdditiveAttention


def _grad_bufs(params):
    bufs = tuple(getattr(p, "main_grad", None) for p in params)
    return bufs if any(b is not None for b in bufs) else None


class MHSAAddAtt(nn.Module):
    """Embedding lookup -> dropout -> multi-head self-attention over tokens -> dropout -> additive
    attention, as ONE fused HIP pipeline (``nrl_news_encoder_fwd``/``_bwd``)."""

    def __init__(self, pretrained_embeddings: torch.Tensor, embed_dim: int, num_heads: int,
                 query_dim: int, dropout_probability: float) -> None:
        super().__init__()
        if not isinstance(dropout_probability, float):
            raise ValueError(
                f"Expected keyword argument `dropout_probability` to be a `float` but got {dropout_probability}")
        self.embedding_layer = nn.Embedding.from_pretrained(
            torch.as_tensor(pretrained_embeddings, dtype=torch.float32), freeze=False, padding_idx=0)
        # nn.MultiheadAttention is used only as the parameter container (same names, shapes and
        # default initialisation as the reference, text.py:218); its forward is never called.
        self.multihead_attention = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads)
        self.additive_attention = AdditiveAttention(input_dim=embed_dim, query_dim=query_dim)
        self.dropout = nn.Dropout(dropout_probability)  # holds p; the kernels draw the mask
        self.num_heads = num_heads
        self.table_grad_hook = None   # optional callable(grad_tensor), see trainer.NRMSTrainer

    def _params(self):
        mha, att = self.multihead_attention, self.additive_attention
        return (self.embedding_layer.weight, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                mha.out_proj.bias, att.linear.weight, att.linear.bias, att.query)

    def forward(self, text: torch.Tensor, seed: Optional[int] = None,
                order: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 and seed is None:
            # host-side draw from torch's CPU generator (no device sync); reproducible under
            # torch.manual_seed
            seed = int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 62))
        params = self._params()
        return ops.NewsEncoderFn.apply(text, *params, self.num_heads, p, seed or 0, 0, _grad_bufs(params), order,
                                       self.table_grad_hook)


class PLM(nn.Module):
    """Text encoder over a pretrained language model, mirroring the reference ``PLM`` (text.py:15-109)
    for the NRMS configuration ``use_mhsa=True, apply_reduce_dim=False``: transformer body -> dropout
    -> multi-head self-attention -> dropout -> additive attention.

    The transformer body is HF ``AutoModel`` running on PyTorch-ROCm (hipBLASLt / SDPA; SURVEY.md build
    plan step 8); everything after ``last_hidden_state`` is ONE call into the HIP library
    (``nrl_user_encoder_fwd`` with dropouts).  The reference feeds the (N, L, D) hidden states to a
    seq-first ``nn.MultiheadAttention``, so attention runs ACROSS THE NEWS ITEMS of the call for each
    token position (SURVEY.md headline fact 3) -- reproduced, and no attention mask is applied after
    the body, exactly as in the reference."""

    def __init__(self, plm_model, frozen_layers: Optional[List[int]], embed_dim: int, use_mhsa: bool,
                 apply_reduce_dim: bool, reduced_embed_dim: Optional[int], num_heads: Optional[int],
                 query_dim: Optional[int], dropout_probability: float) -> None:
        super().__init__()
        if not isinstance(plm_model, str):
            raise ValueError(f"Expected keyword argument `plm_model` to be a `str` but got {plm_model}")
        if not isinstance(dropout_probability, float):
            raise ValueError(
                f"Expected keyword argument `dropout_probability` to be a `float` but got {dropout_probability}")
        if not use_mhsa or apply_reduce_dim:
            raise NotImplementedError("newsreclib_amd.PLM covers use_mhsa=True, apply_reduce_dim=False (NRMS)")
        from transformers import AutoModel
        self.use_mhsa, self.apply_reduce_dim = use_mhsa, apply_reduce_dim
        self.plm_model = AutoModel.from_pretrained(plm_model)
        for name, param in self.plm_model.base_model.named_parameters():   # text.py:69-73
            for layer in (frozen_layers or []):
                if "layer." + str(layer) + "." in name:
                    param.requires_grad = False
        assert isinstance(num_heads, int) and num_heads > 0
        self.multihead_attention = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads)
        self.additive_attention = AdditiveAttention(input_dim=embed_dim, query_dim=query_dim)
        self.dropout = nn.Dropout(p=dropout_probability)
        self.num_heads = num_heads

    def forward(self, text: Dict[str, torch.Tensor], seed: Optional[int] = None) -> torch.Tensor:
        hidden = self.plm_model(**text)[0]                      # (N, L, D)
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 and seed is None:
            seed = int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 62))
        mha, att = self.multihead_attention, self.additive_attention
        params = (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                  att.linear.weight, att.linear.bias, att.query)
        return ops.UserEncoderFn.apply(hidden, *params, self.num_heads, _grad_bufs(params), p, seed or 0)


class NewsEncoder(nn.Module):
    """Dispatches news attributes to their encoders (news.py:134-183).  NRMS uses exactly one text
    encoder (``attributes2encode=["title"]``), for which the reference returns that encoder's output
    unchanged (news.py:159-160); the multi-attribute combinations belong to other recommenders and
    are out of this build's scope (they raise)."""

    def __init__(self, dataset_attributes: List[str], attributes2encode: List[str],
                 concatenate_inputs: bool, text_encoder: Optional[nn.Module],
                 category_encoder: Optional[nn.Module], entity_encoder: Optional[nn.Module],
                 combine_vectors: bool, combine_type: Optional[str], input_dim: Optional[int],
                 query_dim: Optional[int], output_dim: Optional[int]) -> None:
        super().__init__()
        assert len(dataset_attributes) > 0
        self.concatenate_inputs = concatenate_inputs
        if category_encoder is not None or entity_encoder is not None or combine_vectors:
            raise NotImplementedError("newsreclib_amd.NewsEncoder covers the NRMS configuration only "
                                      "(text attributes, no category/entity encoders, no combine layer)")
        assert isinstance(text_encoder, nn.Module)
        if not concatenate_inputs:
            names = sorted(set(dataset_attributes) & set(attributes2encode) & {"title", "abstract"})
            if len(names) != 1:
                raise NotImplementedError("newsreclib_amd.NewsEncoder needs exactly one text attribute "
                                          f"to encode, got {names}")
            self.text_encoders = nn.ModuleDict({name: text_encoder for name in names})
        else:
            self.text_encoders = nn.ModuleDict({"text": text_encoder})
        self.encode_text = True
        self.encode_category = False
        self.encode_entity = False

    def forward(self, news: Dict[str, torch.Tensor], seed: Optional[int] = None) -> torch.Tensor:
        (name, encoder), = self.text_encoders.items()
        kw = {}
        if seed is not None:
            kw["seed"] = seed
        if news.get(name + "_order") is not None:      # optional argsort of the flat ids (prepare_batch)
            kw["order"] = news[name + "_order"]
        return encoder(news[name], **kw)
