"""News-encoder modules with the reference's interfaces, running on the HIP kernels.

``MHSAAddAtt`` / ``CNNAddAtt`` / ``PLM`` mirror ``newsreclib.models.components.encoders.news.text``
(text.py:179-236, 112-176, 15-109), ``LinearEncoder`` mirrors ``...news.category.LinearEncoder``
(category.py:9-80) and ``NewsEncoder`` mirrors ``...encoders.news.news.NewsEncoder`` (news.py:9-183)
for the NRMS configuration (one text attribute) and the LSTUR one (shared text encoder over title +
abstract, category embedding, ``combine_type="concat"``).  Constructor signatures, attribute names
and ``state_dict`` keys are the reference's.
"""
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib, ops, ops_blocks, ops_lstur
from .attention import AdditiveAttention

# dropout stream pair of each text attribute (fixed by NAME: the reference iterates its text encoders
# in Python-set order, which must not change which random numbers an attribute sees)
TEXT_STREAMS = {"title": 0, "abstract": 2, "text": 0}


def _draw_seed() -> int:
    # host-side draw from torch's CPU generator (no device sync); reproducible under torch.manual_seed
    return int(torch.empty((), dtype=torch.int64).random_(0, 2 ** 62))


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
        self.table_grad_hook = None   # optional callable(table_grad, ids) (or callable(table_grad)): called from the backward once the
        # table gradient is complete, before the weight-gradient GEMMs (trainer.NRMSTrainer starts its all-reduce there)
        self._tt_buf = self._tt_key = self._tt_seen_key = None       # token q|k|v table of evaluation forwards (below)
        self._tt_seen, self._tt_pinned = 0, False

    def _params(self):
        mha, att = self.multihead_attention, self.additive_attention
        return (self.embedding_layer.weight, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight,
                mha.out_proj.bias, att.linear.weight, att.linear.bias, att.query)

    # ---- evaluation forwards from a per-token q|k|v table (round 6; ops.token_table_build) ------------------------------
    # Without dropout the q|k|v rows of a token position depend on its token id alone, and validation / test encode every news
    # of every impression under frozen weights (rec_dataset.py:98-121, nrms_module.py:398-535): the in-projection runs once per
    # VOCABULARY id and per weight version instead of once per position and forward.  The result is torch.equal to the table-less
    # forward.  Two ways in:
    #   * ``with encoder.token_table():`` -- the caller vouches that the weights do not change inside (an evaluation epoch:
    #     ``AbstractRecommender.on_validation_epoch_start`` / ``_end``, ``evaluation.NewsVectorCache.build``);
    #   * automatically, for forwards with the module in eval mode under ``torch.no_grad()``, keyed on every parameter's storage
    #     and version counter and on this library's optimizer-step generation -- for TRAINABLE weights only while this
    #     library's optimizer is the writer (``ops_blocks.step_images_allowed``: a foreign ``p.data`` writer moves no counter),
    #     and only once the positions seen under the current weights reach the vocabulary size (a table build costs one
    #     in-projection over V rows).  ``NRL_TOKEN_TABLE=0`` turns the automatic route off.
    TOKEN_TABLE_USES = {"built": 0, "forwards": 0}

    def _token_table_key(self, params):
        return (tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in params[:-1]), _lib.engine_code(), _lib.options_word(),
                ops_blocks._IMAGE_GENERATION[0], ops_blocks._STEP_GENERATION[0], int(self.num_heads))

    def token_table(self):
        """Context manager: forwards inside (eval mode, no grad) run from ONE table built at the first of them."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            prev = self._tt_pinned
            self._tt_pinned = True
            try:
                yield self
            finally:
                self._tt_pinned = prev
        return scope()

    def drop_token_table(self) -> None:
        self._tt_buf = self._tt_key = None

    def _forward_from_token_table(self, text: torch.Tensor, params) -> Optional[torch.Tensor]:
        if not (torch.is_tensor(text) and text.is_cuda and text.dim() == 2 and text.dtype == torch.int64):
            return None
        emb = params[0]
        V, D = emb.shape
        N, L = text.shape
        if not ops.token_table_supported(L, D, self.num_heads, params[5].shape[0]):
            return None
        pinned = self._tt_pinned
        if not pinned:
            if os.environ.get("NRL_TOKEN_TABLE", "1") == "0":
                return None
            if any(t.requires_grad for t in params[:-1]) and not ops_blocks.step_images_allowed(*params[:-1]):
                return None
        key = self._token_table_key(params)
        if self._tt_buf is None or self._tt_key != key:
            if not pinned:
                if self._tt_seen_key != key:
                    self._tt_seen_key, self._tt_seen = key, 0
                self._tt_seen += N * L
                if self._tt_seen < V:
                    return None
            self._tt_key = None                 # (a failed build must not leave the old key over a half-written table)
            self._tt_buf = ops.token_table_build(params, self.num_heads, self._tt_buf)
            if self._tt_buf is None:
                return None
            self._tt_key = key
            MHSAAddAtt.TOKEN_TABLE_USES["built"] += 1
        MHSAAddAtt.TOKEN_TABLE_USES["forwards"] += 1
        return ops.news_encoder_fwd_table(text, self._tt_buf, V, params[1:], self.num_heads)

    def forward(self, text: torch.Tensor, seed: Optional[int] = None,
                order: Optional[torch.Tensor] = None, stream0: int = 0) -> torch.Tensor:
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 and seed is None:
            seed = _draw_seed()
        params = self._params()
        if not self.training and not torch.is_grad_enabled():
            out = self._forward_from_token_table(text, params)
            if out is not None:
                return out
        return ops.NewsEncoderFn.apply(text, *params, self.num_heads, p, seed or 0, stream0, _grad_bufs(params),
                                       order, self.table_grad_hook)


class CNNAddAtt(nn.Module):
    """Embedding lookup -> dropout -> Conv2d over (window, embed_dim) -> ReLU -> dropout -> additive
    attention (reference text.py:112-176) as one HIP pipeline (``nrl_cnn_encoder_fwd``/``_bwd``): the
    convolution is a single GEMM with K = window * embed_dim over overlapping token rows."""

    def __init__(self, pretrained_embeddings: torch.Tensor, embed_dim: int, num_filters: int, window_size: int,
                 query_dim: int, dropout_probability: float) -> None:
        super().__init__()
        if not isinstance(dropout_probability, float):
            raise ValueError(
                f"Expected keyword argument `dropout_probability` to be a `float` but got {dropout_probability}")
        self.embedding_layer = nn.Embedding.from_pretrained(
            torch.as_tensor(pretrained_embeddings, dtype=torch.float32), freeze=False, padding_idx=0)
        # nn.Conv2d is the parameter container only (names, shapes, default init of text.py:148-153)
        self.cnn = nn.Conv2d(in_channels=1, out_channels=num_filters, kernel_size=(window_size, embed_dim),
                             padding=(int((window_size - 1) / 2), 0))
        self.additive_attention = AdditiveAttention(input_dim=num_filters, query_dim=query_dim)
        self.dropout = nn.Dropout(dropout_probability)

    def _params(self):
        att = self.additive_attention
        return (self.embedding_layer.weight, self.cnn.weight, self.cnn.bias, att.linear.weight, att.linear.bias,
                att.query)

    def forward(self, text: torch.Tensor, seed: Optional[int] = None, order: Optional[torch.Tensor] = None,
                stream0: int = 0) -> torch.Tensor:
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 and seed is None:
            seed = _draw_seed()
        params = self._params()
        return ops_lstur.CnnEncoderFn.apply(text, *params, p, seed or 0, stream0, _grad_bufs(params), order)


class CNNMHSAAddAtt(nn.Module):
    """Embedding lookup -> dropout -> Conv1d over tokens -> ReLU -> dropout -> multi-head self-attention over the
    tokens of each news -> dropout -> additive attention (reference text.py:239-309, CenNewsRec) as one HIP
    pipeline (``nrl_cnn_mhsa_encoder_fwd``/``_bwd``).  ``cnn`` is an ``nn.Conv1d`` as in the reference (weight
    (F, D, W)); the kernels take the (F, W, D) order, so the weight is permuted on the way in (1.4 MB, plumbing)
    and its gradient comes back through the same permutation."""

    def __init__(self, pretrained_embeddings: torch.Tensor, embed_dim: int, num_filters: int, window_size: int,
                 num_heads: int, query_dim: int, dropout_probability: float) -> None:
        super().__init__()
        if not isinstance(dropout_probability, float):
            raise ValueError(
                f"Expected keyword argument `dropout_probability` to be a `float` but got {dropout_probability}")
        if window_size != 3:
            raise NotImplementedError("the reference hard-codes padding=1 (text.py:283): only window_size=3 keeps the "
                                      "token count, which the attention block needs")
        self.embedding_layer = nn.Embedding.from_pretrained(
            torch.as_tensor(pretrained_embeddings, dtype=torch.float32), freeze=False, padding_idx=0)
        self.cnn = nn.Conv1d(in_channels=embed_dim, out_channels=num_filters, kernel_size=window_size, padding=1)
        self.multihead_attention = nn.MultiheadAttention(embed_dim=num_filters, num_heads=num_heads)
        self.additive_attention = AdditiveAttention(input_dim=num_filters, query_dim=query_dim)
        self.dropout = nn.Dropout(dropout_probability)
        self.num_heads = num_heads

    def forward(self, text: torch.Tensor, seed: Optional[int] = None, order: Optional[torch.Tensor] = None,
                stream0: int = 0) -> torch.Tensor:
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 and seed is None:
            seed = _draw_seed()
        mha, att = self.multihead_attention, self.additive_attention
        w_c = self.cnn.weight.permute(0, 2, 1).contiguous().unsqueeze(1)      # (F, D, W) -> (F, 1, W, D)
        block = (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, att.linear.weight,
                 att.linear.bias, att.query)
        emb = self.embedding_layer.weight
        bufs = (getattr(emb, "main_grad", None), None, getattr(self.cnn.bias, "main_grad", None)) + \
            tuple(getattr(t, "main_grad", None) for t in block)
        return ops_lstur.CnnMhsaEncoderFn.apply(text, emb, w_c, self.cnn.bias, *block, self.num_heads, p, seed or 0,
                                                stream0, bufs if any(b is not None for b in bufs) else None, order)


class LinearEncoder(nn.Module):
    """Category encoder (reference category.py:9-80) for the configuration the recommenders in scope use
    (LSTUR, lstur_module.py:173-183): a trainable ``nn.Embedding(padding_idx=0)`` lookup, no dropout, no
    linear transform."""

    def __init__(self, pretrained_embeddings: Optional[torch.Tensor], from_pretrained: bool,
                 freeze_pretrained_emb: bool, num_categories: int, embed_dim: Optional[int], use_dropout: bool,
                 dropout_probability: Optional[float], linear_transform: bool, output_dim: Optional[int]) -> None:
        super().__init__()
        if use_dropout:
            raise NotImplementedError("newsreclib_amd.LinearEncoder covers use_dropout=False (LSTUR, NAML)")
        if from_pretrained:
            assert isinstance(pretrained_embeddings, torch.Tensor)
            self.embedding_layer = nn.Embedding.from_pretrained(
                embeddings=pretrained_embeddings, freeze=freeze_pretrained_emb, padding_idx=0)
        else:
            assert isinstance(embed_dim, int) and embed_dim > 0
            self.embedding_layer = nn.Embedding(num_embeddings=num_categories, embedding_dim=embed_dim, padding_idx=0)
        self.use_dropout, self.linear_transform = use_dropout, linear_transform
        if self.linear_transform:
            assert isinstance(output_dim, int)
            self.linear = nn.Linear(in_features=self.embedding_layer.weight.shape[1], out_features=output_dim)

    def forward(self, category: torch.Tensor) -> torch.Tensor:
        w = self.embedding_layer.weight
        vec = ops_lstur.EmbeddingRowsFn.apply(category, w, 0.0, 0, 0, _grad_bufs((w,)))
        if self.linear_transform:                                   # F.relu(self.linear(x)), category.py:78-80
            lp = (self.linear.weight, self.linear.bias)
            vec = ops_blocks.LinearActFn.apply(vec, *lp, "relu", _grad_bufs(lp))
        return vec


class NrlLinear(nn.Module):
    """Drop-in for an ``nn.Linear`` of a third-party stack (the transformer body of the PLM text encoder): the SAME
    ``weight`` / ``bias`` Parameter objects (state-dict keys, freezing and optimizer membership unchanged), the GEMMs on
    this library's engines (``ops_blocks.LinearFn`` -> ``nrl_linear_fwd`` / ``nrl_linear_bwd``)."""

    def __init__(self, linear: nn.Linear) -> None:
        super().__init__()
        if linear.bias is None or linear.in_features % 4 or linear.out_features % 4:
            raise ValueError("NrlLinear: needs a bias and in/out features that are multiples of 4")
        self.in_features, self.out_features = linear.in_features, linear.out_features
        self.weight, self.bias = linear.weight, linear.bias
        # a frozen weight's matrix-core images are built once, not twice per step (NRL_PLM_IMAGE_CACHE=0: every call, A/B)
        self._images = ops_blocks.FrozenImages() if os.environ.get("NRL_PLM_IMAGE_CACHE", "1") != "0" else None
        # ... and a trainable weight's within one optimizer step (two encoder calls per step; NRL_PLM_STEP_IMAGES=0: rebuilt per call)
        self._step_images = ops_blocks.FrozenImages(allow_trainable=True) \
            if os.environ.get("NRL_PLM_STEP_IMAGES", "1") != "0" and self._images is not None else None

    def _load_from_state_dict(self, *args, **kwargs):
        # load_state_dict copies into the Parameter under no_grad (the version counter moves) -- but a caller may also have
        # swapped storage behind it; a load is rare and an image build is 10 us, so drop the cached images outright
        if self._images is not None:
            self._images.invalidate()
        if self._step_images is not None:
            self._step_images.invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype != torch.float32 or not x.is_cuda:
            # meta / CPU construction-time calls of HF.  Counted: a GPU step that lands here is NOT on this library's engine
            # (bench.py's `plm` extra and test_plm_full_width_step report the counters, which must stay 0 on the device)
            FALLBACK_CALLS["linear_cuda" if x.is_cuda else "linear_host"] += 1
            return nn.functional.linear(x, self.weight, self.bias)
        params = (self.weight, self.bias)
        trainable = self.weight.requires_grad or self.bias.requires_grad
        if trainable and self._images is not None:
            self._images.invalidate()        # trained now, maybe frozen again later: never meet an image of the old values
        # (a trainable weight's images survive a call only when this library's optimizer drives the step: ops_blocks.step_images_allowed)
        images = self._images if not trainable else (self._step_images if ops_blocks.step_images_allowed(self.weight, self.bias) else None)
        return ops_blocks.LinearFn.apply(x.contiguous(), self.weight, self.bias, _grad_bufs(params), images)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, engine=newsreclib_amd"


class NrlEmbedding(nn.Module):
    """Drop-in for an ``nn.Embedding`` of the PLM body: the SAME ``weight`` Parameter (state-dict key, optimizer membership
    unchanged), lookup and gradient on this library (``ops_blocks.EmbeddingFn``)."""

    def __init__(self, emb: nn.Embedding) -> None:
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.padding_idx = emb.num_embeddings, emb.embedding_dim, emb.padding_idx
        self.weight = emb.weight

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        w = self.weight
        if not ids.is_cuda or ids.dtype != torch.int64 or w.dtype != torch.float32:
            # construction-time / CPU calls of HF.  Counted like NrlLinear's: a GPU step that lands here is not on this library
            FALLBACK_CALLS["embedding_cuda" if ids.is_cuda else "embedding_host"] += 1
            return nn.functional.embedding(ids, w, self.padding_idx)
        return ops_blocks.EmbeddingFn.apply(ids.contiguous(), w, self.padding_idx, _grad_bufs((w,)))

    def extra_repr(self) -> str:
        return f"{self.num_embeddings}, {self.embedding_dim}, padding_idx={self.padding_idx}, engine=newsreclib_amd"


def swap_embeddings(module: nn.Module) -> int:
    """Replaces the plain ``nn.Embedding`` tables below ``module`` (dim % 4 == 0, dim <= 1024, <= 2^20 rows, no max_norm /
    scale_grad_by_freq / sparse) by ``NrlEmbedding`` in place; returns how many."""
    n = 0
    for name, child in list(module.named_children()):
        if type(child) is nn.Embedding and child.embedding_dim % 4 == 0 and child.embedding_dim <= 1024 \
                and child.num_embeddings <= (1 << 20) and child.max_norm is None and not child.scale_grad_by_freq \
                and not child.sparse:
            setattr(module, name, NrlEmbedding(child))
            n += 1
        else:
            n += swap_embeddings(child)
    return n


def swap_linears(module: nn.Module) -> int:
    """Replaces every eligible ``nn.Linear`` below ``module`` by ``NrlLinear`` in place; returns how many."""
    n = 0
    for name, child in list(module.named_children()):
        if isinstance(child, nn.Linear) and child.bias is not None and child.in_features % 4 == 0 \
                and child.out_features % 4 == 0:
            setattr(module, name, NrlLinear(child))
            n += 1
        else:
            n += swap_linears(child)
    return n


def _fused_output_forward(self, hidden_states: torch.Tensor, input_tensor: torch.Tensor) -> torch.Tensor:
    """``LayerNorm(dropout(dense(hidden_states)) + input_tensor)`` of a BERT-family output block with the three elementwise
    passes after the projection as ONE launch (``ops_blocks.DropoutAddLayerNormFn``)."""
    return _output_glue(self, self.dense(hidden_states), input_tensor)


def _output_glue(self, h: torch.Tensor, input_tensor: torch.Tensor) -> torch.Tensor:
    ln = self.LayerNorm
    if h.dtype != torch.float32 or not h.is_cuda or ln.weight is None or ln.bias is None or h.shape[-1] % 4 or h.shape[-1] > 2048 \
            or h.numel() >= (1 << 32):
        FALLBACK_CALLS["output_block_cuda" if h.is_cuda else "output_block_host"] += 1
        return ln(self.dropout(h) + input_tensor)
    p = float(self.dropout.p) if self.training else 0.0
    params = (ln.weight, ln.bias)
    return ops_blocks.DropoutAddLayerNormFn.apply(h.contiguous(), input_tensor.contiguous(), ln.weight, ln.bias, float(ln.eps), p,
                                                  _draw_seed() if p > 0.0 else 0, _grad_bufs(params))


def _fused_ffn_chunk(self, attention_output: torch.Tensor) -> torch.Tensor:
    """``RobertaLayer.feed_forward_chunk``: intermediate.dense -> GELU -> output.dense as ONE autograd function with the GELU and its
    derivative inside the GEMM epilogues (``ops_blocks.FfnFn``), then the fused dropout + residual + LayerNorm of the output block."""
    d1, d2 = self.intermediate.dense, self.output.dense
    x = attention_output
    if x.dtype != torch.float32 or not x.is_cuda or not bool(_lib.load().nrl_linear_gelu_supported(d1.out_features)):
        FALLBACK_CALLS["ffn_cuda" if x.is_cuda else "ffn_host"] += 1
        return self.output(self.intermediate(x), x)
    p = (d1.weight, d1.bias, d2.weight, d2.bias)

    def images(lin):
        trainable = lin.weight.requires_grad or lin.bias.requires_grad
        if trainable and lin._images is not None:
            lin._images.invalidate()
        return lin._images if not trainable else (lin._step_images if ops_blocks.step_images_allowed(lin.weight, lin.bias) else None)

    out, ln = self.output, self.output.LayerNorm
    if _FFN_GLUE and ln.weight is not None and ln.bias is not None and x.shape[-1] % 4 == 0 and x.shape[-1] <= 2048 and x.numel() < (1 << 32) \
            and d2.out_features == d1.in_features:
        # ... and the layer's closing dropout + residual + LayerNorm inside the same function (the first projection's activation
        # gradient takes the residual branch's gradient in its epilogue)
        pd = float(out.dropout.p) if out.training else 0.0
        pp = p + (ln.weight, ln.bias)
        return ops_blocks.FfnBlockFn.apply(x.contiguous(), *pp, float(ln.eps), pd, _draw_seed() if pd > 0.0 else 0, _grad_bufs(pp),
                                           images(d1), images(d2))
    y = ops_blocks.FfnFn.apply(x.contiguous(), *p, _grad_bufs(p), images(d1), images(d2))
    return _output_glue(self.output, y, x)


_FFN_GLUE = os.environ.get("NRL_PLM_FFN_GLUE", "1") != "0"      # (A/B: the closing glue inside the feed-forward function)


def swap_ffn_blocks(module: nn.Module) -> int:
    """Gives every layer below ``module`` whose feed-forward half is ``intermediate`` (NrlLinear + exact GELU) -> ``output`` (NrlLinear
    + the fused glue) the one-function form above; returns how many.  Parameters and state-dict keys are untouched."""
    import types
    n = 0
    for m in module.modules():
        inter, out = getattr(m, "intermediate", None), getattr(m, "output", None)
        if inter is None or out is None or not hasattr(m, "feed_forward_chunk"):
            continue
        act = getattr(inter, "intermediate_act_fn", None)
        exact_gelu = act is nn.functional.gelu or type(act).__name__ == "GELUActivation" or isinstance(act, nn.GELU) and act.approximate == "none"
        if isinstance(getattr(inter, "dense", None), NrlLinear) and isinstance(getattr(out, "dense", None), NrlLinear) and exact_gelu \
                and isinstance(getattr(out, "LayerNorm", None), nn.LayerNorm) and isinstance(getattr(out, "dropout", None), nn.Dropout):
            m.feed_forward_chunk = types.MethodType(_fused_ffn_chunk, m)
            n += 1
    return n


def swap_output_blocks(module: nn.Module) -> int:
    """Gives every ``dense -> dropout -> LayerNorm(. + residual)`` block below ``module`` (``RobertaSelfOutput`` /
    ``RobertaOutput`` and their BERT-family namesakes: attributes ``dense``, ``dropout``, ``LayerNorm`` and a
    ``forward(hidden_states, input_tensor)``) the fused forward above; returns how many.  Parameters, state-dict keys and
    freezing are untouched (only ``forward`` of the block instance is rebound)."""
    import inspect
    import types
    n = 0
    for m in module.modules():
        if isinstance(getattr(m, "LayerNorm", None), nn.LayerNorm) and isinstance(getattr(m, "dropout", None), nn.Dropout) \
                and isinstance(getattr(m, "dense", None), nn.Module):
            try:
                names = list(inspect.signature(m.forward).parameters)
            except (TypeError, ValueError):
                continue
            if names[:2] == ["hidden_states", "input_tensor"] and len(names) == 2:
                m.forward = types.MethodType(_fused_output_forward, m)
                n += 1
    return n


NRL_ATTENTION = "nrl_x3"
# calls of the PLM body that did NOT run on this library's kernels (framework fallbacks), by kind; `reset_fallback_calls()` zeroes
FALLBACK_CALLS = {"linear_cuda": 0, "linear_host": 0, "attention": 0, "output_block_cuda": 0, "output_block_host": 0,
                  "embedding_cuda": 0, "embedding_host": 0, "ffn_cuda": 0, "ffn_host": 0, "attention_block_cuda": 0,
                  "attention_block_host": 0}


# `PLM.share_body`: how often ONE body pass served the encoder calls of a step -- with the calls already at one sequence length,
# after padding the shorter calls to the longest -- and how often it was refused (each call then pays its own body pass)
SHARE_BODY_CALLS = {"hit_same_length": 0, "hit_padded": 0, "miss": 0}


def reset_fallback_calls() -> None:
    for k in FALLBACK_CALLS:
        FALLBACK_CALLS[k] = 0
    for k in SHARE_BODY_CALLS:
        SHARE_BODY_CALLS[k] = 0


def _key_padding_keep(module, attention_mask, N: int, L: int):
    """-> (ok, keep): the (N, L) uint8 key-padding mask (1 = attend; None: no mask) behind HF's 4-D mask of a bidirectional encoder,
    or ok == False for a mask that is not one.  HF materialises the encoder's mask as (N, 1, L, L); whether its query rows are all the
    same row is a property of the MODEL (encoder vs decoder), so it is verified on the first call of each module (one device
    comparison + sync) and remembered on the module."""
    if attention_mask is None:
        return True, None
    m = attention_mask
    if not (m.dim() == 4 and m.shape[0] == N and m.shape[1] == 1 and m.shape[3] == L and m.shape[2] in (1, L)):
        return False, None
    if m.shape[2] == L:
        invariant = getattr(module, "_nrl_mask_query_invariant", None)
        if invariant is None:
            invariant = bool(torch.equal(m, m[:, :, :1, :].expand_as(m)))
            try:
                module._nrl_mask_query_invariant = invariant
            except Exception:
                pass
        if not invariant:
            return False, None
    row = m[:, 0, 0, :]
    return True, (row if row.dtype == torch.bool else (row == 0)).to(torch.uint8)


def _nrl_body_attention(module, query, key, value, attention_mask, dropout: float = 0.0, scaling=None, **kwargs):
    """HF attention interface (``ALL_ATTENTION_FUNCTIONS[config._attn_implementation]``) of the PLM body on this library's
    bf16x3 attention kernels: query / key / value arrive as (N, H, L, dh) transposes of the projections' (N, L, H*dh) outputs;
    returns (N, L, H, dh).  The body is a bidirectional encoder, so its 4-D mask is a key-padding mask repeated over the query
    rows: row 0 is read as (N, L).  Shapes, devices or engines the kernels do not cover go to the framework's SDPA path."""
    from transformers.integrations.sdpa_attention import sdpa_attention_forward
    N, H, L, dh = query.shape
    ok = (query.is_cuda and query.dtype == torch.float32 and key.shape == query.shape and value.shape == query.shape
          and _lib.engine_code() == 2 and ops_blocks.sdpa_supported(N, L, H, dh))     # (2 = the bf16x3 engine)
    # a causal / query-dependent mask is NOT a key-padding mask
    if kwargs.get("is_causal") or getattr(module, "is_causal", False):
        ok = False
    keep = None
    if ok:
        ok, keep = _key_padding_keep(module, attention_mask, N, L)
    if ok and float(dropout) > 0.0 and N * H >= (1 << 18):     # (the kernels' 32-bit dropout counter: nrl_sdpa_x3.hip)
        ok = False
    if not ok:
        FALLBACK_CALLS["attention"] += 1
        return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, **kwargs)
    scale = float(scaling) if scaling is not None else float(dh) ** -0.5
    p = float(dropout)
    out = ops_blocks.SdpaFn.apply(query.transpose(1, 2), key.transpose(1, 2), value.transpose(1, 2), keep, scale, p,
                                  _draw_seed() if p > 0.0 else 0)
    return out, None


def _fused_attention_forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None,
                             past_key_values=None, **kwargs):
    """``RobertaAttention.forward`` (self-attention + ``RobertaSelfOutput``) as ONE autograd function, ``ops_blocks.AttnBlockFn``: the
    query / key / value projections as one GEMM each way, the residual branch's gradient added in the activation-gradient epilogue.
    Anything the function does not cover (cross attention, caches, attention weights asked for, causal or query-dependent masks,
    shapes off the kernels) runs the module's own forward -- which still uses this library's per-op replacements."""
    sa, so = self.self, self.output
    x = hidden_states
    ok = (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and encoder_hidden_states is None
          and past_key_values is None and not getattr(self, "is_cross_attention", False) and not kwargs.get("output_attentions")
          and not getattr(sa, "is_causal", False) and not kwargs.get("is_causal")
          and getattr(sa.config, "_attn_implementation", None) == NRL_ATTENTION and _lib.engine_code() == 2)
    keep = None
    if ok:
        N, L, K = x.shape
        H, dh = sa.num_attention_heads, sa.attention_head_size
        ln = so.LayerNorm
        ok = (ops_blocks.sdpa_supported(N, L, H, dh) and bool(_lib.load().nrl_linear3_supported(H * dh, K))
              and so.dense.in_features == H * dh and so.dense.out_features == K
              and ln.weight is not None and ln.bias is not None and K % 4 == 0 and K <= 2048 and x.numel() < (1 << 32))
        p_attn = float(sa.dropout.p) if sa.training else 0.0
        if ok and p_attn > 0.0 and N * H >= (1 << 18):
            ok = False
        if ok:
            ok, keep = _key_padding_keep(sa, attention_mask, N, L)
    if not ok:
        FALLBACK_CALLS["attention_block_cuda" if (torch.is_tensor(x) and x.is_cuda) else "attention_block_host"] += 1
        return self._nrl_module_forward(hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask, past_key_values,
                                        **kwargs)
    lins = (sa.query, sa.key, sa.value)
    trainable = any(l.weight.requires_grad or l.bias.requires_grad for l in lins)
    if trainable:
        self._nrl_qkv_images.invalidate()
    img_qkv = self._nrl_qkv_images if not trainable else (
        self._nrl_qkv_step_images if ops_blocks.step_images_allowed(*[t for l in lins for t in (l.weight, l.bias)]) else None)
    od = so.dense
    o_train = od.weight.requires_grad or od.bias.requires_grad
    if o_train and od._images is not None:
        od._images.invalidate()
    img_o = od._images if not o_train else (od._step_images if ops_blocks.step_images_allowed(od.weight, od.bias) else None)
    p_hid = float(so.dropout.p) if so.training else 0.0
    params = (sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias, od.weight, od.bias,
              ln.weight, ln.bias)
    y = ops_blocks.AttnBlockFn.apply(x.contiguous(), keep, *params, H, float(sa.scaling), p_attn, _draw_seed() if p_attn > 0.0 else 0,
                                     float(ln.eps), p_hid, _draw_seed() if p_hid > 0.0 else 0, _grad_bufs(params), img_qkv, img_o)
    return y, None


def swap_attention_blocks(module: nn.Module) -> int:
    """Gives every ``RobertaAttention``-shaped module below ``module`` (``self`` with NrlLinear query / key / value, ``output`` with an
    NrlLinear ``dense`` + dropout + LayerNorm) the one-function forward above; returns how many.  Parameters and state-dict keys are
    untouched; the module's own forward stays reachable for everything the function does not cover."""
    import types
    n = 0
    for m in module.modules():
        sa, so = getattr(m, "self", None), getattr(m, "output", None)
        if sa is None or so is None or not all(isinstance(getattr(sa, k, None), NrlLinear) for k in ("query", "key", "value")):
            continue
        if not (isinstance(getattr(so, "dense", None), NrlLinear) and isinstance(getattr(so, "LayerNorm", None), nn.LayerNorm)
                and isinstance(getattr(so, "dropout", None), nn.Dropout) and isinstance(getattr(sa, "dropout", None), nn.Dropout)
                and hasattr(sa, "num_attention_heads") and hasattr(sa, "attention_head_size") and hasattr(sa, "scaling")
                and hasattr(sa, "config") and so.dense._images is not None):
            continue
        if sa.query.in_features != sa.key.in_features or sa.query.in_features != sa.value.in_features \
                or not (sa.query.out_features == sa.key.out_features == sa.value.out_features):
            continue
        m._nrl_module_forward = m.forward
        m._nrl_qkv_images = ops_blocks.FrozenImages()
        m._nrl_qkv_step_images = ops_blocks.FrozenImages(allow_trainable=True)
        m.forward = types.MethodType(_fused_attention_forward, m)
        n += 1
    return n


def register_body_attention() -> bool:
    """Registers ``NRL_ATTENTION`` with the HF attention / mask interfaces (idempotent); False when this transformers
    version has no such registry (the body then keeps its own attention)."""
    try:
        from transformers.masking_utils import AttentionMaskInterface, sdpa_mask
        from transformers.modeling_utils import AttentionInterface
    except ImportError:
        return False
    AttentionInterface.register(NRL_ATTENTION, _nrl_body_attention)
    AttentionMaskInterface.register(NRL_ATTENTION, sdpa_mask)
    return True


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
        # the body's projections (q/k/v/out, the two feed-forward layers: > 90 % of a config-4 step) on this library's
        # matrix-core engine instead of fp32 hipBLASLt; NRL_PLM_LINEAR=0 keeps the HF modules (A/B)
        self.nrl_linears = 0
        if os.environ.get("NRL_PLM_LINEAR", "1") != "0" and hasattr(self.plm_model, "encoder"):
            self.nrl_linears = swap_linears(self.plm_model.encoder)
        # ... the word / position / token-type tables (trainable: text.py:70-73 freezes `layer.k.` names only) with this library's
        # sorted-segment embedding gradient instead of ATen's merge sort + segment kernels; NRL_PLM_EMBEDDING=0 keeps them (A/B)
        self.nrl_embeddings = 0
        if os.environ.get("NRL_PLM_EMBEDDING", "1") != "0" and hasattr(self.plm_model, "embeddings"):
            self.nrl_embeddings = swap_embeddings(self.plm_model.embeddings)
        # ... the dropout + residual + LayerNorm that ends both halves of every layer as one launch each way
        # (nrl_dropout_add_layernorm_fwd / _bwd); NRL_PLM_GLUE=0 keeps the three framework kernels (A/B)
        self.nrl_output_blocks = 0
        if os.environ.get("NRL_PLM_GLUE", "1") != "0" and hasattr(self.plm_model, "encoder"):
            self.nrl_output_blocks = swap_output_blocks(self.plm_model.encoder)
        # ... the feed-forward half of every layer as one autograd function with the GELU (and its derivative) inside the GEMM
        # epilogues; NRL_PLM_FFN=0 keeps the two projections + the framework's GELU kernels (A/B)
        self.nrl_ffn_blocks = 0
        if os.environ.get("NRL_PLM_FFN", "1") != "0" and self.nrl_linears and self.nrl_output_blocks and hasattr(self.plm_model, "encoder"):
            self.nrl_ffn_blocks = swap_ffn_blocks(self.plm_model.encoder)
        # ... and its self-attention on this library's bf16x3 kernels (nrl_sdpa_fwd / _bwd: heads of 64 over <= 128 tokens; other
        # shapes fall through to the framework's SDPA inside the interface); NRL_PLM_ATTENTION=0 keeps the HF path (A/B)
        self.nrl_attention = False
        if os.environ.get("NRL_PLM_ATTENTION", "1") != "0" and register_body_attention():
            try:
                self.plm_model.config._attn_implementation = NRL_ATTENTION
                self.nrl_attention = True
            except Exception:      # (a transformers version that validates the name against a closed list)
                self.nrl_attention = False
        # ... and the attention half of every layer as one autograd function (one q | k | v GEMM each way, the residual gradient in the
        # activation-gradient epilogue); NRL_PLM_ATTN_BLOCK=0 keeps the per-op modules (A/B)
        self.nrl_attention_blocks = 0
        if os.environ.get("NRL_PLM_ATTN_BLOCK", "1") != "0" and self.nrl_attention and self.nrl_linears and self.nrl_output_blocks \
                and os.environ.get("NRL_PLM_IMAGE_CACHE", "1") != "0" and hasattr(self.plm_model, "encoder"):
            self.nrl_attention_blocks = swap_attention_blocks(self.plm_model.encoder)
        for name, param in self.plm_model.base_model.named_parameters():   # text.py:69-73
            for layer in (frozen_layers or []):
                if "layer." + str(layer) + "." in name:
                    param.requires_grad = False
        assert isinstance(num_heads, int) and num_heads > 0
        self.multihead_attention = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=num_heads)
        self.additive_attention = AdditiveAttention(input_dim=embed_dim, query_dim=query_dim)
        self.dropout = nn.Dropout(p=dropout_probability)
        self.num_heads = num_heads

    def share_body(self, texts) -> bool:
        """Runs the transformer body ONCE over the news of several upcoming ``forward`` calls (round 5).  The reference calls this
        encoder twice per step (history, candidates: nrms_module.py:232,236) and the two calls must stay two for everything AFTER the
        body -- its seq-first attention runs across the news of a call (text.py:92-96) -- but the body itself treats every news on its
        own (self-attention within a news, key-padding mask per news), so its hidden states over [history; candidates] are row for row
        those of the two calls.  The small call is the inefficient one (3,840 token rows at B = 8: its GEMMs run at a third to two
        thirds of the large call's rate and it doubles the launches and the trainable layers' image builds).

        Real batches pad each call to ITS OWN longest text (rec_dataset.py:181), so the calls' sequence lengths normally differ
        (round 6): the shorter calls are then right-padded to the longest with the body's padding id under attention mask 0 --
        masked keys get weight exactly 0 and the position ids of a roberta-type body come from the ids (padding -> padding_idx), so
        the real columns are unchanged -- and every call gets back ITS OWN columns ``[:, :L_call]`` (the tail attends over token
        positions without a mask: it must not see the extra ones).  Refused (``SHARE_BODY_CALLS["miss"]``; each call then runs its own
        body pass) when the padding would add more than a quarter to the token rows, when the body has no padding id, or when a text
        carries a per-token tensor this function does not know how to pad.  ``forward`` picks its rows up by the identity of
        ``input_ids``.  NRL_PLM_SHARE_BODY=0: off."""
        self._shared = []
        if len(texts) < 2 or os.environ.get("NRL_PLM_SHARE_BODY", "1") == "0":
            return False
        pad_values = {"attention_mask": 0, "token_type_ids": 0}
        pad_id = getattr(getattr(self.plm_model, "config", None), "pad_token_id", None)
        groups = {}
        for t in texts:
            if not (isinstance(t, dict) and torch.is_tensor(t.get("input_ids")) and t["input_ids"].dim() == 2
                    and all(torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == t["input_ids"].shape[0] for v in t.values())):
                continue
            # same keys, dtypes, device and trailing shape PAST the token axis; the token axis itself may differ
            sig = tuple(sorted((k, tuple(v.shape[2:]), str(v.dtype), str(v.device), v.dim()) for k, v in t.items()))
            groups.setdefault(sig, []).append(t)
        shared = False
        for group in groups.values():        # (e.g. title texts of both calls in one pass, abstract texts in another)
            if len(group) < 2:
                continue
            lens = [int(t["input_ids"].shape[1]) for t in group]
            L_max = max(lens)
            padded = any(l != L_max for l in lens)
            if padded:
                rows = sum(int(t["input_ids"].shape[0]) * l for t, l in zip(group, lens))
                extra = sum(int(t["input_ids"].shape[0]) * (L_max - l) for t, l in zip(group, lens))
                per_token = all(v.dim() == 1 or int(v.shape[1]) == l for t, l in zip(group, lens) for v in t.values())
                known = all(k == "input_ids" or k in pad_values or v.dim() == 1 for t in group for k, v in t.items())
                if pad_id is None or not per_token or not known or 4 * extra > rows:
                    SHARE_BODY_CALLS["miss"] += 1
                    continue

                def widen(t, l):
                    if l == L_max:
                        out = dict(t)
                    else:
                        out = {k: (v if v.dim() == 1 else torch.nn.functional.pad(
                            v, (0, L_max - l), value=(pad_id if k == "input_ids" else pad_values[k]))) for k, v in t.items()}
                    if "attention_mask" not in out:           # (a call without a mask attends to every one of ITS columns)
                        m = torch.zeros(out["input_ids"].shape, dtype=torch.int64, device=out["input_ids"].device)
                        m[:, :l] = 1
                        out["attention_mask"] = m
                    return out
                wide = [widen(t, l) for t, l in zip(group, lens)]
                merged = {k: torch.cat([w[k] for w in wide], dim=0) for k in wide[0]}
            else:
                merged = {k: torch.cat([t[k] for t in group], dim=0) for k in group[0]}
            hidden = self.plm_model(**merged)[0]
            # (split, not slices: its backward is ONE concatenation of the calls' gradients)
            for t, l, h in zip(group, lens, torch.split(hidden, [int(t["input_ids"].shape[0]) for t in group], dim=0)):
                self._shared.append((t["input_ids"], h if l == L_max else h[:, :l]))
            SHARE_BODY_CALLS["hit_padded" if padded else "hit_same_length"] += 1
            shared = True
        return shared

    def forward(self, text: Dict[str, torch.Tensor], seed: Optional[int] = None, order=None,
                stream0: int = 0) -> torch.Tensor:
        hidden = None
        shared = getattr(self, "_shared", None)
        if shared:
            for i, (ids, h) in enumerate(shared):
                if ids is text.get("input_ids"):
                    hidden = h.contiguous()
                    del shared[i]
                    break
        if hidden is None:
            hidden = self.plm_model(**text)[0]                      # (N, L, D)
        p = float(self.dropout.p) if self.training else 0.0
        if p > 0.0 and seed is None:
            seed = _draw_seed()
        mha, att = self.multihead_attention, self.additive_attention
        params = (mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                  att.linear.weight, att.linear.bias, att.query)
        return ops.UserEncoderFn.apply(hidden, *params, self.num_heads, _grad_bufs(params), p, seed or 0, True, stream0)


class NewsEncoder(nn.Module):
    """Dispatches news attributes to their encoders and combines the vectors (news.py:134-183).

    Built configurations: NRMS (one text attribute -> that encoder's output unchanged, news.py:159-160),
    LSTUR (ONE text encoder registered under every text attribute, a category encoder,
    ``combine_type="concat"``, news.py:69-79,128,181) and NAML (``combine_type="add_att"``: additive
    attention over the stacked view vectors, news.py:118-121,164-165).  Entity encoders and the ``linear``
    combine layer belong to recommenders that are out of this build's scope (they raise).

    Text-vector order: the reference fills its ``ModuleDict`` from a Python ``set`` (news.py:72-77), so the
    title/abstract order of the concatenation depends on the process's string-hash seed.  Here it is
    deterministic -- the order of ``attributes2encode`` -- and ``set_text_order`` pins any other order
    (e.g. the one a reference checkpoint was trained with)."""

    def __init__(self, dataset_attributes: List[str], attributes2encode: List[str],
                 concatenate_inputs: bool, text_encoder: Optional[nn.Module],
                 category_encoder: Optional[nn.Module], entity_encoder: Optional[nn.Module],
                 combine_vectors: bool, combine_type: Optional[str], input_dim: Optional[int],
                 query_dim: Optional[int], output_dim: Optional[int]) -> None:
        super().__init__()
        assert len(dataset_attributes) > 0
        self.concatenate_inputs = concatenate_inputs
        if entity_encoder is not None:
            raise NotImplementedError("newsreclib_amd.NewsEncoder: entity encoders are not built")
        if combine_vectors and combine_type not in ("concat", "add_att"):
            raise NotImplementedError("newsreclib_amd.NewsEncoder: combine_type must be 'concat' or 'add_att' "
                                      f"(got {combine_type!r})")
        self.encode_text = self.encode_category = self.encode_entity = False
        if ("title" in attributes2encode) or ("abstract" in attributes2encode):
            assert isinstance(text_encoder, nn.Module)
            if not concatenate_inputs:
                names = [a for a in attributes2encode if a in dataset_attributes and a in ("title", "abstract")]
                self.text_encoders = nn.ModuleDict({name: text_encoder for name in names})
            else:
                self.text_encoders = nn.ModuleDict({"text": text_encoder})
            self.encode_text = True
        if ("category" in attributes2encode) or ("subcategory" in attributes2encode):
            assert isinstance(category_encoder, nn.Module)
            names = [a for a in attributes2encode if a in dataset_attributes and a in ("category", "subcategory")]
            self.category_encoders = nn.ModuleDict({name: category_encoder for name in names})
            self.encode_category = True
        n_vec = (len(self.text_encoders) if self.encode_text else 0) + \
            (len(self.category_encoders) if self.encode_category else 0)
        if n_vec == 0:
            raise ValueError("no news attribute to encode")
        if n_vec > 1 and not combine_vectors:
            raise ValueError("several news attributes need combine_vectors=True")
        if combine_vectors:
            self.combine_type = combine_type
            if combine_type == "add_att":                           # news.py:118-121
                assert isinstance(input_dim, int) and input_dim > 0
                assert isinstance(query_dim, int) and query_dim > 0
                self.combine_layer = AdditiveAttention(input_dim=input_dim, query_dim=query_dim)

    def share_plm_bodies(self, *news_dicts) -> int:
        """Before several ``forward`` calls of one step (history news, candidate news): every PLM text encoder runs its transformer
        body once over the texts of all of them (``PLM.share_body``).  Returns how many encoders did."""
        if not self.encode_text:
            return 0
        groups = {}
        for name, enc in self.text_encoders.items():
            if isinstance(enc, PLM):
                texts = groups.setdefault(id(enc), (enc, []))[1]
                for nd in news_dicts:
                    if name in nd:
                        texts.append(nd[name])
        return sum(1 for enc, texts in groups.values() if enc.share_body(texts))

    def set_text_order(self, order) -> None:
        """Re-register the text encoders in the given attribute order (state_dict keys are unchanged)."""
        assert sorted(order) == sorted(self.text_encoders.keys())
        self.text_encoders = nn.ModuleDict({name: self.text_encoders[name] for name in order})

    def forward(self, news: Dict[str, torch.Tensor], seed: Optional[int] = None, stream_base: int = 0) -> torch.Tensor:
        """``stream_base`` shifts the dropout streams of this call (a recommender that must make SEPARATE history /
        candidate calls -- the PLM text encoder -- under one step seed gives the second call its own masks)."""
        vectors = []
        if self.encode_text:
            for name, encoder in self.text_encoders.items():
                kw = {}
                if seed is not None:
                    kw["seed"] = seed
                if news.get(name + "_order") is not None:   # optional argsort of the flat ids (prepare_batch)
                    kw["order"] = news[name + "_order"]
                if len(self.text_encoders) > 1 or stream_base:
                    kw["stream0"] = TEXT_STREAMS[name] + stream_base
                vectors.append(encoder(news[name], **kw))
        if self.encode_category:
            vectors += [encoder(news[name]) for name, encoder in self.category_encoders.items()]
        if len(vectors) == 1:
            return vectors[0]
        if self.combine_type == "add_att":                   # news.py:164-165: attention over the stacked views
            return self.combine_layer(torch.stack(vectors, dim=1))
        return torch.cat(vectors, dim=1)                     # news.py:128
