"""CPU restatement of the reference NRMS train step (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function cites the reference lines it restates (paths relative to the reference repo
``andreeaiana/newsreclib``).  The arithmetic is written out with elementary torch ops on CPU
fp32 (matmul / softmax / tanh) instead of ``nn.MultiheadAttention`` so that it is an independent
statement of SURVEY.md Appendix A; gradients come from torch autograd over those elementary ops.
``TorchGraphNRMS`` at the bottom is the module-for-module graph (``nn.Embedding`` /
``nn.MultiheadAttention`` seq-first / ``nn.Linear``) used as the timed CPU baseline.

Dropout: the reference uses ``nn.Dropout`` (torch RNG; ``text.py:220,225,230``).  The HIP path
cannot reproduce torch's RNG stream, so the product defines a counter-based keep mask
(``dropout_keep_mask`` below is its normative restatement).  With ``p == 0`` / eval mode this
oracle is exactly the reference forward; with explicit masks it is the reference forward with
``nn.Dropout`` replaced by "multiply by mask * 1/(1-p)", which is what ``make_golden.py`` injects
into the imported reference modules to pin the train-mode vectors.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# parameter names == reference state_dict keys (verified by instantiating the
# reference components)
# --------------------------------------------------------------------------------------
NEWS_PREFIX = "news_encoder.text_encoders.title."
USER_PREFIX = "user_encoder."
EMB_KEY = NEWS_PREFIX + "embedding_layer.weight"
BLOCK_KEYS = (
    "multihead_attention.in_proj_weight",
    "multihead_attention.in_proj_bias",
    "multihead_attention.out_proj.weight",
    "multihead_attention.out_proj.bias",
    "additive_attention.linear.weight",
    "additive_attention.linear.bias",
    "additive_attention.query",
)


def param_shapes(vocab: int, embed_dim: int = 300, query_dim: int = 200) -> Dict[str, Tuple[int, ...]]:
    d, q = embed_dim, query_dim
    block = {
        "multihead_attention.in_proj_weight": (3 * d, d),
        "multihead_attention.in_proj_bias": (3 * d,),
        "multihead_attention.out_proj.weight": (d, d),
        "multihead_attention.out_proj.bias": (d,),
        "additive_attention.linear.weight": (q, d),
        "additive_attention.linear.bias": (q,),
        "additive_attention.query": (q,),
    }
    shapes = {EMB_KEY: (vocab, d)}
    for pre in (NEWS_PREFIX, USER_PREFIX):
        for k, s in block.items():
            shapes[pre + k] = s
    return shapes


def make_params(vocab: int, embed_dim: int = 300, query_dim: int = 200, seed: int = 0,
                emb_scale: float = 1.0, w_scale: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """Portable parameter set: numpy ``default_rng(seed)`` draws in sorted key order.

    Embedding ~ N(0, emb_scale) (mirrors ``data_utils.py:56``); weights ~ N(0, w_scale) with
    ``w_scale`` defaulting to 1/sqrt(embed_dim); biases ~ N(0, 0.05); query ~ U(-0.1, 0.1)
    (``attention.py:22``).
    """
    rng = np.random.default_rng(seed)
    if w_scale is None:
        w_scale = 1.0 / math.sqrt(embed_dim)
    out = {}
    for key, shape in sorted(param_shapes(vocab, embed_dim, query_dim).items()):
        if key == EMB_KEY:
            a = rng.standard_normal(shape) * emb_scale
        elif key.endswith("query"):
            a = rng.uniform(-0.1, 0.1, shape)
        elif key.endswith("bias"):
            a = rng.standard_normal(shape) * 0.05
        else:
            a = rng.standard_normal(shape) * w_scale
        out[key] = torch.from_numpy(a.astype(np.float32))
    return out


# --------------------------------------------------------------------------------------
# dropout keep-mask specification (normative; the HIP kernels implement the same integer math)
# --------------------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _lowbias32(x: np.ndarray) -> np.ndarray:
    """32-bit integer finaliser; ``x`` is uint64 holding values < 2**32."""
    x = x & _M32
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846CA68B)) & _M32
    x = x ^ (x >> np.uint64(16))
    return x


def dropout_key(seed: int, stream: int) -> int:
    """Per-(seed, stream) 32-bit key, computed on the host side of the C-ABI."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    s = _lowbias32(np.uint64((int(stream) + 0x9E3779B9) & 0xFFFFFFFF))
    s = _lowbias32(np.uint64(hi) ^ s)
    s = _lowbias32(np.uint64(lo) ^ s)
    return int(s)


def dropout_threshold(p: float) -> int:
    return int(math.floor(float(p) * 4294967296.0))


def dropout_scale(p: float) -> np.float32:
    return np.float32(1.0 / (1.0 - float(p)))


def dropout_keep_mask(seed: int, stream: int, p: float, n_elems: int) -> np.ndarray:
    """keep[i] for flat element index i (row * D + col) of an (M, D) activation."""
    assert n_elems < 2 ** 32, "dropout index space is 32-bit"
    key = np.uint64(dropout_key(seed, stream))
    idx = np.arange(n_elems, dtype=np.uint64)
    u = _lowbias32(((idx * np.uint64(0x9E3779B1)) + key) & _M32)
    return u >= np.uint64(dropout_threshold(p))


def dropout_multiplier(seed: int, stream: int, p: float, shape) -> torch.Tensor:
    """float32 tensor of {0, 1/(1-p)} with the given shape (row-major flat indexing)."""
    n = int(np.prod(shape))
    if p <= 0.0:
        return torch.ones(shape, dtype=torch.float32)
    keep = dropout_keep_mask(seed, stream, p, n)
    return torch.from_numpy(keep.astype(np.float32) * dropout_scale(p)).reshape(shape)


# --------------------------------------------------------------------------------------
# to_dense_batch  (third-party torch_geometric==2.3.0; call sites nrms_module.py:233,237,277-284)
# --------------------------------------------------------------------------------------
def to_dense_batch(x: torch.Tensor, batch: torch.Tensor, batch_size: Optional[int] = None,
                   max_num: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """x (N, ...) + sorted assignment vector batch (N,) -> (B, max_count, ...) zero-filled, mask."""
    if batch_size is None:
        batch_size = int(batch.max()) + 1 if batch.numel() > 0 else 0
    counts = torch.bincount(batch, minlength=batch_size)
    if max_num is None:
        max_num = int(counts.max()) if batch_size > 0 else 0
    starts = torch.cumsum(counts, 0) - counts
    pos = torch.arange(batch.numel()) - starts[batch]
    out = x.new_zeros((batch_size, max_num) + tuple(x.shape[1:]))
    mask = torch.zeros((batch_size, max_num), dtype=torch.bool)
    out[batch, pos] = x
    mask[batch, pos] = True
    return out, mask


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def _mhsa_seq_first(x: torch.Tensor, w_in, b_in, w_o, b_o, num_heads: int) -> torch.Tensor:
    """``nn.MultiheadAttention(x, x, x)`` with batch_first=False: x is (S, Bt, D), attention over S.

    Restates torch's explicit path (need_weights=True): q scaled by 1/sqrt(d_h) before QK^T,
    softmax over keys, no masks, attention dropout 0 (``text.py:218``, ``user/nrms.py:27-29``).
    """
    S, Bt, D = x.shape
    dh = D // num_heads
    qkv = x @ w_in.t() + b_in                      # (S, Bt, 3D)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    q = q * (1.0 / math.sqrt(dh))

    def heads(t):                                   # (S, Bt, D) -> (Bt, h, S, dh)
        return t.reshape(S, Bt, num_heads, dh).permute(1, 2, 0, 3)

    s = heads(q) @ heads(k).transpose(-1, -2)       # (Bt, h, S, S)
    p = torch.softmax(s, dim=-1)
    o = p @ heads(v)                                # (Bt, h, S, dh)
    o = o.permute(2, 0, 1, 3).reshape(S, Bt, D)
    return o @ w_o.t() + b_o


def additive_attention(y: torch.Tensor, w_a, b_a, q_a) -> torch.Tensor:
    """``AdditiveAttention.forward`` (``attention.py:24-42``): y (G, S, D) -> (G, D); no mask."""
    a = torch.tanh(y @ w_a.t() + b_a) @ q_a         # (G, S)
    w = torch.softmax(a, dim=1)
    return (w.unsqueeze(-1) * y).sum(dim=1)


def news_encoder_fwd(ids: torch.Tensor, params: Dict[str, torch.Tensor], num_heads: int,
                     mult1: Optional[torch.Tensor] = None,
                     mult2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``MHSAAddAtt.forward`` (``text.py:222-236``) behind ``NewsEncoder.forward`` (``news.py:134-160``,
    one text encoder => passthrough).  ids (N, L) int64 -> (N, D).

    mult1 / mult2: optional (N, L, D) dropout multipliers ({0, 1/(1-p)}) for the two dropouts.
    """
    P = NEWS_PREFIX
    x = params[EMB_KEY][ids]                        # bit-exact gather; id 0 is an ordinary row
    if mult1 is not None:
        x = x * mult1
    # the reference permutes to (L, N, D) so attention runs over the L tokens of each news
    y = _mhsa_seq_first(x.permute(1, 0, 2),
                        params[P + "multihead_attention.in_proj_weight"],
                        params[P + "multihead_attention.in_proj_bias"],
                        params[P + "multihead_attention.out_proj.weight"],
                        params[P + "multihead_attention.out_proj.bias"], num_heads).permute(1, 0, 2)
    if mult2 is not None:
        y = y * mult2
    return additive_attention(y, params[P + "additive_attention.linear.weight"],
                              params[P + "additive_attention.linear.bias"],
                              params[P + "additive_attention.query"])


def user_encoder_fwd(hist: torch.Tensor, params: Dict[str, torch.Tensor], num_heads: int) -> torch.Tensor:
    """NRMS ``UserEncoder.forward`` (``user/nrms.py:32-41``): hist (B, H, D) -> (B, D).

    The reference feeds (B, H, D) to a seq-first MHA, so attention runs ACROSS THE B USERS for
    each history slot (SURVEY.md headline fact 3); reproduced on purpose.
    """
    P = USER_PREFIX
    y = _mhsa_seq_first(hist,                       # S = B, Bt = H
                        params[P + "multihead_attention.in_proj_weight"],
                        params[P + "multihead_attention.in_proj_bias"],
                        params[P + "multihead_attention.out_proj.weight"],
                        params[P + "multihead_attention.out_proj.bias"], num_heads)
    return additive_attention(y, params[P + "additive_attention.linear.weight"],
                              params[P + "additive_attention.linear.bias"],
                              params[P + "additive_attention.query"])


def plm_tail_fwd(hidden: torch.Tensor, params: Dict[str, torch.Tensor], num_heads: int,
                 mult1: Optional[torch.Tensor] = None, mult2: Optional[torch.Tensor] = None,
                 prefix: str = "") -> torch.Tensor:
    """Tail of ``PLM.forward`` with ``use_mhsa=True`` (``text.py:92-99``): last_hidden_state (N, L, D)
    -> dropout -> seq-first MHA (attention ACROSS THE N NEWS for each token position, the same
    batch_first quirk as the user encoder) -> dropout -> additive attention over L -> (N, D).
    The transformer body itself is third-party (HF ``AutoModel``) and is not restated."""
    x = hidden if mult1 is None else hidden * mult1
    y = _mhsa_seq_first(x, params[prefix + "multihead_attention.in_proj_weight"],
                        params[prefix + "multihead_attention.in_proj_bias"],
                        params[prefix + "multihead_attention.out_proj.weight"],
                        params[prefix + "multihead_attention.out_proj.bias"], num_heads)
    if mult2 is not None:
        y = y * mult2
    return additive_attention(y, params[prefix + "additive_attention.linear.weight"],
                              params[prefix + "additive_attention.linear.bias"],
                              params[prefix + "additive_attention.query"])


def click_scores(user: torch.Tensor, cand: torch.Tensor) -> torch.Tensor:
    """``DotProduct.forward`` as called at ``nrms_module.py:251-253``: (B, D), (B, C, D) -> (B, C)."""
    return torch.bmm(user.unsqueeze(1), cand.permute(0, 2, 1)).squeeze(1)


def ce_loss(scores: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
    """``CrossEntropyLoss()(scores, y_true)`` with float (probability-form) targets
    (``nrms_module.py:287-288``): mean_b( -sum_c y * log_softmax(scores)_c )."""
    return -(y_true * torch.log_softmax(scores, dim=1)).sum(dim=1).mean()


def nrms_forward(batch: dict, params: Dict[str, torch.Tensor], num_heads: int = 15,
                 p_drop: float = 0.0, seed: int = 0, fused_news_call: bool = True,
                 late_fusion: bool = False) -> dict:
    """``NRMSModule.forward`` (``nrms_module.py:230-255``) + the loss line (``:277,287-288``).

    batch keys: x_hist["title"] (N_hist, L) int64, batch_hist (N_hist,), x_cand["title"],
    batch_cand, labels (N_cand,) float32, optional "batch_size".

    Dropout streams (p_drop > 0): the product encodes history and candidate news in ONE encoder
    call over the concatenation [hist rows; cand rows] (row-independent, so identical to the
    reference's two calls apart from which random numbers each element sees); stream 0 is the
    post-embedding dropout, stream 1 the post-attention dropout, indexed over the concatenated
    (N_hist + N_cand, L, D) activation.
    """
    ids_h = batch["x_hist"]["title"]
    ids_c = batch["x_cand"]["title"]
    B = int(batch.get("batch_size", int(batch["batch_hist"].max()) + 1))
    ids = torch.cat([ids_h, ids_c], 0)
    D = params[EMB_KEY].shape[1]
    m1 = m2 = None
    if p_drop > 0.0:
        shape = (ids.shape[0], ids.shape[1], D)
        m1 = dropout_multiplier(seed, 0, p_drop, shape)
        m2 = dropout_multiplier(seed, 1, p_drop, shape)
    news = news_encoder_fwd(ids, params, num_heads, m1, m2)
    hist_vec, cand_vec = news[: ids_h.shape[0]], news[ids_h.shape[0]:]
    hist_dense, mask_hist = to_dense_batch(hist_vec, batch["batch_hist"], B)
    cand_dense, mask_cand = to_dense_batch(cand_vec, batch["batch_cand"], B)
    if not late_fusion:
        user = user_encoder_fwd(hist_dense, params, num_heads)
    else:   # nrms_module.py:243-248: sum over the zero-padded slots / true history size
        user = hist_dense.sum(dim=1) / mask_hist.sum(dim=1, keepdim=True).to(hist_dense.dtype)
    scores = click_scores(user, cand_dense)
    y_true, _ = to_dense_batch(batch["labels"], batch["batch_cand"], B)
    loss = ce_loss(scores, y_true)
    return dict(hist_vec=hist_vec, cand_vec=cand_vec, hist_dense=hist_dense, cand_dense=cand_dense,
                user_vec=user, scores=scores, y_true=y_true, loss=loss,
                mask_hist=mask_hist, mask_cand=mask_cand)


def collect_model_outputs(vector: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """``_collect_model_outputs`` (``abstract_recommender.py:126-130``) == row-major boolean index."""
    return torch.cat([vector[n][mask[n]] for n in range(mask.shape[0])], dim=0)


# --------------------------------------------------------------------------------------
# Adam (``torch.optim.Adam(lr=1e-4)``, ``configs/model/nrms.yaml:49-52``; dense over ALL params)
# --------------------------------------------------------------------------------------
def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
              lr: float = 1e-4, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """In-place single-tensor Adam exactly as torch's ``_single_tensor_adam`` (no weight decay,
    no amsgrad); ``step`` is the 1-based step count AFTER increment."""
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class NRMSOracle:
    """Whole train step on CPU: forward, CE loss, autograd backward, dense Adam."""

    def __init__(self, params: Dict[str, torch.Tensor], num_heads: int = 15, p_drop: float = 0.0,
                 lr: float = 1e-4):
        self.params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        self.num_heads, self.p_drop, self.lr = num_heads, p_drop, lr
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.step_count = 0

    def forward(self, batch, train: bool = False, seed: int = 0):
        return nrms_forward(batch, self.params, self.num_heads,
                            self.p_drop if train else 0.0, seed)

    def loss_and_grads(self, batch, train: bool = True, seed: int = 0):
        for p in self.params.values():
            p.grad = None
        out = self.forward(batch, train, seed)
        out["loss"].backward()
        grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p))
                 for k, p in self.params.items()}
        # nn.Embedding(padding_idx=0): the gradient row of id 0 is zeroed (text.py:215-217)
        grads[EMB_KEY] = grads[EMB_KEY].clone()
        grads[EMB_KEY][0].zero_()
        return out, grads

    def train_step(self, batch, seed: int = 0, grads_override=None):
        out, grads = self.loss_and_grads(batch, True, seed)
        if grads_override is not None:
            grads = grads_override
        self.step_count += 1
        with torch.no_grad():
            for k, p in self.params.items():
                adam_step(p, grads[k], self.m[k], self.v[k], self.step_count, self.lr)
        return out, grads


# --------------------------------------------------------------------------------------
# module-for-module torch graph (the timed CPU baseline; same nn graph as the reference)
# --------------------------------------------------------------------------------------
class _AddAtt(torch.nn.Module):
    def __init__(self, d, q):
        super().__init__()
        self.linear = torch.nn.Linear(d, q)
        self.query = torch.nn.Parameter(torch.empty(q).uniform_(-0.1, 0.1))

    def forward(self, x):
        a = torch.tanh(self.linear(x))
        w = torch.softmax(torch.matmul(a, self.query), dim=1)
        return torch.bmm(w.unsqueeze(1), x).squeeze(1)


class TorchGraphNRMS(torch.nn.Module):
    """Same ``nn`` graph as the reference's NRMS (Embedding -> Dropout -> seq-first MHA -> Dropout
    -> additive attention; seq-first MHA user encoder; bmm scorer; float-target CE), used with
    ``torch.optim.Adam`` as the CPU baseline ("port") that ``bench.py`` times on host cores."""

    def __init__(self, emb: torch.Tensor, d=300, heads=15, q=200, p_drop=0.2):
        super().__init__()
        self.embedding_layer = torch.nn.Embedding.from_pretrained(emb.clone(), freeze=False, padding_idx=0)
        self.news_mha = torch.nn.MultiheadAttention(d, heads)
        self.news_att = _AddAtt(d, q)
        self.dropout = torch.nn.Dropout(p_drop)
        self.user_mha = torch.nn.MultiheadAttention(d, heads)
        self.user_att = _AddAtt(d, q)
        self.criterion = torch.nn.CrossEntropyLoss()

    def load_oracle_params(self, params):
        sd = {"embedding_layer.weight": params[EMB_KEY]}
        for pre, mha, att in ((NEWS_PREFIX, "news_mha", "news_att"), (USER_PREFIX, "user_mha", "user_att")):
            sd[mha + ".in_proj_weight"] = params[pre + "multihead_attention.in_proj_weight"]
            sd[mha + ".in_proj_bias"] = params[pre + "multihead_attention.in_proj_bias"]
            sd[mha + ".out_proj.weight"] = params[pre + "multihead_attention.out_proj.weight"]
            sd[mha + ".out_proj.bias"] = params[pre + "multihead_attention.out_proj.bias"]
            sd[att + ".linear.weight"] = params[pre + "additive_attention.linear.weight"]
            sd[att + ".linear.bias"] = params[pre + "additive_attention.linear.bias"]
            sd[att + ".query"] = params[pre + "additive_attention.query"]
        self.load_state_dict(sd)

    def encode_news(self, ids):
        x = self.dropout(self.embedding_layer(ids)).permute(1, 0, 2)
        x, _ = self.news_mha(x, x, x)
        return self.news_att(self.dropout(x).permute(1, 0, 2))

    def forward(self, batch):
        B = int(batch.get("batch_size", int(batch["batch_hist"].max()) + 1))
        hist = self.encode_news(batch["x_hist"]["title"])       # two calls, as the reference
        hist_dense, _ = to_dense_batch(hist, batch["batch_hist"], B)
        cand = self.encode_news(batch["x_cand"]["title"])
        cand_dense, _ = to_dense_batch(cand, batch["batch_cand"], B)
        u, _ = self.user_mha(hist_dense, hist_dense, hist_dense)
        user = self.user_att(u)
        return torch.bmm(user.unsqueeze(1), cand_dense.permute(0, 2, 1)).squeeze(1)

    def loss(self, batch):
        scores = self.forward(batch)
        B = scores.shape[0]
        y_true, _ = to_dense_batch(batch["labels"], batch["batch_cand"], B)
        return self.criterion(scores, y_true)
