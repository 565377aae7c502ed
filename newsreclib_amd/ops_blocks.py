"""``torch.autograd.Function`` wrappers of the generic blocks of the C ABI (standalone additive attention,
nn.Linear + activation).  Same conventions as ``ops.py``."""
from __future__ import annotations

import contextlib
import ctypes
import weakref

import torch

from . import _lib
from ._lib import NrlAddAttGrads, NrlAddAttParams
from .ops import GradAwareFunction, _chk, _grad_targets, _stream, saving

ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2}


class AdditiveAttentionFn(GradAwareFunction):
    """``AdditiveAttention.forward`` (reference layers/attention.py:24-42): (G, S, D) -> (G, D)."""

    @staticmethod
    def forward(ctx, y, w_a, b_a, q_a, grad_bufs):
        lib = _lib.load()
        y = _chk(y, torch.float32, "input_vector")
        params = [_chk(t, torch.float32, n) for t, n in zip((w_a, b_a, q_a), ("linear.weight", "linear.bias", "query"))]
        if y.dim() != 3:
            raise ValueError("newsreclib_amd: additive attention expects (groups, length, dim)")
        G, S, D = y.shape
        Q = params[0].shape[0]
        if params[0].shape != (Q, D) or params[1].shape != (Q,) or params[2].shape != (Q,):
            raise ValueError("newsreclib_amd: inconsistent additive-attention parameter shapes")
        ap = NrlAddAttParams(*[p.data_ptr() for p in params], D, Q)
        ws = torch.empty(max(lib.nrl_additive_attention_workspace_bytes(G, S, D, Q), 256), dtype=torch.uint8,
                         device=y.device)
        out = torch.empty((G, D), dtype=torch.float32, device=y.device)
        save = saving(ctx)
        _lib.check(lib.nrl_additive_attention_fwd(ctypes.byref(ap), y.data_ptr(), G, S, int(save), out.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), _stream()), "nrl_additive_attention_fwd")
        if save:
            ctx.save_for_backward(y, *params)
            ctx.ws, ctx.grad_bufs, ctx.engine, ctx.options = ws, grad_bufs, _lib.engine_code(), _lib.options_mask()
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        _lib.require_engine(ctx.engine, "additive attention")
        _lib.require_options(ctx.options, "additive attention")
        y, *params = ctx.saved_tensors
        G, S, D = y.shape
        Q = params[0].shape[0]
        d_out = _chk(d_out, torch.float32, "d_out")
        ap = NrlAddAttParams(*[p.data_ptr() for p in params], D, Q)
        bufs, rets = _grad_targets(params, ctx.grad_bufs)
        ag = NrlAddAttGrads(*[b.data_ptr() for b in bufs])
        d_y = torch.empty_like(y)
        _lib.check(lib.nrl_additive_attention_bwd(ctypes.byref(ap), ctypes.byref(ag), y.data_ptr(), G, S,
                                                  d_out.data_ptr(), d_y.data_ptr(), ctx.ws.data_ptr(), ctx.ws.numel(),
                                                  _stream()), "nrl_additive_attention_bwd")
        ctx.ws = None
        return (d_y, *rets, None)


class LinearActFn(GradAwareFunction):
    """``act(nn.Linear(x))``, act in {none, tanh, relu}: (M, K) -> (M, N)."""

    @staticmethod
    def forward(ctx, a, w, bias, act, grad_bufs):
        lib = _lib.load()
        a, w, bias = _chk(a, torch.float32, "input"), _chk(w, torch.float32, "weight"), _chk(bias, torch.float32, "bias")
        if a.dim() != 2 or w.dim() != 2 or a.shape[1] != w.shape[1] or bias.shape != (w.shape[0],):
            raise ValueError("newsreclib_amd: inconsistent linear shapes")
        M, K = a.shape
        N = w.shape[0]
        code = ACT[act]
        ws = torch.empty(max(lib.nrl_linear_act_workspace_bytes(M, N, K), 256), dtype=torch.uint8, device=a.device)
        c = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _lib.check(lib.nrl_linear_act_fwd(a.data_ptr(), w.data_ptr(), bias.data_ptr(), M, N, K, code, c.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _stream()), "nrl_linear_act_fwd")
        if saving(ctx):
            ctx.save_for_backward(a, w, bias, c)
            ctx.ws, ctx.code, ctx.grad_bufs, ctx.engine = ws, code, grad_bufs, _lib.engine_code()
            ctx.options = _lib.options_mask()
        return c

    @staticmethod
    def backward(ctx, d_c):
        lib = _lib.load()
        _lib.require_engine(ctx.engine, "linear + activation")
        _lib.require_options(ctx.options, "linear + activation")
        a, w, bias, c = ctx.saved_tensors
        M, K = a.shape
        N = w.shape[0]
        d_c = _chk(d_c, torch.float32, "d_out")
        bufs, rets = _grad_targets([w, bias], ctx.grad_bufs)
        need_da = ctx.needs_input_grad[0]
        d_a = torch.empty_like(a) if need_da else None
        _lib.check(lib.nrl_linear_act_bwd(a.data_ptr(), w.data_ptr(), c.data_ptr(), d_c.data_ptr(), M, N, K, ctx.code,
                                          d_a.data_ptr() if need_da else None, bufs[0].data_ptr(), bufs[1].data_ptr(),
                                          ctx.ws.data_ptr(), ctx.ws.numel(), _stream()), "nrl_linear_act_bwd")
        ctx.ws = None
        return (d_a, rets[0], rets[1], None, None)


class LinearFn(GradAwareFunction):
    """``nn.Linear`` on this library's projection engines (bf16x3 / exact fp32 matrix cores): (..., K) -> (..., N).
    The weight gradient is skipped for a frozen weight (the PLM body: layers 0-7 frozen while their inputs still need
    gradients, reference text.py:69-73); the bf16 weight planes are rebuilt by the backward (10 us) rather than kept
    alive for every layer of a transformer between its forward and backward."""

    @staticmethod
    def forward(ctx, a, w, bias, grad_bufs, images=None):
        lib = _lib.load()
        a, w, bias = _chk(a, torch.float32, "input"), _chk(w, torch.float32, "weight"), _chk(bias, torch.float32, "bias")
        if w.dim() != 2 or a.shape[-1] != w.shape[1] or bias.shape != (w.shape[0],):
            raise ValueError("newsreclib_amd: inconsistent linear shapes")
        N, K = w.shape
        a2 = a.reshape(-1, K)
        M = a2.shape[0]
        # `images` (FrozenImages, a frozen weight only): the matrix-core images of the weight kept across calls
        ws, ready, key = images.buffer("fwd", w, lib, a.device) if images is not None else (None, 0, None)
        if ws is None:
            ws = torch.empty(max(lib.nrl_linear_workspace_bytes(N, K), 256), dtype=torch.uint8, device=a.device)
        c = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _lib.check(lib.nrl_linear_fwd_img(a2.data_ptr(), w.data_ptr(), bias.data_ptr(), M, N, K, c.data_ptr(), ws.data_ptr(),
                                          ws.numel(), ready, _stream()), "nrl_linear_fwd")
        if key is not None:
            images.commit("fwd", key)
        if saving(ctx):
            need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
            ctx.save_for_backward(a2 if need_w else None, w, bias)
            ctx.grad_bufs, ctx.engine, ctx.in_shape = grad_bufs, _lib.engine_code(), tuple(a.shape)
            ctx.images = images          # (a FrozenImages decides per weight whether it may keep an image)
        return c.view(*a.shape[:-1], N)

    @staticmethod
    def backward(ctx, d_c):
        lib = _lib.load()
        _lib.require_engine(ctx.engine, "linear")
        a2, w, bias = ctx.saved_tensors
        N, K = w.shape
        d_c = _chk(d_c.reshape(-1, N), torch.float32, "d_out")
        M = d_c.shape[0]
        need_a = ctx.needs_input_grad[0]
        need_w = a2 is not None
        rets = [None, None]
        dw = db = None
        if need_w:
            bufs, rets = _grad_targets([w, bias], ctx.grad_bufs)
            dw, db = bufs[0].data_ptr(), bufs[1].data_ptr()
        d_a = torch.empty((M, K), dtype=torch.float32, device=d_c.device) if need_a else None
        ws, ready, key = ctx.images.buffer("bwd", w, lib, d_c.device) if ctx.images is not None else (None, 0, None)
        if ws is None:
            ws = torch.empty(max(lib.nrl_linear_workspace_bytes(N, K), 256), dtype=torch.uint8, device=d_c.device)
        _lib.check(lib.nrl_linear_bwd_img(a2.data_ptr() if need_w else None, w.data_ptr(), d_c.data_ptr(), M, N, K,
                                          d_a.data_ptr() if need_a else None, dw, db, ws.data_ptr(), ws.numel(), ready,
                                          _stream()), "nrl_linear_bwd")
        if key is not None:
            ctx.images.commit("bwd", key)
        return (d_a.view(ctx.in_shape) if need_a else None, rets[0], rets[1], None, None)


def _image_ws(images, which, w, lib, device):
    """-> (workspace, ready flag, key to commit) of a projection's matrix-core images (FrozenImages or per call)."""
    ws, ready, key = images.buffer(which, w, lib, device) if images is not None else (None, 0, None)
    if ws is None:
        N, K = w.shape
        ws = torch.empty(max(lib.nrl_linear_workspace_bytes(N, K), 256), dtype=torch.uint8, device=device)
    return ws, ready, key


class FfnFn(GradAwareFunction):
    """The feed-forward half of a BERT-family layer up to its second projection (HF ``RobertaIntermediate`` + the ``dense`` of
    ``RobertaOutput``, inside ``self.plm_model(**text)``, reference text.py:89): y = gelu(x W1^T + b1) W2^T + b2 with the exact GELU
    in the first GEMM's epilogue (``nrl_linear_gelu_fwd_img``: h saved, g handed on) and its derivative in the epilogue of the second
    projection's activation gradient (``nrl_linear_dgrad_gelu_img``: d_h straight from d_y W2) -- the framework's two elementwise
    passes over the (rows, 3072) activation, 5.4 ms of a config-4 step, are gone.  A frozen layer keeps neither x nor g."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, grad_bufs, images1, images2):
        lib = _lib.load()
        x = _chk(x, torch.float32, "input")
        w1, b1, w2, b2 = (_chk(t, torch.float32, "ffn parameter") for t in (w1, b1, w2, b2))
        N1, K1 = w1.shape
        N2, K2 = w2.shape
        if x.shape[-1] != K1 or K2 != N1 or b1.shape != (N1,) or b2.shape != (N2,):
            raise ValueError("newsreclib_amd: inconsistent feed-forward shapes")
        x2 = x.reshape(-1, K1)
        M = x2.shape[0]
        dev = x.device
        h = torch.empty((M, N1), dtype=torch.float32, device=dev)
        g = torch.empty((M, N1), dtype=torch.float32, device=dev)
        ws, ready, key = _image_ws(images1, "fwd", w1, lib, dev)
        _lib.check(lib.nrl_linear_gelu_fwd_img(x2.data_ptr(), w1.data_ptr(), b1.data_ptr(), M, N1, K1, h.data_ptr(), g.data_ptr(),
                                               ws.data_ptr(), ws.numel(), ready, _stream()), "nrl_linear_gelu_fwd")
        if key is not None:
            images1.commit("fwd", key)
        y = torch.empty((M, N2), dtype=torch.float32, device=dev)
        ws, ready, key = _image_ws(images2, "fwd", w2, lib, dev)
        _lib.check(lib.nrl_linear_fwd_img(g.data_ptr(), w2.data_ptr(), b2.data_ptr(), M, N2, K2, y.data_ptr(), ws.data_ptr(),
                                          ws.numel(), ready, _stream()), "nrl_linear_fwd")
        if key is not None:
            images2.commit("fwd", key)
        if saving(ctx):
            need_w1 = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
            need_w2 = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
            ctx.save_for_backward(x2 if need_w1 else None, h, g if need_w2 else None, w1, b1, w2, b2)
            ctx.grad_bufs, ctx.engine, ctx.in_shape = grad_bufs, _lib.engine_code(), tuple(x.shape)
            ctx.images = (images1, images2)
        return y.view(*x.shape[:-1], N2)

    @staticmethod
    def backward(ctx, d_y):
        lib = _lib.load()
        _lib.require_engine(ctx.engine, "feed-forward block")
        x2, h, g, w1, b1, w2, b2 = ctx.saved_tensors
        N1, K1 = w1.shape
        N2, K2 = w2.shape
        d_y = _chk(d_y.reshape(-1, N2), torch.float32, "d_out")
        M = d_y.shape[0]
        dev = d_y.device
        need_x = ctx.needs_input_grad[0]
        need_w1, need_w2 = x2 is not None, g is not None
        images1, images2 = ctx.images
        rets = [None] * 4
        gb = ctx.grad_bufs
        if need_w2:
            bufs, r = _grad_targets([w2, b2], gb[2:4] if gb is not None else None)
            rets[2:4] = r
            ws = torch.empty(256, dtype=torch.uint8, device=dev)          # (weight gradient only: no image)
            _lib.check(lib.nrl_linear_bwd_img(g.data_ptr(), w2.data_ptr(), d_y.data_ptr(), M, N2, K2, None, bufs[0].data_ptr(),
                                              bufs[1].data_ptr(), ws.data_ptr(), ws.numel(), 0, _stream()), "nrl_linear_bwd")
        d_x = None
        if need_x or need_w1:
            d_h = torch.empty((M, N1), dtype=torch.float32, device=dev)
            ws, ready, key = _image_ws(images2, "bwd", w2, lib, dev)
            _lib.check(lib.nrl_linear_dgrad_gelu_img(w2.data_ptr(), d_y.data_ptr(), h.data_ptr(), M, N2, K2, d_h.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), ready, _stream()), "nrl_linear_dgrad_gelu")
            if key is not None:
                images2.commit("bwd", key)
            dw = db = None
            if need_w1:
                bufs, r = _grad_targets([w1, b1], gb[0:2] if gb is not None else None)
                rets[0:2] = r
                dw, db = bufs[0].data_ptr(), bufs[1].data_ptr()
            d_x = torch.empty((M, K1), dtype=torch.float32, device=dev) if need_x else None
            ws, ready, key = _image_ws(images1, "bwd", w1, lib, dev)
            _lib.check(lib.nrl_linear_bwd_img(x2.data_ptr() if need_w1 else None, w1.data_ptr(), d_h.data_ptr(), M, N1, K1,
                                              d_x.data_ptr() if need_x else None, dw, db, ws.data_ptr(), ws.numel(), ready,
                                              _stream()), "nrl_linear_bwd")
            if key is not None:
                images1.commit("bwd", key)
        return (d_x.view(ctx.in_shape) if need_x else None, rets[0], rets[1], rets[2], rets[3], None, None, None)


class EmbeddingFn(GradAwareFunction):
    """``nn.Embedding`` of a third-party stack (the PLM body's word / position / token-type tables): bit-exact gather forward,
    and ``embedding_dense_backward`` as the library's counting sort + sorted-segment reduction (``nrl_sort_positions`` +
    ``nrl_embedding_grad``) instead of ATen's merge sort + three segment kernels; ``padding_idx`` receives no gradient."""

    @staticmethod
    def forward(ctx, ids, weight, padding_idx, grad_bufs):
        from . import ops
        ids = _chk(ids, torch.int64, "ids")
        weight = _chk(weight, torch.float32, "embedding weight")
        out = ops.embedding_gather(weight, ids)
        if saving(ctx):
            ctx.save_for_backward(ids)
            ctx.shape, ctx.padding_idx, ctx.grad_bufs = tuple(weight.shape), padding_idx, grad_bufs
        return out

    @staticmethod
    def backward(ctx, d_out):
        from . import ops
        lib = _lib.load()
        (ids,) = ctx.saved_tensors
        V, D = ctx.shape
        d_out = _chk(d_out.reshape(-1, D), torch.float32, "d_out")
        flat = ids.reshape(-1)
        order = ops.sort_positions(flat, V)
        gb = ctx.grad_bufs[0] if ctx.grad_bufs is not None else None
        if gb is not None:
            target, ret = gb, None
        else:
            target = ret = torch.zeros((V, D), dtype=torch.float32, device=d_out.device)
        pad = -1 if ctx.padding_idx is None else int(ctx.padding_idx)
        _lib.check(lib.nrl_embedding_grad(d_out.data_ptr(), flat.data_ptr(), order.data_ptr(), flat.numel(), D, pad,
                                          target.data_ptr(), _stream()), "nrl_embedding_grad")
        return None, ret, None, None


class SdpaFn(GradAwareFunction):
    """``F.scaled_dot_product_attention`` of a transformer body on the bf16x3 matrix-core kernels (``nrl_sdpa_fwd`` / ``_bwd``):
    q, k, v (N, L, H, dh) -- the projections' outputs viewed per head, NOT transposed -- -> (N, L, H, dh).  ``keep`` (N, L) uint8
    or None: the key-padding mask (1 = attend).  Attention-probability dropout ``p_drop`` under the library's counter-based
    mask spec (``seed``); L <= 128, dh == 64 (``nrl_sdpa_supported``)."""

    @staticmethod
    def forward(ctx, q, k, v, keep, scale, p_drop, seed):
        lib = _lib.load()
        q, k, v = _chk(q, torch.float32, "query"), _chk(k, torch.float32, "key"), _chk(v, torch.float32, "value")
        if q.dim() != 4 or k.shape != q.shape or v.shape != q.shape:
            raise ValueError("newsreclib_amd: sdpa expects q, k, v of one shape (batch, seq, heads, head_dim)")
        N, L, H, dh = q.shape
        if keep is not None:
            keep = _chk(keep, torch.uint8, "key mask")
            if tuple(keep.shape) != (N, L):
                raise ValueError("newsreclib_amd: sdpa key mask must be (batch, seq)")
        save = saving(ctx, slice(0, 3))
        out = torch.empty_like(q)
        lse = torch.empty((N * H, L), dtype=torch.float32, device=q.device) if save else None
        _lib.check(lib.nrl_sdpa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), keep.data_ptr() if keep is not None else None,
                                    N, L, H, dh, float(scale), float(p_drop), int(seed), 0, out.data_ptr(),
                                    lse.data_ptr() if save else None, _stream()), "nrl_sdpa_fwd")
        if save:
            ctx.save_for_backward(q, k, v, out, lse, *([keep] if keep is not None else []))
            ctx.cfg = (float(scale), float(p_drop), int(seed))
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        q, k, v, out, lse, *rest = ctx.saved_tensors
        keep = rest[0] if rest else None
        N, L, H, dh = q.shape
        scale, p_drop, seed = ctx.cfg
        d_out = _chk(d_out, torch.float32, "d_out")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        _lib.check(lib.nrl_sdpa_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), keep.data_ptr() if keep is not None else None,
                                    out.data_ptr(), d_out.data_ptr(), lse.data_ptr(), N, L, H, dh, scale, p_drop, seed, 0,
                                    dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), _stream()), "nrl_sdpa_bwd")
        return dq, dk, dv, None, None, None, None


class DropoutAddLayerNormFn(GradAwareFunction):
    """``LayerNorm(dropout(x) + residual)`` -- the line that ends both halves of every layer of a BERT-family encoder
    (``RobertaSelfOutput`` / ``RobertaOutput``; the PLM body of text.py:89-109) -- as ONE launch each way
    (``nrl_dropout_add_layernorm_fwd`` / ``_bwd``) instead of dropout + add + layer norm (7 passes over the activation forward,
    8 backward -> 4 and 4).  x, residual (..., dim); gamma, beta (dim).  The dropout mask is the library's counter-based one."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, p_drop, seed, grad_bufs):
        lib = _lib.load()
        x, residual = _chk(x, torch.float32, "input"), _chk(residual, torch.float32, "residual")
        gamma, beta = _chk(gamma, torch.float32, "LayerNorm.weight"), _chk(beta, torch.float32, "LayerNorm.bias")
        dim = x.shape[-1]
        if residual.shape != x.shape or gamma.shape != (dim,) or beta.shape != (dim,):
            raise ValueError("newsreclib_amd: dropout_add_layernorm expects x and residual of one shape and (dim,) parameters")
        rows = x.numel() // dim
        save = saving(ctx)
        y = torch.empty_like(x)
        z = torch.empty_like(x) if save else None
        stats = torch.empty((2, rows), dtype=torch.float32, device=x.device) if save else None
        _lib.check(lib.nrl_dropout_add_layernorm_fwd(
            x.data_ptr(), residual.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, dim, float(eps), float(p_drop), int(seed), 0,
            z.data_ptr() if save else None, stats[0].data_ptr() if save else None, stats[1].data_ptr() if save else None,
            y.data_ptr(), _stream()), "nrl_dropout_add_layernorm_fwd")
        if save:
            ctx.save_for_backward(z, stats, gamma, beta)
            ctx.cfg, ctx.grad_bufs = (float(p_drop), int(seed)), grad_bufs
        return y

    @staticmethod
    def backward(ctx, d_y):
        lib = _lib.load()
        z, stats, gamma, beta = ctx.saved_tensors
        p_drop, seed = ctx.cfg
        d_y = _chk(d_y, torch.float32, "d_out")
        dim = z.shape[-1]
        rows = z.numel() // dim
        need_params = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        rets = [None, None]
        dg = db = None
        if need_params:
            bufs, rets = _grad_targets([gamma, beta], ctx.grad_bufs)
            dg, db = bufs[0].data_ptr(), bufs[1].data_ptr()
        d_res = torch.empty_like(z)
        d_x = torch.empty_like(z) if p_drop > 0.0 else None
        _lib.check(lib.nrl_dropout_add_layernorm_bwd(
            d_y.data_ptr(), z.data_ptr(), gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), rows, dim, p_drop, seed, 0,
            d_x.data_ptr() if d_x is not None else None, d_res.data_ptr(), dg, db, _stream()), "nrl_dropout_add_layernorm_bwd")
        return (d_x if d_x is not None else d_res, d_res, rets[0], rets[1], None, None, None, None)


def _glue_fwd(lib, h, residual, gamma, beta, eps, p_drop, seed, save):
    """LayerNorm(dropout(h) + residual) on (rows, dim) -> (y, z, stats); z, stats None unless ``save``."""
    rows, dim = h.shape
    y = torch.empty_like(h)
    z = torch.empty_like(h) if save else None
    stats = torch.empty((2, rows), dtype=torch.float32, device=h.device) if save else None
    _lib.check(lib.nrl_dropout_add_layernorm_fwd(
        h.data_ptr(), residual.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rows, dim, float(eps), float(p_drop), int(seed), 0,
        z.data_ptr() if save else None, stats[0].data_ptr() if save else None, stats[1].data_ptr() if save else None,
        y.data_ptr(), _stream()), "nrl_dropout_add_layernorm_fwd")
    return y, z, stats


def _glue_bwd(lib, d_y, z, stats, gamma, p_drop, seed, dg, db):
    """-> (gradient of the dropout's input, gradient of the residual); the same tensor when there is no dropout."""
    rows, dim = z.shape
    d_res = torch.empty_like(z)
    d_h = torch.empty_like(z) if p_drop > 0.0 else None
    _lib.check(lib.nrl_dropout_add_layernorm_bwd(
        d_y.data_ptr(), z.data_ptr(), gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), rows, dim, p_drop, seed, 0,
        d_h.data_ptr() if d_h is not None else None, d_res.data_ptr(), dg, db, _stream()), "nrl_dropout_add_layernorm_bwd")
    return (d_h if d_h is not None else d_res), d_res


def _wgrad_only(lib, a2, w, d_c, M, dw, db, dev):
    """d_w += d_c^T a, d_bias += colsum(d_c) of one nn.Linear (no activation gradient, no image)."""
    N, K = w.shape
    ws = torch.empty(256, dtype=torch.uint8, device=dev)
    _lib.check(lib.nrl_linear_bwd_img(a2.data_ptr(), w.data_ptr(), d_c.data_ptr(), M, N, K, None, dw.data_ptr(), db.data_ptr(),
                                      ws.data_ptr(), ws.numel(), 0, _stream()), "nrl_linear_bwd")


class AttnBlockFn(GradAwareFunction):
    """The attention half of a BERT-family layer as ONE autograd function (HF ``RobertaAttention`` = ``RobertaSelfAttention`` +
    ``RobertaSelfOutput`` inside ``self.plm_model(**text)``, reference text.py:89):
    y = LayerNorm(dropout(sdpa(x Wq^T + bq, x Wk^T + bk, x Wv^T + bv) Wo^T + bo) + x).
    Round 5: the three projections are one GEMM (``nrl_linear3_fwd_img``: x streamed once, 9 column panels in one launch), their
    activation gradient is one GEMM over the concatenated reduction (``nrl_linear3_dgrad_img``) whose epilogue adds the residual
    branch's gradient -- the framework's two gradient adds per layer for q / k / v and the one for the residual stream are gone."""

    @staticmethod
    def forward(ctx, x, keep, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, heads, scale, p_attn, seed_attn, eps, p_hid, seed_hid,
                grad_bufs, images_qkv, images_o):
        lib = _lib.load()
        x = _chk(x, torch.float32, "input")
        params = [_chk(t, torch.float32, "attention-block parameter") for t in (wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta)]
        wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta = params
        if x.dim() != 3:
            raise ValueError("newsreclib_amd: attention block expects (batch, seq, dim)")
        Nb, L, K = x.shape
        n = wq.shape[0]
        if wq.shape != (n, K) or wk.shape != (n, K) or wv.shape != (n, K) or wo.shape != (K, n) or n % heads or \
                any(b.shape != (n,) for b in (bq, bk, bv)) or bo.shape != (K,) or gamma.shape != (K,) or beta.shape != (K,):
            raise ValueError("newsreclib_amd: inconsistent attention-block shapes")
        dh = n // heads
        if keep is not None:
            keep = _chk(keep, torch.uint8, "key mask")
            if tuple(keep.shape) != (Nb, L):
                raise ValueError("newsreclib_amd: attention-block key mask must be (batch, seq)")
        dev = x.device
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        save = saving(ctx)
        # q | k | v stacked: (3, M, n)
        qkv = torch.empty((3, M, n), dtype=torch.float32, device=dev)
        nbytes = lib.nrl_linear3_workspace_bytes(n, K)
        ws, ready, key = images_qkv.buffer_multi("fwd", (wq, wk, wv), nbytes, dev) if images_qkv is not None else (None, 0, None)
        if ws is None:
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
        _lib.check(lib.nrl_linear3_fwd_img(x2.data_ptr(), wq.data_ptr(), wk.data_ptr(), wv.data_ptr(), bq.data_ptr(), bk.data_ptr(),
                                           bv.data_ptr(), M, n, K, qkv.data_ptr(), ws.data_ptr(), ws.numel(), ready, _stream()),
                   "nrl_linear3_fwd")
        if key is not None:
            images_qkv.commit("fwd", key)
        o = torch.empty((M, n), dtype=torch.float32, device=dev)
        lse = torch.empty((Nb * heads, L), dtype=torch.float32, device=dev) if save else None
        _lib.check(lib.nrl_sdpa_fwd(qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(), keep.data_ptr() if keep is not None else None,
                                    Nb, L, heads, dh, float(scale), float(p_attn), int(seed_attn), 0, o.data_ptr(),
                                    lse.data_ptr() if save else None, _stream()), "nrl_sdpa_fwd")
        h = torch.empty((M, K), dtype=torch.float32, device=dev)
        ws, ready, key = _image_ws(images_o, "fwd", wo, lib, dev)
        _lib.check(lib.nrl_linear_fwd_img(o.data_ptr(), wo.data_ptr(), bo.data_ptr(), M, K, n, h.data_ptr(), ws.data_ptr(), ws.numel(),
                                          ready, _stream()), "nrl_linear_fwd")
        if key is not None:
            images_o.commit("fwd", key)
        y, z, stats = _glue_fwd(lib, h, x2, gamma, beta, eps, p_hid, seed_hid, save)
        if save:
            nig = ctx.needs_input_grad
            need_wqkv = any(nig[2:8])
            ctx.save_for_backward(x2 if need_wqkv else None, qkv, o, lse, z, stats, keep, *params)
            ctx.cfg = (heads, float(scale), float(p_attn), int(seed_attn), float(p_hid), int(seed_hid), (Nb, L, K))
            ctx.grad_bufs, ctx.engine, ctx.images = grad_bufs, _lib.engine_code(), (images_qkv, images_o)
        return y.view(Nb, L, K)

    @staticmethod
    def backward(ctx, d_y):
        lib = _lib.load()
        _lib.require_engine(ctx.engine, "attention block")
        x2, qkv, o, lse, z, stats, keep, wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta = ctx.saved_tensors
        heads, scale, p_attn, seed_attn, p_hid, seed_hid, (Nb, L, K) = ctx.cfg
        images_qkv, images_o = ctx.images
        n = wq.shape[0]
        dh = n // heads
        M = Nb * L
        dev = d_y.device
        d_y = _chk(d_y.reshape(M, K), torch.float32, "d_out")
        nig = ctx.needs_input_grad
        gb = ctx.grad_bufs
        rets = [None] * 10

        def targets(i0, ps):
            bufs, r = _grad_targets(ps, gb[i0:i0 + len(ps)] if gb is not None else None)
            rets[i0:i0 + len(ps)] = r
            return bufs

        dg = dbt = None
        if nig[10] or nig[11]:
            bufs = targets(8, [gamma, beta])
            dg, dbt = bufs[0].data_ptr(), bufs[1].data_ptr()
        d_h, d_res = _glue_bwd(lib, d_y, z, stats, gamma, p_hid, seed_hid, dg, dbt)
        # out-projection: weight gradient + d_o
        dw = db = None
        need_wo = nig[8] or nig[9]
        if need_wo:
            bufs = targets(6, [wo, bo])
            dw, db = bufs[0].data_ptr(), bufs[1].data_ptr()
        d_o = torch.empty((M, n), dtype=torch.float32, device=dev)
        ws, ready, key = _image_ws(images_o, "bwd", wo, lib, dev)
        _lib.check(lib.nrl_linear_bwd_img(o.data_ptr() if need_wo else None, wo.data_ptr(), d_h.data_ptr(), M, K, n, d_o.data_ptr(), dw, db,
                                          ws.data_ptr(), ws.numel(), ready, _stream()), "nrl_linear_bwd")
        if key is not None:
            images_o.commit("bwd", key)
        dqkv = torch.empty((3, M, n), dtype=torch.float32, device=dev)
        _lib.check(lib.nrl_sdpa_bwd(qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(), keep.data_ptr() if keep is not None else None,
                                    o.data_ptr(), d_o.data_ptr(), lse.data_ptr(), Nb, L, heads, dh, scale, p_attn, seed_attn, 0,
                                    dqkv[0].data_ptr(), dqkv[1].data_ptr(), dqkv[2].data_ptr(), _stream()), "nrl_sdpa_bwd")
        for q, (w, b) in enumerate(((wq, bq), (wk, bk), (wv, bv))):
            if nig[2 + 2 * q] or nig[3 + 2 * q]:
                bufs = targets(2 * q, [w, b])
                _wgrad_only(lib, x2, w, dqkv[q], M, bufs[0], bufs[1], dev)
        d_x = None
        if nig[0]:
            d_x = torch.empty((M, K), dtype=torch.float32, device=dev)
            nbytes = lib.nrl_linear3_workspace_bytes(n, K)
            ws, ready, key = images_qkv.buffer_multi("bwd", (wq, wk, wv), nbytes, dev) if images_qkv is not None else (None, 0, None)
            if ws is None:
                ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
            _lib.check(lib.nrl_linear3_dgrad_img(dqkv.data_ptr(), wq.data_ptr(), wk.data_ptr(), wv.data_ptr(), M, n, K, d_res.data_ptr(),
                                                 d_x.data_ptr(), ws.data_ptr(), ws.numel(), ready, _stream()), "nrl_linear3_dgrad")
            if key is not None:
                images_qkv.commit("bwd", key)
            d_x = d_x.view(Nb, L, K)
        return (d_x, None, *rets, *([None] * 10))


class FfnBlockFn(GradAwareFunction):
    """``FfnFn`` + the dropout / residual / LayerNorm that ends the layer (HF ``RobertaOutput``) as one function:
    LayerNorm(dropout(gelu(x W1^T + b1) W2^T + b2) + x).  The first projection's activation gradient takes the residual branch's
    gradient in its epilogue (``nrl_linear_dgrad_add_img``) instead of leaving the sum to a framework add."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gamma, beta, eps, p_hid, seed_hid, grad_bufs, images1, images2):
        lib = _lib.load()
        x = _chk(x, torch.float32, "input")
        w1, b1, w2, b2, gamma, beta = (_chk(t, torch.float32, "ffn parameter") for t in (w1, b1, w2, b2, gamma, beta))
        N1, K1 = w1.shape
        N2, K2 = w2.shape
        if x.shape[-1] != K1 or K2 != N1 or N2 != K1 or b1.shape != (N1,) or b2.shape != (N2,) or gamma.shape != (N2,) or beta.shape != (N2,):
            raise ValueError("newsreclib_amd: inconsistent feed-forward block shapes")
        x2 = x.reshape(-1, K1)
        M = x2.shape[0]
        dev = x.device
        save = saving(ctx)
        hh = torch.empty((M, N1), dtype=torch.float32, device=dev)
        g = torch.empty((M, N1), dtype=torch.float32, device=dev)
        ws, ready, key = _image_ws(images1, "fwd", w1, lib, dev)
        _lib.check(lib.nrl_linear_gelu_fwd_img(x2.data_ptr(), w1.data_ptr(), b1.data_ptr(), M, N1, K1, hh.data_ptr(), g.data_ptr(),
                                               ws.data_ptr(), ws.numel(), ready, _stream()), "nrl_linear_gelu_fwd")
        if key is not None:
            images1.commit("fwd", key)
        t = torch.empty((M, N2), dtype=torch.float32, device=dev)
        ws, ready, key = _image_ws(images2, "fwd", w2, lib, dev)
        _lib.check(lib.nrl_linear_fwd_img(g.data_ptr(), w2.data_ptr(), b2.data_ptr(), M, N2, K2, t.data_ptr(), ws.data_ptr(),
                                          ws.numel(), ready, _stream()), "nrl_linear_fwd")
        if key is not None:
            images2.commit("fwd", key)
        y, z, stats = _glue_fwd(lib, t, x2, gamma, beta, eps, p_hid, seed_hid, save)
        if save:
            nig = ctx.needs_input_grad
            need_w1, need_w2 = nig[1] or nig[2], nig[3] or nig[4]
            ctx.save_for_backward(x2 if need_w1 else None, hh, g if need_w2 else None, z, stats, w1, b1, w2, b2, gamma, beta)
            ctx.cfg = (float(p_hid), int(seed_hid), tuple(x.shape))
            ctx.grad_bufs, ctx.engine, ctx.images = grad_bufs, _lib.engine_code(), (images1, images2)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, d_y):
        lib = _lib.load()
        _lib.require_engine(ctx.engine, "feed-forward block")
        x2, hh, g, z, stats, w1, b1, w2, b2, gamma, beta = ctx.saved_tensors
        p_hid, seed_hid, in_shape = ctx.cfg
        images1, images2 = ctx.images
        N1, K1 = w1.shape
        N2, K2 = w2.shape
        M = z.shape[0]
        dev = d_y.device
        d_y = _chk(d_y.reshape(M, N2), torch.float32, "d_out")
        nig = ctx.needs_input_grad
        gb = ctx.grad_bufs
        rets = [None] * 6

        def targets(i0, ps):
            bufs, r = _grad_targets(ps, gb[i0:i0 + len(ps)] if gb is not None else None)
            rets[i0:i0 + len(ps)] = r
            return bufs

        dg = dbt = None
        if nig[5] or nig[6]:
            bufs = targets(4, [gamma, beta])
            dg, dbt = bufs[0].data_ptr(), bufs[1].data_ptr()
        d_t, d_res = _glue_bwd(lib, d_y, z, stats, gamma, p_hid, seed_hid, dg, dbt)
        if g is not None:
            bufs = targets(2, [w2, b2])
            _wgrad_only(lib, g, w2, d_t, M, bufs[0], bufs[1], dev)
        d_x = None
        if not nig[0] and x2 is None:
            return (None, *rets, None, None, None, None, None, None)
        d_h = torch.empty((M, N1), dtype=torch.float32, device=dev)
        ws, ready, key = _image_ws(images2, "bwd", w2, lib, dev)
        _lib.check(lib.nrl_linear_dgrad_gelu_img(w2.data_ptr(), d_t.data_ptr(), hh.data_ptr(), M, N2, K2, d_h.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), ready, _stream()), "nrl_linear_dgrad_gelu")
        if key is not None:
            images2.commit("bwd", key)
        if x2 is not None:
            bufs = targets(0, [w1, b1])
            _wgrad_only(lib, x2, w1, d_h, M, bufs[0], bufs[1], dev)
        if nig[0]:
            d_x = torch.empty((M, K1), dtype=torch.float32, device=dev)
            ws, ready, key = _image_ws(images1, "bwd", w1, lib, dev)
            _lib.check(lib.nrl_linear_dgrad_add_img(d_h.data_ptr(), w1.data_ptr(), M, N1, K1, d_res.data_ptr(), d_x.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), ready, _stream()), "nrl_linear_dgrad_add")
            if key is not None:
                images1.commit("bwd", key)
            d_x = d_x.view(in_shape)
        return (d_x, *rets, None, None, None, None, None, None)


def sdpa_supported(n_batch: int, seq_len: int, heads: int, head_dim: int) -> bool:
    return bool(_lib.load().nrl_sdpa_supported(int(n_batch), int(seq_len), int(heads), int(head_dim)))


_IMAGE_GENERATION = [0]
_STEP_GENERATION = [0]      # bumped by an optimizer that writes parameters through raw pointers (trainer.FusedAdam.begin_step)


# Optimizers of THIS library that are alive (held weakly: the permission below ends with the optimizer -- round-5 advisor: a
# process-wide counter that only ever grew kept trainable-weight images for every module of the process once ANY trainer had
# existed, including modules a foreign optimizer drives through ``p.data``).  Every parameter such an optimizer owns carries a weak
# reference to it (``_nrl_step_driver``).
_STEP_DRIVERS = weakref.WeakSet()
_ANON_DRIVERS = []          # drivers registered without an object (a process-wide promise; tests)


class _AnonDriver:
    pass


def next_optimizer_step() -> None:
    """Tells the per-step weight-image caches (``FrozenImages(allow_trainable=True)``) that trainable weights are about to change."""
    _STEP_GENERATION[0] += 1


def register_step_driver(driver=None, params=()) -> None:
    """Called by an optimizer that promises ``next_optimizer_step()`` before every parameter write (trainer.FusedAdam): `driver` is
    the optimizer (kept weakly), `params` the parameters it writes.  Without arguments: a process-wide promise for every weight."""
    if driver is None:
        driver = _AnonDriver()
        _ANON_DRIVERS.append(driver)
    _STEP_DRIVERS.add(driver)
    ref = weakref.ref(driver)
    for p in params:
        p._nrl_step_driver = ref


def _owned_by_live_driver(w) -> bool:
    ref = getattr(w, "_nrl_step_driver", None)
    return ref is not None and ref() is not None


@contextlib.contextmanager
def no_step_drivers():
    """The process as if no optimizer of this library existed (tests)."""
    global _STEP_DRIVERS
    saved, anon = _STEP_DRIVERS, list(_ANON_DRIVERS)
    _STEP_DRIVERS = weakref.WeakSet()
    del _ANON_DRIVERS[:]
    try:
        yield
    finally:
        _STEP_DRIVERS = saved
        _ANON_DRIVERS[:] = anon


def step_images_allowed(*weights) -> bool:
    """Whether images (or tables) built from TRAINABLE weights may be kept within an optimizer step.  Only under this library's own
    optimizer: a foreign writer that updates through ``p.data`` (apex, legacy AdamW forks, EMA weight swaps, manual
    ``.data.copy_``) moves neither the parameter's version counter nor the step generation, and the kernels would keep multiplying
    by the old weights (round-4 advisor finding).  With `weights` given, every trainable one of them must be OWNED by a live
    optimizer of this library (or a process-wide promise must stand); without, any live driver answers for the process.  Outside
    ``trainer.NRMSTrainer`` -- the Lightning / Hydra drop-in path, whose optimizer comes from the config -- a trainable weight's
    images are rebuilt on every call."""
    if len(_STEP_DRIVERS) == 0:
        return False
    if _ANON_DRIVERS:
        return True
    trainable = [w for w in weights if w is not None and getattr(w, "requires_grad", False)]
    return all(_owned_by_live_driver(w) for w in trainable)


def invalidate_frozen_images() -> None:
    """Drops every cached weight image of every ``FrozenImages`` (a generation counter that is part of their keys): call it
    after writing a frozen weight through a path no version counter sees (``w.data.copy_()``, raw-pointer writes)."""
    _IMAGE_GENERATION[0] += 1


class FrozenImages:
    """The matrix-core weight images of ONE frozen ``nn.Linear`` weight, kept across calls (``nrl_linear_fwd_img`` /
    ``nrl_linear_bwd_img``): the forward's and the backward's (transposed) image each in its own buffer, rebuilt when the
    weight tensor was modified in place or replaced (its version counter / storage address), when the engine or the
    kernel-selection switches changed, after ``invalidate()`` (``NrlLinear`` calls it from ``load_state_dict`` and whenever
    it sees the weight trainable, so a trained-then-refrozen layer never meets an old image) or after the module-wide
    ``invalidate_frozen_images()``.  For weights NO optimizer updates: the fused Adam writes parameters through raw
    pointers, which no version counter sees -- ``NrlLinear`` therefore uses this for ``requires_grad == False`` weights only
    (the PLM body's frozen layers, text.py:69-73: two thirds of a config-4 step's image builds).
    A buffer is marked ready only by ``commit`` AFTER the C call that built its image returned NRL_OK."""

    def __init__(self, allow_trainable: bool = False):
        # allow_trainable: images of a TRAINABLE weight, valid within one optimizer step (the PLM encoder is called twice per
        # step -- history, candidates -- and every trainable projection rebuilt its forward and its backward image in both
        # calls).  Sound only because every writer of such a weight moves the key: torch.optim's in-place updates bump the
        # tensor's version counter, this library's FusedAdam bumps the module-wide generation (``begin_step``).
        self._buf = {}
        self._key = {}
        self._allow_trainable = bool(allow_trainable)

    def invalidate(self) -> None:
        self._key = {}

    def buffer(self, which: str, w: torch.Tensor, lib, device):
        """-> (buffer, ready flag, key to ``commit`` once the call succeeded); (None, 0, None) for a trainable weight."""
        if w.requires_grad and not self._allow_trainable:
            self.invalidate()
            return None, 0, None
        key = (w.data_ptr(), w._version, tuple(w.shape), _lib.engine_code(), _lib.options_word(), str(device),
               _IMAGE_GENERATION[0], _STEP_GENERATION[0] if self._allow_trainable else 0, bool(w.requires_grad))
        buf = self._buf.get(which)
        if buf is None or buf.device != device:
            N, K = w.shape
            buf = self._buf[which] = torch.empty(max(lib.nrl_linear_workspace_bytes(N, K), 256), dtype=torch.uint8, device=device)
            self._key[which] = None
        ready = 1 if self._key.get(which) == key else 0
        if not ready:
            self._key[which] = None          # a failed build must not leave the previous key standing over a half-written image
        return buf, ready, key

    def buffer_multi(self, which: str, weights, nbytes: int, device):
        """``buffer`` for ONE image built from several weights (the fused query / key / value projections): any trainable one makes
        the image a trainable weight's."""
        trainable = any(w.requires_grad for w in weights)
        if trainable and not self._allow_trainable:
            self.invalidate()
            return None, 0, None
        key = (tuple((w.data_ptr(), w._version, tuple(w.shape), bool(w.requires_grad)) for w in weights), _lib.engine_code(),
               _lib.options_word(), str(device), _IMAGE_GENERATION[0], _STEP_GENERATION[0] if self._allow_trainable else 0)
        buf = self._buf.get(which)
        if buf is None or buf.device != device or buf.numel() < nbytes:
            buf = self._buf[which] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
            self._key[which] = None
        ready = 1 if self._key.get(which) == key else 0
        if not ready:
            self._key[which] = None
        return buf, ready, key

    def commit(self, which: str, key) -> None:
        if key is not None:
            self._key[which] = key


class MhaFn(GradAwareFunction):
    """``nn.MultiheadAttention(x, x, x)[0]`` with ``batch_first=False``: x (S, Bt, D) -> (S, Bt, D), attention over
    S.  ``scale`` = the factor applied to q (None: 1/sqrt(D / heads))."""

    @staticmethod
    def forward(ctx, x, w_in, b_in, w_o, b_o, heads, scale, grad_bufs):
        from ._lib import NrlMhaParams
        lib = _lib.load()
        x = _chk(x, torch.float32, "input")
        params = [_chk(t, torch.float32, n) for t, n in zip(
            (w_in, b_in, w_o, b_o), ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias"))]
        if x.dim() != 3:
            raise ValueError("newsreclib_amd: multi-head attention expects (seq, batch, dim)")
        S, Bt, D = x.shape
        if params[0].shape != (3 * D, D) or params[1].shape != (3 * D,) or params[2].shape != (D, D) or \
                params[3].shape != (D,):
            raise ValueError("newsreclib_amd: inconsistent attention parameter shapes")
        engine = _lib.engine_code()
        mp = NrlMhaParams(*[p.data_ptr() for p in params], D, int(heads), float(scale or 0.0), engine)
        ws = torch.empty(max(lib.nrl_mha_workspace_bytes(S, Bt, D, int(heads)), 256), dtype=torch.uint8, device=x.device)
        out = torch.empty_like(x)
        save = saving(ctx)
        _lib.check(lib.nrl_mha_fwd(ctypes.byref(mp), x.data_ptr(), S, Bt, int(save), out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), _stream()), "nrl_mha_fwd")
        if save:
            ctx.save_for_backward(x, *params)
            ctx.ws, ctx.cfg, ctx.grad_bufs, ctx.engine = ws, (int(heads), float(scale or 0.0)), grad_bufs, engine
            ctx.options = _lib.options_mask()
        return out

    @staticmethod
    def backward(ctx, d_out):
        from ._lib import NrlMhaGrads, NrlMhaParams
        lib = _lib.load()
        x, *params = ctx.saved_tensors
        S, Bt, D = x.shape
        heads, scale = ctx.cfg
        d_out = _chk(d_out, torch.float32, "d_out")
        _lib.require_options(ctx.options, "multi-head attention")
        mp = NrlMhaParams(*[p.data_ptr() for p in params], D, heads, scale, ctx.engine)
        bufs, rets = _grad_targets(params, ctx.grad_bufs)
        mg = NrlMhaGrads(*[b.data_ptr() for b in bufs])
        d_x = torch.empty_like(x)
        _lib.check(lib.nrl_mha_bwd(ctypes.byref(mp), ctypes.byref(mg), x.data_ptr(), S, Bt, d_out.data_ptr(),
                                   d_x.data_ptr(), ctx.ws.data_ptr(), ctx.ws.numel(), _stream()), "nrl_mha_bwd")
        ctx.ws = None
        return (d_x, *rets, None, None, None)
