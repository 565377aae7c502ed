"""``torch.autograd.Function`` wrappers around the LSTUR half of the C ABI (include/newsreclib_amd.h):
CNN text encoder, row-masked embedding lookup, GRU.  Same conventions as ``ops.py`` (no eager
fallback; optional ``grad_bufs`` to accumulate parameter gradients in place)."""
from __future__ import annotations

import ctypes
from typing import Sequence

import torch

from . import _lib
from ._lib import NrlCnnGrads, NrlCnnParams, NrlGruGrads, NrlGruParams
from .ops import GradAwareFunction, _chk, _grad_targets, _stream, saving
from .ops import sort_positions as _sort_positions


def _cnn_params(tensors: Sequence[torch.Tensor], embed_dim: int) -> NrlCnnParams:
    w_c, b_c, w_a, b_a, q_a = tensors
    if w_c.dim() != 4 or w_c.shape[1] != 1 or w_c.shape[3] != embed_dim:
        raise ValueError("newsreclib_amd: cnn.weight must be (num_filters, 1, window, embed_dim)")
    F, _, W, D = w_c.shape
    Q = w_a.shape[0]
    if b_c.shape != (F,) or w_a.shape != (Q, F) or b_a.shape != (Q,) or q_a.shape != (Q,):
        raise ValueError("newsreclib_amd: inconsistent CNN/additive-attention parameter shapes")
    return NrlCnnParams(w_c.data_ptr(), b_c.data_ptr(), w_a.data_ptr(), b_a.data_ptr(), q_a.data_ptr(), D, F, W, Q)


class CnnEncoderFn(GradAwareFunction):
    """``CNNAddAtt.forward`` (reference text.py:163-176): ids (N, L) -> (N, F)."""

    @staticmethod
    def forward(ctx, ids, emb, w_c, b_c, w_a, b_a, q_a, p_drop, seed, stream0, grad_bufs, order=None):
        lib = _lib.load()
        ids = _chk(ids, torch.int64, "ids")
        params = [_chk(t, torch.float32, n) for t, n in zip(
            (emb, w_c, b_c, w_a, b_a, q_a),
            ("embedding", "cnn.weight", "cnn.bias", "linear.weight", "linear.bias", "query"))]
        emb = params[0]
        if ids.dim() != 2:
            raise ValueError("newsreclib_amd: token ids must be (num_news, num_tokens)")
        N, L = ids.shape
        V, D = emb.shape
        cp = _cnn_params(params[1:], D)
        save = saving(ctx)
        ws_bytes = lib.nrl_cnn_encoder_workspace_bytes(N, L, D, cp.num_filters, cp.window, cp.query_dim)
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=ids.device)
        out = torch.empty((N, cp.num_filters), dtype=torch.float32, device=ids.device)
        _lib.check(lib.nrl_cnn_encoder_fwd(ctypes.byref(cp), emb.data_ptr(), V, ids.data_ptr(), N, L, float(p_drop),
                                           int(seed), int(stream0), int(save), out.data_ptr(), ws.data_ptr(),
                                           ws.numel(), _stream()), "nrl_cnn_encoder_fwd")
        if save:
            if order is None:
                order = _sort_positions(ids, V)     # counting sort over the vocabulary (ops.sort_positions)
            from .ops import order_event
            ctx.order_ready = order_event(order)
            order = _chk(order, torch.int64, "order")
            if order.numel() == ids.numel():          # an order from elsewhere (argsort): append the count of id-0 positions
                order = torch.cat([order.reshape(-1), (ids == 0).sum().reshape(1)])
            ctx.save_for_backward(ids, order, *params)
            ctx.ws, ctx.cfg, ctx.grad_bufs = ws, (float(p_drop), int(seed), int(stream0)), grad_bufs
            ctx.engine, ctx.options = _lib.engine_code(), _lib.options_mask()
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        _lib.require_engine(ctx.engine, "the CNN text encoder")
        _lib.require_options(ctx.options, "the CNN text encoder")
        ids, order, *params = ctx.saved_tensors
        from .ops import wait_order
        wait_order(ctx.order_ready)
        emb = params[0]
        p_drop, seed, stream0 = ctx.cfg
        N, L = ids.shape
        V, D = emb.shape
        d_out = _chk(d_out, torch.float32, "d_out")
        cp = _cnn_params(params[1:], D)
        bufs, rets = _grad_targets(params, ctx.grad_bufs)
        cg = NrlCnnGrads(*[b.data_ptr() for b in bufs[1:]])
        _lib.check(lib.nrl_cnn_encoder_bwd(ctypes.byref(cp), ctypes.byref(cg), bufs[0].data_ptr(), V, ids.data_ptr(),
                                           order.data_ptr(), N, L, p_drop, seed, stream0, d_out.data_ptr(),
                                           ctx.ws.data_ptr(), ctx.ws.numel(), _stream()), "nrl_cnn_encoder_bwd")
        ctx.ws = None
        return (None, *rets, None, None, None, None, None)


class EmbeddingRowsFn(GradAwareFunction):
    """``nn.Embedding(padding_idx=0)`` lookup + row mask: category encoder (category.py:72-73, p_row = 0) and
    the masked long-term user vector (user/lstur.py:70-71, ``nn.Dropout2d`` over whole users)."""

    @staticmethod
    def forward(ctx, ids, table, p_row, seed, stream_id, grad_bufs):
        lib = _lib.load()
        ids = _chk(ids, torch.int64, "ids")
        table = _chk(table, torch.float32, "embedding")
        if ids.dim() != 1:
            raise ValueError("newsreclib_amd: ids must be 1-D")
        n, D = ids.shape[0], table.shape[1]
        out = torch.empty((n, D), dtype=torch.float32, device=ids.device)
        _lib.check(lib.nrl_embedding_rows_fwd(table.data_ptr(), ids.data_ptr(), n, D, float(p_row), int(seed),
                                              int(stream_id), out.data_ptr(), _stream()), "nrl_embedding_rows_fwd")
        ctx.save_for_backward(ids, table)
        ctx.cfg, ctx.grad_bufs = (float(p_row), int(seed), int(stream_id)), grad_bufs
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        ids, table = ctx.saved_tensors
        p_row, seed, stream_id = ctx.cfg
        d_out = _chk(d_out, torch.float32, "d_out")
        bufs, rets = _grad_targets([table], ctx.grad_bufs)
        _lib.check(lib.nrl_embedding_rows_bwd(d_out.data_ptr(), ids.data_ptr(), ids.shape[0], table.shape[1], p_row,
                                              seed, stream_id, bufs[0].data_ptr(), _stream()),
                   "nrl_embedding_rows_bwd")
        return (None, rets[0], None, None, None, None)


class GruFn(GradAwareFunction):
    """``nn.GRU`` over a packed batch-first sequence, last hidden state (user/lstur.py:74-83):
    hist (B, T, Din), lengths (B) int64, h0 (B, Hd) or None -> (B, Hd)."""

    @staticmethod
    def forward(ctx, hist, lengths, h0, w_ih, w_hh, b_ih, b_hh, grad_bufs):
        lib = _lib.load()
        hist = _chk(hist, torch.float32, "hist")
        lengths = _chk(lengths, torch.int64, "lengths")
        params = [_chk(t, torch.float32, n) for t, n in zip((w_ih, w_hh, b_ih, b_hh),
                                                            ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))]
        if hist.dim() != 3:
            raise ValueError("newsreclib_amd: hist must be (batch, max_len, input_dim)")
        B, T, Din = hist.shape
        Hd = params[1].shape[1]
        if params[0].shape != (3 * Hd, Din) or params[1].shape != (3 * Hd, Hd) or params[2].shape != (3 * Hd,) \
                or params[3].shape != (3 * Hd,):
            raise ValueError("newsreclib_amd: inconsistent GRU parameter shapes")
        if h0 is not None:
            h0 = _chk(h0, torch.float32, "h0")
            if h0.shape != (B, Hd):
                raise ValueError("newsreclib_amd: h0 must be (batch, hidden_dim)")
        gp = NrlGruParams(*[p.data_ptr() for p in params], Din, Hd)
        ws_bytes = lib.nrl_gru_workspace_bytes(B, T, Din, Hd)
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=hist.device)
        out = torch.empty((B, Hd), dtype=torch.float32, device=hist.device)
        save = saving(ctx)
        _lib.check(lib.nrl_gru_fwd(ctypes.byref(gp), hist.data_ptr(), lengths.data_ptr(),
                                   h0.data_ptr() if h0 is not None else None, B, T, int(save), out.data_ptr(),
                                   ws.data_ptr(), ws.numel(), _stream()), "nrl_gru_fwd")
        if save:
            ctx.save_for_backward(hist, lengths, *params)
            ctx.has_h0 = h0 is not None
            ctx.ws, ctx.grad_bufs, ctx.engine, ctx.options = ws, grad_bufs, _lib.engine_code(), _lib.options_mask()
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        _lib.require_engine(ctx.engine, "the GRU")
        _lib.require_options(ctx.options, "the GRU")
        hist, lengths, *params = ctx.saved_tensors
        B, T, Din = hist.shape
        Hd = params[1].shape[1]
        d_out = _chk(d_out, torch.float32, "d_out")
        gp = NrlGruParams(*[p.data_ptr() for p in params], Din, Hd)
        bufs, rets = _grad_targets(params, ctx.grad_bufs)
        gg = NrlGruGrads(*[b.data_ptr() for b in bufs])
        d_hist = torch.empty_like(hist)
        d_h0 = torch.empty((B, Hd), dtype=torch.float32, device=hist.device) if ctx.has_h0 else None
        _lib.check(lib.nrl_gru_bwd(ctypes.byref(gp), ctypes.byref(gg), hist.data_ptr(), lengths.data_ptr(), None, B, T,
                                   d_out.data_ptr(), d_hist.data_ptr(),
                                   d_h0.data_ptr() if d_h0 is not None else None, ctx.ws.data_ptr(), ctx.ws.numel(),
                                   _stream()), "nrl_gru_bwd")
        ctx.ws = None
        return (d_hist, None, d_h0, *rets, None)


class CnnMhsaEncoderFn(GradAwareFunction):
    """``CNNMHSAAddAtt.forward`` (reference text.py:291-309): ids (N, L) -> (N, F).  ``w_c`` in the
    (F, 1, W, D) layout of ``CnnEncoderFn`` (the module permutes its ``nn.Conv1d`` weight (F, D, W))."""

    @staticmethod
    def forward(ctx, ids, emb, w_c, b_c, w_in, b_in, w_o, b_o, w_a, b_a, q_a, heads, p_drop, seed, stream0, grad_bufs,
                order=None):
        from .ops import _block_grads, _block_params
        lib = _lib.load()
        ids = _chk(ids, torch.int64, "ids")
        params = [_chk(t, torch.float32, "cnn-mhsa encoder parameter") for t in
                  (emb, w_c, b_c, w_in, b_in, w_o, b_o, w_a, b_a, q_a)]
        emb, w_c, b_c = params[:3]
        N, L = ids.shape
        V, D = emb.shape
        F_, _, W, _ = w_c.shape
        engine = _lib.engine_code()
        options = _lib.options_word()
        bp = _block_params(params[3:], heads, engine, options)
        cp = NrlCnnParams(w_c.data_ptr(), b_c.data_ptr(), None, None, None, D, F_, W, bp.query_dim)
        save = saving(ctx)
        ws = torch.empty(max(lib.nrl_cnn_mhsa_encoder_workspace_bytes(N, L, D, F_, W, heads, bp.query_dim), 256),
                         dtype=torch.uint8, device=ids.device)
        out = torch.empty((N, F_), dtype=torch.float32, device=ids.device)
        _lib.check(lib.nrl_cnn_mhsa_encoder_fwd(ctypes.byref(cp), ctypes.byref(bp), emb.data_ptr(), V, ids.data_ptr(), N,
                                                L, float(p_drop), int(seed), int(stream0), int(save), out.data_ptr(),
                                                ws.data_ptr(), ws.numel(), _stream()), "nrl_cnn_mhsa_encoder_fwd")
        if save:
            if order is None:
                order = _sort_positions(ids, V)     # counting sort over the vocabulary (ops.sort_positions)
            from .ops import order_event
            ctx.order_ready = order_event(order)
            ctx.save_for_backward(ids, _chk(order, torch.int64, "order"), *params)
            ctx.ws, ctx.cfg, ctx.grad_bufs = ws, (heads, float(p_drop), int(seed), int(stream0)), grad_bufs
            ctx.engine, ctx.options = engine, options
        return out

    @staticmethod
    def backward(ctx, d_out):
        from .ops import _block_grads, _block_params
        lib = _lib.load()
        ids, order, *params = ctx.saved_tensors
        from .ops import wait_order
        wait_order(ctx.order_ready)
        emb, w_c, b_c = params[:3]
        heads, p_drop, seed, stream0 = ctx.cfg
        N, L = ids.shape
        V, D = emb.shape
        F_, _, W, _ = w_c.shape
        d_out = _chk(d_out, torch.float32, "d_out")
        bp = _block_params(params[3:], heads, ctx.engine, ctx.options)   # (engine and switches of the forward)
        cp = NrlCnnParams(w_c.data_ptr(), b_c.data_ptr(), None, None, None, D, F_, W, bp.query_dim)
        bufs, rets = _grad_targets(params, ctx.grad_bufs)
        cg = NrlCnnGrads(bufs[1].data_ptr(), bufs[2].data_ptr(), None, None, None)
        bg = _block_grads(bufs[3:])
        _lib.check(lib.nrl_cnn_mhsa_encoder_bwd(ctypes.byref(cp), ctypes.byref(cg), ctypes.byref(bp), ctypes.byref(bg),
                                                bufs[0].data_ptr(), V, ids.data_ptr(), order.data_ptr(), N, L, p_drop,
                                                seed, stream0, d_out.data_ptr(), ctx.ws.data_ptr(), ctx.ws.numel(),
                                                _stream()), "nrl_cnn_mhsa_encoder_bwd")
        ctx.ws = None
        return (None, *rets, None, None, None, None, None, None)
