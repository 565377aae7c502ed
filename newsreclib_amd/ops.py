"""``torch.autograd.Function`` wrappers around the C ABI (include/newsreclib_amd.h).

PyTorch is plumbing here: it owns device memory, streams and the autograd tape; every FLOP of the
path runs in the HIP library.  There is deliberately no eager fallback.

Parameter-gradient convention: when a ``grad_bufs`` tuple is supplied (the persistent ``.grad`` /
flat data-parallel gradient buffer of each parameter, see ``trainer.FlatParams``), the kernels
accumulate straight into those buffers and autograd receives ``None`` for the parameters --
no 84 MB zero-fill + add per encoder call.  Without it the functions return ordinary gradients.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import NrlBlockGrads, NrlBlockParams


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """Raw handle of the current HIP stream of the current device.  (Through the private fast path when this torch has it:
    ``torch.cuda.current_stream()`` builds a Stream object per call, ~9 us, and every ctypes wrapper asks once -- 16 calls = 0.15 ms
    of host time per train step, a sixth of the whole issue time at B = 32 where host and device are neck and neck.)"""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"newsreclib_amd: `{name}` must live on the GPU (got {t.device}); "
                           "there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"newsreclib_amd: `{name}` must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _block_params(tensors: Sequence[torch.Tensor], heads: int, engine: int = 0, options: int = 0) -> NrlBlockParams:
    w_in, b_in, w_o, b_o, w_a, b_a, q_a = tensors
    D, Q = w_o.shape[0], w_a.shape[0]
    if w_in.shape != (3 * D, D) or b_in.shape != (3 * D,) or b_o.shape != (D,) or \
            w_a.shape != (Q, D) or b_a.shape != (Q,) or q_a.shape != (Q,):
        raise ValueError("newsreclib_amd: inconsistent MHSA/additive-attention parameter shapes")
    return NrlBlockParams(w_in.data_ptr(), b_in.data_ptr(), w_o.data_ptr(), b_o.data_ptr(),
                          w_a.data_ptr(), b_a.data_ptr(), q_a.data_ptr(), D, heads, Q, int(engine), int(options))


def _block_grads(bufs: Sequence[torch.Tensor]) -> NrlBlockGrads:
    return NrlBlockGrads(*[b.data_ptr() for b in bufs])


def _grad_targets(params: Sequence[torch.Tensor], grad_bufs: Optional[Sequence[Optional[torch.Tensor]]]):
    """-> (buffers the kernels add into, gradients to hand back to autograd)."""
    bufs, rets = [], []
    for i, p in enumerate(params):
        gb = grad_bufs[i] if grad_bufs is not None else None
        if gb is not None:
            if gb.shape != p.shape or not gb.is_contiguous() or gb.dtype != torch.float32:
                raise ValueError("newsreclib_amd: bad grad buffer")
            bufs.append(gb)
            rets.append(None)
        else:
            z = torch.zeros_like(p)
            bufs.append(z)
            rets.append(z)
    return bufs, rets


class deferred_weight_grads:
    """While active, ``UserEncoderFn.backward`` issues its three weight gradients (phase 2 of ``nrl_user_encoder_bwd_phase``)
    on ``stream`` instead of the caller's: they are ~100 us of few-row launches that nothing in the rest of the backward reads,
    so they run beside the chip-filling news-encoder backward instead of in front of it.  Leaving the context makes the
    current stream wait for ``stream`` (the join) -- do it before anything reads the gradient buffers (all-reduce, Adam).
    Only calls that accumulate into caller-owned ``grad_bufs`` defer (gradients handed back to autograd are consumed on the
    caller's stream right after the backward returns).  Used by ``trainer.NRMSTrainer``; off everywhere else."""

    _active = None

    def __init__(self, stream: "torch.cuda.Stream"):
        self.stream = stream
        self.keep = []          # tensors the side stream still reads: kept alive until the join

    def __enter__(self):
        deferred_weight_grads._active = self
        return self

    def __exit__(self, *exc):
        deferred_weight_grads._active = None
        self.joined = bool(self.keep)      # (the caller's stream now waits for everything issued on `stream` so far)
        if self.keep:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.keep = []
        return False


import threading

_TLS = threading.local()


class GradAwareFunction(torch.autograd.Function):
    """``torch.autograd.Function`` whose ``forward`` can tell whether the CALLER had gradients enabled.  Inside ``forward``
    autograd has switched grad mode off, and ``ctx.needs_input_grad`` only reflects the inputs' ``requires_grad`` flags -- so
    a forward under ``torch.no_grad()`` over trainable parameters looked like a training forward and ran the training shapes of
    the kernels (activations saved, 0.57 + 0.23 ms instead of 0.41 + 0.18 ms for the two news-encoder launches of an evaluation
    batch: found in round 4 with a kernel trace of the evaluation forward).  ``apply`` records the caller's grad mode;
    ``saving(ctx)`` = "a backward can follow"."""

    @classmethod
    def apply(cls, *args, **kwargs):
        prev = getattr(_TLS, "grad", True)
        _TLS.grad = torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _TLS.grad = prev


def saving(ctx, inputs=None) -> bool:
    """True when the forward must keep what its backward needs: the caller has gradients enabled AND some (listed) input
    requires one."""
    need = ctx.needs_input_grad if inputs is None else ctx.needs_input_grad[inputs]
    return bool(getattr(_TLS, "grad", True)) and any(need)


class NewsEncoderFn(GradAwareFunction):
    """``MHSAAddAtt.forward`` (reference text.py:222-236): ids (N, L) -> (N, D)."""

    @staticmethod
    def forward(ctx, ids, emb, w_in, b_in, w_o, b_o, w_a, b_a, q_a, heads, p_drop, seed, stream0,
                grad_bufs, order=None, table_grad_hook=None):
        lib = _lib.load()
        ids = _chk(ids, torch.int64, "ids")
        params = [_chk(t, torch.float32, n) for t, n in zip(
            (emb, w_in, b_in, w_o, b_o, w_a, b_a, q_a),
            ("embedding", "in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias",
             "linear.weight", "linear.bias", "query"))]
        emb = params[0]
        if ids.dim() != 2:
            raise ValueError("newsreclib_amd: token ids must be (num_news, num_tokens)")
        N, L = ids.shape
        V, D = emb.shape
        engine, options = _lib.engine_code(), _lib.options_word()
        bp = _block_params(params[1:], heads, engine, options)
        save = saving(ctx)
        ws_bytes = lib.nrl_news_encoder_workspace_bytes(N, L, D, heads, bp.query_dim)
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=ids.device)
        out = torch.empty((N, D), dtype=torch.float32, device=ids.device)
        _lib.check(lib.nrl_news_encoder_fwd(ctypes.byref(bp), emb.data_ptr(), V, ids.data_ptr(), N, L,
                                            float(p_drop), int(seed), int(stream0), int(save),
                                            out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
                   "nrl_news_encoder_fwd")
        if save:
            if order is None:
                # id-sorted visiting order for the table gradient (index bookkeeping on the int64 ids;
                # `prepare_batch` precomputes it once per batch so the step does not pay the sort)
                order = sort_positions_async(ids, V) if os.environ.get("NRL_SORT_ASYNC", "1") == "2" else sort_positions(ids, V)
            ctx.order_ready = order_event(order)      # (side-stream sort: the backward waits for it)
            order = _chk(order, torch.int64, "order")
            if order.numel() == ids.numel():          # an order from elsewhere (argsort): append the count of id-0 positions
                order = torch.cat([order.reshape(-1), (ids == 0).sum().reshape(1)])
            ctx.save_for_backward(ids, order, *params)
            ctx.ws, ctx.cfg, ctx.grad_bufs = ws, (heads, float(p_drop), int(seed), int(stream0)), grad_bufs
            ctx.table_grad_hook, ctx.engine, ctx.options = table_grad_hook, engine, options
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        ids, order, *params = ctx.saved_tensors
        emb = params[0]
        heads, p_drop, seed, stream0 = ctx.cfg
        N, L = ids.shape
        V, D = emb.shape
        d_out = _chk(d_out, torch.float32, "d_out")
        # the engine and the kernel-selection switches the forward ran under travel with the call: the backward reads the
        # workspace in the format the forward wrote, whatever the process defaults are by now
        bp = _block_params(params[1:], heads, ctx.engine, ctx.options)
        bufs, rets = _grad_targets(params, ctx.grad_bufs)
        bg = _block_grads(bufs[1:])
        ws = ctx.ws
        wait_order(ctx.order_ready)
        def run(phase):
            _lib.check(lib.nrl_news_encoder_bwd(ctypes.byref(bp), ctypes.byref(bg), emb.data_ptr(), bufs[0].data_ptr(), V,
                                                ids.data_ptr(), order.data_ptr(), N, L, p_drop, seed, stream0,
                                                d_out.data_ptr(), phase, ws.data_ptr(), ws.numel(), _stream()),
                       "nrl_news_encoder_bwd")

        hook = ctx.table_grad_hook
        if hook is None:
            run(0)
        else:
            # the embedding-table gradient (>96 % of all gradient bytes) is complete after phase 1: let the
            # data-parallel trainer start its all-reduce now, under the weight-gradient GEMMs of phase 2
            run(1)
            _call_table_grad_hook(hook, bufs[0], ids)
            run(2)
        ctx.ws = None
        return (None, *rets, None, None, None, None, None, None, None)


def _call_table_grad_hook(hook, grad: torch.Tensor, ids: torch.Tensor) -> None:
    """``hook(grad, ids)`` -- or ``hook(grad)`` for a callable that takes one positional argument (the signature the hook had
    before the touched-row exchange needed the ids)."""
    import inspect
    try:
        params = [p for p in inspect.signature(hook).parameters.values()
                  if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD, p.VAR_POSITIONAL)]
        two = any(p.kind == p.VAR_POSITIONAL for p in params) or len(params) >= 2
    except (TypeError, ValueError):
        two = True
    if two:
        hook(grad, ids)
    else:
        hook(grad)


class UserEncoderFn(GradAwareFunction):
    """NRMS ``UserEncoder.forward`` (reference user/nrms.py:32-41): hist (B, H, D) -> (B, D)."""

    @staticmethod
    def forward(ctx, hist, w_in, b_in, w_o, b_o, w_a, b_a, q_a, heads, grad_bufs, p_drop=0.0, seed=0,
                input_dropout=True, stream0=0):
        lib = _lib.load()
        hist = _chk(hist, torch.float32, "hist_news_vector")
        params = [_chk(t, torch.float32, "user encoder parameter") for t in (w_in, b_in, w_o, b_o, w_a, b_a, q_a)]
        if hist.dim() != 3:
            raise ValueError("newsreclib_amd: hist_news_vector must be (batch, history, dim)")
        B, H, D = hist.shape
        engine = _lib.engine_code()
        options = _lib.options_word()
        bp = _block_params(params, heads, engine, options)
        if bp.embed_dim != D:
            raise ValueError("newsreclib_amd: hist feature dim does not match the encoder")
        save = saving(ctx)
        ws_bytes = lib.nrl_user_encoder_workspace_bytes(B, H, D, heads, bp.query_dim)
        ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=hist.device)
        out = torch.empty((B, D), dtype=torch.float32, device=hist.device)
        _lib.check(lib.nrl_user_encoder_fwd(ctypes.byref(bp), hist.data_ptr(), B, H, float(p_drop), int(seed),
                                            int(stream0), int(bool(input_dropout)), int(save), out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), _stream()), "nrl_user_encoder_fwd")
        if save:
            ctx.save_for_backward(hist, *params)
            ctx.ws, ctx.heads, ctx.grad_bufs, ctx.engine = ws, heads, grad_bufs, engine
            ctx.options = options
            ctx.drop = (float(p_drop), int(seed), int(stream0), int(bool(input_dropout)))
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        hist, *params = ctx.saved_tensors
        B, H, D = hist.shape
        d_out = _chk(d_out, torch.float32, "d_out")
        bp = _block_params(params, ctx.heads, ctx.engine, ctx.options)   # (engine and switches of the forward)
        bufs, rets = _grad_targets(params, ctx.grad_bufs)
        bg = _block_grads(bufs)
        d_hist = torch.empty_like(hist)
        ws = ctx.ws

        def run(phase):
            _lib.check(lib.nrl_user_encoder_bwd_phase(ctypes.byref(bp), ctypes.byref(bg), hist.data_ptr(), B, H,
                                                      ctx.drop[0], ctx.drop[1], ctx.drop[2], ctx.drop[3], d_out.data_ptr(),
                                                      d_hist.data_ptr(), phase, ws.data_ptr(), ws.numel(), _stream()),
                       "nrl_user_encoder_bwd_phase")

        defer = deferred_weight_grads._active
        if defer is None or any(r is not None for r in rets):
            run(0)
        else:
            run(1)
            defer.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(defer.stream):
                run(2)
            defer.keep.append((ws, hist, params, bufs))
        ctx.ws = None
        return (d_hist, *rets, None, None, None, None, None, None)


class ToDenseBatchFn(GradAwareFunction):
    """``to_dense_batch`` values (torch_geometric; nrms_module.py:233,237): (N, D) -> (B, max_len, D)."""

    @staticmethod
    def forward(ctx, x, offsets, batch_size, max_len):
        lib = _lib.load()
        x = _chk(x, torch.float32, "x")
        offsets = _chk(offsets, torch.int64, "offsets")
        N, D = x.shape
        dense = torch.empty((batch_size, max_len, D), dtype=torch.float32, device=x.device)
        _lib.check(lib.nrl_to_dense_batch_fwd(x.data_ptr(), offsets.data_ptr(), batch_size, max_len, D,
                                              dense.data_ptr(), _stream()), "nrl_to_dense_batch_fwd")
        ctx.save_for_backward(offsets)
        ctx.shape = (N, D, batch_size, max_len)
        return dense

    @staticmethod
    def backward(ctx, d_dense):
        lib = _lib.load()
        (offsets,) = ctx.saved_tensors
        N, D, B, max_len = ctx.shape
        d_dense = _chk(d_dense, torch.float32, "d_dense")
        d_x = torch.empty((N, D), dtype=torch.float32, device=d_dense.device)
        _lib.check(lib.nrl_to_dense_batch_bwd(d_dense.data_ptr(), offsets.data_ptr(), B, max_len, D, N,
                                              d_x.data_ptr(), _stream()), "nrl_to_dense_batch_bwd")
        return d_x, None, None, None


class HistMeanFn(GradAwareFunction):
    """late fusion (nrms_module.py:243-248): (B, max_len, D) zero-padded history -> (B, D) mean over the TRUE
    number of clicks."""

    @staticmethod
    def forward(ctx, hist, offsets):
        lib = _lib.load()
        hist = _chk(hist, torch.float32, "hist_news_vector")
        offsets = _chk(offsets, torch.int64, "offsets")
        B, H, D = hist.shape
        user = torch.empty((B, D), dtype=torch.float32, device=hist.device)
        _lib.check(lib.nrl_hist_mean_fwd(hist.data_ptr(), offsets.data_ptr(), B, H, D, user.data_ptr(), _stream()),
                   "nrl_hist_mean_fwd")
        ctx.save_for_backward(offsets)
        ctx.shape = (B, H, D)
        return user

    @staticmethod
    def backward(ctx, d_user):
        lib = _lib.load()
        (offsets,) = ctx.saved_tensors
        B, H, D = ctx.shape
        d_user = _chk(d_user, torch.float32, "d_user")
        d_hist = torch.empty((B, H, D), dtype=torch.float32, device=d_user.device)
        _lib.check(lib.nrl_hist_mean_bwd(d_user.data_ptr(), offsets.data_ptr(), B, H, D, d_hist.data_ptr(), _stream()),
                   "nrl_hist_mean_bwd")
        return d_hist, None


class DotScoresFn(GradAwareFunction):
    """``DotProduct.forward`` (click_predictor.py:9-11): user (B, D) x cand (B, C, D) -> (B, C)."""

    @staticmethod
    def forward(ctx, user, cand):
        lib = _lib.load()
        user = _chk(user, torch.float32, "user_vec")
        cand = _chk(cand, torch.float32, "cand_news_vector")
        B, C, D = cand.shape
        scores = torch.empty((B, C), dtype=torch.float32, device=user.device)
        _lib.check(lib.nrl_dot_scores_fwd(user.data_ptr(), cand.data_ptr(), B, C, D, scores.data_ptr(),
                                          _stream()), "nrl_dot_scores_fwd")
        ctx.save_for_backward(user, cand)
        return scores

    @staticmethod
    def backward(ctx, d_scores):
        lib = _lib.load()
        user, cand = ctx.saved_tensors
        B, C, D = cand.shape
        d_scores = _chk(d_scores, torch.float32, "d_scores")
        d_user, d_cand = torch.empty_like(user), torch.empty_like(cand)
        _lib.check(lib.nrl_dot_scores_bwd(d_scores.data_ptr(), user.data_ptr(), cand.data_ptr(), B, C, D,
                                          d_user.data_ptr(), d_cand.data_ptr(), _stream()),
                   "nrl_dot_scores_bwd")
        return d_user, d_cand


class CrossEntropyFn(GradAwareFunction):
    """``CrossEntropyLoss()(scores, y_true)`` with float targets (nrms_module.py:287-288)."""

    @staticmethod
    def forward(ctx, scores, y_true, grad_scale):
        lib = _lib.load()
        scores = _chk(scores, torch.float32, "scores")
        y_true = _chk(y_true, torch.float32, "y_true")
        B, C = scores.shape
        loss = torch.empty((), dtype=torch.float32, device=scores.device)
        d_scores = torch.empty_like(scores)
        _lib.check(lib.nrl_ce_loss_fwd_bwd(scores.data_ptr(), y_true.data_ptr(), B, C, float(grad_scale),
                                           loss.data_ptr(), d_scores.data_ptr(), _stream()),
                   "nrl_ce_loss_fwd_bwd")
        ctx.save_for_backward(d_scores)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_scores,) = ctx.saved_tensors
        if g.data_ptr() in _UNIT_GRADS:      # the root gradient of a trainer (a registered tensor holding 1.0): d_scores as is
            return d_scores, None, None
        return d_scores * g, None, None


# Root gradients known to hold exactly 1.0 (``register_unit_grad``): a loss that receives one of them as its incoming gradient
# IS the root of the backward and hands its stored gradient on without the multiply (one small launch per step; autograd's own
# ``ones_like`` fill is the other one a trainer saves by passing the tensor to ``loss.backward(gradient=...)``).  The registry
# keeps the tensors alive, so an address can never come back as some other tensor's.
_UNIT_GRADS = {}


def register_unit_grad(t: torch.Tensor) -> torch.Tensor:
    _UNIT_GRADS[t.data_ptr()] = t
    return t


class SupConFn(GradAwareFunction):
    """``SupConLoss()(embeddings=scores, indices_tuple=...)`` as called at nrms_module.py:289-318: scores, y_true
    (B, C) dense, cand_sizes (B,) int64 = real candidates per row."""

    @staticmethod
    def forward(ctx, scores, y_true, cand_sizes, temperature):
        lib = _lib.load()
        scores = _chk(scores, torch.float32, "scores")
        y_true = _chk(y_true, torch.float32, "y_true")
        cand_sizes = _chk(cand_sizes, torch.int64, "cand_sizes")
        B, C = scores.shape
        if y_true.shape != (B, C) or cand_sizes.shape != (B,):
            raise ValueError("newsreclib_amd: inconsistent sup-con loss shapes")
        loss = torch.empty((), dtype=torch.float32, device=scores.device)
        d_scores = torch.empty_like(scores)
        _lib.check(lib.nrl_supcon_loss_fwd_bwd(scores.data_ptr(), y_true.data_ptr(), cand_sizes.data_ptr(), B, C,
                                               float(temperature), 1.0, loss.data_ptr(), d_scores.data_ptr(),
                                               _stream()), "nrl_supcon_loss_fwd_bwd")
        ctx.save_for_backward(d_scores)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_scores,) = ctx.saved_tensors
        return d_scores * g, None, None, None


class SplitRowsFn(GradAwareFunction):
    """(x[:n], x[n:]) whose backward is ONE concatenation.  (Plain slicing makes autograd zero-fill two full-size
    gradients, copy a slice into each and add them: five launches where one does.)"""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = int(n)
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, g_head, g_tail):
        return torch.cat((g_head, g_tail), dim=0), None


def split_rows(x: torch.Tensor, n: int):
    return SplitRowsFn.apply(x, n) if x.requires_grad else (x[:n], x[n:])


# ---- evaluation forwards of the news encoder from a per-token q|k|v table (include/newsreclib_amd.h, ABI v16) ------------
def token_table_supported(seq_len: int, embed_dim: int, heads: int, query_dim: int) -> bool:
    return _lib.engine_code() == 2 and bool(_lib.load().nrl_token_table_supported(int(seq_len), int(embed_dim), int(heads),
                                                                                 int(query_dim)))


def token_table_build(params: Sequence[torch.Tensor], heads: int, buf: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """``params`` = (embedding table, in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, linear.weight, linear.bias,
    query) of an ``MHSAAddAtt``: runs the in-projection once per vocabulary id and builds the back half's weight images into
    ``buf`` (reused when large enough).  None when the geometry has no table (``nrl_token_table_bytes`` == 0)."""
    lib = _lib.load()
    params = [_chk(t.detach(), torch.float32, "parameter") for t in params]
    emb = params[0]
    V, D = emb.shape
    bp = _block_params(params[1:], heads, _lib.engine_code(), _lib.options_word())
    nbytes = int(lib.nrl_token_table_bytes(V, D, int(heads), bp.query_dim))
    if nbytes == 0:
        return None
    if buf is None or buf.numel() < nbytes or buf.device != emb.device:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=emb.device)
    _lib.check(lib.nrl_token_table_build(ctypes.byref(bp), emb.data_ptr(), V, buf.data_ptr(), buf.numel(), _stream()),
               "nrl_token_table_build")
    return buf


def news_encoder_fwd_table(ids: torch.Tensor, table: torch.Tensor, vocab: int, params: Sequence[torch.Tensor],
                           heads: int) -> torch.Tensor:
    """``MHSAAddAtt.forward`` in evaluation mode from a table of ``token_table_build`` (``params`` without the embedding table:
    the seven block parameters): ids (N, L) -> (N, D), EQUAL to the table-less evaluation forward."""
    lib = _lib.load()
    ids = _chk(ids, torch.int64, "ids")
    if ids.dim() != 2:
        raise ValueError("newsreclib_amd: token ids must be (num_news, num_tokens)")
    N, L = ids.shape
    params = [_chk(t.detach(), torch.float32, "parameter") for t in params]
    bp = _block_params(params, heads, _lib.engine_code(), _lib.options_word())
    out = torch.empty((N, bp.embed_dim), dtype=torch.float32, device=ids.device)
    ws = torch.empty(max(int(lib.nrl_news_encoder_fwd_table_workspace_bytes(N, L, int(heads))), 256), dtype=torch.uint8,
                     device=ids.device)
    _lib.check(lib.nrl_news_encoder_fwd_table(ctypes.byref(bp), table.data_ptr(), table.numel(), int(vocab), ids.data_ptr(), N, L,
                                              out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "nrl_news_encoder_fwd_table")
    return out


# ---- plain (non-autograd) entry points ----------------------------------------------------------
def embedding_gather(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    table, ids = _chk(table, torch.float32, "table"), _chk(ids, torch.int64, "ids")
    out = torch.empty(tuple(ids.shape) + (table.shape[1],), dtype=torch.float32, device=table.device)
    _lib.check(lib.nrl_embedding_gather(table.data_ptr(), ids.data_ptr(), ids.numel(), table.shape[1],
                                        out.data_ptr(), _stream()), "nrl_embedding_gather")
    return out


def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    a, w = _chk(a, torch.float32, "a"), _chk(w, torch.float32, "w")
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    b = _chk(bias, torch.float32, "bias").data_ptr() if bias is not None else None
    ws = torch.empty(max(lib.nrl_linear_workspace_bytes(N, K), 256), dtype=torch.uint8, device=a.device)
    _lib.check(lib.nrl_linear_fwd(a.data_ptr(), w.data_ptr(), b, M, N, K, c.data_ptr(), ws.data_ptr(), ws.numel(),
                                  _stream()), "nrl_linear_fwd")
    return c


def dropout_mask(n: int, p: float, seed: int, stream: int, device) -> torch.Tensor:
    lib = _lib.load()
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    _lib.check(lib.nrl_dropout_mask(keep.data_ptr(), n, float(p), int(seed), int(stream), _stream()),
               "nrl_dropout_mask")
    return keep


def adam_step_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
               step: int, lr: float = 1e-4, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
               grad_scale: float = 1.0, zero_grad: bool = False) -> None:
    lib = _lib.load()
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise ValueError("newsreclib_amd.adam_step_: contiguous float32 GPU tensors required")
    _lib.check(lib.nrl_adam_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(),
                                 exp_avg_sq.data_ptr(), param.numel(), lr, betas[0], betas[1], eps,
                                 int(step), float(grad_scale), int(zero_grad), _stream()), "nrl_adam_step")


def adam_rows_mark_(ids: torch.Tensor, mark: torch.Tensor, step: int) -> None:
    """mark[id] = step for the ids of a step's batch (``nrl_adam_rows_mark``; lazy table Adam, trainer.LazyTableAdam)."""
    lib = _lib.load()
    ids = _chk(ids, torch.int64, "ids").reshape(-1)
    _lib.check(lib.nrl_adam_rows_mark(ids.data_ptr(), ids.numel(), mark.numel(), mark.data_ptr(), int(step), _stream()),
               "nrl_adam_rows_mark")


def adam_rows_advance_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                       last_step: torch.Tensor, mark: Optional[torch.Tensor], status: torch.Tensor, upto_step: int,
                       with_grad, lr: float, betas: Tuple[float, float], eps: float, grad_scale: float = 1.0,
                       stride: int = 1, offset: int = 0, exclude_mark: Optional[torch.Tensor] = None, exclude_tag: int = 0) -> None:
    """``nrl_adam_rows_advance`` over a (rows, dim) table and its flat gradient / moment views (see include/newsreclib_amd.h)."""
    lib = _lib.load()
    rows, dim = param.shape
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape) == (rows, dim)):
            raise ValueError("newsreclib_amd.adam_rows_advance_: contiguous float32 GPU tensors of one (rows, dim) shape required")
    _lib.check(lib.nrl_adam_rows_advance(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), rows, dim,
                                         last_step.data_ptr(), mark.data_ptr() if mark is not None else None,
                                         status.data_ptr(), int(stride), int(offset), int(upto_step), int(with_grad),
                                         lr, betas[0], betas[1], eps, float(grad_scale),
                                         exclude_mark.data_ptr() if exclude_mark is not None else None, int(exclude_tag),
                                         _stream()), "nrl_adam_rows_advance")


def sort_positions(ids: torch.Tensor, vocab: Optional[int] = None) -> torch.Tensor:
    """Positions of the flat id vector grouped by ascending id: the visiting order of the embedding-table gradient
    (what ``embedding_dense_backward`` gets from its own sort).  With ``vocab`` (exclusive upper bound of the ids) a
    three-launch counting sort in the HIP library; the order INSIDE one id's run is unspecified.  Without it,
    ``torch.argsort`` (index bookkeeping on int64, no arithmetic).  The result has n + 1 entries: the last one is the number
    of positions holding id 0 (they sort first), which lets the news-encoder backward skip the rows of the padding id."""
    flat = _chk(ids, torch.int64, "ids").reshape(-1)
    n = flat.numel()
    if not vocab or vocab > (1 << 20):
        # (n + 1) entries like the library's sort: the positions by ascending id, then the number of id-0 positions
        return torch.cat([torch.argsort(flat, stable=True), (flat == 0).sum().reshape(1)])
    lib = _lib.load()
    order = torch.empty(n + 1, dtype=torch.int64, device=flat.device)
    ws = torch.empty(max(lib.nrl_sort_positions_workspace_bytes(n, int(vocab)), 256), dtype=torch.uint8,
                     device=flat.device)
    _lib.check(lib.nrl_sort_positions(flat.data_ptr(), n, int(vocab), order.data_ptr(), ws.data_ptr(), ws.numel(),
                                      _stream()), "nrl_sort_positions")
    return order


_SIDE_STREAMS = {}


def sort_positions_async(ids: torch.Tensor, vocab: Optional[int] = None) -> torch.Tensor:
    """``sort_positions`` on a side stream: the visiting order is needed by the news encoder's BACKWARD only, so the three
    small, latency-bound launches of the counting sort (~45 us at B = 128) run beside the forward instead of in front of
    it.  The completion event travels WITH the tensor (``order._nrl_ready``): every autograd forward that receives the
    tensor reads it (``order_event``, which does not consume it) and its backward waits for it on the launch stream
    (``wait_order``) -- a second encoder call on the same order, or a re-forwarded prepared batch, waits just the same.
    Ids that are not on a GPU are sorted in place (no stream to fork from)."""
    dev = ids.device
    if dev.type != "cuda":
        return sort_positions(ids, vocab)
    main = torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get(dev.index)
    if side is None:
        side = _SIDE_STREAMS[dev.index] = torch.cuda.Stream(dev)
    side.wait_stream(main)                       # the ids were produced on the launch stream
    with torch.cuda.stream(side):
        order = sort_positions(ids, vocab)
        ready = torch.cuda.Event()
        ready.record(side)
    ids.record_stream(side)                      # allocator: neither tensor may be recycled under the other stream
    order.record_stream(main)
    order._nrl_ready = ready                     # lives and dies with the tensor: no stale event can meet a recycled address
    return order


def order_event(order: Optional[torch.Tensor]):
    """The side-stream completion event of an order tensor (None for an order computed on the launch stream)."""
    return getattr(order, "_nrl_ready", None) if order is not None else None


def wait_order(event) -> None:
    if event is not None:
        torch.cuda.current_stream().wait_event(event)


def offsets_from_sorted_batch(batch: torch.Tensor, batch_size: int) -> torch.Tensor:
    """Prefix sums of the sorted assignment vector (``rec_dataset.py:289-293``) -> (B + 1,) int64."""
    counts = torch.bincount(batch, minlength=batch_size)
    off = torch.zeros(batch_size + 1, dtype=torch.int64, device=batch.device)
    torch.cumsum(counts, 0, out=off[1:])
    return off
