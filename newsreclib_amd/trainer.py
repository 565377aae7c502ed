"""Minimal train loop around ``NRMSModule`` for benchmarks and tests: flat parameter / gradient /
Adam-state buffers, fused dense Adam, and data-parallel gradient all-reduce over RCCL.

This replaces, for this path only, what Lightning's ``Trainer`` + ``ddp`` strategy do around the
reference module: ``loss.backward()`` -> bucketed gradient all-reduce ->
``Adam.step()``.  One process per GPU; ``torch.distributed`` backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests of the communication logic.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch
import torch.distributed as dist

from . import ops


class FlatParams:
    """Re-homes a module's parameters into ONE contiguous fp32 buffer (and a matching gradient
    buffer), keeping every ``nn.Parameter`` object alive as a view.  ``p.main_grad`` views let the
    backward kernels accumulate straight into the flat gradient (no per-call zero-fill + add), and
    the whole-model all-reduce / Adam step become single launches over 288 GB-class HBM."""

    def __init__(self, params: Iterable[torch.nn.Parameter], align: int = 64):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("FlatParams needs fp32 parameters on one device")
            offs.append(n)
            n += (p.numel() + align - 1) // align * align   # keep every view 256-byte aligned
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            view = self.flat[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.main_grad = self.grad[o:o + p.numel()].view_as(p)
            p.grad = None
        self.offsets = offs

    def grads_by_param(self) -> List[torch.Tensor]:
        return [p.main_grad for p in self.params]


class FusedAdam:
    """``torch.optim.Adam`` arithmetic (lr, betas, eps; no weight decay) as one HIP kernel over the
    flat buffer (reference: configs/model/nrms.yaml:49-52, abstract_recommender.py:96).  Dense on
    purpose: rows of the embedding table with a zero gradient still move through their decaying
    moments, exactly as in the reference."""

    def __init__(self, flat: FlatParams, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0

    def step(self, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        self.step_count += 1
        ops.adam_step_(self.flat.flat, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.step_count,
                       self.lr, self.betas, self.eps, grad_scale, zero_grad)


def allreduce_gradients(grad: torch.Tensor, group=None) -> float:
    """Sum-all-reduce the flat gradient across ranks (one RCCL call; the (V, 300) table gradient is
    >96 % of the bytes, so bucketing buys nothing).  Returns the scale (1/world) the optimizer must
    apply -- folded into the Adam kernel instead of a separate pass over the buffer."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1.0
    world = dist.get_world_size(group)
    if world == 1:
        return 1.0
    dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


class OverlappedGradReduce:
    """Two-piece gradient all-reduce over the flat buffer.

    The first parameter of the module (the (V, D) embedding table, >96 % of the gradient bytes) is
    complete as soon as the news encoder's activation-gradient chain has run; `start_head` is called
    at that point (from inside the backward, see ``ops.NewsEncoderFn``) and launches an ASYNC all-reduce
    of that leading segment, which RCCL carries over xGMI while the weight-gradient GEMMs still run.
    `finish` reduces the small tail and waits for the head.  With no process group (or world 1) both
    are no-ops and the scale is 1."""

    def __init__(self, flat: "FlatParams", head_numel: int, group=None):
        self.flat, self.head, self.group = flat, int(head_numel), group
        self._work = None

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def start_head(self, _grad=None) -> None:
        if self._active() and self.head > 0:
            self._work = dist.all_reduce(self.flat.grad[: self.head], op=dist.ReduceOp.SUM, group=self.group,
                                         async_op=True)

    def finish(self) -> float:
        if not self._active():
            return 1.0
        if self._work is None:                      # the hook did not fire: reduce everything now
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
        else:
            if self.head < self.flat.numel:
                dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group)
            self._work.wait()
            self._work = None
        return 1.0 / dist.get_world_size(self.group)


class NRMSTrainer:
    """forward -> CE loss -> backward (table-gradient all-reduce overlapped) -> fused Adam."""

    def __init__(self, module, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, group=None):
        self.module = module
        self.flat = FlatParams(module.parameters())
        self.opt = FusedAdam(self.flat, lr, betas, eps)
        self.group = group
        # leading segment = first parameter = the embedding table when the news encoder is MHSAAddAtt
        head = 0
        te = None
        enc = getattr(module, "news_encoder", None)
        if enc is not None and hasattr(enc, "text_encoders"):
            te = next(iter(enc.text_encoders.values()))
        # (only the fused MHSA encoder exposes the two-phase backward; a text encoder shared by several
        # attributes -- LSTUR -- finishes its table gradient only after the LAST of its backward calls)
        if te is not None and hasattr(te, "table_grad_hook") and len(enc.text_encoders) == 1 \
                and self.flat.params[0] is te.embedding_layer.weight:
            head = self.flat.offsets[1] if len(self.flat.offsets) > 1 else self.flat.numel
        self.reduce = OverlappedGradReduce(self.flat, head, group)
        if head > 0:
            te.table_grad_hook = self.reduce.start_head

    def step(self, batch: Dict) -> torch.Tensor:
        self.module.train()
        # model_step directly: training_step would append preds / targets to training_step_outputs every step, and
        # nothing here runs the epoch-end hook that clears them (use `epoch_end()` for the epoch metrics instead)
        loss = self.module.model_step(batch)[0]
        loss.backward()
        # parameters whose gradient came through ordinary autograd (``.grad``: a transformer body, a small
        # head fed through torch ops) rather than through a kernel writing ``main_grad``: fold them in
        for p in self.flat.params:
            if p.grad is not None:
                p.main_grad.add_(p.grad)
                p.grad = None
        scale = self.reduce.finish()
        self.opt.step(grad_scale=scale, zero_grad=True)
        return loss.detach()
