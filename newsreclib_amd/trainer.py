"""Minimal train loop around ``NRMSModule`` for benchmarks and tests: flat parameter / gradient /
Adam-state buffers, fused dense Adam, and data-parallel gradient all-reduce over RCCL.

This replaces, for this path only, what Lightning's ``Trainer`` + ``ddp`` strategy do around the
reference module: ``loss.backward()`` -> bucketed gradient all-reduce ->
``Adam.step()``.  One process per GPU; ``torch.distributed`` backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests of the communication logic.
"""
from __future__ import annotations

import os
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
        self.begin_step()
        self.step_range(0, self.flat.numel, grad_scale, zero_grad)

    def begin_step(self) -> None:
        self.step_count += 1

    def step_range(self, lo: int, hi: int, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        """The Adam update of elements [lo, hi) of the flat buffer (after ``begin_step``): lets the trainer update a slice
        of the table as soon as ITS all-reduce has landed, while the next slice is still on the wire."""
        if hi > lo:
            ops.adam_step_(self.flat.flat[lo:hi], self.flat.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                           self.step_count, self.lr, self.betas, self.eps, grad_scale, zero_grad)


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

    def __init__(self, flat: "FlatParams", head_numel: int, group=None, chunks: int = 1):
        self.flat, self.head, self.group = flat, int(head_numel), group
        self._work = None
        # the head goes out as `chunks` all-reduces over equal, 256-byte aligned slices: `finish_pipelined` hands each
        # slice to the optimizer as soon as it has landed, so the Adam pass over slice i runs under the transfer of
        # slice i + 1.  (The table gradient itself cannot leave earlier: its last contribution is the LAST kernel of the
        # activation-gradient chain, whichever id range one looks at.)
        self.chunks = max(1, int(chunks))
        self.bounds = [(self.head * i // self.chunks) // 64 * 64 for i in range(self.chunks)] + [self.head]

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def start_head(self, _grad=None, _ids=None) -> None:
        if self._active() and self.head > 0:
            self._work = [dist.all_reduce(self.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                          for lo, hi in zip(self.bounds[:-1], self.bounds[1:]) if hi > lo]

    def finish(self) -> float:
        if not self._active():
            return 1.0
        if self._work is None:                      # the hook did not fire: reduce everything now
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
        else:
            if self.head < self.flat.numel:
                dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group)
            for w in self._work:
                w.wait()
            self._work = None
        return 1.0 / dist.get_world_size(self.group)

    def finish_pipelined(self):
        """Generator form of ``finish``: yields (lo, hi, scale) slices of the flat gradient in the order they become final --
        first the small tail (reduced here), then the head slices as their async all-reduces land."""
        if not self._active():
            yield 0, self.flat.numel, 1.0
            return
        scale = 1.0 / dist.get_world_size(self.group)
        if self._work is None:
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
            yield 0, self.flat.numel, scale
            return
        if self.head < self.flat.numel:
            dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group)
            yield self.head, self.flat.numel, scale
        live = [(lo, hi) for lo, hi in zip(self.bounds[:-1], self.bounds[1:]) if hi > lo]
        for (lo, hi), w in zip(live, self._work):
            w.wait()
            yield lo, hi, scale
        self._work = None

    def info(self) -> Dict:
        return {"mode": "dense", "payload_bytes_per_rank": 4 * self.flat.numel, "head_chunks": self.chunks,
                "note": "sum all-reduce of the whole flat fp32 gradient (table head async under the weight gradients, in "
                        "`head_chunks` slices whose Adam updates pipeline with the transfers)"}


class TouchedRowsExchange:
    """Gradient exchange that ships only the embedding-table rows a rank TOUCHED this step (SURVEY.md section 8e,
    "optional later"): the (V, D) table gradient is > 96 % of the flat gradient and a rank's batch touches at most
    B * (H + C) * L of its V rows.

    After phase 1 of the news-encoder backward (``start_head``, called from inside the backward with the token ids):
    each rank gathers its unique sorted ids and their gradient rows, all ranks all-gather (counts, ids, rows) -- async,
    under the weight-gradient GEMMs.  ``finish`` all-reduces the small dense rest, zeroes the rank's own touched rows and
    adds EVERY rank's rows (its own included, from the gathered buffer) in ascending rank order: each replica performs the
    same additions in the same order, so the replicas stay bit-identical, and with two ranks the result is bit-identical to
    the dense all-reduce (a + b == b + a).  Dense Adam is unchanged (rows nobody touched carry a zero gradient).
    Costs one host sync per step (the unique counts size the gather buffers); the dense path has none."""

    def __init__(self, flat: "FlatParams", head_numel: int, table: torch.Tensor, group=None):
        self.flat, self.head, self.group = flat, int(head_numel), group
        self.rows, self.dim = int(table.shape[0]), int(table.shape[1])
        self._pending = None
        self._payload = 0

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _table_grad(self) -> torch.Tensor:
        return self.flat.grad[: self.rows * self.dim].view(self.rows, self.dim)

    def start_head(self, _grad=None, ids=None) -> None:
        if not self._active() or self.head <= 0 or ids is None:
            return
        world = dist.get_world_size(self.group)
        g = self._table_grad()
        uniq = torch.unique(ids.reshape(-1))                                  # sorted unique ids of this rank
        count = torch.tensor([uniq.numel()], dtype=torch.int64, device=g.device)
        counts = [torch.zeros_like(count) for _ in range(world)]
        dist.all_gather(counts, count, group=self.group)
        counts = [int(c) for c in torch.cat(counts).tolist()]                 # (host sync: sizes of the gather buffers)
        cap = max(max(counts), 1)
        ids_pad = torch.zeros(cap, dtype=torch.int64, device=g.device)
        ids_pad[: uniq.numel()] = uniq
        rows_pad = torch.zeros(cap, self.dim, dtype=g.dtype, device=g.device)
        rows_pad[: uniq.numel()] = g.index_select(0, uniq)
        ids_all = [torch.empty_like(ids_pad) for _ in range(world)]
        rows_all = [torch.empty_like(rows_pad) for _ in range(world)]
        w1 = dist.all_gather(ids_all, ids_pad, group=self.group, async_op=True)
        w2 = dist.all_gather(rows_all, rows_pad, group=self.group, async_op=True)
        self._pending = (uniq, counts, ids_all, rows_all, w1, w2)
        self._payload = cap * (8 + 4 * self.dim) + 8 + 4 * (self.flat.numel - self.head)

    def finish(self) -> float:
        if not self._active():
            return 1.0
        world = dist.get_world_size(self.group)
        if self._pending is None:                   # the hook did not fire (or carried no ids): dense fallback
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
            self._payload = 4 * self.flat.numel
            return 1.0 / world
        if self.head < self.flat.numel:
            dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group)
        uniq, counts, ids_all, rows_all, w1, w2 = self._pending
        self._pending = None
        w1.wait()
        w2.wait()
        g = self._table_grad()
        g.index_fill_(0, uniq, 0.0)                 # own rows out, then every rank's rows in, in rank order
        for r in range(world):
            n = counts[r]
            if n > 0:
                g.index_add_(0, ids_all[r][:n], rows_all[r][:n])    # unique indices: no contention, deterministic
        return 1.0 / world

    def info(self) -> Dict:
        return {"mode": "rows", "payload_bytes_per_rank": int(self._payload), "dense_payload_bytes_per_rank": 4 * self.flat.numel,
                "note": "all-gather of (unique ids, their table-gradient rows) padded to the largest rank + dense all-reduce of "
                        "the non-table gradient; last step's sizes"}


class NRMSTrainer:
    """forward -> CE loss -> backward (table-gradient all-reduce overlapped) -> fused Adam."""

    def __init__(self, module, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, group=None,
                 grad_exchange: str = "dense", head_chunks: int = 4):
        if grad_exchange not in ("dense", "rows"):
            raise ValueError("grad_exchange must be 'dense' (all-reduce of the flat gradient) or 'rows' (touched table rows)")
        self.module = module
        self.flat = FlatParams(module.parameters())
        self.opt = FusedAdam(self.flat, lr, betas, eps)
        self.group = group
        self._losses: List[torch.Tensor] = []
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
        if grad_exchange == "rows" and head > 0:
            self.reduce = TouchedRowsExchange(self.flat, head, te.embedding_layer.weight, group)
        else:
            self.reduce = OverlappedGradReduce(self.flat, head, group, chunks=head_chunks)
        if head > 0:
            te.table_grad_hook = self.reduce.start_head
        self._side = None
        dev = self.flat.params[0].device
        # d(loss)/d(loss): one cached tensor instead of autograd's per-step fill; losses recognise it (ops.register_unit_grad)
        self._unit_root = os.environ.get("NRL_UNIT_ROOT_GRAD", "1") not in ("", "0")     # (0: plain loss.backward(), A/B runs)
        self._one = ops.register_unit_grad(torch.ones((), dtype=torch.float32, device=dev))
        if dev.type == "cuda" and os.environ.get("NRL_DEFER_USER_WGRAD", "1") not in ("", "0"):
            self._side = torch.cuda.Stream(device=dev)

    def epoch_end(self) -> Dict[str, float]:
        """Mean train loss over the steps since the last call (all ranks' steps under data parallelism), then reset --
        what ``on_train_epoch_end`` logs as train/loss when Lightning drives the module (nrms_module.py:380-396)."""
        if not self._losses:
            return {}
        tot = torch.stack(self._losses).sum().reshape(1).double()
        cnt = torch.tensor([float(len(self._losses))], dtype=torch.float64, device=tot.device)
        self._losses = []
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            both = torch.cat([tot, cnt])
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=self.group)
            tot, cnt = both[:1], both[1:]
        return {"train/loss": float(tot / cnt)}

    def exchange_info(self) -> Dict:
        """What the data-parallel gradient exchange ships per rank and step (bench.py prints it for N > 1)."""
        return self.reduce.info()

    def step(self, batch: Dict) -> torch.Tensor:
        self.module.train()
        # model_step directly: training_step would append preds / targets to training_step_outputs every step, and
        # nothing here runs the epoch-end hook that clears them; the loss is still tracked for `epoch_end()`
        loss = self.module.model_step(batch)[0]
        self._losses.append(loss.detach())
        root = self._one if self._unit_root and loss.dim() == 0 and loss.dtype == torch.float32 and loss.device == self._one.device else None
        if self._side is not None:
            # the user encoder's weight gradients on a side stream beside the news-encoder backward; joined on exit
            with ops.deferred_weight_grads(self._side):
                loss.backward(gradient=root)
        else:
            loss.backward(gradient=root)
        # parameters whose gradient came through ordinary autograd (``.grad``: a transformer body, a small
        # head fed through torch ops) rather than through a kernel writing ``main_grad``: fold them in
        for p in self.flat.params:
            if p.grad is not None:
                p.main_grad.add_(p.grad)
                p.grad = None
        if hasattr(self.reduce, "finish_pipelined"):
            # dense exchange: Adam over each slice of the flat buffer as soon as its all-reduce has landed
            self.opt.begin_step()
            for lo, hi, scale in self.reduce.finish_pipelined():
                self.opt.step_range(lo, hi, grad_scale=scale, zero_grad=True)
        else:
            scale = self.reduce.finish()
            self.opt.step(grad_scale=scale, zero_grad=True)
        return loss.detach()
