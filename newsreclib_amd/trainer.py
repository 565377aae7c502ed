"""Minimal train loop around ``NRMSModule`` for benchmarks and tests: flat parameter / gradient /
Adam-state buffers, fused dense Adam, and data-parallel gradient all-reduce over RCCL.

This replaces, for this path only, what Lightning's ``Trainer`` + ``ddp`` strategy do around the
reference module: ``loss.backward()`` -> bucketed gradient all-reduce ->
``Adam.step()``.  One process per GPU; ``torch.distributed`` backend "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests of the communication logic.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist

from . import _lib, ops


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
        from . import ops_blocks
        # this optimizer announces every parameter write (begin_step): images of the weights it OWNS may live for a step
        ops_blocks.register_step_driver(self, flat.params)

    def step(self, grad_scale: float = 1.0, zero_grad: bool = True) -> None:
        self.begin_step()
        self.step_range(0, self.flat.numel, grad_scale, zero_grad)

    def begin_step(self) -> None:
        self.step_count += 1
        # this optimizer writes parameters through raw pointers, which no tensor version counter sees: the within-step image
        # caches of trainable weights (ops_blocks.FrozenImages(allow_trainable=True)) are keyed on this generation
        from . import ops_blocks
        ops_blocks.next_optimizer_step()

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


class _ExposedWait:
    """Measures what the launch stream spends BLOCKED on the gradient exchange (the part of the wire time the backward did not
    hide): an event pair on the launch stream around every wait for a collective.  Off by default (``measure = False``: nothing is
    recorded); bench.py switches it on for its instrumented pass.  ``exposed_wait_ms()`` -> mean per step, steps measured."""
    measure = False

    def _timed(self, fn):
        if not self.measure or not torch.cuda.is_available():
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self.__dict__.setdefault("_wait_events", []).append((a, b))
        return out

    def _step_measured(self) -> None:
        if self.measure:
            ev = self.__dict__.setdefault("_wait_events", [])
            self.__dict__.setdefault("_wait_steps", []).append(ev[:])
            del ev[:]

    def exposed_wait_ms(self):
        steps = self.__dict__.get("_wait_steps", [])
        if not steps:
            return 0.0, 0
        torch.cuda.synchronize()
        per_step = [sum(a.elapsed_time(b) for a, b in ev) for ev in steps]
        return float(sum(per_step) / len(per_step)), len(per_step)

    def reset_exposed_wait(self) -> None:
        self.__dict__["_wait_steps"], self.__dict__["_wait_events"] = [], []


class OverlappedGradReduce(_ExposedWait):
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
            self._timed(lambda: dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group))
        else:
            if self.head < self.flat.numel:
                self._timed(lambda: dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group))
            for w in self._work:
                self._timed(w.wait)
            self._work = None
        self._step_measured()
        return 1.0 / dist.get_world_size(self.group)

    def finish_pipelined(self):
        """Generator form of ``finish``: yields (lo, hi, scale) slices of the flat gradient in the order they become final --
        first the small tail (reduced here), then the head slices as their async all-reduces land."""
        if not self._active():
            yield 0, self.flat.numel, 1.0
            return
        scale = 1.0 / dist.get_world_size(self.group)
        if self._work is None:
            self._timed(lambda: dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group))
            self._step_measured()
            yield 0, self.flat.numel, scale
            return
        if self.head < self.flat.numel:
            self._timed(lambda: dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group))
            yield self.head, self.flat.numel, scale
        live = [(lo, hi) for lo, hi in zip(self.bounds[:-1], self.bounds[1:]) if hi > lo]
        for (lo, hi), w in zip(live, self._work):
            self._timed(w.wait)
            yield lo, hi, scale
        self._work = None
        self._step_measured()

    def info(self) -> Dict:
        world = dist.get_world_size(self.group) if self._active() else 1
        return {"mode": "dense", "payload_bytes_per_rank": 4 * self.flat.numel, "head_chunks": self.chunks,
                "predicted_wire_ms": {"dense": predicted_wire_ms("dense", 4 * self.flat.numel, world), "link_GBps": 153.0,
                                      "unmeasured": True},
                "note": "sum all-reduce of the whole flat fp32 gradient (table head async under the weight gradients, in "
                        "`head_chunks` slices whose Adam updates pipeline with the transfers)"}


XGMI_LINK_GBPS = 153.0      # MI355X: 7 point-to-point xGMI links per GPU, ~153 GB/s each (MI355X_MICROARCH.md)


def predicted_wire_ms(mode: str, payload_bytes: float, world: int, gather_bytes: float = 0.0) -> Dict[str, float]:
    """Back-of-envelope wire time of one gradient exchange over xGMI (no RCCL protocol overhead):
    dense = sum all-reduce of `payload_bytes`; rows = all-gather where every rank contributes `payload_bytes`; owners =
    all-to-all in which every rank sends `payload_bytes` in total, followed by an all-gather where every rank contributes
    `gather_bytes`.  The all-to-all's pieces (payload / world to each peer) leave over all links at once under `direct`
    (payload / world / bw); on a ring every piece crosses every hop towards its peer, i.e. the link next to a rank carries
    ~ (world - 1) / 2 pieces of each of its ~ 2 neighbour directions: priced at (world - 1) / world * payload / bw (round-5 advisor:
    payload / world under BOTH models biased `auto` towards this exchange).
    `ring`: every byte crosses one link per hop; `direct`: reduce-scatter + all-gather (or the all-gather) spread over all
    world - 1 links of the fully connected node."""
    if world <= 1:
        return {"ring": 0.0, "direct": 0.0}
    bw = XGMI_LINK_GBPS * 1e9
    if mode == "dense":
        ring = 2.0 * (world - 1) / world * payload_bytes / bw
        direct = 2.0 * (payload_bytes / world) / bw
    elif mode == "owners":
        ring = (world - 1) / world * payload_bytes / bw + (world - 1) * gather_bytes / bw
        direct = payload_bytes / world / bw + gather_bytes / bw
    else:
        ring = (world - 1) * payload_bytes / bw
        direct = payload_bytes / bw
    return {"ring": round(ring * 1e3, 6), "direct": round(direct * 1e3, 6)}


def wire_model() -> str:
    """Which column of ``predicted_wire_ms`` the `auto` exchange decides by: "ring" (RCCL's ring algorithms, the conservative
    default) or "direct" (NRL_WIRE_MODEL=direct)."""
    return "direct" if os.environ.get("NRL_WIRE_MODEL", "ring") == "direct" else "ring"


class TouchedRowsExchange(_ExposedWait):
    """Gradient exchange that ships only the embedding-table rows a rank TOUCHED this step (SURVEY.md section 8e,
    "optional later"): the (V, D) table gradient is > 96 % of the flat gradient and a rank's batch touches at most
    B * (H + C) * L of its V rows.

    ``prepare(ids, order)`` -- called by the trainer BEFORE the forward, as soon as the step's token ids exist -- builds the
    rank's sorted unique ids into a fixed-size buffer with device-side arithmetic only (flags of the id-sorted positions,
    prefix sum, scatter: no ``torch.unique``, no read-back), all-gathers the unique COUNTS of all ranks asynchronously and
    copies them to pinned host memory behind an event.  After phase 1 of the news-encoder backward (``start_head``, called from
    inside the backward): the host waits for THAT event only -- recorded a whole forward + backward chain earlier, so the wait
    is over before it starts and the launch queue never drains (the round-3 form called ``torch.unique`` and ``.tolist()`` here:
    two device-wide syncs in the middle of the backward) -- sizes the buffers from the largest count, gathers the rank's
    gradient rows and all-gathers (ids, rows) asynchronously under the weight-gradient GEMMs.  ``finish`` all-reduces the small
    dense rest, zeroes the rank's own touched rows and adds EVERY rank's rows (its own included, from the gathered buffer) in
    ascending rank order: each replica performs the same additions in the same order, so the replicas stay bit-identical,
    and with two ranks the result is bit-identical to the dense all-reduce (a + b == b + a).  Dense Adam is unchanged (rows
    nobody touched carry a zero gradient).

    ``auto=True``: per step, from the gathered counts (identical on every rank, so every rank takes the same branch):
    whichever of {rows, dense} ``predicted_wire_ms`` prices lower under ``wire_model()`` -- an all-gather moves
    (w - 1) * payload per rank on a ring, a ring all-reduce 2 (w - 1) / w * S, so the break-even is payload * world < 2 S
    (round 4 shipped ``< 0.5 S``, which sent the configs[2] rank shape to the dense all-reduce its own predictor priced 4x
    slower).  ``--grad-exchange auto`` itself is ``OwnerRowsExchange(mode="auto")``, which also has the owner-partitioned form."""

    def __init__(self, flat: "FlatParams", head_numel: int, table: torch.Tensor, group=None, auto: bool = False):
        self.flat, self.head, self.group, self.auto = flat, int(head_numel), group, bool(auto)
        self.rows, self.dim = int(table.shape[0]), int(table.shape[1])
        self._prepared = None
        self._pending = None
        self._dense_work = None
        self._payload = 0
        self.last_gathered = None       # the unique ids of every rank at the last `finish` (None: that step went dense)
        self._last = {"choice": "rows", "max_unique_rows": 0}

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _table_grad(self) -> torch.Tensor:
        return self.flat.grad[: self.rows * self.dim].view(self.rows, self.dim)

    def prepare(self, ids: torch.Tensor, order: Optional[torch.Tensor] = None) -> None:
        """Unique ids of this rank (fixed-size buffer + device count) and the async exchange of the counts.  ``order``: the
        id-sorted visiting order of ``ids`` if the caller has it (``ops.sort_positions``: n + 1 entries), else sorted here."""
        if not self._active() or self.head <= 0 or ids is None:
            self._prepared = None
            return
        world = dist.get_world_size(self.group)
        flat_ids = ids.reshape(-1)
        n = flat_ids.numel()
        dev = flat_ids.device
        if order is not None:
            ev = getattr(order, "_nrl_ready", None)
            if ev is not None:
                torch.cuda.current_stream(dev).wait_event(ev)
            sorted_ids = flat_ids.index_select(0, order[:n])
        else:
            sorted_ids = torch.sort(flat_ids).values
        first = torch.ones(n, dtype=torch.bool, device=dev)
        if n > 1:
            first[1:] = sorted_ids[1:] != sorted_ids[:-1]
        slot = torch.cumsum(first, 0) - 1                       # rank of each position's id among the unique ids
        uniq = torch.zeros(n, dtype=torch.int64, device=dev)
        uniq.scatter_(0, slot, sorted_ids)                      # (equal ids write equal values: deterministic)
        count = (slot[-1:] + 1) if n > 0 else torch.zeros(1, dtype=torch.int64, device=dev)
        parts = [torch.zeros_like(count) for _ in range(world)]
        work = dist.all_gather(parts, count.contiguous(), group=self.group, async_op=True)
        work.wait()                                               # (RCCL: a stream-side wait, the host does not block)
        counts_dev = torch.cat(parts)
        if dev.type == "cuda":
            counts_host = torch.empty(world, dtype=torch.int64, pin_memory=True)
            counts_host.copy_(counts_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        else:
            counts_host, ev = counts_dev, None
        self._prepared = (uniq, counts_host, ev, counts_dev)

    def start_head(self, _grad=None, ids=None) -> None:
        if not self._active() or self.head <= 0:
            return
        if self._prepared is None:
            if ids is None:
                return
            self.prepare(ids)                                     # (callers without the early hook: same result, later)
        uniq, counts_host, ev, _keep = self._prepared
        self._prepared = None
        if ev is not None:
            ev.synchronize()                                      # an event of the step's first microseconds: no drain
        world = dist.get_world_size(self.group)
        counts = [int(c) for c in counts_host.tolist()]
        cap = max(max(counts), 1)
        g = self._table_grad()
        rows_bytes = cap * (8 + 4 * self.dim)
        self._last = {"choice": "rows", "max_unique_rows": cap}
        m = wire_model()
        if self.auto and predicted_wire_ms("rows", rows_bytes, world)[m] >= predicted_wire_ms("dense", 4 * self.head, world)[m]:
            self._last["choice"] = "dense"
            self._dense_work = dist.all_reduce(self.flat.grad[: self.head], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._payload = 4 * self.flat.numel
            return
        # padded to the LARGEST rank's count: `uniq` has one slot per token of THIS rank's batch, which a ragged / partial
        # batch can leave shorter than another rank's unique count (entries past the rank's own count: id 0, never added)
        ids_pad = torch.zeros(cap, dtype=uniq.dtype, device=uniq.device)
        k = min(int(uniq.numel()), cap)
        ids_pad[:k] = uniq[:k]
        rows_pad = g.index_select(0, ids_pad)
        ids_all = [torch.empty_like(ids_pad) for _ in range(world)]
        rows_all = [torch.empty_like(rows_pad) for _ in range(world)]
        w1 = dist.all_gather(ids_all, ids_pad, group=self.group, async_op=True)
        w2 = dist.all_gather(rows_all, rows_pad, group=self.group, async_op=True)
        own = counts[dist.get_rank(self.group)]
        self._pending = (ids_pad[:own], counts, ids_all, rows_all, w1, w2)
        self._payload = rows_bytes + 8 + 4 * (self.flat.numel - self.head)

    def finish(self) -> float:
        self.last_gathered = None
        if not self._active():
            return 1.0
        world = dist.get_world_size(self.group)
        if self._dense_work is not None:            # auto chose the dense all-reduce for this step
            if self.head < self.flat.numel:
                dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group)
            self._dense_work.wait()
            self._dense_work = None
            return 1.0 / world
        if self._pending is None:                   # the hook did not fire (or carried no ids): dense fallback
            self._prepared = None
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
            self._payload = 4 * self.flat.numel
            self._last = {"choice": "dense (hook did not fire)", "max_unique_rows": 0}
            return 1.0 / world
        if self.head < self.flat.numel:
            dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group)
        uniq, counts, ids_all, rows_all, w1, w2 = self._pending
        self._pending = None
        w1.wait()
        w2.wait()
        g = self._table_grad()
        self.last_gathered = [ids_all[r][: counts[r]] for r in range(world) if counts[r] > 0]
        g.index_fill_(0, uniq, 0.0)                 # own rows out, then every rank's rows in, in rank order
        for r in range(world):
            n = counts[r]
            if n > 0:
                g.index_add_(0, ids_all[r][:n], rows_all[r][:n])    # unique indices: no contention, deterministic
        return 1.0 / world

    def info(self) -> Dict:
        world = dist.get_world_size(self.group) if self._active() else 1
        dense = 4 * self.flat.numel
        rows = self._last["max_unique_rows"] * (8 + 4 * self.dim)
        return {"mode": "auto" if self.auto else "rows", "last_step_choice": self._last["choice"],
                "max_unique_rows_per_rank": self._last["max_unique_rows"],
                "payload_bytes_per_rank": int(self._payload), "dense_payload_bytes_per_rank": dense,
                "rule": f"whichever of rows / dense predicted_wire_ms prices lower ({wire_model()} model)" if self.auto else None,
                "predicted_wire_ms": {"dense": predicted_wire_ms("dense", dense, world),
                                      "rows": predicted_wire_ms("rows", rows, world),
                                      "link_GBps": XGMI_LINK_GBPS, "unmeasured": True},
                "note": "all-gather of (unique ids, their table-gradient rows) padded to the largest rank + dense all-reduce of "
                        "the non-table gradient; last step's sizes; no host sync beyond one early event"}


def _sorted_unique_ids(ids: torch.Tensor, order: Optional[torch.Tensor] = None):
    """-> (uniq, count): the ascending unique ids of `ids` at the front of an n-slot buffer (zeros behind) and their number
    as a (1,) device tensor -- device arithmetic only (flags of the id-sorted positions, prefix sum, scatter): no
    ``torch.unique``, no read-back.  ``order``: the id-sorted visiting order if the caller has it (``ops.sort_positions``)."""
    flat_ids = ids.reshape(-1)
    n = flat_ids.numel()
    dev = flat_ids.device
    if order is not None:
        ev = getattr(order, "_nrl_ready", None)
        if ev is not None:
            torch.cuda.current_stream(dev).wait_event(ev)
        sorted_ids = flat_ids.index_select(0, order[:n])
    else:
        sorted_ids = torch.sort(flat_ids).values
    first = torch.ones(n, dtype=torch.bool, device=dev)
    if n > 1:
        first[1:] = sorted_ids[1:] != sorted_ids[:-1]
    slot = torch.cumsum(first, 0) - 1
    uniq = torch.zeros(n, dtype=torch.int64, device=dev)
    uniq.scatter_(0, slot, sorted_ids)                      # (equal ids write equal values: deterministic)
    count = (slot[-1:] + 1) if n > 0 else torch.zeros(1, dtype=torch.int64, device=dev)
    return uniq, count


class OwnerRowsExchange(_ExposedWait):
    """Owner-partitioned exchange of the touched embedding-table rows (``--grad-exchange owners`` / ``auto``; DESIGN section 6;
    reference leg: ``configs/trainer/ddp.yaml:4``, whose DDP all-reduces the dense (V, D) gradient).

    Rank r OWNS the table rows with ``id % world == r``.  Per step:

    ``prepare`` (before the forward, on the exchange's own stream): the rank's sorted unique ids go out in ONE all-gather of a
    fixed-capacity message ``[count | ids ...]`` (``id_capacity`` slots: every rank must post the same size without asking
    anyone).  From it every rank derives -- identically, on the device -- the per-owner UNION of touched ids (the canonical
    order of the reduced rows), the all-to-all split matrix and, for itself as an owner, the slot of every row it will
    receive.  The three small integer tables the host needs (counts, union sizes, splits) are copied to pinned memory behind
    one event.  No ids travel after this point.

    ``start_head`` (from inside the backward, after the table gradient is complete): the host reads those tables (an event of
    the step's first microseconds), gathers the rank's gradient rows in (owner, id) order and posts ONE async all-to-all:
    each row goes to its owner, under the weight-gradient GEMMs.

    ``finish``: the owner sums what it received IN RANK ORDER into its union slots (one ``index_add_`` per source, unique
    slots each: a fixed order, so a rerun gives the same bits), all-gathers the reduced rows (padded to the largest owner's
    union, no ids), and every replica copies them into its table gradient -- every replica receives the SAME bytes, so the
    replicas stay bit-identical at any world size, and at two ranks the result is bit-identical to the dense all-reduce
    (0 + a + b).  Rows nobody touched keep their zero gradient.  ``last_gathered`` = the union (the lazy table optimizer's marks).

    Against the all-gather of (ids, rows) of ``TouchedRowsExchange`` the second leg carries each touched row ONCE (the
    Zipf head of a news vocabulary is touched by every rank), and the summation work is split eight ways.

    ``mode="auto"``: per step, from the gathered counts: whichever of {owners, rows, dense} ``predicted_wire_ms`` prices
    lowest under ``wire_model()``; the ``rows`` branch here needs no id traffic either.  A step in which some rank has more
    unique ids than ``id_capacity`` takes the dense all-reduce (every rank sees every count, so all take the same branch)."""

    def __init__(self, flat: "FlatParams", head_numel: int, table: torch.Tensor, group=None, mode: str = "owners",
                 id_capacity: int = 32768):
        if mode not in ("owners", "auto"):
            raise ValueError("OwnerRowsExchange: mode must be 'owners' or 'auto'")
        self.flat, self.head, self.group, self.mode = flat, int(head_numel), group, mode
        self.rows, self.dim = int(table.shape[0]), int(table.shape[1])
        self.capacity = int(id_capacity)
        self._prepared = None
        self._pending = None
        self._payload = 0
        self._stream = None
        self.last_gathered = None
        self._last = {"choice": mode, "max_unique_rows": 0, "union_rows": 0}

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _table_grad(self) -> torch.Tensor:
        return self.flat.grad[: self.rows * self.dim].view(self.rows, self.dim)

    def prepare(self, ids: torch.Tensor, order: Optional[torch.Tensor] = None) -> None:
        if not self._active() or self.head <= 0 or ids is None:
            self._prepared = None
            return
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        dev = ids.device
        cuda = dev.type == "cuda"
        main = torch.cuda.current_stream(dev) if cuda else None
        if cuda and self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        if cuda:
            self._stream.wait_stream(main)                     # the ids (and a side-stream sort's event) come from there
        with (torch.cuda.stream(self._stream) if cuda else _NullCtx()):
            V, cap0 = self.rows, self.capacity
            uniq, count = _sorted_unique_ids(ids, order)
            msg = torch.zeros(cap0 + 1, dtype=torch.int64, device=dev)
            msg[:1] = count
            k = min(int(uniq.numel()), cap0)
            msg[1:1 + k] = uniq[:k]
            allmsg = torch.empty(world, cap0 + 1, dtype=torch.int64, device=dev)
            _all_gather_rows(allmsg, msg, self.group)
            counts = allmsg[:, 0]
            ids_mat = allmsg[:, 1:]
            valid = torch.arange(cap0, device=dev)[None, :] < counts[:, None]
            owner = ids_mat % world
            sentinel = world * V
            key = torch.where(valid, owner * V + ids_mat, torch.full_like(ids_mat, sentinel))     # (owner, id) order
            ks = torch.sort(key.reshape(-1)).values
            first = torch.ones_like(ks, dtype=torch.bool)
            first[1:] = ks[1:] != ks[:-1]
            first &= ks < sentinel
            upos = torch.cumsum(first, 0) - 1
            union_keys = torch.full((ks.numel() + 1,), sentinel, dtype=torch.int64, device=dev)
            union_keys.scatter_(0, torch.where(first, upos, torch.full_like(upos, ks.numel())), ks)
            union_keys = union_keys[:-1]                       # ascending (owner, id) keys of every touched row, sentinels behind
            own_of = torch.where(first, ks // V, torch.full_like(ks, world))
            U = torch.bincount(own_of, minlength=world + 1)[:world]               # union size per owner
            ustart = torch.cumsum(U, 0) - U
            pair = torch.arange(world, device=dev)[:, None] * (world + 1) + torch.where(valid, owner, torch.full_like(owner, world))
            C = torch.bincount(pair.reshape(-1), minlength=world * (world + 1)).view(world, world + 1)[:, :world]
            # this rank as a SOURCE: its ids in (owner, id) order
            my_key = key[rank]
            send_ids = torch.sort(my_key).values % V           # (sentinels -> id 0 behind the real ones)
            # this rank as an OWNER: for every (source, entry) it receives, in (source, id) order, the slot in its union list
            slot_glob = torch.searchsorted(union_keys, key.reshape(-1))
            mine = valid & (owner == rank)
            pick = torch.where(mine, torch.arange(world * cap0, device=dev).view(world, cap0), torch.full_like(ids_mat, world * cap0))
            pick = torch.sort(pick.reshape(-1)).values         # the first sum(C[:, rank]) entries are the received rows' (source, entry)
            recv_slots = slot_glob[pick.clamp_max(world * cap0 - 1)] - ustart[rank]
            stats = torch.cat([counts, U, C.reshape(-1)])
            if cuda:
                host = torch.empty(stats.numel(), dtype=torch.int64, pin_memory=True)
                host.copy_(stats, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            else:
                host, ev = stats, None
        keep = (uniq, allmsg, union_keys, ustart, send_ids, recv_slots, stats)
        if cuda:
            for t in keep:
                t.record_stream(main)                          # allocator: read on the launch stream later in the step
        self._prepared = (host, ev, ids_mat, union_keys, ustart, send_ids, recv_slots, keep)

    def _decide(self, counts, U, C, world, rank) -> str:
        cap = max(max(counts), 1)
        row_b = 4 * self.dim
        rows_bytes = cap * row_b
        own = counts[rank]
        a2a = max(counts) * row_b                              # the largest sender bounds the all-to-all
        ag = max(max(U), 1) * row_b
        m = wire_model()
        price = {"dense": predicted_wire_ms("dense", 4 * self.head, world)[m],
                 "rows": predicted_wire_ms("rows", rows_bytes, world)[m],
                 "owners": predicted_wire_ms("owners", a2a, world, ag)[m]}
        self._last = {"choice": self.mode, "max_unique_rows": cap, "union_rows": int(sum(U)), "priced_ms": price,
                      "own_rows": int(own)}
        if max(counts) > self.capacity:
            return "dense (a rank's unique ids exceed id_capacity)"
        if self.mode == "owners":
            return "owners"
        return min(("owners", "rows", "dense"), key=lambda k: price[k])

    def start_head(self, _grad=None, ids=None) -> None:
        if not self._active() or self.head <= 0:
            return
        if self._prepared is None:
            if ids is None:
                return
            self.prepare(ids)
        host, ev, ids_mat, union_keys, ustart, send_ids, recv_slots, keep = self._prepared
        self._prepared = None
        if ev is not None:
            ev.synchronize()                                   # an event of the step's first microseconds: no drain
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        st = [int(x) for x in host.tolist()]
        counts, U = st[:world], st[world:2 * world]
        C = [st[2 * world + s * world: 2 * world + (s + 1) * world] for s in range(world)]
        choice = self._decide(counts, U, C, world, rank)
        self._last["choice"] = choice
        g = self._table_grad()
        D = self.dim
        if choice.startswith("dense"):
            work = dist.all_reduce(self.flat.grad[: self.head], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending = ("dense", work)
            self._payload = 4 * self.flat.numel
            return
        if choice == "rows":
            cap = max(max(counts), 1)
            ids_pad = ids_mat[rank, :cap]
            rows_pad = g.index_select(0, ids_pad)
            rows_all = torch.empty(world, cap, D, dtype=g.dtype, device=g.device)
            work = _all_gather_rows(rows_all, rows_pad, self.group, async_op=True)
            self._pending = ("rows", work, counts, ids_mat, rows_all, keep)
            self._payload = cap * 4 * D + 8 * (self.capacity + 1) + 4 * (self.flat.numel - self.head)
            return
        own = counts[rank]
        send_splits, recv_splits = C[rank], [C[s][rank] for s in range(world)]
        rows_send = g.index_select(0, send_ids[:own])
        rows_recv = torch.empty(sum(recv_splits), D, dtype=g.dtype, device=g.device)
        work = _all_to_all_rows(rows_recv, rows_send, recv_splits, send_splits, self.group)
        self._pending = ("owners", work, U, recv_splits, rows_recv, rows_send, recv_slots, union_keys, ustart, keep)
        self._payload = (own - C[rank][rank]) * 4 * D + max(max(U), 1) * 4 * D + 8 * (self.capacity + 1) + \
            4 * (self.flat.numel - self.head)

    def finish(self) -> float:
        self.last_gathered = None
        if not self._active():
            return 1.0
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        pend, self._pending = self._pending, None
        if pend is None:                                        # the hook did not fire (or carried no ids): dense fallback
            self._prepared = None
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
            self._payload = 4 * self.flat.numel
            self._last = dict(self._last, choice="dense (hook did not fire)")
            return 1.0 / world
        if self.head < self.flat.numel:
            dist.all_reduce(self.flat.grad[self.head:], op=dist.ReduceOp.SUM, group=self.group)
        g = self._table_grad()
        D = self.dim
        if pend[0] == "dense":
            pend[1].wait()
            return 1.0 / world
        if pend[0] == "rows":
            _, work, counts, ids_mat, rows_all, _keep = pend
            work.wait()
            self.last_gathered = [ids_mat[r, : counts[r]] for r in range(world) if counts[r] > 0]
            g.index_fill_(0, ids_mat[rank, : counts[rank]], 0.0)     # own rows out, then every rank's rows in, in rank order
            for r in range(world):
                # one launch per source ON PURPOSE: within a source the ids are unique, and the sources are added in a fixed
                # order -- one concatenated index_add_ would leave the order of the (up to `world`) additions into a shared
                # row to the atomics, and replicas that round differently drift apart
                if counts[r] > 0:
                    g.index_add_(0, ids_mat[r, : counts[r]], rows_all[r, : counts[r]])
            return 1.0 / world
        _, work, U, recv_splits, rows_recv, _rows_send, recv_slots, union_keys, ustart, _keep = pend
        work.wait()
        Umax, total = max(max(U), 1), sum(U)
        acc = torch.zeros(Umax, D, dtype=g.dtype, device=g.device)
        off = 0
        for s in range(world):                                  # the owner's sum, in rank order (unique slots per source)
            n_s = recv_splits[s]
            if n_s:
                acc.index_add_(0, recv_slots[off: off + n_s], rows_recv[off: off + n_s])
                off += n_s
        rows_all = torch.empty(world, Umax, D, dtype=g.dtype, device=g.device)
        _all_gather_rows(rows_all, acc, self.group)
        if total > 0:
            ukeys = union_keys[:total]
            own_of = ukeys // self.rows
            pos = own_of * Umax + (torch.arange(total, device=g.device) - ustart.index_select(0, own_of))
            union_ids = ukeys % self.rows
            g.index_copy_(0, union_ids, rows_all.view(-1, D).index_select(0, pos))     # unique ids: every row written once
            self.last_gathered = [union_ids]
        return 1.0 / world

    def info(self) -> Dict:
        world = dist.get_world_size(self.group) if self._active() else 1
        dense = 4 * self.flat.numel
        row_b = 4 * self.dim
        cap, uni = self._last.get("max_unique_rows", 0), self._last.get("union_rows", 0)
        return {"mode": self.mode, "last_step_choice": self._last["choice"], "max_unique_rows_per_rank": cap,
                "union_rows": uni, "id_capacity": self.capacity,
                "payload_bytes_per_rank": int(self._payload), "dense_payload_bytes_per_rank": dense,
                "rule": f"min over predicted_wire_ms of owners / rows / dense ({wire_model()} model)" if self.mode == "auto" else None,
                "predicted_wire_ms": {"dense": predicted_wire_ms("dense", dense, world),
                                      "rows": predicted_wire_ms("rows", cap * row_b, world),
                                      "owners": predicted_wire_ms("owners", cap * row_b, world, -(-uni // max(world, 1)) * row_b),
                                      "link_GBps": XGMI_LINK_GBPS, "unmeasured": True},
                "note": "rank r owns the table rows id % world == r: one early all-gather of the unique ids (fixed capacity), "
                        "all-to-all of the touched gradient rows to their owners, sum in rank order, all-gather of the reduced "
                        "rows (each touched row once, no ids); dense all-reduce of the non-table gradient; last step's sizes"}


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _DoneWork:
    def wait(self):
        return True


def _is_rccl(group=None) -> bool:
    try:
        return str(dist.get_backend(group)).lower() == "nccl"
    except Exception:
        return False


def _all_gather_rows(out: torch.Tensor, piece: torch.Tensor, group=None, async_op: bool = False):
    """all-gather of equally sized pieces into the leading axis of `out`: ``all_gather_into_tensor`` over RCCL, the list
    form otherwise (gloo: the CPU tests of the orchestration, and the two-ranks-on-one-GPU test)."""
    piece = piece.contiguous()
    if out.is_cuda and _is_rccl(group):
        return dist.all_gather_into_tensor(out, piece, group=group, async_op=async_op)
    parts = [torch.empty_like(piece) for _ in range(out.shape[0])]
    dist.all_gather(parts, piece, group=group)
    for r, part in enumerate(parts):
        out[r].copy_(part)
    return _DoneWork() if async_op else None


def _all_to_all_rows(recv: torch.Tensor, send: torch.Tensor, recv_splits, send_splits, group=None):
    """async ``all_to_all_single`` (RCCL, or gloo over CPU tensors); gloo has no all-to-all for GPU tensors, so the
    two-ranks-on-one-GPU test stages through the host."""
    send = send.contiguous()
    if send.is_cuda and not _is_rccl(group):
        r_cpu = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r_cpu, send.cpu(), recv_splits, send_splits, group=group)
        recv.copy_(r_cpu)
        return _DoneWork()
    return dist.all_to_all_single(recv, send, recv_splits, send_splits, group=group, async_op=True)


class LazyTableAdam:
    """Dense-Adam semantics for the (V, D) embedding table WITHOUT touching every row every step (VERDICT round 3 item 8;
    reference: ``abstract_recommender.py:96`` -- ``torch.optim.Adam`` over all parameters, so rows with a zero gradient still
    move through their decaying moments).  A row whose gradient is zero evolves by a deterministic fp32 recurrence in
    (p, m, v, step): its missed steps are replayed -- same instruction sequence, same bits -- when the row is next gathered
    (``begin``: the rows of the step's ids, before the forward), when it next receives a gradient (``finish``) or when the state
    is exported (``flush``).  ``last[r]`` records how far row r has been advanced.  A rolling flush (rows r with
    r % period == step % period, on the trainer's side stream) bounds every row's lag by ``period`` < 128, the window of bias
    corrections one kernel call carries.  Against the dense kernel at V = 70k, B = 128: 0.70 GB / ~100 us of every step become
    ~0.2 GB (the touched rows + the 0.84 M non-table parameters); the result after ``flush`` is bit-identical
    (tests/test_gpu_parity.py::test_lazy_table_adam_is_bit_identical_to_dense_adam)."""

    def __init__(self, flat: FlatParams, opt: FusedAdam, table: torch.Tensor, period: int = 64):
        self.flat, self.opt = flat, opt
        self.rows, self.dim = int(table.shape[0]), int(table.shape[1])
        idx = [i for i, p in enumerate(flat.params) if p is table]
        if len(idx) != 1 or table.dim() != 2 or self.dim % 4:
            raise ValueError("LazyTableAdam: the table must be one 2-D parameter of the flat buffer with dim % 4 == 0")
        self.offset = int(flat.offsets[idx[0]])
        self.numel = self.rows * self.dim
        self.head = self.offset + self.numel       # (end of the table in the flat buffer)
        dev = flat.flat.device
        self.last = torch.zeros(self.rows, dtype=torch.int32, device=dev)
        # marks of step t live in marks[t & 1]: the EARLY catch-up of step t + 1 (``hint``) writes the other array while step t's
        # marks are still in use
        self.marks = [torch.zeros(self.rows, dtype=torch.int32, device=dev) for _ in range(2)]
        self._hinted = 0               # the step whose marks + catch-up were issued early (0: none)
        self.hint_hits = self.hint_misses = 0   # hints the next ``begin`` recognised / did not (other ids came)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.period = int(period)
        assert 1 <= self.period < 127
        self.flushed_upto = 0          # every row is at least this far (== opt.step_count: nothing pending)

    def _views(self):
        cached = getattr(self, "_view_cache", None)
        if cached is None:          # (the four flat buffers never move: build the views once, not three times a step)
            v = lambda t: t[self.offset: self.head].view(self.rows, self.dim)
            cached = self._view_cache = (v(self.flat.flat), v(self.flat.grad), v(self.opt.exp_avg), v(self.opt.exp_avg_sq))
        return cached

    def _advance(self, mark, upto, with_grad, grad_scale=1.0, stride=1, offset=0, exclude=None, exclude_tag=0):
        ops.adam_rows_advance_(*self._views(), self.last, mark, self.status, upto, with_grad, self.opt.lr, self.opt.betas,
                               self.opt.eps, grad_scale, stride, offset, exclude, exclude_tag)

    @property
    def mark(self) -> torch.Tensor:
        """The marks of the step in flight (t = step_count + 1)."""
        return self.marks[(self.opt.step_count + 1) & 1]

    @property
    def pending(self) -> bool:
        return self.flushed_upto < self.opt.step_count

    def begin(self, ids: torch.Tensor, side=None) -> None:
        """Before the forward of step t = step_count + 1: the rows of `ids` are brought to step t - 1 (what the gather must
        read); then the step's slice of the rolling flush, on `side` if given (it only moves rows the catch-up left behind)."""
        t = self.opt.step_count + 1
        self._join_side()                # the previous step's flush slice (side stream) before anything here moves a row
        h = getattr(self, "_hint_ids", None)
        # (a hint is recognised by the identity of its id tensor -- the trainer keeps it on the prepared batch -- or, for callers
        #  that pass views of one buffer, by address and length)
        hinted = self._hinted == t and h is not None and (h is ids or (h.data_ptr() == ids.data_ptr() and h.numel() == ids.numel()))
        if self._hinted == t:                  # (counted: a hint that never matches is wasted side-stream work -- round-5 advisor)
            self.hint_hits += int(hinted)
            self.hint_misses += int(not hinted)
        self._hinted, self._hint_ids = 0, None
        if not hinted:
            # (a hint for other ids than the ones that came: its rows were advanced to t - 1 and carry this step's tag -- their
            #  update at the end of the step is a zero-gradient step, i.e. the replay they would get later: harmless)
            ops.adam_rows_mark_(ids, self.mark, t)
            self._advance(self.mark, t - 1, False)
        mode = os.environ.get("NRL_LAZY_FLUSH", "side")        # (A/B: "side" | "main" = right here | "end" = after the update)
        if mode == "end":
            self._slice_pending = t
        elif side is not None and mode == "side":
            self._side = side
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._advance(None, t - 1, False, stride=self.period, offset=t % self.period)
        else:
            self._advance(None, t - 1, False, stride=self.period, offset=t % self.period)

    def hint(self, next_ids: torch.Tensor, side) -> None:
        """The mark + catch-up of step t + 1, issued on `side` WHILE step t runs (call after ``begin`` of step t): the rows of
        `next_ids` that step t does not own are advanced to step t -- a row outside step t's batch has a zero gradient at step
        t ON ONE RANK, so the replay is already determined -- and the rows both steps touch are brought to t by step t's own
        update.  The next ``begin`` then launches nothing on the caller's stream (41 us of 2.8 ms at B = 128).  The caller joins
        `side` before the next forward (the trainer's end-of-backward join covers it)."""
        t = self.opt.step_count + 1
        nxt = self.marks[(t + 1) & 1]
        self._side = side
        with torch.cuda.stream(side):
            ops.adam_rows_mark_(next_ids, nxt, t + 1)
            self._advance(nxt, t, False, exclude=self.marks[t & 1], exclude_tag=t)
        self._hinted, self._hint_ids = t + 1, next_ids

    def _join_side(self) -> None:
        side = getattr(self, "_side", None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            self._side = None

    def mark_more(self, ids: torch.Tensor) -> None:
        """Rows other ranks touched this step (touched-row exchange): same tag as ``begin``."""
        ops.adam_rows_mark_(ids, self.mark, self.opt.step_count + 1)

    def update(self, grad_scale: float) -> None:
        """After the backward (and the gradient exchange), BEFORE ``opt.begin_step()``: step t for the marked rows with their
        gradient rows (cleared)."""
        # (rows another rank touched -- mark_more -- were not advanced by begin(): the replay they get here must not race the
        #  rolling-flush slice of the side stream over the same rows; a no-op when the trainer's join already covered it)
        self._join_side()
        self._advance(self.mark, self.opt.step_count, True, grad_scale)
        self._run_pending_slice()

    def _run_pending_slice(self) -> None:
        t = getattr(self, "_slice_pending", None)
        if t is not None:                # the flush slice on the launch stream, after the update: rows nobody marked, to step t - 1
            self._slice_pending = None
            self._advance(None, t - 1, False, stride=self.period, offset=t % self.period)

    def update_scan(self, grad_scale: float) -> None:
        """``update`` without a list of touched rows (after a DENSE all-reduce any row may carry a gradient): every row whose
        gradient row has a non-zero element gets step t (after its missed steps), all-zero rows stay lazy -- their dense update
        IS the replay they get later.  Reads the whole table gradient once (84 MB at V = 70k) instead of 0.70 GB of Adam state."""
        self._join_side()
        self._advance(None, self.opt.step_count, 2, grad_scale)
        self._run_pending_slice()          # (NRL_LAZY_FLUSH=end: the rolling flush must run under the dense exchange too)

    def finish(self, grad_scale: float) -> None:
        """``update`` + the dense kernel over everything else of the flat buffer (the single-table case)."""
        self.update(grad_scale)
        self.opt.begin_step()
        self.opt.step_range(0, self.offset, grad_scale, zero_grad=True)
        self.opt.step_range(self.head, self.flat.numel, grad_scale, zero_grad=True)

    def advance_all_before_dense(self) -> None:
        """A step whose table gradient may be non-zero ANYWHERE (a dense all-reduce fallback): every row to t - 1; the caller
        then runs the dense kernel over the whole flat buffer and calls ``mark_all_current``."""
        self._join_side()
        self._advance(None, self.opt.step_count, False)

    def mark_all_current(self) -> None:
        self.last.fill_(self.opt.step_count)
        self.flushed_upto = self.opt.step_count

    def finish_dense(self, grad_scale: float) -> None:
        self.advance_all_before_dense()
        self.opt.begin_step()
        self.opt.step_range(0, self.flat.numel, grad_scale, zero_grad=True)
        self.mark_all_current()

    def flush(self) -> None:
        """Every row to the current step: after this the flat buffers hold exactly what dense Adam would."""
        if self.pending:
            self._join_side()
            self._advance(None, self.opt.step_count, False)
            self.flushed_upto = self.opt.step_count

    def check(self) -> None:
        """(sync) raises if a row was ever found further behind than the bias-correction window."""
        if int(self.status.item()) != 0:
            raise RuntimeError("LazyTableAdam: a table row lagged more than 127 steps (rolling flush period too long?)")


class NRMSTrainer:
    """forward -> CE loss -> backward (table-gradient all-reduce overlapped) -> fused Adam."""

    def __init__(self, module, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, group=None,
                 grad_exchange: str = "dense", head_chunks: int = 4, lazy_adam: Optional[bool] = None,
                 id_capacity: int = 32768):
        if grad_exchange not in ("dense", "rows", "owners", "auto"):
            raise ValueError("grad_exchange must be 'dense' (all-reduce of the flat gradient), 'rows' (all-gather of the touched "
                             "table rows), 'owners' (touched rows reduced by their owner rank, then all-gathered) or 'auto' (per "
                             "step, whichever of the three the wire model prices lowest)")
        self.module = module
        self.flat = FlatParams(module.parameters())
        self.opt = FusedAdam(self.flat, lr, betas, eps)
        self.group = group
        self._losses: List[torch.Tensor] = []        # folded into (_loss_sum, _loss_count) every LOSS_FOLD steps: bounded
        self._loss_sum, self._loss_count = None, 0
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
        if grad_exchange in ("owners", "auto") and head > 0:
            self.reduce = OwnerRowsExchange(self.flat, head, te.embedding_layer.weight, group, mode=grad_exchange,
                                            id_capacity=id_capacity)
        elif grad_exchange == "rows" and head > 0:
            self.reduce = TouchedRowsExchange(self.flat, head, te.embedding_layer.weight, group)
        else:
            self.reduce = OverlappedGradReduce(self.flat, head, group, chunks=head_chunks)
        if head > 0:
            te.table_grad_hook = self.reduce.start_head
        self._side = None
        self._table_encoder = te if head > 0 else None
        dev = self.flat.params[0].device
        # d(loss)/d(loss): one cached tensor instead of autograd's per-step fill; losses recognise it (ops.register_unit_grad)
        self._unit_root = os.environ.get("NRL_UNIT_ROOT_GRAD", "1") not in ("", "0")     # (0: plain loss.backward(), A/B runs)
        self._one = ops.register_unit_grad(torch.ones((), dtype=torch.float32, device=dev))
        if dev.type == "cuda" and os.environ.get("NRL_DEFER_USER_WGRAD", "1") not in ("", "0"):
            self._side = torch.cuda.Stream(device=dev)
        # lazy dense Adam for the embedding table (bit-identical to the dense kernel after a flush): on one GPU, and under the
        # touched-row exchange (every rank then knows every touched row); the dense all-reduce leaves any row possibly non-zero
        self.lazy = None               # the word-embedding table's lazy optimizer (first of `lazy_tables`), or None
        self.lazy_tables = []          # [(LazyTableAdam, ids_of_batch)]
        self._dense_ranges = [(0, self.flat.numel)]
        self._in_step = False
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if world > 1 and dev.type == "cuda" and "NRL_NEWS_FORK" not in os.environ:
            # N > 1: the weight-gradient phase of the news-encoder backward is the window the gradient exchange hides in (it starts when
            # the table gradient is complete).  `news_fork` (default on since round 6: -70 ... -130 us of the one-GPU step) moves its
            # three GEMMs OUT of that window, beside the activation-gradient chain: 0.60 ms of window at B = 128 -> none.  With
            # 87 MB to exchange per step the window is worth more than the fork: keep the three weight gradients behind the table
            # gradient on more than one rank (NRL_NEWS_FORK=0|1 in the environment decides instead, for A/B runs on real nodes).
            _lib.set_option("news_fork", False)
        if lazy_adam is None:
            lazy_adam = os.environ.get("NRL_LAZY_ADAM", "1") not in ("", "0")
        # world > 1: under the touched-row exchange the gathered ids name the rows; under the dense all-reduce (and `auto`) the
        # update scans the reduced gradient for non-zero rows (`update_scan`) -- either way the optimizer of N ranks is the one
        # of one rank
        if lazy_adam and dev.type == "cuda":
            self._find_lazy_tables(module, enc, single=False)
        if self.lazy_tables:
            self.lazy = self.lazy_tables[0][0]
            cuts = sorted((t.offset, t.head) for t, _ in self.lazy_tables)
            self._dense_ranges, lo = [], 0
            for a, b in cuts:
                if a > lo:
                    self._dense_ranges.append((lo, a))
                lo = b
            if lo < self.flat.numel:
                self._dense_ranges.append((lo, self.flat.numel))
            if hasattr(module, "register_state_dict_pre_hook"):
                module.register_state_dict_pre_hook(lambda *_a, **_k: self._table_read())
            # load_state_dict on a live trainer (restoring the best weights): every row is brought to the current step FIRST, so
            # the loaded values are not followed by a replay of zero-gradient steps the old values had missed
            reg = getattr(module, "register_load_state_dict_pre_hook", None) or getattr(module, "_register_load_state_dict_pre_hook", None)
            if reg is not None:
                reg(lambda *_a, **_k: self._table_read())

    def _find_lazy_tables(self, module, enc, single: bool) -> None:
        """Embedding tables whose gradient is zero outside the rows a step's batch names: the word table of every text encoder
        (ids: the token ids of the attributes it encodes, ``batch['x_all']``) and the LSTUR user encoder's long-term user table
        (ids: ``batch['user_idx']``).  Anything that reads such a table outside a step sees dense-Adam values (forward pre-hooks)."""
        in_flat = {id(p) for p in self.flat.params}
        seen = set()
        if enc is not None and hasattr(enc, "text_encoders"):
            for name, te in enc.text_encoders.items():
                emb = getattr(te, "embedding_layer", None)
                w = getattr(emb, "weight", None)
                if w is None or id(w) not in in_flat or w.dim() != 2 or w.shape[1] % 4 or not w.requires_grad:
                    continue
                attrs = [n for n, t2 in enc.text_encoders.items() if t2 is te]
                if id(w) in seen:
                    continue
                seen.add(id(w))

                def ids_of(batch, _attrs=tuple(attrs)):
                    xa = batch.get("x_all", {})
                    parts = [xa[a].reshape(-1) for a in _attrs if torch.is_tensor(xa.get(a))]
                    if len(parts) != len(_attrs):
                        raise RuntimeError("NRMSTrainer: the lazy table optimizer needs the step's token ids (batch['x_all'])")
                    return parts[0] if len(parts) == 1 else torch.cat(parts)

                self.lazy_tables.append((LazyTableAdam(self.flat, self.opt, w), ids_of))
                te.register_forward_pre_hook(lambda _m, _a: self._table_read())
                if single:
                    return
        ue = getattr(module, "user_encoder", None)
        w = getattr(getattr(ue, "long_term_user_embedding", None), "weight", None)
        if not single and w is not None and id(w) in in_flat and w.dim() == 2 and w.shape[1] % 4 == 0 and w.requires_grad:
            self.lazy_tables.append((LazyTableAdam(self.flat, self.opt, w), lambda batch: batch["user_idx"].reshape(-1)))
            ue.register_forward_pre_hook(lambda _m, _a: self._table_read())

    LOSS_FOLD = 256

    def _fold_losses(self) -> None:
        if self._losses:
            tot = torch.stack(self._losses).sum().double()
            self._loss_sum = tot if self._loss_sum is None else self._loss_sum + tot
            self._loss_count += len(self._losses)
            self._losses = []

    def epoch_end(self) -> Dict[str, float]:
        """Mean train loss over the steps since the last call (all ranks' steps under data parallelism), then reset --
        what ``on_train_epoch_end`` logs as train/loss when Lightning drives the module (nrms_module.py:380-396).  A caller
        that never asks (bench loops) holds at most LOSS_FOLD loss scalars: ``step`` folds them into a running device sum."""
        self._fold_losses()
        if self._loss_count == 0:
            return {}
        tot = self._loss_sum.reshape(1)
        cnt = torch.tensor([float(self._loss_count)], dtype=torch.float64, device=tot.device)
        self._loss_sum, self._loss_count = None, 0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            both = torch.cat([tot, cnt])
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=self.group)
            tot, cnt = both[:1], both[1:]
        return {"train/loss": float(tot / cnt)}

    def state_dict(self) -> Dict:
        """Everything a resumed run needs that the module's own ``state_dict`` does not hold: Adam's step count and its two
        moment buffers (flat, in ``FlatParams`` order -- ``layout`` names the slices), plus the hyper-parameters they were
        built under.  (Lightning checkpoints ``optimizer.state_dict()`` next to the module; this is that half for the flat
        optimizer of the bench / test loop.)"""
        if getattr(self, "lazy_tables", None):
            self.flush()                                   # moments of EVERY row at step_count
        names = {id(p): n for n, p in self.module.named_parameters()}
        layout = [(names.get(id(p), f"param{i}"), int(o), int(p.numel())) for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets))]
        return {"step_count": int(self.opt.step_count), "exp_avg": self.opt.exp_avg.detach().clone(),
                "exp_avg_sq": self.opt.exp_avg_sq.detach().clone(), "lr": self.opt.lr, "betas": tuple(self.opt.betas),
                "eps": self.opt.eps, "layout": layout, "numel": int(self.flat.numel)}

    def load_state_dict(self, state: Dict) -> None:
        names = {id(p): n for n, p in self.module.named_parameters()}
        layout = [(names.get(id(p), f"param{i}"), int(o), int(p.numel())) for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets))]
        if int(state["numel"]) != self.flat.numel or [tuple(x) for x in state["layout"]] != layout:
            raise ValueError("NRMSTrainer.load_state_dict: the optimizer state was saved for a different parameter layout")
        self.opt.step_count = int(state["step_count"])
        self.opt.exp_avg.copy_(state["exp_avg"])
        self.opt.exp_avg_sq.copy_(state["exp_avg_sq"])
        self.opt.lr, self.opt.betas, self.opt.eps = float(state["lr"]), tuple(state["betas"]), float(state["eps"])
        for tab, _ in getattr(self, "lazy_tables", []):      # a saved state is a flushed state: every row stands at step_count
            tab.mark_all_current()

    def _table_read(self) -> None:
        if not self._in_step:
            self.flush()

    def flush(self) -> None:
        """Brings every lazily updated table row to the current step (no-op without the lazy table optimizer)."""
        for tab, _ in self.lazy_tables:
            pending = tab.pending
            tab.flush()
            if pending:
                tab.check()        # (one scalar read-back, outside the step: a row that ever lagged past the window raises here)

    def exchange_info(self) -> Dict:
        """What the data-parallel gradient exchange ships per rank and step (bench.py prints it for N > 1), with -- when
        ``reduce.measure`` was on -- the MEASURED time the launch stream spent blocked on it next to the predicted wire time."""
        info = self.reduce.info()
        ms, n = self.reduce.exposed_wait_ms() if hasattr(self.reduce, "exposed_wait_ms") else (0.0, 0)
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        info["measured"] = {
            "world": world, "exposed_wait_ms_per_step": round(ms, 4), "steps_measured": n,
            "what": "event pairs on the launch stream around every wait for a collective of the step (dense: the tail all-reduce + "
                    "the waits for the head slices; rows / owners: the whole finish, i.e. waits + the kernels adding the gathered "
                    "rows): the part of the exchange the backward did NOT hide.  0 at world 1 (nothing is exchanged)."}
        return info

    def _prefetch(self, next_batch: Dict) -> None:
        """The id work of the NEXT step, issued on the side stream while this one runs: concatenating its history / candidate
        ids, their counting sort (for its embedding gradient) and -- one rank, lazy table optimizer -- the mark + catch-up of
        the table rows it will gather (``LazyTableAdam.hint``).  ~50 us of small launches leave the launch stream; the
        end-of-backward join covers them."""
        if next_batch is None or self._side is None or not hasattr(self.module, "_prepare"):
            self._next = None
            return
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)           # (this step's begin(): marks written, rows caught up; last step's update)
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        with torch.cuda.stream(self._side):
            nb = self.module._prepare(next_batch)
            for t in (nb.get("x_all") or {}).values():
                if torch.is_tensor(t):
                    t.record_stream(main)      # allocated under the side stream, read on the launch stream next step
            if world == 1:
                # the ids of every lazy table are built HERE, on the stream that just produced x_all (an encoder registered under
                # several attributes concatenates them: on the launch stream that read raced the side stream's writes, and the
                # fresh tensor never matched the hint -- round-5 advisor), and kept on the prepared batch: the next ``step``
                # hands ``begin`` the SAME tensor, which is how a hint is recognised
                cache = nb.setdefault("_lazy_ids", {})
                for i, (tab, ids_of) in enumerate(self.lazy_tables):
                    ids = ids_of(nb)
                    ids.record_stream(main)
                    cache[i] = ids
                    tab.hint(ids, self._side)
        self._next = (next_batch, nb)

    def _prefetch_pending(self) -> None:
        nb, self._pending_next = getattr(self, "_pending_next", None), None
        if nb is not None:
            self._prefetch(nb)

    def step(self, batch: Dict, next_batch: Optional[Dict] = None) -> torch.Tensor:
        """One train step.  ``next_batch`` (optional): the batch the NEXT call will be given -- a loader that prefetches has it --
        lets the trainer run that step's id bookkeeping beside this step (``_prefetch``): same kernels on the same operands, and
        the optimizer arithmetic is bit-identical to the in-line form (``test_lazy_table_adam_early_catch_up_is_bit_identical_to_
        dense_adam``); two RUNS of a step still differ at the 1e-7 level, as any two runs do (the backward's atomics)."""
        nxt = getattr(self, "_next", None)
        if nxt is not None:
            if not getattr(self, "_side_joined", False) and self._side is not None:
                torch.cuda.current_stream().wait_stream(self._side)     # (no end-of-backward join happened: join here)
            if nxt[0] is batch:
                batch = nxt[1]
        self._next = None
        self._side_joined = False
        # (nn.Module.train() walks every submodule and rebinds the flag: 80 us of host time per step when called blindly; the
        #  cached list makes the check a few microseconds and still catches a submodule somebody left in eval())
        mods = getattr(self, "_mods", None)
        if mods is None:
            mods = self._mods = list(self.module.modules())
        if not all(m.training for m in mods):
            self.module.train()
            self._mods = None
        rows_mode = hasattr(self.reduce, "prepare") and self.reduce._active()
        if rows_mode or self.lazy_tables:
            # the step's token ids exist now.  Touched-row exchange: unique ids + the async exchange of their counts go out before
            # the forward, so the backward's hook finds the sizes on the host without draining the launch queue.  Lazy table
            # Adam: the rows the forward gathers are brought up to date first.
            batch = self.module._prepare(batch) if hasattr(self.module, "_prepare") else batch
            if rows_mode:
                xa = batch.get("x_all", {})
                if torch.is_tensor(xa.get("title")):
                    self.reduce.prepare(xa["title"], xa.get("title_order"))
            cached = batch.get("_lazy_ids") or {}
            for i, (tab, ids_of) in enumerate(self.lazy_tables):
                ids = cached.get(i)
                tab.begin(ids if ids is not None else ids_of(batch), self._side)
        # the next step's id work goes out from a forward hook of the news encoder: right AFTER the chip-filling kernels of the news
        # forward, so that it runs beside the user encoder's few-row launches (issued at the top of the step it ran beside the
        # fused forward and cost it what it saved: 0.545 -> 0.606 ms per launch, profiles/r05_ab.txt)
        self._pending_next = next_batch if os.environ.get("NRL_PREFETCH_IDS", "1") not in ("", "0") else None
        if self._pending_next is not None and not getattr(self, "_hooked", False):
            enc = getattr(self.module, "news_encoder", None)
            if enc is not None and os.environ.get("NRL_PREFETCH_AT", "hook") == "hook":
                enc.register_forward_hook(lambda *_a, **_k: self._prefetch_pending())
            self._hooked = True
            if enc is None or os.environ.get("NRL_PREFETCH_AT", "hook") == "top":
                self._prefetch_pending()
        elif self._pending_next is not None and os.environ.get("NRL_PREFETCH_AT", "hook") == "top":
            self._prefetch_pending()
        self._in_step = True
        try:
            # model_step directly: training_step would append preds / targets to training_step_outputs every step, and
            # nothing here runs the epoch-end hook that clears them; the loss is still tracked for `epoch_end()`
            loss = self.module.model_step(batch)[0]
        finally:
            self._in_step = False
        self._prefetch_pending()               # (a module whose news encoder never ran through the hook)
        self._losses.append(loss.detach())
        if len(self._losses) >= self.LOSS_FOLD:
            self._fold_losses()
        root = self._one if self._unit_root and loss.dim() == 0 and loss.dtype == torch.float32 and loss.device == self._one.device else None
        if self._side is not None:
            # the user encoder's weight gradients on a side stream beside the news-encoder backward; joined on exit
            with ops.deferred_weight_grads(self._side) as deferred:
                loss.backward(gradient=root)
            if deferred.joined:
                self._side_joined = True
                # that join also covers the lazy optimizer's flush slice (issued on the same side stream before the forward): one
                # cross-stream wait per step instead of two (each is a barrier packet of ~6 us on the launch queue)
                for tab, _ in self.lazy_tables:
                    tab._side = None
        else:
            loss.backward(gradient=root)
        # parameters whose gradient came through ordinary autograd (``.grad``: a transformer body, a small
        # head fed through torch ops) rather than through a kernel writing ``main_grad``: fold them in
        for p in self.flat.params:
            if p.grad is not None:
                p.main_grad.add_(p.grad)
                p.grad = None
        if self.lazy_tables:
            scale = self._finish_exchange()
            world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
            scan = world > 1                               # other ranks' rows carry gradients too
            if rows_mode:
                gathered = getattr(self.reduce, "last_gathered", None)
                if gathered is not None:                   # (None: the exchange went dense this step -> scan)
                    for ids_r in gathered:
                        self.lazy.mark_more(ids_r)
                    scan = False
            for i, (tab, _) in enumerate(self.lazy_tables):
                # (only the first table's marks are completed by the exchange; any other table scans under data parallelism)
                if scan or (world > 1 and i > 0):
                    tab.update_scan(scale)
                else:
                    tab.update(scale)
            self.opt.begin_step()
            for lo, hi in self._dense_ranges:
                self.opt.step_range(lo, hi, scale, zero_grad=True)
        elif hasattr(self.reduce, "finish_pipelined"):
            # dense exchange: Adam over each slice of the flat buffer as soon as its all-reduce has landed
            self.opt.begin_step()
            for lo, hi, scale in self.reduce.finish_pipelined():
                self.opt.step_range(lo, hi, grad_scale=scale, zero_grad=True)
        else:
            scale = self._finish_exchange()
            self.opt.step(grad_scale=scale, zero_grad=True)
        return loss.detach()

    def _finish_exchange(self) -> float:
        """``reduce.finish()``; a row exchange's finish -- its waits AND the kernels that add the gathered rows -- is timed as one
        piece when measuring is on (the dense exchange times its waits itself)."""
        r = self.reduce
        if isinstance(r, OverlappedGradReduce) or not getattr(r, "measure", False) or not r._active():
            return r.finish()
        scale = r._timed(r.finish)
        r._step_measured()
        return scale
