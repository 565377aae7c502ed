"""``to_dense_batch`` (torch_geometric 2.3.0; reference call sites nrms_module.py:233,237,277-284)."""
import os
from typing import Optional, Tuple

import torch

from . import ops

# NRL_FULL_RESHAPE=0: always run the dense-batch kernels, also over full batches (A/B runs)
_FULL_IS_RESHAPE = os.environ.get("NRL_FULL_RESHAPE", "1") not in ("", "0")


def to_dense_batch(x: torch.Tensor, batch: torch.Tensor, batch_size: Optional[int] = None,
                   max_num_nodes: Optional[int] = None,
                   offsets: Optional[torch.Tensor] = None,
                   flat_idx: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """x (N, D) or (N,), sorted assignment vector batch (N,) -> dense (B, max, D)/(B, max), bool mask.

    ``batch_size`` / ``max_num_nodes`` / ``offsets`` may be supplied to avoid the two device->host
    syncs the shapes otherwise cost (the reference pays them inside torch_geometric)."""
    batch_size, max_num_nodes, offsets = _resolve(batch, batch_size, max_num_nodes, offsets)
    counts = offsets[1:] - offsets[:-1]
    mask = torch.arange(max_num_nodes, device=x.device)[None, :] < counts[:, None]
    return dense_rows(x, batch, batch_size, max_num_nodes, offsets, flat_idx), mask


def _resolve(batch, batch_size, max_num_nodes, offsets):
    if batch_size is None:
        batch_size = int(batch.max()) + 1 if batch.numel() else 0
    if offsets is None:
        offsets = ops.offsets_from_sorted_batch(batch, batch_size)
    if max_num_nodes is None:
        max_num_nodes = int((offsets[1:] - offsets[:-1]).max()) if batch_size else 0
    return batch_size, max_num_nodes, offsets


def dense_rows(x: torch.Tensor, batch: torch.Tensor, batch_size: Optional[int] = None,
               max_num_nodes: Optional[int] = None, offsets: Optional[torch.Tensor] = None,
               flat_idx: Optional[torch.Tensor] = None, max_is_exact: bool = False) -> torch.Tensor:
    """The dense tensor of ``to_dense_batch`` without the mask (every call site on the step path discards it, and
    building it costs three small launches per call).

    ``max_is_exact``: the caller vouches that ``max_num_nodes`` is the true largest group (the layout metadata of
    ``attach_layout`` / ``input_pipeline.build_batch``, not a truncating cap).  Then ``N == B * max`` means every group is
    full and the dense tensor IS the row-major reshape of ``x``: no launch forward or backward (training candidates are
    always 1 + K per impression; full histories whenever every user has >= max_history_len clicks)."""
    batch_size, max_num_nodes, offsets = _resolve(batch, batch_size, max_num_nodes, offsets)
    if max_is_exact and _FULL_IS_RESHAPE and batch_size > 0 and x.shape[0] == batch_size * max_num_nodes:
        return x.reshape((batch_size, max_num_nodes) + tuple(x.shape[1:]))
    if x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] % 4 == 0:
        return ops.ToDenseBatchFn.apply(x, offsets, batch_size, max_num_nodes)
    # label / category vectors: (N,) of any dtype -- tiny index bookkeeping, not arithmetic;
    # explicit slot indices instead of a boolean-mask store (which would sync the host)
    if flat_idx is None:
        flat_idx = dense_slot_index(batch, offsets, max_num_nodes)
    dense = x.new_zeros((batch_size * max_num_nodes,) + tuple(x.shape[1:]))
    dense[flat_idx] = x
    return dense.view((batch_size, max_num_nodes) + tuple(x.shape[1:]))


def dense_slot_index(batch: torch.Tensor, offsets: torch.Tensor, max_num_nodes: int) -> torch.Tensor:
    """Flat (b * max + position) slot of every ragged row; row-major order == concatenation order."""
    pos = torch.arange(batch.numel(), device=batch.device) - offsets[batch]
    return batch * max_num_nodes + pos
