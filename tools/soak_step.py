#!/usr/bin/env python3
"""Soak of the train step's two-stream paths (side-stream counting sort, deferred user-encoder weight gradients): N steps over
rotating batches, device memory and the pending-event table before / after, loss finite throughout."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    from newsreclib_amd import ops
    from newsreclib_amd.nrms_module import attach_layout
    from newsreclib_amd.synthetic import make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    dev = torch.device("cuda", 0)
    mod = bench.build_module(dev)
    tr = NRMSTrainer(mod, lr=1e-4)
    batches = [attach_layout(make_batch(128, 70_000, "ragged" if i % 2 else "fixed", seed=10 + i, device=dev)) for i in range(6)]
    for i in range(20):
        tr.step(batches[i % 6])
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    losses = []
    for i in range(steps):
        losses.append(tr.step(batches[i % 6]))
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_allocated()
    ls = torch.stack(losses[-600:]).float()
    print(f"steps {steps}: memory allocated {m0 / 2**20:.1f} -> {m1 / 2**20:.1f} MiB (max {torch.cuda.max_memory_allocated() / 2**20:.1f}), "
          f"pending order events {len(ops._ORDER_EVENTS)}, last losses finite {bool(torch.isfinite(ls).all())}, "
          f"mean of the last 600 losses {float(ls.mean()):.4f} (first 20: {float(torch.stack(losses[:20]).mean()):.4f})")
    assert abs(m1 - m0) < 64 * 2**20 and len(ops._ORDER_EVENTS) <= 64 and bool(torch.isfinite(ls).all())


if __name__ == "__main__":
    main()
