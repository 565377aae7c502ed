#!/usr/bin/env python3
"""Soak of the train step's two-stream paths (side-stream counting sort, deferred user-encoder weight gradients, the lazy table
optimizer's rolling flush): N steps over rotating batches, device memory before / after, loss finite throughout, no table row ever
beyond the lazy optimizer's bias-correction window, and -- over the first 300 steps -- the loss curve of a second trainer with the
dense optimizer (NRMSTrainer(lazy_adam=False)) on the same batches and dropout seeds."""
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

    def make(lazy):
        mod = bench.build_module(dev)
        te = mod.news_encoder.text_encoders["title"]
        orig = te.forward
        counter = [0]

        def fwd(text, seed=None, _o=orig, **kw):          # the same dropout draw per step in both trainers
            counter[0] += 1
            return _o(text, seed=1000 + counter[0], **kw)

        te.forward = fwd
        return mod, NRMSTrainer(mod, lr=1e-4, lazy_adam=lazy)

    mod, tr = make(True)
    batches = [attach_layout(make_batch(128, 70_000, "ragged" if i % 2 else "fixed", seed=10 + i, device=dev)) for i in range(6)]
    assert tr.lazy is not None
    mod_d, tr_d = make(False)
    la, ld = [], []
    for i in range(300):
        la.append(tr.step(batches[i % 6]))
        ld.append(tr_d.step(batches[i % 6]))
    la, ld = torch.stack(la).float(), torch.stack(ld).float()
    gap = float((la - ld).abs().max())
    print(f"lazy vs dense optimizer, 300 steps: max |loss difference| {gap:.2e} (loss {float(la[:20].mean()):.4f} -> {float(la[-20:].mean()):.4f})")
    del mod_d, tr_d
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    losses = []
    for i in range(steps):
        losses.append(tr.step(batches[i % 6]))
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_allocated()
    tr.flush()
    tr.lazy.check()
    ls = torch.stack(losses[-600:]).float()
    print(f"steps {steps}: memory allocated {m0 / 2**20:.1f} -> {m1 / 2**20:.1f} MiB (max {torch.cuda.max_memory_allocated() / 2**20:.1f}), "
          f"last losses finite {bool(torch.isfinite(ls).all())}, mean of the last 600 losses {float(ls.mean()):.4f} "
          f"(first 20: {float(torch.stack(losses[:20]).mean()):.4f}); table rows all at step {int(tr.lazy.last.min())} == {tr.opt.step_count}")
    assert abs(m1 - m0) < 64 * 2**20 and bool(torch.isfinite(ls).all()) and gap < 5e-3
    assert int(tr.lazy.last.min()) == int(tr.lazy.last.max()) == tr.opt.step_count


if __name__ == "__main__":
    main()
