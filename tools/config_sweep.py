#!/usr/bin/env python3
"""Train-step and forward-only (eval) throughput of the NRMS path over the BASELINE.json configurations on one
GPU: configs[0] (B=32, V=70k), configs[1] (B=128, V=70k; the bench.py workload), the per-rank shapes of
configs[2] (V=150k, B=64 and B=128), plus the ragged MIND-like batch.  SURVEY.md section 8(d) asks for the
forward-only rate next to the train rate."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--engine", default="bf16x3")
    args = ap.parse_args()
    from newsreclib_amd import _lib
    from newsreclib_amd.nrms_module import attach_layout
    from newsreclib_amd.synthetic import make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    _lib.set_gemm_engine(args.engine)
    dev = torch.device("cuda", 0)
    print(f"engine={args.engine}  steps={args.steps}")
    print(f"{'config':44s} {'train ms':>9s} {'train imp/s':>12s} {'eval ms':>9s} {'eval imp/s':>11s}")
    for name, vocab, B, mode in (("configs[0]  B=32   V=70k  fixed", 70000, 32, "fixed"),
                                 ("configs[1]  B=128  V=70k  fixed", 70000, 128, "fixed"),
                                 ("configs[1]  B=128  V=70k  ragged", 70000, 128, "ragged"),
                                 ("configs[2]  B=64   V=150k fixed (per rank)", 150000, 64, "fixed"),
                                 ("configs[2]  B=128  V=150k fixed (weak)", 150000, 128, "fixed"),
                                 ("            B=512  V=70k  fixed", 70000, 512, "fixed")):
        bench.VOCAB = vocab
        mod = bench.build_module(dev)
        tr = NRMSTrainer(mod, lr=1e-4)
        batches = [attach_layout(make_batch(B, vocab, mode, seed=1234 + i, device=dev)) for i in range(4)]   # id concat + grouping run inside the step, as in bench.py
        it = iter(range(10 ** 9))
        dt_train = timed(lambda: tr.step(batches[next(it) % 4]), args.steps, args.warmup)
        mod.eval()
        with torch.no_grad():
            dt_eval = timed(lambda: mod.forward(batches[next(it) % 4]), args.steps, args.warmup)
        print(f"{name:44s} {dt_train * 1e3:9.3f} {B / dt_train:12.0f} {dt_eval * 1e3:9.3f} {B / dt_eval:11.0f}")
        del mod, tr, batches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
