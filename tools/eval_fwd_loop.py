#!/usr/bin/env python3
"""Evaluation-mode forward of the bench workload in a loop (for `rocprofv3 --kernel-trace --stats`, and a wall-clock figure):
    python tools/eval_fwd_loop.py [--iters 50] [--batch 128]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--mode", default="fixed")
    ap.add_argument("--token-table", action="store_true",
                    help="forwards from the per-token q|k|v table (news_encoder.MHSAAddAtt.token_table: what an evaluation epoch runs under)")
    a = ap.parse_args()
    from newsreclib_amd import _lib
    from newsreclib_amd.nrms_module import attach_layout
    from newsreclib_amd.synthetic import make_batch
    _lib.load()
    dev = torch.device("cuda", 0)
    mod = bench.build_module(dev).eval()
    batches = [attach_layout(make_batch(a.batch, bench.VOCAB, a.mode, seed=1234 + 1000 * i, device=dev)) for i in range(4)]
    import contextlib
    te = mod.news_encoder.text_encoders["title"]
    with torch.no_grad(), (te.token_table() if a.token_table else contextlib.nullcontext()):
        for i in range(5):
            out = mod.forward(batches[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.iters):
            out = mod.forward(batches[i % 4])
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    print(f"eval forward B={a.batch} {a.mode}{' [token table]' if a.token_table else ''}: {dt * 1e3:.4f} ms = {a.batch / dt:.0f} impressions/s; checksum {float(out.double().sum()):.9f}")


if __name__ == "__main__":
    main()
