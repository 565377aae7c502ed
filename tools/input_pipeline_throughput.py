#!/usr/bin/env python3
"""Input-pipeline throughput at MINDsmall shape (65 k news, 150 k train impressions, <= 50 clicks, ~37 candidates
of which ~1.5 clicked, 4 negatives per positive, B = 128): batches/s of ``TrainBatchLoader`` alone, of the NRMS
train step on a pre-built batch, and of loader + train step together.  (The reference-shaped pandas collate it is
compared with is timed by tests/bench_collate_baseline.py -- the restatement is test infrastructure.)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--news", type=int, default=65000)
    ap.add_argument("--impressions", type=int, default=150000)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--batches", type=int, default=200)
    ap.add_argument("--vocab", type=int, default=70000)
    args = ap.parse_args()

    from newsreclib_amd import input_pipeline as IP
    from newsreclib_amd.evaluation import DeviceNewsTable
    from newsreclib_amd.nrms_module import prepare_batch
    from newsreclib_amd.trainer import NRMSTrainer
    import bench
    rng = np.random.default_rng(0)
    n, m = args.news, args.impressions
    lens = np.clip(np.round(rng.normal(11.5, 3.5, n)), 3, 30).astype(int)
    titles = [list(rng.integers(1, args.vocab, k)) for k in lens]
    t0 = time.perf_counter()
    title = torch.from_numpy(IP.pad_token_lists(titles, 30))
    t_tok = time.perf_counter() - t0
    table = DeviceNewsTable({"news_ids": torch.arange(n) + 1, "title": title,
                             "category": torch.from_numpy(rng.integers(0, 18, n)),
                             "subcategory": torch.from_numpy(rng.integers(0, 200, n))}, device="cuda")
    hist_len = np.clip(np.round(rng.lognormal(3.0, 0.8, m)), 1, 50).astype(np.int64)
    ncand = np.clip(np.round(rng.lognormal(3.3, 0.7, m)), 6, 299).astype(np.int64)
    hist_ptr = np.concatenate(([0], np.cumsum(hist_len)))
    cand_ptr = np.concatenate(([0], np.cumsum(ncand)))
    labels = np.zeros(cand_ptr[-1], dtype=np.int64)
    npos = np.minimum(1 + rng.poisson(0.35, m), ncand // 5)
    for i in range(m):
        labels[cand_ptr[i] + rng.choice(ncand[i], npos[i], replace=False)] = 1
    bt = IP.BehaviorTable(hist_ptr, rng.integers(1, n, hist_ptr[-1]), cand_ptr, rng.integers(1, n, cand_ptr[-1]), labels,
                          rng.integers(1, 10 ** 6, m), rng.integers(0, 45000, m), device="cuda")
    print(f"tables: {n} news pre-tokenised in {t_tok:.2f} s; {m} impressions, {hist_ptr[-1]} clicks, "
          f"{cand_ptr[-1]} candidates ({labels.mean() * 100:.1f} % clicked)")
    loader = IP.TrainBatchLoader(table, bt, args.batch, 4, seed=0)

    def timed(fn, k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn(k)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / k

    def loader_only(k):
        for i, b in enumerate(loader):
            if i + 1 == k:
                break

    loader_only(20)
    dt_load = timed(loader_only, args.batches)
    bench.VOCAB = args.vocab
    mod = bench.build_module(torch.device("cuda", 0))
    tr = NRMSTrainer(mod, lr=1e-4)
    fixed = prepare_batch(next(iter(loader)))

    def step_only(k):
        for _ in range(k):
            tr.step(fixed)

    def both(k):
        for i, b in enumerate(loader):
            tr.step(b)
            if i + 1 == k:
                break

    step_only(10)
    dt_step = timed(step_only, args.batches)
    both(10)
    dt_both = timed(both, args.batches)
    B = args.batch
    print(f"loader alone      : {dt_load * 1e3:7.3f} ms/batch  {B / dt_load:9.0f} impressions/s")
    print(f"train step alone  : {dt_step * 1e3:7.3f} ms/batch  {B / dt_step:9.0f} impressions/s (ragged batch)")
    print(f"loader + step     : {dt_both * 1e3:7.3f} ms/batch  {B / dt_both:9.0f} impressions/s")


if __name__ == "__main__":
    main()
