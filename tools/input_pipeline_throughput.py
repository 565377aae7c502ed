#!/usr/bin/env python3
"""Input-pipeline throughput at MINDsmall shape (65 k news, 150 k train impressions, <= 50 clicks, ~37 candidates
of which ~1.5 clicked, 4 negatives per positive, B = 128): batches/s of ``TrainBatchLoader`` alone, of the NRMS
train step on a pre-built batch, and of loader + train step together; plus the reference-shaped pandas collate
(oracle/input_oracle.py, one core, ``num_workers: 0`` as in configs/data/mind_rec.yaml:66) on a bounded sample."""
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
    ap.add_argument("--cpu-batches", type=int, default=3)
    args = ap.parse_args()
    import pandas as pd

    from newsreclib_amd import input_pipeline as IP
    from newsreclib_amd.evaluation import DeviceNewsTable
    from newsreclib_amd.nrms_module import prepare_batch
    from newsreclib_amd.trainer import NRMSTrainer
    from tests.helpers import build_module
    from oracle import nrms_oracle as O
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
    mod = build_module(O.make_params(args.vocab, seed=1), p_drop=0.2)
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
    # reference-shaped host collate on a bounded sample (pandas .loc + concat + per-row pad), one core
    from oracle import input_oracle as IO
    news_df = pd.DataFrame({"tokenized_title": titles, "category_class": table.attrs["category"].cpu().numpy(),
                            "subcategory_class": table.attrs["subcategory"].cpu().numpy()},
                           index=[f"N{i + 1}" for i in range(n)])
    hr, cr = bt.hist_rows.cpu().numpy(), bt.cand_rows.cpu().numpy()
    idx = np.arange(B * args.cpu_batches)
    bhv = pd.DataFrame({"uid": [f"U{i}" for i in idx], "user": idx,
                        "history": [[f"N{r + 1}" for r in hr[hist_ptr[i]:hist_ptr[i + 1]]] for i in idx],
                        "candidates": [[f"N{r + 1}" for r in cr[cand_ptr[i]:cand_ptr[i + 1]]] for i in idx],
                        "labels": [list(labels[cand_ptr[i]:cand_ptr[i + 1]]) for i in idx]})
    nrng = np.random.default_rng(0)
    t = time.perf_counter()
    for b in range(args.cpu_batches):
        items = [IO.get_item(news_df, bhv, int(i), 50, IO.sample_candidates(np.array(bhv.iloc[int(i)]["labels"]), 4, nrng))
                 for i in range(b * B, (b + 1) * B)]
        IO.collate(items, ["title", "category"], 30)
    dt_cpu = (time.perf_counter() - t) / args.cpu_batches
    print(f"pandas collate    : {dt_cpu * 1e3:7.1f} ms/batch  {B / dt_cpu:9.0f} impressions/s "
          f"(reference-shaped restatement, 1 core, {args.cpu_batches} batches)")


if __name__ == "__main__":
    main()
