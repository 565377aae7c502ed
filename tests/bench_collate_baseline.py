#!/usr/bin/env python3
"""CPU baseline of the host input pipeline: the reference-shaped dataset + collate restatement
(oracle/input_oracle.py: pandas ``.loc`` per impression, numpy negative sampling, ``pd.concat``, one padded row per
news; ``num_workers: 0`` as in configs/data/mind_rec.yaml:66) timed on one core over a bounded sample of
MINDsmall-shaped logs.  Not a test (not collected); lives here because only tests/ may use oracle/.

    python tests/bench_collate_baseline.py [--batches 3]
"""
import argparse
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import input_oracle as IO  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--news", type=int, default=65000)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--batches", type=int, default=3)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    n, B = args.news, args.batch
    m = B * args.batches
    lens = np.clip(np.round(rng.normal(11.5, 3.5, n)), 3, 30).astype(int)
    news = pd.DataFrame({"tokenized_title": [list(rng.integers(1, 70000, k)) for k in lens],
                         "category_class": rng.integers(0, 18, n), "subcategory_class": rng.integers(0, 200, n)},
                        index=[f"N{i + 1}" for i in range(n)])
    rows = []
    for i in range(m):
        nc = int(np.clip(round(rng.lognormal(3.3, 0.7)), 6, 299))
        labels = np.zeros(nc, dtype=np.int64)
        labels[rng.choice(nc, min(1 + rng.poisson(0.35), nc // 5), replace=False)] = 1
        rows.append({"uid": f"U{i}", "user": i, "labels": list(labels),
                     "history": [f"N{r}" for r in rng.integers(1, n, int(np.clip(round(rng.lognormal(3.0, 0.8)), 1, 50)))],
                     "candidates": [f"N{r}" for r in rng.integers(1, n, nc)]})
    bhv = pd.DataFrame(rows)
    t = time.perf_counter()
    for b in range(args.batches):
        items = [IO.get_item(news, bhv, i, 50, IO.sample_candidates(np.array(bhv.iloc[i]["labels"]), 4, rng))
                 for i in range(b * B, (b + 1) * B)]
        IO.collate(items, ["title", "category"], 30)
    dt = (time.perf_counter() - t) / args.batches
    print(f"pandas collate    : {dt * 1e3:7.1f} ms/batch  {B / dt:9.0f} impressions/s "
          f"(reference-shaped restatement, 1 core, {args.batches} batches of {B})")


if __name__ == "__main__":
    main()
