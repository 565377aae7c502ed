#!/usr/bin/env python3
"""Evaluation-path throughput on one GPU: full-impression scoring of a MIND-dev-shaped
synthetic set (65k unique news, ~37 candidates and <= 50 clicks per impression) --
  (a) the reference's flow: every batch re-encodes all of its history + candidate news;
  (b) encode-once: NewsVectorCache (table encoded once, impressions scored from gathered vectors).
Prints impressions/s for both (cache build time included in (b)) and checks that the scores agree."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impressions", type=int, default=20000)
    ap.add_argument("--news", type=int, default=65000)
    ap.add_argument("--vocab", type=int, default=70000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--uncached-batches", type=int, default=12)
    args = ap.parse_args()
    from functools import partial

    from newsreclib_amd.evaluation import DeviceNewsTable, NewsVectorCache
    from newsreclib_amd.nrms_module import NRMSModule
    from newsreclib_amd.synthetic import _titles
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    mod = NRMSModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title"],
        outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
        loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=False, pretrained_embeddings_path=None,
        plm_model=None, frozen_layers=None, embed_dim=300, num_heads=15, query_dim=200, dropout_probability=0.2,
        top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False, recs_fpath=None,
        optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None,
        pretrained_embeddings=torch.randn(args.vocab, 300) * 0.3).cuda().eval()
    table = DeviceNewsTable({"title": torch.from_numpy(_titles(rng, args.news, args.vocab, 30)),
                             "category": torch.from_numpy(rng.integers(1, 19, args.news))})
    n = args.impressions
    hs = np.clip(np.rint(rng.lognormal(3.0, 0.8, n)), 1, 50).astype(np.int64)
    cs = np.clip(np.rint(rng.lognormal(3.3, 0.7, n)), 2, 300).astype(np.int64)
    hist = torch.from_numpy(rng.integers(1, args.news, int(hs.sum()))).cuda()
    cand = torch.from_numpy(rng.integers(1, args.news, int(cs.sum()))).cuda()
    hs_t, cs_t = torch.from_numpy(hs).cuda(), torch.from_numpy(cs).cuda()
    ho = np.concatenate([[0], np.cumsum(hs)])
    co = np.concatenate([[0], np.cumsum(cs)])
    labels = torch.zeros(int(cs.sum()), device="cuda")
    print(f"{n} impressions, {args.news} unique news; rows to encode per epoch without cache: "
          f"{int(hs.sum() + cs.sum())} ({(hs.sum() + cs.sum()) / args.news:.1f}x the table)")

    def batch_slices(b):
        lo, hi = b * args.batch, min((b + 1) * args.batch, n)
        return (hist[ho[lo]:ho[hi]], hs_t[lo:hi], cand[co[lo]:co[hi]], cs_t[lo:hi], labels[co[lo]:co[hi]])

    nb = (n + args.batch - 1) // args.batch
    # (a) uncached, on a sample of batches
    k = min(args.uncached_batches, nb)
    with torch.no_grad():
        mod.forward(table.build_batch(*batch_slices(0)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ref = [mod.forward(table.build_batch(*batch_slices(b))) for b in range(k)]
        torch.cuda.synchronize()
    t_unc = time.perf_counter() - t0
    imp_unc = min(k * args.batch, n)
    # (b) encode once + score everything
    cache = NewsVectorCache(mod, table)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cache.build()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = [cache.scores(*batch_slices(b)[:4]) for b in range(nb)]
    torch.cuda.synchronize()
    t_score = time.perf_counter() - t0
    same = all(torch.equal(a, b) for a, b in zip(ref, got[:k]))
    print(f"uncached (reference flow): {imp_unc / t_unc:9.0f} impressions/s  ({k} batches of {args.batch})")
    print(f"encode-once:               {n / (t_build + t_score):9.0f} impressions/s  (table build {t_build * 1e3:.0f} ms "
          f"+ scoring {t_score * 1e3:.0f} ms for {n} impressions; scoring alone {n / t_score:.0f}/s)")
    print(f"scores identical to the uncached forward on the sampled batches: {same}")


if __name__ == "__main__":
    main()
