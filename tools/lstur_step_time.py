#!/usr/bin/env python3
"""Times the LSTUR (BASELINE config 5) or NAML (--model naml) train step on one GPU: B users x 50 clicks, title 30 + abstract 50
tokens, CNN 300 filters x window 3, GRU 700.  Prints ms/step and impressions/s; with --breakdown also the
per-kernel time from torch.profiler-free HIP events around the module's stages."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--vocab", type=int, default=70000)
    ap.add_argument("--engine", default="bf16x3")
    ap.add_argument("--model", default="lstur", choices=["lstur", "naml", "cen", "mins"])
    args = ap.parse_args()
    from functools import partial

    from newsreclib_amd import _lib
    from newsreclib_amd.lstur_module import LSTURModule
    from newsreclib_amd.nrms_module import prepare_batch
    from newsreclib_amd.synthetic import add_lstur_fields, make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    _lib.set_gemm_engine(args.engine)
    torch.manual_seed(0)
    emb = torch.randn(args.vocab, 300) * 0.3
    mod = LSTURModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title", "abstract", "category"],
        outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
        loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=False,
        pretrained_embeddings_path=None, plm_model=None, frozen_layers=None, text_embed_dim=300, num_heads=15,
        num_filters=300, window_size=3, query_dim=200, categ_embed_dim=100, dropout_probability=0.2,
        num_users=45214, user_masking_probability=0.5, long_short_term_method="ini", top_k_list=[5, 10],
        num_categ_classes=18, num_sent_classes=3, save_recs=False, recs_fpath=None,
        optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None, pretrained_embeddings=emb).cuda()
    if args.model == "naml":   # configs/model/naml.yaml: 400 filters, category view, add_att combination
        from newsreclib_amd.naml_module import NAMLModule
        mod = NAMLModule(
            dataset_attributes=["title", "abstract", "category"], attributes2encode=["title", "abstract", "category"],
            outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
            loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=False,
            pretrained_embeddings_path=None, plm_model=None, frozen_layers=None, text_embed_dim=300, num_heads=15,
            num_filters=400, window_size=3, query_dim=200, categ_embed_dim=100, dropout_probability=0.2,
            top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False, recs_fpath=None,
            optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None, pretrained_embeddings=emb).cuda()
    if args.model == "cen":    # configs/model/cen_news_rec.yaml: title only, 400 filters, 20 heads, GRU 400 over 20 recent
        from newsreclib_amd.cen_news_rec_module import CenNewsRecModule
        mod = CenNewsRecModule(
            dataset_attributes=["title", "abstract", "category"], attributes2encode=["title"],
            outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
            loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=False,
            pretrained_embeddings_path=None, plm_model=None, frozen_layers=None, embed_dim=300, num_heads=20,
            num_filters=400, window_size=3, query_dim=200, dropout_probability=0.2, gru_hidden_dim=400,
            num_recent_news=20, top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False,
            recs_fpath=None, optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None,
            pretrained_embeddings=emb).cuda()
    if args.model == "mins":   # configs/model/mins.yaml: MHSA text encoder on title + abstract, 6 GRU channels
        from newsreclib_amd.mins_module import MINSModule
        mod = MINSModule(
            dataset_attributes=["title", "abstract", "category"], attributes2encode=["title", "abstract", "category"],
            outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
            loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=False,
            pretrained_embeddings_path=None, plm_model=None, frozen_layers=None, text_embed_dim=300,
            categ_embed_dim=100, num_heads=15, query_dim=200, dropout_probability=0.2, num_filters=300,
            num_gru_channels=6, top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False,
            recs_fpath=None, optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None,
            pretrained_embeddings=emb).cuda()
    trainer = NRMSTrainer(mod, lr=1e-4)
    batch = add_lstur_fields(make_batch(args.batch, vocab=args.vocab, mode="fixed", seed=1, device="cuda"), args.vocab,
                             19, 45215, 50, seed=2)
    batch = prepare_batch(batch)
    for _ in range(args.warmup):
        trainer.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"{args.model} B={args.batch} engine={args.engine}: {dt * 1e3:.3f} ms/step, {args.batch / dt:.1f} impressions/s, "
          f"loss={float(loss):.4f}")


if __name__ == "__main__":
    main()
