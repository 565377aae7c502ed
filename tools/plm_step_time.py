#!/usr/bin/env python3
"""Times the NRMS-PLM train step (BASELINE config 3: roberta-base-shaped news encoder, d=768, 16 heads, L=96)
on one GPU with a RANDOM-INIT body of roberta-base's shape (no network for the checkpoint): HF transformer on
PyTorch-ROCm (third-party body) + the encoder tail, user encoder, scorer, loss through the C ABI, fused Adam.
Layers 0-7 frozen as in configs/model/nrms_plm*.yaml.  Prints the split body vs tail from HIP events."""
import argparse
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--len", type=int, default=96)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--layers", type=int, default=12)
    args = ap.parse_args()
    from functools import partial

    from transformers import RobertaConfig, RobertaModel

    from newsreclib_amd.nrms_module import NRMSModule, prepare_batch
    from newsreclib_amd.synthetic import make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=args.layers, num_attention_heads=12,
                        intermediate_size=3072, max_position_embeddings=514, type_vocab_size=1, pad_token_id=1,
                        bos_token_id=0, eos_token_id=2)
    tmp = tempfile.mkdtemp()
    RobertaModel(cfg, add_pooling_layer=False).save_pretrained(tmp)
    mod = NRMSModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title"],
        outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
        loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=True, pretrained_embeddings_path=None,
        plm_model=tmp, frozen_layers=list(range(8)), embed_dim=768, num_heads=16, query_dim=200,
        dropout_probability=0.2, top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False,
        recs_fpath=None, optimizer=partial(torch.optim.Adam, lr=1e-5), scheduler=None).cuda()
    trainer = NRMSTrainer(mod, lr=1e-5)
    b = make_batch(args.batch, vocab=50000, mode="fixed", seed=1, L=args.len, device="cuda")
    for part in ("x_hist", "x_cand"):          # tokenizer-style inputs (rec_dataset.py:180-190)
        ids = b[part]["title"].clamp_min(3)
        b[part]["title"] = {"input_ids": ids, "attention_mask": torch.ones_like(ids)}
    batch = prepare_batch(b)
    n_news = sum(batch[p]["title"]["input_ids"].shape[0] for p in ("x_hist", "x_cand"))
    trainer.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"nrms-plm B={args.batch} ({n_news} news x {args.len} tokens, roberta-base-shaped random body): "
          f"{dt * 1e3:.0f} ms/step, {args.batch / dt:.1f} impressions/s, loss={float(loss):.4f}")


if __name__ == "__main__":
    main()
