"""BASELINE configs[3] END TO END AT FULL WIDTH against the reference.

The fixture ``tests/golden/plm_full.npz`` was produced in the build container by the REFERENCE's own ``PLM`` (text.py:15-109),
NRMS ``UserEncoder`` and ``DotProduct`` on the host (cast to fp64; the distance of their fp32 run is stored beside it) over a roberta-base-SHAPED random body (d = 768, 12 layers, 12 body
heads, feed-forward 3072, layers 0-7 frozen, 16 tail heads, query 200) -- ``tests/golden/make_golden_plm_full.py``.  Here the
product module (every body projection, attention, feed-forward, embedding and layer-norm block on this library's kernels, then the
HIP tail, dense batching, user encoder, scorer and loss) runs the same ragged batch: 18 history news padded to 96 tokens, 10
candidate news padded to 64 (each call padded to its own longest text, as the reference's collate does), padded tails inside.

Contract: news vectors and scores within 1e-3 (|scores| <= 4.2 in the fixture); every trainable parameter's gradient -- the tail,
the user encoder, the unfrozen layers 8-11 AND the body's embedding tables, which the reference leaves trainable (text.py:70-73) --
within 2e-4 of the parameter's largest gradient entry on a strided sample (+ 128 x the reference's own fp32-vs-fp64 distance for
that parameter, where it is ill-conditioned; never above 5e-3), norms within 1e-3.  One body pass over both calls
(``share_body``, with the shorter call padded) and two separate passes must both meet it, and the framework-fallback counters
stay 0."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import (PLM_FULL_FROZEN, PLM_FULL_HEADS, PLM_FULL_OUT_SCALE, PLM_FULL_Q, batch_to, load_golden,
                           make_full_roberta, make_plm_tail_params, plm_full_inputs)

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-3


@pytest.fixture(scope="module")
def full_module(tmp_path_factory):
    from functools import partial

    from newsreclib_amd import _lib
    from newsreclib_amd.nrms_module import NRMSModule
    _lib.set_gemm_engine("bf16x3")
    path = make_full_roberta(str(tmp_path_factory.mktemp("roberta_full")))
    mod = NRMSModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title"],
        outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
        loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=True, pretrained_embeddings_path=None,
        plm_model=path, frozen_layers=PLM_FULL_FROZEN, embed_dim=768, num_heads=PLM_FULL_HEADS, query_dim=PLM_FULL_Q,
        dropout_probability=0.2, top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False,
        recs_fpath=None, optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None)
    tail = make_plm_tail_params(dim=768, query_dim=PLM_FULL_Q, seed=23, out_scale=PLM_FULL_OUT_SCALE)
    utail = make_plm_tail_params(dim=768, query_dim=PLM_FULL_Q, seed=29, out_scale=PLM_FULL_OUT_SCALE)
    sd = {"news_encoder.text_encoders.title." + k: v for k, v in tail.items()}
    sd.update({"user_encoder." + k: v for k, v in utail.items()})
    assert not mod.load_state_dict(sd, strict=False).unexpected_keys
    return mod.to(DEV)


def _sample_idx(n, n_sample):
    return np.unique(np.linspace(0, n - 1, min(n, n_sample)).astype(np.int64))


@pytest.mark.parametrize("share", [True, False], ids=["one_body_pass_padded", "two_body_passes"])
@pytest.mark.parametrize("tag", ["eval", "train"])
def test_plm_full_width_matches_the_reference_end_to_end(full_module, tag, share):
    from newsreclib_amd import _lib
    from newsreclib_amd import news_encoder as ne
    from newsreclib_amd import ops
    from newsreclib_amd.dense_batch import dense_rows
    from newsreclib_amd.nrms_module import prepare_batch
    _lib.set_gemm_engine("bf16x3")
    g = load_golden("plm_full")
    mod = full_module
    te = mod.news_encoder.text_encoders["title"]
    assert te.nrl_linears == 72 and te.nrl_attention and te.nrl_output_blocks == 24
    assert te.nrl_ffn_blocks == 12 and te.nrl_attention_blocks == 12 and te.nrl_embeddings >= 2
    mod.train(tag == "train")
    mod.zero_grad(set_to_none=True)
    batch = prepare_batch(batch_to(plm_full_inputs(), DEV))
    assert batch["x_hist"]["title"]["input_ids"].shape[1] == 96 and batch["x_cand"]["title"]["input_ids"].shape[1] == 64
    ne.reset_fallback_calls()
    old = os.environ.get("NRL_PLM_SHARE_BODY")
    os.environ["NRL_PLM_SHARE_BODY"] = "1" if share else "0"
    try:
        # nrms_module.py:230-255 as the product module runs it (NRMSModule.forward), with the two calls' dropout seeds pinned
        shared = mod.news_encoder.share_plm_bodies(batch["x_hist"], batch["x_cand"])
        assert shared == (1 if share else 0)
        hist_vec = mod.news_encoder(batch["x_hist"], seed=int(g["cfg_seed_hist"]))
        cand_vec = mod.news_encoder(batch["x_cand"], seed=int(g["cfg_seed_cand"]))
        assert not te._shared                                 # (both calls picked their rows up)
        scores = mod.score_news_vectors(hist_vec, cand_vec, batch)
    finally:
        if old is None:
            del os.environ["NRL_PLM_SHARE_BODY"]
        else:
            os.environ["NRL_PLM_SHARE_BODY"] = old
    assert ne.SHARE_BODY_CALLS == {"hit_same_length": 0, "hit_padded": 1 if share else 0, "miss": 0}
    B = batch["batch_size"]
    y_true = dense_rows(batch["labels"], batch["batch_cand"], B, batch["max_cand"], batch["cand_offsets"], batch["cand_flat_idx"])
    loss = mod._loss(scores, y_true.float(), batch)
    errs = {}
    with torch.no_grad():                                     # (nrms_module.py:241: no dropout in the user encoder, so a second call is the same vector)
        user_vec = mod.user_encoder(dense_rows(hist_vec.detach(), batch["batch_hist"], B, batch["max_hist"], batch["hist_offsets"],
                                               max_is_exact=True))
    for k, v in (("hist_vec", hist_vec), ("cand_vec", cand_vec), ("user_vec", user_vec), ("scores", scores), ("loss", loss)):
        ref = g[f"out_{tag}/{k}"]
        errs[k] = float(np.abs(v.detach().cpu().numpy().reshape(ref.shape) - ref).max())
    print(f"plm full width [{tag}, share={share}]: max abs err " + ", ".join(f"{k} {e:.2e}" for k, e in errs.items()))
    assert all(e <= TOL for e in errs.values()), errs
    loss.backward()
    assert all(v == 0 for v in ne.FALLBACK_CALLS.values()), dict(ne.FALLBACK_CALLS)
    n_sample = int(g["cfg_n_sample"])
    worst, worst_key, checked, bad = 0.0, None, 0, []
    params = dict(mod.named_parameters())
    for key in [k[len(f"gnorm_{tag}/"):] for k in g if k.startswith(f"gnorm_{tag}/")]:
        p = params[key]
        got = getattr(p, "main_grad", None)
        got = got if got is not None else p.grad
        assert got is not None, key
        got = got.detach().reshape(-1)
        ref_s = g[f"gsample_{tag}/{key}"]
        scale = float(g[f"gmax_{tag}/{key}"])
        if key.endswith("attention.self.key.bias"):
            # softmax is invariant to a key bias: the TRUE gradient is zero (fixture: ~1e-18) and both sides hold rounding noise;
            # bound it by the tolerance of the query bias next to it
            scale = float(g[f"gmax_{tag}/{key[:-len('key.bias')]}query.bias"])
            ref_s = np.zeros_like(ref_s)
        idx = torch.from_numpy(_sample_idx(got.numel(), n_sample)).to(DEV)
        d = np.abs(got[idx].cpu().numpy() - ref_s)
        rel = float(d.max()) / scale
        # the bar: 2e-4 of the parameter's largest gradient entry, plus what this engine's 2^-17 products may add where the
        # quantity is ill-conditioned -- 128 x the distance of the reference's OWN fp32 run (2^-24 products) from its fp64 run
        noise = float(g[f"gnoise_{tag}/{key}"]) / scale
        tol = min(2e-4 + 128.0 * noise, 5e-3)
        ref_n = float(g[f"gnorm_{tag}/{key}"])
        nrm = 0.0 if key.endswith("attention.self.key.bias") else abs(float(got.double().norm()) - ref_n) / max(1e-30, ref_n)
        if rel > worst:
            worst, worst_key = rel, key
        if not (rel <= tol and nrm <= 1e-3):
            bad.append((key, rel, tol, nrm, int((d > 2e-4 * scale).sum()), int(d.size)))
        checked += 1
    for b in bad:
        print("  off: %s sample err %.2e of max (bar %.2e), norm err %.2e, %d of %d samples beyond 2e-4" % b)
    assert not bad, bad
    assert checked >= 70                                      # 7 tail + 7 user-encoder + 4 x 16 layer + 5 embedding tensors
    frozen = [k for k, p in te.named_parameters() if not p.requires_grad]
    assert len(frozen) == int(g["cfg_n_frozen"]) and all(p.grad is None for k, p in te.named_parameters() if not p.requires_grad)
    print(f"plm full width [{tag}, share={share}]: {checked} parameter gradients, worst sample error {worst:.2e} of the "
          f"parameter's largest entry ({worst_key})")
