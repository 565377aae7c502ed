#!/usr/bin/env python3
"""Generate tests/golden/plm_full.npz: BASELINE configs[3] END TO END AT FULL WIDTH from the REFERENCE's own components.

Runs only in the build container (needs the read-only checkout at /root/reference); the fixture holds seeds, inputs'
shapes and the reference's outputs -- no reference source text, and no weights (the roberta-base-SHAPED random body,
d = 768 / 12 layers / 12 heads / 3072, is regenerated on both sides from ``numpy.random.default_rng`` seeds by
``tests/helpers.make_full_roberta``; there is no network for the pretrained checkpoint).

What is imported from the reference: ``PLM`` (text.py:15-109; frozen layers 0-7 as the experiment file has them, 16 tail heads,
L = 96), the NRMS ``UserEncoder`` (user/nrms.py:7-41) and ``DotProduct`` (click_predictor.py:5-11).  ``NRMSModule`` itself cannot
be imported here (lightning / torch_geometric / torchmetrics are absent), so its forward glue (nrms_module.py:230-255: TWO encoder
calls, history then candidates; ``to_dense_batch``; the loss line :287-288) is restated with plain loops as in make_golden.py.

Two passes over the same ragged batch (18 history news padded to 96 tokens + 10 candidate news padded to 64, each with its own
padded tail): evaluation mode, and
train mode with the tail's two ``nn.Dropout`` calls replaced by the product's counter-based masks (known Bernoulli draw; the
body's own dropouts are 0 in the config).  The modules run TWICE: as the reference runs them (fp32) and cast to fp64; stored are the
fp64 results -- news vectors of both calls, user vectors, scores, loss, and for every trainable parameter its gradient's L2 norm,
largest magnitude and a strided sample -- plus, per quantity, the distance of the reference's own fp32 run from them (`noise`).

Usage:  python tests/golden/make_golden_plm_full.py   (from the repo root; ~5 min on 8 cores)
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from newsreclib.models.components.encoders.news.text import PLM  # noqa: E402
from newsreclib.models.components.encoders.user.nrms import UserEncoder  # noqa: E402
from newsreclib.models.components.layers.click_predictor import DotProduct  # noqa: E402

from oracle.nrms_oracle import dropout_multiplier  # noqa: E402

spec = importlib.util.spec_from_file_location("nrl_test_helpers", os.path.join(REPO, "tests", "helpers.py"))
th = importlib.util.module_from_spec(spec)          # (the reference ships its own tests/helpers package)
spec.loader.exec_module(th)

OUT = os.path.dirname(os.path.abspath(__file__))
D, P_DROP = 768, 0.2
SEED_HIST, SEED_CAND = 1001, 2002
N_SAMPLE = 384


class InjectedDropout(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.mults, self.k = [], 0

    def arm(self, mults):
        self.mults, self.k = list(mults), 0

    def forward(self, x):
        if not self.mults:
            return x
        m = self.mults[self.k]
        self.k += 1
        return x * m


def dense_batch_loops(x, batch, B):
    counts = [int((batch == b).sum()) for b in range(B)]
    out = x.new_zeros((B, max(counts)) + tuple(x.shape[1:]))
    rows, start = [], 0
    for b in range(B):
        rows.append(out[b].clone())
        rows[b][: counts[b]] = x[start:start + counts[b]]
        start += counts[b]
    return torch.stack(rows)


def sample_idx(n):
    return np.unique(np.linspace(0, n - 1, min(n, N_SAMPLE)).astype(np.int64))


def run_reference(tmp, dtype, batch, arrays=None):
    """The reference components in `dtype` over the batch, evaluation then train mode -> {tag: (outputs, {key: gradient})}."""
    enc = PLM(plm_model=tmp, frozen_layers=th.PLM_FULL_FROZEN, embed_dim=D, use_mhsa=True, apply_reduce_dim=False,
              reduced_embed_dim=None, num_heads=th.PLM_FULL_HEADS, query_dim=th.PLM_FULL_Q, dropout_probability=P_DROP)
    tail = th.make_plm_tail_params(dim=D, query_dim=th.PLM_FULL_Q, seed=23, out_scale=th.PLM_FULL_OUT_SCALE)
    missing = enc.load_state_dict(tail, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("plm_model.") for k in missing.missing_keys)
    user = UserEncoder(news_embed_dim=D, num_heads=th.PLM_FULL_HEADS, query_dim=th.PLM_FULL_Q)
    utail = th.make_plm_tail_params(dim=D, query_dim=th.PLM_FULL_Q, seed=29, out_scale=th.PLM_FULL_OUT_SCALE)
    user.load_state_dict(utail, strict=True)
    enc, user = enc.to(dtype), user.to(dtype)
    click = DotProduct()
    inj = InjectedDropout()
    enc.dropout = inj
    B = batch["batch_size"]
    hist, cand = batch["x_hist"]["title"], batch["x_cand"]["title"]
    nh, nc = hist["input_ids"].shape[0], cand["input_ids"].shape[0]
    named = [("news_encoder.text_encoders.title." + k, p) for k, p in enc.named_parameters() if p.requires_grad]
    named += [("user_encoder." + k, p) for k, p in user.named_parameters()]
    res = {}
    for tag in ("eval", "train"):
        enc.train(tag == "train")
        user.train(tag == "train")
        enc.zero_grad()
        user.zero_grad()
        for call, (text, n, seed) in enumerate(((hist, nh, SEED_HIST), (cand, nc, SEED_CAND))):
            L = text["input_ids"].shape[1]
            if tag == "train":
                inj.arm([dropout_multiplier(seed, 0, P_DROP, (n, L, D)).to(dtype), dropout_multiplier(seed, 1, P_DROP, (n, L, D)).to(dtype)])
            else:
                inj.arm([])
            vec = enc(text)                                    # (two calls: nrms_module.py:232,236)
            if call == 0:
                hist_vec = vec
            else:
                cand_vec = vec
        hist_dense = dense_batch_loops(hist_vec, batch["batch_hist"], B)
        cand_dense = dense_batch_loops(cand_vec, batch["batch_cand"], B)
        uvec = user(hist_dense)
        scores = click(uvec.unsqueeze(dim=1), cand_dense.permute(0, 2, 1))
        y_true = dense_batch_loops(batch["labels"].to(dtype), batch["batch_cand"], B)
        loss = torch.nn.CrossEntropyLoss()(scores, y_true)
        loss.backward()
        outs = {k: v.detach().double() for k, v in (("hist_vec", hist_vec), ("cand_vec", cand_vec), ("user_vec", uvec),
                                                    ("scores", scores), ("loss", loss))}
        grads = {}
        for k, p in named:
            if p.grad is None:          # (AutoModel adds a pooler the saved body does not have: initialised, never used -- text.py:91 reads [0])
                assert ".pooler." in k, k
                continue
            grads[k] = p.grad.detach().double().reshape(-1).clone()
        res[tag] = (outs, grads)
        print(str(dtype), tag, "loss", float(loss.detach()), "|scores|max", float(scores.abs().max()), "|hist_vec|max", float(hist_vec.abs().max()))
    frozen = [k for k, p in enc.named_parameters() if not p.requires_grad]
    return res, len(frozen)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp()
    th.make_full_roberta(tmp)
    batch = th.plm_full_inputs()
    # the reference as it runs (fp32) and the same modules in fp64: the fixture stores the fp64 results as the truth and, per
    # quantity, how far the reference's OWN fp32 run is from them (`noise`: what fp32 rounding does to this quantity)
    r32, n_frozen = run_reference(tmp, torch.float32, batch)
    r64, _ = run_reference(tmp, torch.float64, batch)
    arrays = {"cfg_p_drop": np.float64(P_DROP), "cfg_seed_hist": np.int64(SEED_HIST), "cfg_seed_cand": np.int64(SEED_CAND),
              "cfg_n_sample": np.int64(N_SAMPLE), "cfg_n_frozen": np.int64(n_frozen)}
    for tag in ("eval", "train"):
        o32, g32 = r32[tag]
        o64, g64 = r64[tag]
        for k, v in o64.items():
            arrays[f"out_{tag}/{k}"] = v.numpy().astype(np.float32)
            arrays[f"outnoise_{tag}/{k}"] = np.float64((o32[k] - v).abs().max())
        worst = (0.0, None)
        for k, g in g64.items():
            idx = torch.from_numpy(sample_idx(g.numel()))
            arrays[f"gnorm_{tag}/{k}"] = np.float64(g.norm())
            arrays[f"gmax_{tag}/{k}"] = np.float64(g.abs().max())
            arrays[f"gsample_{tag}/{k}"] = g[idx].numpy().astype(np.float32)
            noise = float((g32[k] - g).abs().max())
            arrays[f"gnoise_{tag}/{k}"] = np.float64(noise)
            rel = noise / max(1e-30, float(g.abs().max()))
            if rel > worst[0] and not k.endswith("key.bias"):
                worst = (rel, k)
        print(tag, "reference fp32 vs fp64: outputs", {k: float(arrays[f"outnoise_{tag}/{k}"]) for k in o64},
              "worst gradient (of the parameter's largest entry, key biases aside): %.2e %s" % worst)
    np.savez_compressed(os.path.join(OUT, "plm_full.npz"), **arrays)
    print("plm_full:", len(r64["eval"][1]), "trainable parameters summarised,", n_frozen, "frozen")


if __name__ == "__main__":
    main()
