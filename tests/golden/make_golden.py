#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own components.

Runs only in the build container (needs the read-only checkout at /root/reference); the GPU box
never sees the reference, only the arrays written here.  The fixtures contain inputs, seeds and
the reference's outputs -- no reference source text.

What is imported from the reference: ``MHSAAddAtt`` (text.py:179-236),
``NewsEncoder`` (news.py:9-183), NRMS ``UserEncoder`` (user/nrms.py:7-41), ``DotProduct``
(click_predictor.py:5-11).  ``NRMSModule`` itself cannot be imported here (lightning,
torch_geometric, torchmetrics are not installed), so its 25-line forward glue
(nrms_module.py:230-255), the loss line (:287-288), ``to_dense_batch`` (torch_geometric 2.3.0) and
``torch.optim.Adam`` wiring (abstract_recommender.py:96) are restated below with plain loops.

Train-mode cases: ``nn.Dropout`` draws from torch's RNG, which the HIP path cannot reproduce.  The
product defines its own counter-based keep mask (oracle.nrms_oracle.dropout_keep_mask is the
normative numpy statement); here the reference's ``dropout`` sub-module is swapped for a module
that multiplies by that mask * 1/(1-p), i.e. the reference forward with a known Bernoulli draw.

Usage:  python tests/golden/make_golden.py   (from the repo root)
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from newsreclib.models.components.encoders.news.news import NewsEncoder  # noqa: E402
from newsreclib.models.components.encoders.news.text import PLM, MHSAAddAtt  # noqa: E402
from newsreclib.models.components.encoders.user.nrms import UserEncoder  # noqa: E402
from newsreclib.models.components.layers.click_predictor import DotProduct  # noqa: E402

from newsreclib_amd.synthetic import batch_from_sizes, make_batch  # noqa: E402
from oracle.nrms_oracle import EMB_KEY, dropout_multiplier, make_params  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
D, HEADS, Q = 300, 15, 200
SAMPLE_STRIDE = 97
ROW_STRIDE = 9


class InjectedDropout(torch.nn.Module):
    """Stands in for ``nn.Dropout``: call k multiplies by ``mults[k]`` (identity when empty)."""

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
        if m.shape != x.shape:          # second dropout sees (L, N, D)
            m = m.permute(1, 0, 2)
        return x * m


class RefNRMS(torch.nn.Module):
    """Reference components under the reference's attribute names (=> reference state_dict keys)."""

    def __init__(self, params, p_drop=0.2):
        super().__init__()
        emb = params[EMB_KEY].numpy()
        text_encoder = MHSAAddAtt(pretrained_embeddings=emb, embed_dim=D, num_heads=HEADS,
                                  query_dim=Q, dropout_probability=float(p_drop))
        self.news_encoder = NewsEncoder(
            dataset_attributes=["title", "abstract", "category"], attributes2encode=["title"],
            concatenate_inputs=False, text_encoder=text_encoder, category_encoder=None,
            entity_encoder=None, combine_vectors=False, combine_type=None, input_dim=None,
            query_dim=None, output_dim=None)
        self.user_encoder = UserEncoder(news_embed_dim=D, num_heads=HEADS, query_dim=Q)
        self.click_predictor = DotProduct()
        missing = self.load_state_dict(params, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        self.inj = InjectedDropout()
        text_encoder.dropout = self.inj
        self.criterion = torch.nn.CrossEntropyLoss()


def dense_batch_loops(x, batch, B):
    """to_dense_batch restated with plain loops (third-party; call sites nrms_module.py:233,237)."""
    counts = [int((batch == b).sum()) for b in range(B)]
    mx = max(counts)
    out = x.new_zeros((B, mx) + tuple(x.shape[1:]))
    mask = torch.zeros(B, mx, dtype=torch.bool)
    rows = []
    start = 0
    for b in range(B):
        rows.append(out[b].clone())
        if counts[b]:
            rows[b][: counts[b]] = x[start:start + counts[b]]
            mask[b, : counts[b]] = True
        start += counts[b]
    return torch.stack(rows), mask


def ref_forward(model, batch, p_drop=0.0, seed=0, late_fusion=False):
    """nrms_module.py:230-255 + :277,287-288 glued around the imported components."""
    B = batch["batch_size"]
    ids_h, ids_c = batch["x_hist"]["title"], batch["x_cand"]["title"]
    nh, nc, L = ids_h.shape[0], ids_c.shape[0], ids_h.shape[1]
    if p_drop > 0:
        m1 = dropout_multiplier(seed, 0, p_drop, (nh + nc, L, D))
        m2 = dropout_multiplier(seed, 1, p_drop, (nh + nc, L, D))
        model.inj.arm([m1[:nh], m2[:nh], m1[nh:], m2[nh:]])
    else:
        model.inj.arm([])
    hist_vec = model.news_encoder({"title": ids_h})
    hist_dense, mask_hist = dense_batch_loops(hist_vec, batch["batch_hist"], B)
    cand_vec = model.news_encoder({"title": ids_c})
    cand_dense, mask_cand = dense_batch_loops(cand_vec, batch["batch_cand"], B)
    if not late_fusion:
        user = model.user_encoder(hist_dense)
    else:   # nrms_module.py:243-248
        hist_size = torch.tensor([torch.where(mask_hist[i])[0].shape[0] for i in range(mask_hist.shape[0])])
        user = torch.div(hist_dense.sum(dim=1), hist_size.unsqueeze(dim=-1))
    scores = model.click_predictor(user.unsqueeze(dim=1), cand_dense.permute(0, 2, 1))
    y_true, _ = dense_batch_loops(batch["labels"], batch["batch_cand"], B)
    loss = model.criterion(scores, y_true)
    return dict(hist_vec=hist_vec, cand_vec=cand_vec, user_vec=user, scores=scores,
                y_true=y_true, loss=loss, hist_dense=hist_dense, cand_dense=cand_dense)


def batch_arrays(batch):
    return {
        "in_ids_hist": batch["x_hist"]["title"].numpy(), "in_ids_cand": batch["x_cand"]["title"].numpy(),
        "in_batch_hist": batch["batch_hist"].numpy(), "in_batch_cand": batch["batch_cand"].numpy(),
        "in_labels": batch["labels"].numpy(), "in_batch_size": np.int64(batch["batch_size"]),
    }


def grad_summary(model, full_embedding):
    out = {}
    for k, p in model.state_dict(keep_vars=True).items():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        flat = g.detach().reshape(-1).double()
        out["gnorm/" + k] = np.float64(flat.norm())
        out["gsum/" + k] = np.float64(flat.sum())
        if k == EMB_KEY and not full_embedding:
            rows = torch.nonzero(g.abs().sum(1) > 0).reshape(-1)[:8]
            out["grows_idx/" + k] = rows.numpy()
            out["grows/" + k] = g[rows].detach().numpy()
        elif k == EMB_KEY:
            out["gfull/" + k] = g.detach().numpy()
        else:
            out["gsample/" + k] = g.detach().reshape(-1)[::SAMPLE_STRIDE].numpy().copy()
    return out


def run_case(name, batch, vocab, param_seed, p_drop=0.0, seed=0, full_embedding=False, extra=None,
             late_fusion=False):
    params = make_params(vocab, D, Q, seed=param_seed)
    model = RefNRMS(params)
    model.train()                      # dropout handled by injection; nothing else is mode-dependent
    out = ref_forward(model, batch, p_drop, seed, late_fusion)
    out["loss"].backward()
    arrays = batch_arrays(batch)
    arrays.update(cfg_vocab=np.int64(vocab), cfg_param_seed=np.int64(param_seed),
                  cfg_p_drop=np.float64(p_drop), cfg_seed=np.int64(seed),
                  cfg_sample_stride=np.int64(SAMPLE_STRIDE))
    for k in ("user_vec", "scores", "y_true", "loss"):
        arrays["out_" + k] = out[k].detach().numpy()
    rs = 1 if full_embedding else ROW_STRIDE      # big cases keep every ROW_STRIDE-th news vector
    arrays["cfg_row_stride"] = np.int64(rs)
    for k in ("hist_vec", "cand_vec"):
        arrays["out_" + k] = out[k].detach().numpy()[::rs].copy()
    arrays.update(grad_summary(model, full_embedding))
    if extra:
        arrays.update(extra)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: loss={float(out['loss'].detach()):.6f} scores{tuple(out['scores'].shape)} "
          f"-> {os.path.getsize(path) / 1024:.1f} KiB")
    return model, out


def tiny_batch():
    # hist sizes (1, 4, 2); cand sizes (5, 10, 5); user 1 has two positives (multi-positive CE)
    labels = [0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1]
    return batch_from_sizes([1, 4, 2], [5, 10, 5], labels, vocab=64, seed=11)


def case_quirks():
    """Evidence for the reproduced reference quirks (SURVEY.md headline facts 3 and 4)."""
    params = make_params(64, D, Q, seed=3)
    model = RefNRMS(params)
    full = tiny_batch()
    out_full = ref_forward(model, full)
    # same users 0 and 1, but user 2 removed from the batch: seq-first user MHA => scores change
    sub = batch_from_sizes([1, 4], [5, 10], full["labels"][:15].numpy(), vocab=64, seed=11)
    sub["x_hist"]["title"] = full["x_hist"]["title"][:5].clone()
    sub["x_cand"]["title"] = full["x_cand"]["title"][:15].clone()
    out_sub = ref_forward(model, sub)
    # pad-token influence: change embedding row 0 only
    params2 = {k: v.clone() for k, v in params.items()}
    params2[EMB_KEY][0] += 1.0
    out_pad = ref_forward(RefNRMS(params2), full)
    arrays = batch_arrays(full)
    arrays.update(cfg_vocab=np.int64(64), cfg_param_seed=np.int64(3),
                  out_scores_full=out_full["scores"].detach().numpy(),
                  out_scores_sub=out_sub["scores"].detach().numpy(),
                  out_scores_pad_row_changed=out_pad["scores"].detach().numpy(),
                  out_hist_vec_full=out_full["hist_vec"].detach().numpy(),
                  out_hist_vec_pad_row_changed=out_pad["hist_vec"].detach().numpy())
    np.savez_compressed(os.path.join(OUT, "quirks.npz"), **arrays)
    d_sub = (out_full["scores"][:2, :] - torch.nn.functional.pad(
        out_sub["scores"], (0, out_full["scores"].shape[1] - out_sub["scores"].shape[1])))[:, :5].abs().max()
    print(f"quirks: |scores(full)-scores(sub)|max={float(d_sub):.4f} "
          f"|hist_vec pad-row delta|max={float((out_full['hist_vec'] - out_pad['hist_vec']).abs().max()):.4f}")


def case_adam(n_steps=3, lr=1e-4):
    """n optimizer steps with torch.optim.Adam (abstract_recommender.py:96), dropout off."""
    batch = tiny_batch()
    params = make_params(64, D, Q, seed=5)
    model = RefNRMS(params)
    opt = torch.optim.Adam(params=model.parameters(), lr=lr)
    losses = []
    for _ in range(n_steps):
        opt.zero_grad()
        out = ref_forward(model, batch)
        out["loss"].backward()
        opt.step()
        losses.append(float(out["loss"]))
    arrays = batch_arrays(batch)
    arrays.update(cfg_vocab=np.int64(64), cfg_param_seed=np.int64(5), cfg_lr=np.float64(lr),
                  cfg_steps=np.int64(n_steps), out_losses=np.asarray(losses, dtype=np.float64),
                  cfg_sample_stride=np.int64(SAMPLE_STRIDE))
    for k, p in model.state_dict().items():
        arrays["psum/" + k] = np.float64(p.double().sum())
        arrays["pnorm/" + k] = np.float64(p.double().norm())
        arrays["psample/" + k] = p.reshape(-1)[::SAMPLE_STRIDE].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "adam3.npz"), **arrays)
    print("adam3: losses", losses)


def case_plm():
    """Config-4 path: the reference ``PLM`` text encoder (text.py:15-109) over a tiny random-init
    roberta-shaped body (no network for roberta-base).  Eval forward + a train forward with injected
    dropout masks, and gradients of the encoder tail."""
    import tempfile

    import importlib.util
    spec = importlib.util.spec_from_file_location("nrl_test_helpers", os.path.join(REPO, "tests", "helpers.py"))
    th = importlib.util.module_from_spec(spec)      # (the reference ships its own tests/helpers package)
    spec.loader.exec_module(th)
    PLM_HEADS, PLM_Q, make_plm_tail_params, make_tiny_roberta = (th.PLM_HEADS, th.PLM_Q, th.make_plm_tail_params,
                                                                  th.make_tiny_roberta)
    tmp = tempfile.mkdtemp()
    make_tiny_roberta(tmp)
    enc = PLM(plm_model=tmp, frozen_layers=[0], embed_dim=96, use_mhsa=True, apply_reduce_dim=False,
              reduced_embed_dim=None, num_heads=PLM_HEADS, query_dim=PLM_Q, dropout_probability=0.2)
    tail = make_plm_tail_params()
    missing = enc.load_state_dict(tail, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("plm_model.") for k in missing.missing_keys)
    inj = InjectedDropout()
    enc.dropout = inj
    rng = np.random.default_rng(31)
    N, L = 7, 12
    ids = rng.integers(3, 200, (N, L))
    lens = rng.integers(4, L + 1, N)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    ids = np.where(mask == 1, ids, 1)
    text = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    arrays = {"in_input_ids": ids, "in_attention_mask": mask}
    d_out = torch.from_numpy(rng.standard_normal((N, 96)).astype(np.float32))
    arrays["in_d_out"] = d_out.numpy()
    for tag, p_drop, seed in (("eval", 0.0, 0), ("train", 0.2, 5)):
        enc.zero_grad()
        if p_drop > 0:
            inj.arm([dropout_multiplier(seed, 0, p_drop, (N, L, 96)), dropout_multiplier(seed, 1, p_drop, (N, L, 96))])
        else:
            inj.arm([])
        out = enc(text)
        (out * d_out).sum().backward()
        arrays[f"out_{tag}"] = out.detach().numpy()
        for k in tail:
            obj = enc
            for part in k.split("."):
                obj = getattr(obj, part)
            arrays[f"grad_{tag}/{k}"] = obj.grad.detach().numpy().copy()
        emb_g = enc.plm_model.embeddings.word_embeddings.weight.grad
        arrays[f"grad_{tag}/plm_word_embeddings_norm"] = np.float64(emb_g.double().norm())
        arrays[f"cfg_{tag}_p_drop"], arrays[f"cfg_{tag}_seed"] = np.float64(p_drop), np.int64(seed)
    with torch.no_grad():
        arrays["out_hidden"] = enc.plm_model(**text)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "plm_tiny.npz"), **arrays)
    print("plm_tiny: out", arrays["out_eval"].shape, "|out|max", float(np.abs(arrays["out_eval"]).max()))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run_case("tiny_eval", tiny_batch(), vocab=64, param_seed=1, full_embedding=True)
    run_case("tiny_train", tiny_batch(), vocab=64, param_seed=1, p_drop=0.2, seed=7, full_embedding=True)
    b32 = make_batch(32, vocab=2000, mode="ragged", seed=21)
    run_case("mind32_eval", b32, vocab=2000, param_seed=2)
    run_case("mind32_train", b32, vocab=2000, param_seed=2, p_drop=0.2, seed=99)
    run_case("tiny_late_fusion", tiny_batch(), vocab=64, param_seed=1, p_drop=0.2, seed=7, full_embedding=True,
             late_fusion=True)
    case_quirks()
    case_adam()
    case_plm()


if __name__ == "__main__":
    main()
