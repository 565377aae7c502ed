#!/usr/bin/env python3
"""Generate tests/golden/naml_plm_tiny.npz: the REFERENCE's own components wired as ``NAMLModule`` wires them with
``use_plm=True`` (naml_module.py:149-207) -- ``PLM`` text encoder (text.py:15-109, one instance shared by title and abstract)
over a tiny random-init roberta-shaped body, ``LinearEncoder(linear_transform=True, output_dim=text_embed_dim)``,
``NewsEncoder(combine_type="add_att")``, NAML ``UserEncoder``, ``DotProduct`` -- and run the reference's way
(naml_module.py:261-286: news_encoder(x_hist), to_dense_batch, news_encoder(x_cand), user encoder, scorer, CE).
``NAMLModule`` itself needs lightning / torch_geometric / torchmetrics, so its wiring is restated around the imported
components (same rules as make_golden_naml.py).  The fixture holds inputs, the parameters that are not re-creatable from a
seed, and reference outputs only; the body comes from tests/helpers.make_tiny_roberta (portable numpy rng).

Usage:  python tests/golden/make_golden_naml_plm.py   (from the repo root; BUILD container only)
"""
import importlib.util
import os
import sys
import tempfile

if os.environ.get("PYTHONHASHSEED") != "0":      # the reference fills its text-encoder ModuleDict from a Python set
    os.execvpe(sys.executable, [sys.executable] + sys.argv, dict(os.environ, PYTHONHASHSEED="0"))

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

from newsreclib.models.components.encoders.news.category import LinearEncoder  # noqa: E402
from newsreclib.models.components.encoders.news.news import NewsEncoder  # noqa: E402
from newsreclib.models.components.encoders.news.text import PLM  # noqa: E402
from newsreclib.models.components.encoders.user.naml import UserEncoder  # noqa: E402
from newsreclib.models.components.layers.click_predictor import DotProduct  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("nrl_test_helpers", os.path.join(REPO, "tests", "helpers.py"))
th = importlib.util.module_from_spec(spec)          # (the reference ships its own tests/helpers package)
spec.loader.exec_module(th)

DIM, CATEG_DIM, N_CATEG = 96, 16, 7


class RefNAMLPLM(torch.nn.Module):
    def __init__(self, plm_path):
        super().__init__()
        text_encoder = PLM(plm_model=plm_path, frozen_layers=[0], embed_dim=DIM, use_mhsa=True, apply_reduce_dim=False,
                           reduced_embed_dim=None, num_heads=th.PLM_HEADS, query_dim=th.PLM_Q, dropout_probability=0.2)
        category_encoder = LinearEncoder(pretrained_embeddings=None, from_pretrained=False, freeze_pretrained_emb=False,
                                         num_categories=N_CATEG, embed_dim=CATEG_DIM, use_dropout=False, dropout_probability=None,
                                         linear_transform=True, output_dim=DIM)
        self.news_encoder = NewsEncoder(
            dataset_attributes=["title", "abstract", "category"], attributes2encode=["title", "abstract", "category"],
            concatenate_inputs=False, text_encoder=text_encoder, category_encoder=category_encoder, entity_encoder=None,
            combine_vectors=True, combine_type="add_att", input_dim=DIM, query_dim=th.PLM_Q, output_dim=None)
        self.user_encoder = UserEncoder(news_embed_dim=DIM, query_dim=th.PLM_Q)
        self.click_predictor = DotProduct()
        self.criterion = torch.nn.CrossEntropyLoss()
        self.text_order = list(self.news_encoder.text_encoders.keys())


def dense_batch_loops(x, batch, B):
    counts = [int((batch == b).sum()) for b in range(B)]
    mx = max(counts)
    rows, start = [], 0
    for b in range(B):
        r = x.new_zeros((mx,) + tuple(x.shape[1:]))
        if counts[b]:
            r[: counts[b]] = x[start:start + counts[b]]
        rows.append(r)
        start += counts[b]
    return torch.stack(rows)


def toks(rng, n, L):
    ids = rng.integers(3, 200, (n, L))
    lens = rng.integers(3, L + 1, n)
    m = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    return {"input_ids": torch.from_numpy(np.where(m == 1, ids, 1)), "attention_mask": torch.from_numpy(m)}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tmp = tempfile.mkdtemp()
    th.make_tiny_roberta(tmp)
    model = RefNAMLPLM(tmp)
    # parameters outside the body: the PLM tail from tests/helpers.make_plm_tail_params (as plm_tiny.npz), the rest drawn here
    # and stored in the fixture
    sd = model.state_dict()
    tail = th.make_plm_tail_params()
    rng = np.random.default_rng(77)
    new, stored = {}, {}
    for k in sorted(sd.keys()):
        if ".plm_model." in k:
            continue
        hit = [t for t in tail if k.endswith("text_encoders.title." + t) or k.endswith("text_encoders.abstract." + t)]
        if hit:
            new[k] = tail[hit[0]].clone()
            continue
        v = sd[k]
        scale = 0.3 if v.dim() == 1 else (1.5 / np.sqrt(v.shape[-1]))
        new[k] = torch.from_numpy((scale * rng.standard_normal(tuple(v.shape))).astype(np.float32))
        if k.endswith("embedding_layer.weight") and "categ" in k:
            new[k][0] = 0.0                                   # padding_idx row
        stored[k] = new[k].numpy()
    res = model.load_state_dict(new, strict=False)
    assert not res.unexpected_keys and all(".plm_model." in k for k in res.missing_keys), res
    model.eval()                                              # dropout off: the body's HF dropouts are 0 in this config anyway

    hist_sizes, cand_sizes = [2, 3, 1, 4], [5, 10, 5, 5]
    nh, nc = sum(hist_sizes), sum(cand_sizes)
    rng = np.random.default_rng(5)
    x_hist = {"title": toks(rng, nh, 9), "abstract": toks(rng, nh, 13), "category": torch.from_numpy(rng.integers(1, N_CATEG, nh))}
    x_cand = {"title": toks(rng, nc, 10), "abstract": toks(rng, nc, 12), "category": torch.from_numpy(rng.integers(1, N_CATEG, nc))}
    B = len(hist_sizes)
    batch_hist = torch.repeat_interleave(torch.arange(B), torch.tensor(hist_sizes))
    batch_cand = torch.repeat_interleave(torch.arange(B), torch.tensor(cand_sizes))
    labels = torch.zeros(nc)
    start = 0
    for c in cand_sizes:
        labels[start + int(rng.integers(0, c))] = 1.0
        start += c

    hist_vec = model.news_encoder(x_hist)
    cand_vec = model.news_encoder(x_cand)
    hist_dense = dense_batch_loops(hist_vec, batch_hist, B)
    cand_dense = dense_batch_loops(cand_vec, batch_cand, B)
    user = model.user_encoder(hist_dense)
    scores = model.click_predictor(user.unsqueeze(dim=1), cand_dense.permute(0, 2, 1))
    # (a padded candidate slot scores 0 against the zero vector, as in the reference's dense batch)
    y_true = dense_batch_loops(labels, batch_cand, B)
    loss = model.criterion(scores, y_true)
    loss.backward()

    arrays = {"in_batch_hist": batch_hist.numpy(), "in_batch_cand": batch_cand.numpy(), "in_labels": labels.numpy(),
              "cfg_text_order": np.array(model.text_order), "cfg_dim": np.int64(DIM), "cfg_categ_dim": np.int64(CATEG_DIM),
              "cfg_n_categ": np.int64(N_CATEG)}
    for part, x in (("hist", x_hist), ("cand", x_cand)):
        for a in ("title", "abstract"):
            arrays[f"in_{a}_{part}_input_ids"] = x[a]["input_ids"].numpy()
            arrays[f"in_{a}_{part}_attention_mask"] = x[a]["attention_mask"].numpy()
        arrays[f"in_category_{part}"] = x["category"].numpy()
    for k, v in stored.items():
        arrays["param/" + k] = v
    for k, v in (("hist_vec", hist_vec), ("cand_vec", cand_vec), ("user_vec", user), ("scores", scores), ("y_true", y_true),
                 ("loss", loss)):
        arrays["out_" + k] = v.detach().numpy()
    named = dict(model.named_parameters())                    # (shared text encoder: listed once, under its first name)
    for k, p in named.items():
        if p.grad is None:
            continue
        if ".plm_model." in k:
            arrays["gnorm/" + k] = np.float64(p.grad.double().norm())
        else:
            arrays["gfull/" + k] = p.grad.detach().numpy()
    path = os.path.join(OUT, "naml_plm_tiny.npz")
    np.savez_compressed(path, **arrays)
    print(f"naml_plm_tiny: order={model.text_order} loss={float(loss.detach()):.6f} scores {tuple(scores.shape)} -> "
          f"{os.path.getsize(path) / 1024:.1f} KiB; body grads {sum(1 for k in arrays if k.startswith('gnorm/'))}, "
          f"other grads {sum(1 for k in arrays if k.startswith('gfull/'))}")


if __name__ == "__main__":
    main()
