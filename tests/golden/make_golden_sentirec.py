#!/usr/bin/env python3
"""Generate tests/golden/sentirec_*.npz with the REFERENCE's own NRMS components (same rules as make_golden.py,
whose ``RefNRMS`` / ``ref_forward`` -- reference ``MHSAAddAtt``, ``NewsEncoder``, NRMS ``UserEncoder``, ``DotProduct``
-- are reused).  ``SentiRecModule`` needs lightning / torch_geometric / torchmetrics, so what it adds to the NRMS
wiring -- the ``nn.Linear`` sentiment predictor (sentirec_module.py:186-188), the row order of its input (:271)
and the two extra loss terms (:347-364) -- is restated here, line by line.

Usage:  python tests/golden/make_golden_sentirec.py   (from the repo root)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (puts the repo root and /root/reference on sys.path)

from newsreclib_amd.synthetic import make_batch  # noqa: E402
from oracle.sentirec_oracle import make_sentirec_params  # noqa: E402

N_SENT = 4          # num_sent_classes + 1 (sentirec_module.py:113)


def add_sentiment(batch, seed):
    g = torch.Generator().manual_seed(seed)
    for part in ("x_hist", "x_cand"):
        n = batch[part]["title"].shape[0]
        batch[part]["sentiment_score"] = (torch.rand(n, generator=g) * 2 - 1).float()      # VADER compound in [-1, 1]
        batch[part]["sentiment"] = torch.randint(0, N_SENT, (n,), generator=g)
    return batch


class RefSentiRec(MG.RefNRMS):
    def __init__(self, params):
        nrms = {k: v for k, v in params.items() if not k.startswith("sent_predictor.")}
        super().__init__(nrms)
        self.sent_predictor = torch.nn.Linear(MG.D, N_SENT)                               # :186-188
        with torch.no_grad():
            self.sent_predictor.weight.copy_(params["sent_predictor.weight"])
            self.sent_predictor.bias.copy_(params["sent_predictor.bias"])
        self.sent_pred_loss = torch.nn.L1Loss()                                             # :129


def ref_forward(model, batch, p_drop, seed, pred_coef, div_coef):
    out = MG.ref_forward(model, batch, p_drop, seed)
    B = batch["batch_size"]
    sent_scores = model.sent_predictor(torch.cat((out["cand_vec"], out["hist_vec"]), dim=0))       # :271
    loss = out["loss"]
    # sentirec_module.py:347-364 (note: `sent_scores` is REBOUND to the labels before the L1 loss)
    labels = torch.cat((batch["x_cand"]["sentiment_score"], batch["x_hist"]["sentiment_score"]))
    sent_pred_loss = model.sent_pred_loss(labels.flatten(), labels)
    loss = loss + pred_coef * sent_pred_loss
    sent_hist, mask_hist = MG.dense_batch_loops(batch["x_hist"]["sentiment_score"], batch["batch_hist"], B)
    sent_cand, _ = MG.dense_batch_loops(batch["x_cand"]["sentiment_score"], batch["batch_cand"], B)
    hist_news_size = torch.tensor([torch.where(mask_hist[n])[0].shape[0] for n in range(mask_hist.shape[0])])
    user_mean_sent_score = torch.div(sent_hist.sum(dim=1), hist_news_size)
    sent_div_loss = torch.nn.functional.relu(user_mean_sent_score.unsqueeze(dim=-1) * sent_cand * out["scores"]).mean()
    loss = loss + div_coef * sent_div_loss
    out = dict(out)
    out.update(loss=loss, sent_scores=sent_scores, sent_div_loss=sent_div_loss)
    return out


def run_case(name, batch, vocab, param_seed, p_drop=0.0, seed=0, full_embedding=False, pred_coef=0.4, div_coef=10.0):
    params = make_sentirec_params(vocab, N_SENT, MG.D, MG.Q, seed=param_seed)
    model = RefSentiRec(params)
    model.train()
    out = ref_forward(model, batch, p_drop, seed, pred_coef, div_coef)
    out["loss"].backward()
    arrays = MG.batch_arrays(batch)
    for part in ("hist", "cand"):
        arrays[f"in_sentiment_score_{part}"] = batch["x_" + part]["sentiment_score"].numpy()
    arrays.update(cfg_vocab=np.int64(vocab), cfg_param_seed=np.int64(param_seed), cfg_p_drop=np.float64(p_drop),
                  cfg_seed=np.int64(seed), cfg_sample_stride=np.int64(MG.SAMPLE_STRIDE), cfg_n_sent=np.int64(N_SENT),
                  cfg_pred_coef=np.float64(pred_coef), cfg_div_coef=np.float64(div_coef))
    for k in ("user_vec", "scores", "y_true", "loss", "sent_div_loss"):
        arrays["out_" + k] = out[k].detach().numpy()
    rs = 1 if full_embedding else MG.ROW_STRIDE
    arrays["cfg_row_stride"] = np.int64(rs)
    for k in ("hist_vec", "cand_vec"):
        arrays["out_" + k] = out[k].detach().numpy()[::rs].copy()
    arrays["out_sent_scores"] = out["sent_scores"].detach().numpy()[::rs].copy()
    arrays.update(MG.grad_summary(model, full_embedding))
    path = os.path.join(MG.OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: loss={float(out['loss'].detach()):.6f} (div term {float(out['sent_div_loss'].detach()):.6f}) "
          f"-> {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    run_case("sentirec_tiny_eval", add_sentiment(MG.tiny_batch(), 3), 64, param_seed=1, full_embedding=True)
    run_case("sentirec_tiny_train", add_sentiment(MG.tiny_batch(), 3), 64, param_seed=1, p_drop=0.2, seed=7,
             full_embedding=True)
    run_case("sentirec32_train", add_sentiment(make_batch(32, vocab=5000, mode="ragged", seed=21), 4), 5000,
             param_seed=2, p_drop=0.2, seed=13)


if __name__ == "__main__":
    main()
