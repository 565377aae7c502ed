"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_batch(g, device="cpu"):
    """Rebuild the RecommendationBatch dict stored in a golden file."""
    t = lambda a: torch.as_tensor(a).to(device)  # noqa: E731
    return {
        "x_hist": {"title": t(g["in_ids_hist"])}, "x_cand": {"title": t(g["in_ids_cand"])},
        "batch_hist": t(g["in_batch_hist"]), "batch_cand": t(g["in_batch_cand"]),
        "labels": t(g["in_labels"]), "batch_size": int(g["in_batch_size"]),
        "user_ids": torch.arange(int(g["in_batch_size"])) + 1,
        "user_idx": torch.arange(int(g["in_batch_size"])),
    }


def check_grads_against_golden(g, grads, rtol=2e-4, atol=2e-5):
    """grads: dict reference-state_dict-key -> tensor.  Compares norms, sums, samples, rows."""
    stride = int(g["cfg_sample_stride"])
    for key in [k[len("gnorm/"):] for k in g if k.startswith("gnorm/")]:
        gr = grads[key].detach().cpu().double()
        ref_norm = float(g["gnorm/" + key])
        assert abs(float(gr.norm()) - ref_norm) <= rtol * ref_norm + atol, (key, float(gr.norm()), ref_norm)
        if "gsample/" + key in g:
            ref = torch.from_numpy(g["gsample/" + key]).double()
            got = gr.reshape(-1)[::stride]
            scale = max(1.0, float(ref.abs().max()))
            assert float((got - ref).abs().max()) <= 2e-4 * scale, (key, float((got - ref).abs().max()))
        if "gfull/" + key in g:
            ref = torch.from_numpy(g["gfull/" + key]).double()
            scale = max(1.0, float(ref.abs().max()))
            assert float((gr - ref).abs().max()) <= 2e-4 * scale, key
            # padding_idx=0: the reference zeroes the gradient row of id 0 (text.py:215-217)
            assert float(gr[0].abs().max()) == 0.0
        if "grows/" + key in g:
            rows = torch.from_numpy(g["grows_idx/" + key])
            ref = torch.from_numpy(g["grows/" + key]).double()
            scale = max(1.0, float(ref.abs().max()))
            assert float((gr[rows] - ref).abs().max()) <= 2e-4 * scale, key
