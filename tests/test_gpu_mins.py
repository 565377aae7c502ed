"""GPU parity of the standalone multi-head attention block and of the MINS module against the CPU oracle and the
golden vectors made from the reference's own components."""
import math

import numpy as np
import pytest
import torch

from tests.helpers import (MINS_CASES, batch_to, build_mins_module, check_lstur_grads, load_golden, lstur_golden_batch,
                           mins_golden_cfg, mins_golden_params, module_grads)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f32", "bf16x3"])
def engine(request):
    from newsreclib_amd import _lib
    prev = _lib.get_gemm_engine()
    _lib.set_gemm_engine(request.param)
    yield request.param
    _lib.set_gemm_engine(prev)


def _tols(engine):
    return (2e-5, 2e-4) if engine == "f32" else (1e-4, 5e-4)


@pytest.mark.parametrize("shape", [(5, 3, 48, 3), (128, 50, 300, 15), (70, 7, 384, 6), (1, 1, 32, 2),
                                   # 32 < S < 64: the 64-lanes-per-group kernels, every head dim
                                   (50, 9, 300, 15), (40, 5, 128, 2), (33, 4, 96, 2), (63, 3, 64, 2), (45, 3, 32, 2)])
def test_mha_matches_oracle(shape, engine):
    from newsreclib_amd.ops_blocks import MhaFn
    from oracle.nrms_oracle import _mhsa_seq_first
    S, Bt, D, heads = shape
    rng = np.random.default_rng(S + D)
    t = lambda *s, scale=1.0: torch.from_numpy((rng.standard_normal(s) * scale).astype(np.float32))  # noqa: E731
    x, w_in, b_in, w_o, b_o = t(S, Bt, D, scale=0.5), t(3 * D, D, scale=D ** -0.5), t(3 * D, scale=0.05), \
        t(D, D, scale=D ** -0.5), t(D, scale=0.05)
    d_out = t(S, Bt, D)
    leaves = [v.clone().requires_grad_(True) for v in (x, w_in, b_in, w_o, b_o)]
    ref = _mhsa_seq_first(*leaves, heads)
    ref.backward(d_out)
    dev = [v.cuda().requires_grad_(True) for v in (x, w_in, b_in, w_o, b_o)]
    out = MhaFn.apply(*dev, heads, None, None)
    out.backward(d_out.cuda())
    ftol, gtol = _tols(engine)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= ftol * 5
    for name, got, want in zip(("x", "w_in", "b_in", "w_o", "b_o"), dev, leaves):
        assert float((got.grad.cpu() - want.grad).abs().max()) <= gtol * max(1.0, float(want.grad.abs().max())), name


@pytest.mark.parametrize("shape", [(4, 5, 48, 4, 16), (16, 50, 300, 6, 32), (3, 2, 64, 4, 8)])
def test_user_encoder_matches_oracle(shape, engine):
    """incl. the reference shape: 300 / 6 channels = head dim 50 (zero-padded to 64), GRU width 50 (padded to 52)."""
    from newsreclib_amd.user_encoder_mins import UserEncoder
    from oracle import mins_oracle as MO
    B, H, D, C, Q = shape
    params = {k: v for k, v in MO.make_mins_params(8, 4, D, Q, 8, C, seed=B).items() if k.startswith(MO.USER)}
    rng = np.random.default_rng(H)
    hist = torch.from_numpy((rng.standard_normal((B, H, D)) * 0.5).astype(np.float32))
    sizes = torch.from_numpy(rng.integers(1, H + 1, B))
    sizes[0] = H
    for b in range(B):
        hist[b, int(sizes[b]):] = 0.0
    d_out = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32))
    keys = [k for k in params if "multi_channel_gru" not in k]
    leaves = {k: params[k].clone().requires_grad_(True) for k in keys}
    full = {k: leaves[k if "multi_channel_gru" not in k else MO.USER + "gru." + k.rsplit(".", 1)[1]] for k in params}
    hl = hist.clone().requires_grad_(True)
    ref = MO.mins_user_encoder_fwd(hl, sizes, full, C)
    ref.backward(d_out)
    enc = UserEncoder(D, Q, D, C)
    enc.load_state_dict({k[len(MO.USER):]: v for k, v in params.items()})
    enc = enc.cuda()
    hd = hist.cuda().requires_grad_(True)
    out = enc(hd, sizes.cuda())
    out.backward(d_out.cuda())
    ftol, gtol = _tols(engine)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= ftol * 5
    assert float((hd.grad.cpu() - hl.grad).abs().max()) <= gtol * max(1.0, float(hl.grad.abs().max()))
    for k, p in enc.named_parameters():
        want = leaves[MO.USER + k].grad
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        want = want if want is not None else torch.zeros_like(p).cpu()
        assert float((got.cpu() - want).abs().max()) <= gtol * max(1.0, float(want.abs().max())), k
    assert math.isfinite(float(out.sum()))


@pytest.mark.parametrize("name", MINS_CASES)
def test_mins_module_matches_reference_golden(name, engine):
    from newsreclib_amd.dense_batch import to_dense_batch
    from newsreclib_amd.nrms_module import prepare_batch
    g = load_golden(name)
    cfg = mins_golden_cfg(g)
    mod = build_mins_module(cfg, mins_golden_params(cfg))
    mod.train() if cfg["p_drop"] > 0 else mod.eval()
    pb = prepare_batch(batch_to(lstur_golden_batch(g), "cuda"))
    scores = mod.forward(pb, seed=cfg["seed"])
    ftol, gtol = _tols(engine)
    assert float(np.abs(scores.detach().cpu().numpy() - g["out_scores"]).max()) <= max(ftol * 5, 1e-4)   # contract 1e-3
    y_true, _ = to_dense_batch(pb["labels"], pb["batch_cand"], pb["batch_size"], pb["max_cand"], pb["cand_offsets"],
                               pb["cand_flat_idx"])
    loss = mod.criterion(scores, y_true.float())
    assert abs(float(loss.detach()) - float(g["out_loss"])) <= 1e-4
    loss.backward()
    check_lstur_grads(g, module_grads(mod), tol=gtol, rtol=5e-4)
