"""GPU parity of the LSTUR path (BASELINE config 5) against the CPU oracle and the golden vectors made
from the reference's own components.  Every call goes through the C ABI."""
import numpy as np
import pytest
import torch

from tests.helpers import (LSTUR_CASES, batch_to, build_lstur_module, check_lstur_grads, load_golden,
                           lstur_golden_batch, lstur_golden_cfg, lstur_golden_params, module_grads)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f32", "bf16x3"])
def engine(request):
    from newsreclib_amd import _lib
    prev = _lib.get_gemm_engine()
    _lib.set_gemm_engine(request.param)
    yield request.param
    _lib.set_gemm_engine(prev)


def _tols(engine):
    # forward tolerance / gradient tolerance relative to max(1, |ref|max)
    return (2e-5, 2e-4) if engine == "f32" else (1e-4, 5e-4)


def _cnn_params(rng, V, D, F, W, Q):
    t = lambda *s, scale: torch.from_numpy((rng.standard_normal(s) * scale).astype(np.float32))  # noqa: E731
    return {"embedding_layer.weight": t(V, D, scale=0.3), "cnn.weight": t(F, 1, W, D, scale=(W * D) ** -0.5),
            "cnn.bias": t(F, scale=0.05), "additive_attention.linear.weight": t(Q, F, scale=F ** -0.5),
            "additive_attention.linear.bias": t(Q, scale=0.05), "additive_attention.query": t(Q, scale=0.1)}


@pytest.mark.parametrize("shape", [(37, 12, 64, 48, 3, 32), (5, 30, 300, 300, 3, 200), (130, 9, 32, 64, 5, 16),
                                   (3, 4, 16, 16, 1, 8), (1, 31, 20, 36, 3, 8), (9, 32, 16, 16, 3, 8), (4, 63, 24, 20, 3, 8)])
@pytest.mark.parametrize("p_drop", [0.0, 0.2])
def test_cnn_encoder_matches_oracle(shape, p_drop, engine):
    from newsreclib_amd.ops_lstur import CnnEncoderFn
    from oracle.lstur_oracle import cnn_text_encoder_fwd
    from oracle.nrms_oracle import dropout_multiplier
    N, L, D, F, W, Q = shape
    rng = np.random.default_rng(N * 7 + L)
    V = 50
    params = _cnn_params(rng, V, D, F, W, Q)
    ids = torch.from_numpy(rng.integers(0, V, (N, L)))
    ids[:, L - 2:] = 0
    d_out = torch.from_numpy(rng.standard_normal((N, F)).astype(np.float32))
    seed, s0 = 11, 2
    m1 = dropout_multiplier(seed, s0, p_drop, (N, L, D)) if p_drop else None
    m2 = dropout_multiplier(seed, s0 + 1, p_drop, (N, L, F)) if p_drop else None
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = cnn_text_encoder_fwd(ids, leaves, "", m1, m2)
    ref.backward(d_out)
    keys = list(params)
    dev = [params[k].cuda().requires_grad_(True) for k in keys]
    for order in (None, torch.argsort(ids.reshape(-1)).cuda()):
        for t in dev:
            t.grad = None
        out = CnnEncoderFn.apply(ids.cuda(), *dev, p_drop, seed, s0, None, order)
        out.backward(d_out.cuda())
        ftol, gtol = _tols(engine)
        assert float((out.detach().cpu() - ref.detach()).abs().max()) <= ftol * max(1.0, float(ref.abs().max()))
        for k, t in zip(keys, dev):
            r = leaves[k].grad.clone()
            if k == "embedding_layer.weight":
                r[0] = 0.0                                   # padding_idx
                assert float(t.grad[0].abs().max()) == 0.0
            assert float((t.grad.cpu() - r).abs().max()) <= gtol * max(1.0, float(r.abs().max())), k


@pytest.mark.parametrize("p_row", [0.0, 0.5])
def test_embedding_rows(p_row):
    from newsreclib_amd.ops_lstur import EmbeddingRowsFn
    from oracle.nrms_oracle import dropout_multiplier
    rng = np.random.default_rng(5)
    table = torch.from_numpy(rng.standard_normal((40, 100)).astype(np.float32))
    ids = torch.from_numpy(rng.integers(0, 40, 300))
    mult = dropout_multiplier(9, 8, p_row, (300,)) if p_row else torch.ones(300)
    t = table.cuda().requires_grad_(True)
    out = EmbeddingRowsFn.apply(ids.cuda(), t, p_row, 9, 8, None)
    assert torch.equal(out.detach().cpu(), table[ids] * mult[:, None])        # bit-exact (x * {0, 1, 2})
    d_out = torch.from_numpy(rng.standard_normal((300, 100)).astype(np.float32))
    out.backward(d_out.cuda())
    ref = torch.zeros_like(table).index_add_(0, ids, d_out * mult[:, None])
    ref[0] = 0.0
    assert float((t.grad.cpu() - ref).abs().max()) <= 1e-5
    assert float(t.grad[0].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(5, 7, 24, 24), (16, 50, 700, 700), (3, 4, 16, 8), (130, 3, 32, 32)])
@pytest.mark.parametrize("with_h0", [True, False])
def test_gru_matches_oracle(shape, with_h0, engine):
    from newsreclib_amd.ops_lstur import GruFn
    from oracle.lstur_oracle import gru_last_hidden
    B, T, Din, Hd = shape
    rng = np.random.default_rng(B + T)
    t = lambda *s, scale=1.0: torch.from_numpy((rng.standard_normal(s) * scale).astype(np.float32))  # noqa: E731
    hist, h0 = t(B, T, Din, scale=0.5), t(B, Hd, scale=0.5)
    lengths = torch.from_numpy(rng.integers(1, T + 1, B))
    lengths[0] = T
    params = [t(3 * Hd, Din, scale=Din ** -0.5), t(3 * Hd, Hd, scale=Hd ** -0.5), t(3 * Hd, scale=0.05),
              t(3 * Hd, scale=0.05)]
    d_out = t(B, Hd)
    leaves = [x.clone().requires_grad_(True) for x in [hist, h0] + params]
    ref = gru_last_hidden(leaves[0], lengths, leaves[1] if with_h0 else torch.zeros(B, Hd), *leaves[2:])
    ref.backward(d_out)
    dev = [x.cuda().requires_grad_(True) for x in [hist, h0] + params]
    out = GruFn.apply(dev[0], lengths.cuda(), dev[1] if with_h0 else None, *dev[2:], None)
    out.backward(d_out.cuda())
    ftol, gtol = _tols(engine)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= ftol * 5
    names = ["hist", "h0", "weight_ih", "weight_hh", "bias_ih", "bias_hh"]
    for n, a, b in zip(names, dev, leaves):
        if n == "h0" and not with_h0:
            continue
        scale = max(1.0, float(b.grad.abs().max()))
        assert float((a.grad.cpu() - b.grad).abs().max()) <= gtol * scale, n
    # steps past a sequence's length contribute nothing
    pad = (torch.arange(T)[None, :] >= lengths[:, None])
    assert float(dev[0].grad.cpu()[pad].abs().max() if pad.any() else 0.0) == 0.0


def test_gru_rejects_empty_history():
    from newsreclib_amd.user_encoder_lstur import UserEncoder
    enc = UserEncoder(num_users=5, input_dim=16, user_masking_probability=0.5, long_short_term_method="ini").cuda()
    with pytest.raises(RuntimeError):
        enc(torch.tensor([1, 2]).cuda(), torch.zeros(2, 3, 16).cuda(), torch.tensor([2, 0]).cuda())


@pytest.mark.parametrize("name", LSTUR_CASES)
def test_lstur_module_matches_reference_golden(name, engine):
    g = load_golden(name)
    cfg = lstur_golden_cfg(g)
    params = lstur_golden_params(cfg)
    mod = build_lstur_module(cfg, params)
    mod.train()
    batch = batch_to(lstur_golden_batch(g), "cuda")
    seed = cfg["seed"]
    from newsreclib_amd.dense_batch import to_dense_batch
    from newsreclib_amd.nrms_module import prepare_batch
    pb = prepare_batch(batch)
    scores = mod.forward(pb, seed=seed)
    ftol, gtol = _tols(engine)
    assert float(np.abs(scores.detach().cpu().numpy() - g["out_scores"]).max()) <= max(ftol * 5, 1e-4)   # contract 1e-3
    y_true, _ = to_dense_batch(pb["labels"], pb["batch_cand"], pb["batch_size"], pb["max_cand"], pb["cand_offsets"],
                               pb["cand_flat_idx"])
    loss = mod.criterion(scores, y_true.float())
    assert abs(float(loss) - float(g["out_loss"])) <= 1e-4
    loss.backward()
    grads = module_grads(mod)
    check_lstur_grads(g, grads, tol=gtol, rtol=5e-4)


def test_lstur_config5_shapes_train_step():
    """One full-size step (configs/model/lstur.yaml dims, B = 64): finite loss near ln(5), gradients reach
    every parameter, the GRU state of a user with a short history ignores its padding."""
    from newsreclib_amd.synthetic import add_lstur_fields, make_batch
    from oracle.lstur_oracle import make_lstur_params
    cfg = dict(vocab=5000, n_categ=19, n_users=500, D=300, F=300, W=3, Q=200, categ_dim=100,
               text_attrs=("title", "abstract"), text_order=("title", "abstract"), method="ini", p_drop=0.2,
               p_mask=0.5)
    params = make_lstur_params(cfg["vocab"], cfg["n_categ"], cfg["n_users"], seed=3)
    mod = build_lstur_module(cfg, params)
    mod.train()
    batch = add_lstur_fields(make_batch(64, vocab=cfg["vocab"], mode="ragged", seed=5), cfg["vocab"], cfg["n_categ"],
                             cfg["n_users"], 50, seed=6)
    batch = batch_to(batch, "cuda")
    loss, preds, targets, cand_news_size, *_ = mod.model_step(batch)
    loss.backward()
    assert torch.isfinite(loss) and 0.5 < float(loss) < 6.0
    assert preds.shape == targets.shape and int(cand_news_size.sum()) == preds.shape[0]
    for k, gr in module_grads(mod).items():
        assert torch.isfinite(gr).all() and float(gr.abs().max()) > 0.0, k


def test_lstur_trainer_prefetch_hints_match_for_a_multi_attribute_encoder(engine):
    """Round-5 advisor: an encoder registered under several attributes (LSTUR: title + abstract through ONE text encoder)
    concatenates their ids for the lazy table optimizer; the prefetch built that concatenation on the launch stream -- racing the
    side stream that had just produced x_all -- and the fresh tensor never matched the hint, so the early catch-up was silently
    wasted.  The ids are now built on the side stream, kept on the prepared batch and recognised by identity: every announced
    batch that came is a hit, for the word table AND the long-term user table, and the losses follow the plain loop."""
    from newsreclib_amd.nrms_module import attach_layout
    from newsreclib_amd.synthetic import add_lstur_fields, make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    from oracle.lstur_oracle import make_lstur_params
    cfg = dict(vocab=3000, n_categ=19, n_users=300, D=300, F=300, W=3, Q=200, categ_dim=100,
               text_attrs=("title", "abstract"), text_order=("title", "abstract"), method="ini", p_drop=0.0, p_mask=0.0)
    params = make_lstur_params(cfg["vocab"], cfg["n_categ"], cfg["n_users"], seed=4)
    batches = [attach_layout(batch_to(add_lstur_fields(make_batch(12, vocab=cfg["vocab"], mode="ragged", seed=40 + i), cfg["vocab"],
                                                       cfg["n_categ"], cfg["n_users"], 50, seed=60 + i), "cuda")) for i in range(3)]
    losses = {}
    for mode in ("plain", "prefetch"):
        mod = build_lstur_module(cfg, params)
        tr = NRMSTrainer(mod, lr=1e-4)
        assert len(tr.lazy_tables) == 2                       # the word table and the long-term user table
        ls = []
        for i in range(5):
            b, nb = batches[i % 3], batches[(i + 1) % 3]
            ls.append(float(tr.step(b, nb) if mode == "prefetch" else tr.step(b)))
        tr.flush()
        torch.cuda.synchronize()
        losses[mode] = ls
        if mode == "prefetch":
            for tab, _ in tr.lazy_tables:
                assert (tab.hint_hits, tab.hint_misses) == (4, 0), (tab.hint_hits, tab.hint_misses)
                tab.check()
    for a, b in zip(losses["plain"], losses["prefetch"]):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (a, b)


_PERSISTENT_GRU_SCRIPT = r"""
import numpy as np, torch
from newsreclib_amd import _lib
from newsreclib_amd.ops_lstur import GruFn
from oracle.lstur_oracle import gru_last_hidden
_lib.set_gemm_engine("bf16x3")
for (B, T, Din, Hd) in [(16, 50, 700, 700), (128, 20, 64, 400), (5, 7, 24, 24)]:
    rng = np.random.default_rng(B + T)
    t = lambda *s, scale=1.0: torch.from_numpy((rng.standard_normal(s) * scale).astype(np.float32))
    hist, h0 = t(B, T, Din, scale=0.5), t(B, Hd, scale=0.5)
    lengths = torch.from_numpy(rng.integers(1, T + 1, B)); lengths[0] = T
    params = [t(3 * Hd, Din, scale=Din ** -0.5), t(3 * Hd, Hd, scale=Hd ** -0.5), t(3 * Hd, scale=0.05), t(3 * Hd, scale=0.05)]
    ref = gru_last_hidden(hist, lengths, h0, *params)
    out = GruFn.apply(hist.cuda(), lengths.cuda(), h0.cuda(), *[x.cuda() for x in params], None)
    err = float((out.cpu() - ref).abs().max())
    assert err <= 5e-4, (B, T, Din, Hd, err)
print("PERSISTENT_GRU_OK")
"""


def test_gru_persistent_cooperative_launch_matches_oracle():
    # the one-launch recurrence (W_hh resident in registers, grid barrier per step) is an opt-in measurement variant
    # (NRL_GRU_PERSISTENT=1, read once per process): run it in a child process, under a timeout -- a grid barrier
    # that is not co-resident would hang, which the launcher's occupancy check is there to exclude
    import os
    import subprocess
    import sys
    env = dict(os.environ, NRL_GRU_PERSISTENT="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", _PERSISTENT_GRU_SCRIPT], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "PERSISTENT_GRU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


_CONV_WGRAD_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from newsreclib_amd import _lib
from newsreclib_amd.ops_lstur import CnnEncoderFn
_lib.set_gemm_engine("bf16x3")
N, L, D, F, W, Q = (int(v) for v in sys.argv[3:9])
rng = np.random.default_rng(N + L)
V = 500
t = lambda *s, scale: torch.from_numpy((rng.standard_normal(s) * scale).astype(np.float32))
params = [t(V, D, scale=0.3), t(F, 1, W, D, scale=(W * D) ** -0.5), t(F, scale=0.05), t(Q, F, scale=F ** -0.5), t(Q, scale=0.05),
          t(Q, scale=0.1)]
ids = torch.from_numpy(rng.integers(0, V, (N, L)))
ids[::3, L - 3:] = 0
d_out = torch.from_numpy(rng.standard_normal((N, F)).astype(np.float32))
dev = [p.cuda().requires_grad_(True) for p in params]
out = CnnEncoderFn.apply(ids.cuda(), *dev, 0.2, 11, 2, None, torch.argsort(ids.reshape(-1)).cuda())
out.backward(d_out.cuda())
np.savez(sys.argv[2], out=out.detach().cpu().numpy(), **{"g%d" % i: p.grad.cpu().numpy() for i, p in enumerate(dev)})
"""


@pytest.mark.parametrize("shape", [(700, 30, 300, 300, 3, 200), (260, 50, 300, 300, 3, 200), (33, 31, 64, 48, 3, 32),
                                   (40, 63, 20, 36, 3, 8), (1100, 7, 32, 16, 3, 8)])
def test_conv_weight_gradient_from_planes_equals_the_fp32_fed_kernel(shape, tmp_path, engine):
    """The planes form of the convolution weight gradient (wgrad_planes_conv_kernel: padded 32 / 64-row news, the taps as row
    rotations; x planes from the lookup kernel, dc planes from the attention-backward epilogue) against the fp32-fed kernel it
    replaces, same inputs, one process each (the switch is read once per process): every other output bit for bit -- the
    planes are extra outputs -- and the convolution's weight / bias gradients to summation order."""
    import os
    import subprocess
    import sys
    if engine != "bf16x3":
        pytest.skip("bf16x3 path")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # "0": fp32 rows everywhere; "1x": x as fp32 rows, converted for the planes weight gradient (round 4); "1": x ONLY as planes,
    # written by the lookup and read by the convolution forward through KCWindowPlanes too (round 5)
    # "1f": as "1" with the additive attention fed from fp32 rows (NRL_CNN_AA_PLANES=0, rounds 4-5); "1" has it on planes (round 6)
    for flag, xp in (("0", "0"), ("1x", "0"), ("1", "1"), ("1f", "1")):
        f = str(tmp_path / ("g%s.npz" % flag))
        env = dict(os.environ, NRL_CONV_WGRAD_PLANES=flag[0], NRL_CONV_X_PLANES=xp, NRL_CNN_AA_PLANES="0" if flag == "1f" else "1")
        subprocess.run([sys.executable, "-c", _CONV_WGRAD_SCRIPT, root, f] + [str(v) for v in shape], check=True, env=env,
                       timeout=600)
        res[flag] = np.load(f)
    # additive attention on planes (c planes from the convolution's epilogue, d_pre planes from the pooling backward) against the
    # fp32-fed form: the planes hold the very (hi, lo) split the row-panel kernel makes of the fp32 rows, so the tanh projection, the
    # output, dc, the convolution's gradients and the table gradient are EQUAL; the additive attention's own weight / bias gradient
    # comes from another kernel (summation order), its query gradient from atomics in both runs
    p1, f1 = res["1"], res["1f"]
    assert np.array_equal(p1["out"], f1["out"])
    for k in ("g0", "g1", "g2"):
        assert np.array_equal(p1[k], f1[k]), k
    for k in ("g3", "g4", "g5"):
        scale = max(1.0, float(np.abs(f1[k]).max()))
        assert float(np.abs(p1[k] - f1[k]).max()) <= 2e-5 * scale, (k, float(np.abs(p1[k] - f1[k]).max()), scale)
    a, b = res["0"], res["1x"]
    assert np.array_equal(a["out"], b["out"])
    assert np.array_equal(a["g0"], b["g0"])                                  # table gradient: sorted segments, fixed order
    # the x-planes forward multiplies the same (hi, lo) operands in another k order (taps padded to whole k-blocks): equal to rounding
    c = res["1"]
    so = max(1.0, float(np.abs(a["out"]).max()))
    assert float(np.abs(a["out"] - c["out"]).max()) <= 1e-5 * so
    # (a ReLU whose pre-activation is within rounding of 0 may open in one k order and stay shut in the other: each such gate moves
    #  one filter's weight gradient and three embedding rows by ~|dc| -- a handful of entries, bounded; everything else to rounding)
    for k in ("g0", "g1", "g2", "g3", "g4", "g5"):
        scale = max(1.0, float(np.abs(a[k]).max()))
        d = np.abs(a[k] - c[k])
        off = float((d > 2e-5 * scale).mean())
        # (one flipped gate touches three rows of a 500-row table = 0.6 % of its entries, one filter's 900 of the 270 k weight
        #  gradients = 0.33 %, one of the 300 bias gradients)
        assert off <= 2e-2 and float(d.max()) <= 0.05 * scale, (k, off, float(d.max()), scale)
    # ... and the ReLU-gate argument is CHECKED, not assumed (round-5 advisor): the pre-activations are restated on the host in fp64
    # under the product's dropout draw; every filter whose weight-gradient row differs beyond rounding between the two k orders must
    # have a pre-activation within 1e-4 of zero (relative to the largest) somewhere in the batch -- a mismatch on a filter with no
    # such gate would be a real ordering bug, and fails here.
    N, L, D, F, W, Q = shape
    if N * L * F * W * D <= 6e9:
        from oracle.nrms_oracle import dropout_multiplier
        rng = np.random.default_rng(N + L)                                   # (the script's draws, in its order)
        t64 = lambda *s, scale: (rng.standard_normal(s) * scale).astype(np.float32).astype(np.float64)  # noqa: E731
        emb, wc, bc = t64(500, D, scale=0.3), t64(F, 1, W, D, scale=(W * D) ** -0.5), t64(F, scale=0.05)
        t64(Q, F, scale=F ** -0.5), t64(Q, scale=0.05), t64(Q, scale=0.1)
        ids = rng.integers(0, 500, (N, L))
        ids[::3, L - 3:] = 0
        x = emb[ids] * dropout_multiplier(11, 2, 0.2, (N, L, D)).numpy().astype(np.float64)
        xp = np.pad(x, ((0, 0), (1, 1), (0, 0)))
        pre = bc[None, None, :] + sum(xp[:, w:w + L, :] @ wc[:, 0, w, :].T for w in range(W))      # (N, L, F)
        # (two k orders of a 900-term fp32 accumulation differ by ~sqrt(900) * 6e-8 * |partial sums| ~ 2e-6: a gate can only flip inside
        #  a band of that size; 1e-5 of the largest pre-activation leaves a factor of a few.  How many filters have such a gate is
        #  printed: the check has teeth only while that is a minority of the filters)
        near = (np.abs(pre) <= 1e-5 * max(1.0, float(np.abs(pre).max()))).any(axis=(0, 1))         # filters with a gate at the edge
        scale1 = max(1.0, float(np.abs(a["g1"]).max()))
        rows_off = (np.abs(a["g1"] - c["g1"]).reshape(F, -1).max(axis=1) > 2e-5 * scale1)
        print(f"conv planes vs rows {shape}: {int(rows_off.sum())} of {F} filter rows differ beyond rounding; "
              f"{int(near.sum())} filters have a pre-activation within 1e-5 of zero")
        assert not (rows_off & ~near).any(), (int(rows_off.sum()), int(near.sum()), np.nonzero(rows_off & ~near)[0][:8])
    for k in ("g1", "g2"):                                                   # conv weight (F, 1, W, D), conv bias
        scale = max(1.0, float(np.abs(a[k]).max()))
        assert float(np.abs(a[k] - b[k]).max()) <= 2e-5 * scale, (k, float(np.abs(a[k] - b[k]).max()), scale)
    for k in ("g3", "g4", "g5"):                                             # attention parameters: atomics in both runs
        scale = max(1.0, float(np.abs(a[k]).max()))
        assert float(np.abs(a[k] - b[k]).max()) <= 2e-5 * scale, k
