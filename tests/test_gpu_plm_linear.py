"""Config 4 (BASELINE.json configs[3]): the projections of the PLM body on this library's matrix-core engines
(news_encoder.NrlLinear / swap_linears -> nrl_linear_fwd / nrl_linear_bwd) against the same HF layer on fp32 torch."""
import copy

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _one_layer_roberta(layers=1):
    from transformers import RobertaConfig, RobertaModel
    cfg = RobertaConfig(vocab_size=1000, hidden_size=768, num_hidden_layers=layers, num_attention_heads=12,
                        intermediate_size=3072, max_position_embeddings=130, type_vocab_size=1, pad_token_id=1,
                        bos_token_id=0, eos_token_id=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    return RobertaModel(cfg, add_pooling_layer=False)


@pytest.mark.parametrize("engine_name", ["bf16x3", "f32"])
def test_full_width_encoder_layer_forward_backward_matches_torch(engine_name):
    """One roberta-base-width encoder layer (D = 768, 12 heads, FFN 3072, L = 96): hidden states and EVERY parameter
    gradient of the swapped model within 1e-4 x scale of the unswapped fp32 torch model; state-dict keys unchanged;
    a frozen linear gets no weight gradient but still passes the gradient to its input (text.py:69-73)."""
    from newsreclib_amd import _lib
    from newsreclib_amd.news_encoder import NrlLinear, swap_linears
    _lib.set_gemm_engine(engine_name)
    try:
        ref = _one_layer_roberta().to(DEV).train()
        mod = copy.deepcopy(ref)
        keys = list(mod.state_dict().keys())
        n = swap_linears(mod.encoder)
        assert n == 6 and list(mod.state_dict().keys()) == keys
        assert isinstance(mod.encoder.layer[0].intermediate.dense, NrlLinear)
        # freeze the feed-forward output projection in both: its input must still receive a gradient
        for m in (ref, mod):
            m.encoder.layer[0].output.dense.weight.requires_grad_(False)
            m.encoder.layer[0].output.dense.bias.requires_grad_(False)
        g = torch.Generator().manual_seed(1)
        ids = torch.randint(3, 1000, (6, 96), generator=g).to(DEV)
        d_out = torch.randn(6, 96, 768, generator=g).to(DEV)
        outs = []
        for m in (ref, mod):
            h = m(input_ids=ids, attention_mask=torch.ones_like(ids))[0]
            h.backward(d_out)
            outs.append(h.detach())
        scale = float(outs[0].abs().max())
        err = float((outs[0] - outs[1]).abs().max())
        print(f"{engine_name}: hidden max abs err {err:.3e} (scale {scale:.2f})")
        assert err <= 1e-4 * max(1.0, scale)
        worst = 0.0
        for (k, p), (_, q) in zip(ref.named_parameters(), mod.named_parameters()):
            if not p.requires_grad:
                assert q.grad is None, k
                continue
            assert q.grad is not None, k
            s = max(1.0, float(p.grad.abs().max()))
            e = float((p.grad - q.grad).abs().max())
            worst = max(worst, e / s)
            assert e <= 1e-4 * s, (k, e, s)
        print(f"{engine_name}: worst parameter-gradient error {worst:.3e} of the parameter's largest gradient")
    finally:
        _lib.set_gemm_engine("bf16x3")


def test_plm_module_uses_the_library_for_its_body_projections(tmp_path):
    """news_encoder.PLM swaps the body's nn.Linear modules (12 layers x 6) and still loads / saves reference-keyed
    state dicts; NRL_PLM_LINEAR=0 keeps the HF modules."""
    import os
    from newsreclib_amd.news_encoder import PLM
    _one_layer_roberta(layers=2).save_pretrained(str(tmp_path))
    enc = PLM(str(tmp_path), [0], 768, True, False, None, 16, 200, 0.2)
    assert enc.nrl_linears == 12
    assert not enc.plm_model.encoder.layer[0].attention.self.query.weight.requires_grad
    assert enc.plm_model.encoder.layer[1].attention.self.query.weight.requires_grad
    os.environ["NRL_PLM_LINEAR"] = "0"
    try:
        plain = PLM(str(tmp_path), [0], 768, True, False, None, 16, 200, 0.2)
    finally:
        del os.environ["NRL_PLM_LINEAR"]
    assert plain.nrl_linears == 0 and list(plain.state_dict().keys()) == list(enc.state_dict().keys())
    plain.load_state_dict(enc.state_dict())


@pytest.mark.parametrize("M,N,K", [(5, 300, 100), (130, 256, 64), (1000, 520, 772), (33, 768, 3072), (70, 200, 300), (257, 1024, 256)])
@pytest.mark.parametrize("engine_name", ["bf16x3", "f32"])
def test_linear_fn_shapes_against_torch(M, N, K, engine_name):
    """ops_blocks.LinearFn (nrl_linear_fwd / nrl_linear_bwd) over odd shapes: outputs narrower and wider than one 256-column
    panel, a last panel that is only partly filled (N = 300, 520), reductions that are not multiples of 32, few rows;
    forward, input gradient, weight / bias gradients, and the frozen-weight form (no weight gradient, input gradient kept)."""
    from newsreclib_amd import _lib
    from newsreclib_amd.ops_blocks import LinearFn
    _lib.set_gemm_engine(engine_name)
    try:
        g = torch.Generator().manual_seed(M * 7 + N)
        x = torch.randn(M, K, generator=g).to(DEV).requires_grad_(True)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).requires_grad_(True)
        b = torch.randn(N, generator=g).to(DEV).requires_grad_(True)
        d = torch.randn(M, N, generator=g).to(DEV)
        ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
        gx, gw, gb = torch.autograd.grad(ref, (x, w, b), d.double())
        out = LinearFn.apply(x, w, b, None)
        ox, ow, ob = torch.autograd.grad(out, (x, w, b), d)
        tol = 1e-4 if engine_name == "bf16x3" else 2e-5
        for got, want, name in ((out, ref, "out"), (ox, gx, "d_x"), (ow, gw, "d_w"), (ob, gb, "d_b")):
            scale = max(1.0, float(want.abs().max()))
            err = float((got.double() - want).abs().max())
            assert err <= tol * scale, (name, err, scale)
        wf, bf_ = w.detach().clone(), b.detach().clone()            # frozen layer
        out2 = LinearFn.apply(x, wf, bf_, None)
        (ox2,) = torch.autograd.grad(out2, (x,), d)
        assert float((ox2 - ox).abs().max()) <= 1e-6 * max(1.0, float(ox.abs().max()))
    finally:
        _lib.set_gemm_engine("bf16x3")


def test_frozen_linear_keeps_its_weight_images_and_rebuilds_them_when_the_weight_changes():
    """``NrlLinear`` over a FROZEN weight builds the matrix-core images once (``nrl_linear_fwd_img`` / ``_bwd_img``,
    ``ops_blocks.FrozenImages``): repeated calls give bit-identical outputs and input gradients, an in-place change of the
    weight (version counter) is seen by the next call, and a trainable weight never uses the cache."""
    from newsreclib_amd.news_encoder import NrlLinear
    torch.manual_seed(3)
    lin = torch.nn.Linear(768, 768).to(DEV)
    nl = NrlLinear(lin)
    for p in nl.parameters():
        p.requires_grad_(False)
    x = torch.randn(300, 768, device=DEV, requires_grad=True)
    g = torch.randn(300, 768, device=DEV)

    def run():
        x.grad = None
        y = nl(x)
        y.backward(g)
        return y.detach().clone(), x.grad.clone()

    y0, dx0 = run()
    assert nl._images._key.get("fwd") is not None and nl._images._key.get("bwd") is not None
    y1, dx1 = run()                                         # second call: images reused
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    ref = torch.nn.functional.linear(x.detach(), lin.weight, lin.bias)
    assert float((y0 - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    with torch.no_grad():
        lin.weight.mul_(2.0)                                # in place: the version counter moves, the images are rebuilt
    y2, dx2 = run()
    ref2 = torch.nn.functional.linear(x.detach(), lin.weight, lin.bias)
    assert float((y2 - ref2).abs().max()) <= 1e-4 * float(ref2.abs().max())
    assert float((dx2 - 2.0 * dx0).abs().max()) <= 1e-4 * float(dx2.abs().max())
    # a trainable weight: no cache (the fused Adam changes it through raw pointers, which no version counter sees)
    lin.weight.requires_grad_(True)
    lin.bias.requires_grad_(True)
    nl(x).sum().backward()
    assert nl._images._key == {}, "a weight seen trainable must drop its cached images (it may be frozen again later)"
    # ... trained through a path no version counter sees (what the fused Adam's raw-pointer writes look like), frozen again:
    # the next frozen call must rebuild, not meet the image of the old values
    lin.weight.data.mul_(0.5)
    lin.weight.requires_grad_(False)
    lin.bias.requires_grad_(False)
    y3, _ = run()
    ref3 = torch.nn.functional.linear(x.detach(), lin.weight, lin.bias)
    assert float((y3 - ref3).abs().max()) <= 1e-4 * float(ref3.abs().max())
    # explicit invalidation after a `.data` write on a frozen weight (ops_blocks.invalidate_frozen_images / FrozenImages.invalidate)
    from newsreclib_amd import ops_blocks
    lin.weight.data.mul_(3.0)
    ops_blocks.invalidate_frozen_images()
    y4, _ = run()
    ref4 = torch.nn.functional.linear(x.detach(), lin.weight, lin.bias)
    assert float((y4 - ref4).abs().max()) <= 1e-4 * float(ref4.abs().max())


def _sdpa_reference(q, k, v, keep, scale, mult=None):
    """fp64 attention over (N, L, H, dh) tensors: softmax((q k^T) scale + key mask) [o dropout multiplier] v."""
    qh, kh, vh = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))          # (N, H, L, dh)
    s = qh @ kh.transpose(-1, -2) * scale
    if keep is not None:
        s = s.masked_fill(~keep.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    if mult is not None:
        p = p * mult.double()
    return (p @ vh).permute(0, 2, 1, 3)


@pytest.mark.parametrize("N,L,H", [(5, 96, 12), (3, 128, 2), (4, 50, 3), (2, 17, 1)])
@pytest.mark.parametrize("masked", [False, True])
def test_body_attention_kernels_match_fp64_attention(N, L, H, masked):
    """``nrl_sdpa_fwd`` / ``_bwd`` (the PLM body's self-attention, heads of 64 over <= 128 tokens, bf16x3 arithmetic) against
    fp64 attention: outputs and all three input gradients, with and without a key-padding mask, full / partly filled row
    blocks; then with attention dropout, against the same reference under the library's own mask (``ops.dropout_mask`` over
    the documented flat index ((n * H + h) * 128 + query) * 128 + key)."""
    from newsreclib_amd import ops, ops_blocks
    torch.manual_seed(N * 1000 + L)
    dh, scale = 64, 64 ** -0.5
    q, k, v = (torch.randn(N, L, H, dh, device=DEV, requires_grad=True) for _ in range(3))
    keep = None
    if masked:
        lens = torch.randint(max(1, L // 3), L + 1, (N,))
        keep = (torch.arange(L)[None, :] < lens[:, None]).to(torch.uint8).to(DEV)
    g = torch.randn(N, L, H, dh, device=DEV)

    def compare(p_drop, seed):
        mult = None
        if p_drop > 0.0:
            km = ops.dropout_mask(N * H * 128 * 128, p_drop, seed, 0, DEV).view(N, H, 128, 128)[:, :, :L, :L]
            mult = km.float() / (1.0 - p_drop)
        for t in (q, k, v):
            t.grad = None
        out = ops_blocks.SdpaFn.apply(q, k, v, keep, scale, p_drop, seed)
        out.backward(g)
        got = [out.detach()] + [t.grad.clone() for t in (q, k, v)]
        q64, k64, v64 = (t.detach().double().requires_grad_(True) for t in (q, k, v))
        ref = _sdpa_reference(q64, k64, v64, keep, scale, mult)
        ref.backward(g.double())
        want = [ref.detach(), q64.grad, k64.grad, v64.grad]
        for name, a, b in zip(("out", "dq", "dk", "dv"), got, want):
            scale_b = max(1e-3, float(b.abs().max()))
            err = float((a.double() - b).abs().max()) / scale_b
            assert err <= 2e-4, (name, p_drop, err)

    compare(0.0, 0)
    compare(0.1, 1234)


def test_plm_body_runs_its_attention_on_the_library_kernels(tmp_path):
    """The HF body of ``news_encoder.PLM`` is switched to the registered attention interface; in eval mode (no dropout) its
    hidden states agree with the same body on the framework's SDPA, padded batch included."""
    from transformers import RobertaConfig, RobertaModel
    from newsreclib_amd.news_encoder import NRL_ATTENTION, PLM
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=300, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                        max_position_embeddings=80, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    RobertaModel(cfg, add_pooling_layer=False).save_pretrained(tmp_path)
    enc = PLM(plm_model=str(tmp_path), frozen_layers=[0], embed_dim=128, use_mhsa=True, apply_reduce_dim=False,
              reduced_embed_dim=None, num_heads=4, query_dim=32, dropout_probability=0.2).to(DEV).eval()
    assert enc.nrl_attention and enc.plm_model.config._attn_implementation == NRL_ATTENTION
    ids = torch.randint(3, 300, (6, 40), device=DEV)
    am = torch.ones_like(ids)
    am[1, 25:] = 0
    am[4, 9:] = 0
    with torch.no_grad():
        ours = enc.plm_model(input_ids=ids, attention_mask=am)[0]
        enc.plm_model.config._attn_implementation = "sdpa"
        ref = enc.plm_model(input_ids=ids, attention_mask=am)[0]
    valid = am.bool()[:, :, None]
    err = float(((ours - ref) * valid).abs().max())
    assert err <= 2e-4 * max(1.0, float(ref.abs().max())), err


@pytest.mark.gpu
@pytest.mark.parametrize("rows,dim,p", [(300, 768, 0.1), (300, 768, 0.0), (37, 64, 0.3), (5, 2048, 0.1), (1000, 300, 0.1), (4099, 768, 0.1)])
@pytest.mark.parametrize("trainable", [True, False])
def test_dropout_add_layernorm_matches_torch(rows, dim, p, trainable):
    """``nrl_dropout_add_layernorm_fwd`` / ``_bwd`` (the line that ends both halves of every PLM body layer: RobertaSelfOutput /
    RobertaOutput, behind text.py:89-91) against fp64 torch: LayerNorm(x * keep / (1 - p) + residual) under the library's own
    keep mask (exported by ``ops.dropout_mask`` for the call's seed), outputs and all four gradients; frozen LayerNorm
    parameters get none; evaluation (no_grad) saves nothing and gives the same output as p = 0."""
    from newsreclib_amd import ops, ops_blocks
    gen = torch.Generator().manual_seed(rows + dim)
    x = torch.randn(rows, dim, generator=gen).to(DEV).requires_grad_(True)
    r = torch.randn(rows, dim, generator=gen).to(DEV).requires_grad_(True)
    gamma = (1.0 + 0.3 * torch.randn(dim, generator=gen)).to(DEV).requires_grad_(trainable)
    beta = (0.2 * torch.randn(dim, generator=gen)).to(DEV).requires_grad_(trainable)
    dy = torch.randn(rows, dim, generator=gen).to(DEV)
    seed = 1234
    y = ops_blocks.DropoutAddLayerNormFn.apply(x, r, gamma, beta, 1e-5, p, seed, None)
    y.backward(dy)
    keep = ops.dropout_mask(rows * dim, p, seed, 0, DEV).view(rows, dim).double() if p > 0 else torch.ones(rows, dim, device=DEV, dtype=torch.float64)
    xd, rd = x.detach().double().requires_grad_(True), r.detach().double().requires_grad_(True)
    gd, bd = gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd * keep / (1.0 - p) + rd, (dim,), gd, bd, 1e-5)
    ref.backward(dy.double())
    assert float((y.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    for got, want, name in ((x.grad, xd.grad, "d_x"), (r.grad, rd.grad, "d_residual")):
        assert float((got.double() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), name
    if trainable:
        for got, want, name in ((gamma.grad, gd.grad, "d_gamma"), (beta.grad, bd.grad, "d_beta")):
            assert float((got.double() - want).abs().max()) <= 2e-4 * max(1.0, float(want.abs().max())), name
    else:
        assert gamma.grad is None and beta.grad is None
    with torch.no_grad():
        ye = ops_blocks.DropoutAddLayerNormFn.apply(x, r, gamma, beta, 1e-5, 0.0, 0, None)
    ref0 = torch.nn.functional.layer_norm(xd + rd, (dim,), gd, bd, 1e-5)
    assert float((ye.double() - ref0).abs().max()) <= 2e-5 * max(1.0, float(ref0.abs().max()))
    # gradients accumulated straight into caller-owned buffers (the flat-gradient trainer's `main_grad`)
    if trainable:
        bufs = (torch.ones(dim, device=DEV), torch.ones(dim, device=DEV))
        x2 = x.detach().requires_grad_(True)
        y2 = ops_blocks.DropoutAddLayerNormFn.apply(x2, r.detach(), gamma, beta, 1e-5, p, seed, bufs)
        gamma.grad = None
        y2.backward(dy)
        assert gamma.grad is None
        assert float((bufs[0].double() - 1.0 - gd.grad).abs().max()) <= 2e-4 * max(1.0, float(gd.grad.abs().max()))


@pytest.mark.gpu
def test_plm_full_width_step_runs_on_this_library(tmp_path):
    """BASELINE configs[3] at full width (roberta-base SHAPE, random init: d = 768, 12 layers, 12 body heads, L = 96, B = 8,
    layers 0-7 frozen as the reference's experiment file has them; text.py:15-109): three train steps -- finite, decreasing loss --
    with every body projection, self-attention and output block on this library's kernels: the framework-fallback counters of
    the GPU path stay 0."""
    from functools import partial

    from transformers import RobertaConfig, RobertaModel

    from newsreclib_amd import _lib
    from newsreclib_amd import news_encoder as ne
    from newsreclib_amd.nrms_module import NRMSModule, prepare_batch
    from newsreclib_amd.synthetic import make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    _lib.set_gemm_engine("bf16x3")
    torch.manual_seed(0)
    cfg = RobertaConfig(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                        max_position_embeddings=514, type_vocab_size=1, pad_token_id=1, bos_token_id=0, eos_token_id=2)
    RobertaModel(cfg, add_pooling_layer=False).save_pretrained(str(tmp_path))
    mod = NRMSModule(
        dataset_attributes=["title", "abstract", "category"], attributes2encode=["title"],
        outputs={"train": [], "val": [], "test": []}, dual_loss_training=False, dual_loss_coef=None,
        loss="cross_entropy_loss", late_fusion=False, temperature=None, use_plm=True, pretrained_embeddings_path=None,
        plm_model=str(tmp_path), frozen_layers=list(range(8)), embed_dim=768, num_heads=16, query_dim=200,
        dropout_probability=0.2, top_k_list=[5, 10], num_categ_classes=18, num_sent_classes=3, save_recs=False,
        recs_fpath=None, optimizer=partial(torch.optim.Adam, lr=1e-4), scheduler=None).to(DEV)
    te = mod.news_encoder.text_encoders["title"]
    assert te.nrl_linears == 72 and te.nrl_attention and te.nrl_output_blocks == 24
    assert te.nrl_ffn_blocks == 12 and te.nrl_attention_blocks == 12 and te.nrl_embeddings >= 2
    trainer = NRMSTrainer(mod, lr=1e-4)
    b = make_batch(8, vocab=50000, mode="fixed", seed=1, L=96, device=DEV)
    for part in ("x_hist", "x_cand"):
        ids = b[part]["title"].clamp_min(3)
        am = torch.ones_like(ids)
        am[:, 70:] = 0                                        # a padded tail: the body attention gets a key-padding mask
        b[part]["title"] = {"input_ids": ids, "attention_mask": am}
    pb = prepare_batch(b)

    def eval_loss():
        mod.eval()
        with torch.no_grad():
            return float(mod.model_step(pb)[0])

    before = eval_loss()
    ne.reset_fallback_calls()
    losses = [float(trainer.step(pb)) for _ in range(3)]
    assert all(np.isfinite(l) for l in losses), losses
    fb_train = dict(ne.FALLBACK_CALLS)
    # the train losses carry a fresh dropout draw each (noise of the same size as three steps' progress): "decreases" is judged
    # on the dropout-free loss of the same batch before and after the three steps
    after = eval_loss()
    assert np.isfinite(before) and after < before, (before, after, losses)
    ne.FALLBACK_CALLS.update(fb_train)
    fb = dict(ne.FALLBACK_CALLS)
    assert fb["linear_cuda"] == 0 and fb["attention"] == 0 and fb["output_block_cuda"] == 0 and fb["embedding_cuda"] == 0, fb
    assert fb["ffn_cuda"] == 0 and fb["attention_block_cuda"] == 0, fb


@pytest.mark.gpu
def test_trainable_linear_keeps_its_images_within_an_optimizer_step_only():
    """``NrlLinear`` over a TRAINABLE weight (layers 8-11 of the PLM body, two encoder calls per step): the forward / backward
    images are built once per optimizer step and rebuilt after ``FusedAdam.begin_step`` (raw-pointer writes: generation) and
    after an in-place update (torch.optim: version counter) -- outputs and gradients always those of the CURRENT weights."""
    from newsreclib_amd import ops_blocks
    from newsreclib_amd.news_encoder import NrlLinear
    torch.manual_seed(5)
    lin = torch.nn.Linear(768, 768).to(DEV)
    nl = NrlLinear(lin)
    assert nl._step_images is not None
    x = torch.randn(200, 768, device=DEV, requires_grad=True)
    g = torch.randn(200, 768, device=DEV)

    def run():
        x.grad = None
        lin.weight.grad = None
        y = nl(x)
        y.backward(g)
        return y.detach().clone(), x.grad.clone(), lin.weight.grad.clone()

    def check(y, dx, dw):
        ref = torch.nn.functional.linear(x.detach(), lin.weight, lin.bias)
        assert float((y - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        assert float((dx - g @ lin.weight.detach()).abs().max()) <= 1e-4 * float(dx.abs().max())
        assert float((dw - g.t() @ x.detach()).abs().max()) <= 1e-4 * float(dw.abs().max())

    # without this library's optimizer in the process nothing of a trainable weight is kept: a writer through `p.data` (apex, EMA
    # swaps) moves neither the version counter nor the generation, and must still meet its own values (advisor, round 4)
    with ops_blocks.no_step_drivers():
        check(*run())
        assert not nl._step_images._key
        lin.weight.data.mul_(0.75)                      # unannounced raw write
        check(*run())

    # an optimizer of this library that owns OTHER parameters gives this layer nothing (round-5 advisor: the permission used to
    # be a process-wide counter), and the permission ends with the optimizer that granted it
    class _Driver:
        pass
    with ops_blocks.no_step_drivers():
        other = torch.nn.Parameter(torch.zeros(4, device=DEV))
        drv = _Driver()
        ops_blocks.register_step_driver(drv, [other])
        assert ops_blocks.step_images_allowed(other) and not ops_blocks.step_images_allowed(lin.weight, lin.bias)
        check(*run())
        assert not nl._step_images._key
        mine = _Driver()
        ops_blocks.register_step_driver(mine, [lin.weight, lin.bias])
        assert ops_blocks.step_images_allowed(lin.weight, lin.bias)
        del mine
        import gc
        gc.collect()
        assert not ops_blocks.step_images_allowed(lin.weight, lin.bias)
    ops_blocks.register_step_driver()                   # (a process-wide promise: what the tests below run under)
    y0, dx0, dw0 = run()
    check(y0, dx0, dw0)
    k0 = dict(nl._step_images._key)
    assert k0.get("fwd") is not None and k0.get("bwd") is not None
    y1, dx1, _ = run()                                  # second encoder call of the step: same images
    assert nl._step_images._key == k0 and torch.equal(y0, y1) and torch.equal(dx0, dx1)
    lin.weight.data.mul_(1.5)                           # a raw write, as the fused Adam does ...
    ops_blocks.next_optimizer_step()                    # ... announced by its begin_step
    check(*run())
    assert nl._step_images._key != k0
    with torch.no_grad():
        lin.weight.mul_(0.5)                            # torch.optim-style in-place update: the version counter moves
    check(*run())


@pytest.mark.gpu
@pytest.mark.parametrize("V,D,pad,n", [(50265, 768, 1, 38400), (514, 768, 1, 4224), (1, 768, None, 3000), (97, 20, 0, 500), (300, 1024, None, 77)])
def test_embedding_gradient_of_any_table_matches_torch(V, D, pad, n):
    """``ops_blocks.EmbeddingFn`` / ``nrl_embedding_grad`` (ABI v14): lookup bit-exact, gradient == ``embedding_dense_backward`` of
    torch (fp64 reference) for the PLM body's three tables' shapes -- heavy duplicates (position ids, the one-row token-type
    table), a padding row in the middle of the id range, dims up to 1024."""
    from newsreclib_amd import ops_blocks
    torch.manual_seed(V + n)
    w = torch.randn(V, D, device=DEV, requires_grad=True)
    zipf = (1.0 / torch.arange(1, V + 1, dtype=torch.float64)) ** 1.1
    ids = torch.multinomial(zipf, n, replacement=True).to(DEV).reshape(-1, 1 if n % 96 else 96)
    if pad is not None and V > 1:
        ids.view(-1)[::7] = pad
    g = torch.randn(*ids.shape, D, device=DEV)
    out = ops_blocks.EmbeddingFn.apply(ids, w, pad, None)
    assert torch.equal(out, torch.nn.functional.embedding(ids, w.detach(), pad))
    out.backward(g)
    w64 = w.detach().double().requires_grad_(True)
    torch.nn.functional.embedding(ids, w64, pad).backward(g.double())
    scale = max(1.0, float(w64.grad.abs().max()))
    assert float((w.grad.double() - w64.grad).abs().max()) <= 2e-5 * scale
    if pad is not None:
        assert float(w.grad[pad].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("M,D,F", [(1000, 768, 3072), (77, 256, 512), (4100, 320, 1024)])
@pytest.mark.parametrize("frozen", [False, True])
def test_feed_forward_block_with_gelu_in_the_epilogues_matches_torch(M, D, F, frozen):
    """``ops_blocks.FfnFn`` (ABI v14: ``nrl_linear_gelu_fwd_img`` / ``nrl_linear_dgrad_gelu_img``): y = gelu(x W1^T + b1) W2^T + b2 with
    the exact GELU and its derivative inside the GEMM epilogues, against fp64 torch: output, input gradient and -- trainable
    block -- all four parameter gradients; a frozen block (layers 0-7 of the PLM body) still passes the gradient to its input."""
    from newsreclib_amd import _lib, ops_blocks
    _lib.set_gemm_engine("bf16x3")
    torch.manual_seed(M + F)
    l1, l2 = torch.nn.Linear(D, F).to(DEV), torch.nn.Linear(F, D).to(DEV)
    if frozen:
        for p in list(l1.parameters()) + list(l2.parameters()):
            p.requires_grad_(False)
    x = torch.randn(M, D, device=DEV, requires_grad=True)
    gy = torch.randn(M, D, device=DEV)
    y = ops_blocks.FfnFn.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, None, None, None)
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    d1, d2 = torch.nn.Linear(D, F).double().to(DEV), torch.nn.Linear(F, D).double().to(DEV)
    d1.load_state_dict({k: v.double() for k, v in l1.state_dict().items()})
    d2.load_state_dict({k: v.double() for k, v in l2.state_dict().items()})
    ref = d2(torch.nn.functional.gelu(d1(xd)))
    ref.backward(gy.double())

    def close(a, b, tol=1e-4):
        assert float((a.double() - b).abs().max()) <= tol * max(1e-6, float(b.abs().max()))

    close(y, ref.detach())
    close(x.grad, xd.grad)
    if not frozen:
        close(l1.weight.grad, d1.weight.grad); close(l1.bias.grad, d1.bias.grad)
        close(l2.weight.grad, d2.weight.grad); close(l2.bias.grad, d2.bias.grad)
    else:
        assert l1.weight.grad is None and l2.weight.grad is None


@pytest.mark.gpu
@pytest.mark.parametrize("M,D,F", [(1000, 768, 3072), (77, 256, 512)])
@pytest.mark.parametrize("frozen", [False, True])
def test_feed_forward_block_with_its_closing_layer_norm_matches_torch(M, D, F, frozen):
    """``ops_blocks.FfnBlockFn`` (ABI v15: ``nrl_linear_dgrad_add_img``): LayerNorm(gelu(x W1^T + b1) W2^T + b2 + x) -- the residual
    branch's gradient is added in the epilogue of the first projection's activation gradient -- against fp64 torch (no dropout)."""
    from newsreclib_amd import _lib, ops_blocks
    _lib.set_gemm_engine("bf16x3")
    torch.manual_seed(M + F + 1)
    l1, l2, ln = torch.nn.Linear(D, F).to(DEV), torch.nn.Linear(F, D).to(DEV), torch.nn.LayerNorm(D).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.3, 0.3)
    mods = (l1, l2, ln)
    if frozen:
        for m in mods:
            for p in m.parameters():
                p.requires_grad_(False)
    x = torch.randn(M, D, device=DEV, requires_grad=True)
    gy = torch.randn(M, D, device=DEV)
    y = ops_blocks.FfnBlockFn.apply(x, l1.weight, l1.bias, l2.weight, l2.bias, ln.weight, ln.bias, float(ln.eps), 0.0, 0, None, None, None)
    y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    d1, d2, dn = torch.nn.Linear(D, F).double().to(DEV), torch.nn.Linear(F, D).double().to(DEV), torch.nn.LayerNorm(D).double().to(DEV)
    for d, l in zip((d1, d2, dn), mods):
        d.load_state_dict({k: v.double() for k, v in l.state_dict().items()})
    ref = dn(d2(torch.nn.functional.gelu(d1(xd))) + xd)
    ref.backward(gy.double())

    def close(a, b, tol=1e-4):
        assert float((a.double() - b).abs().max()) <= tol * max(1e-6, float(b.abs().max()))

    close(y, ref.detach())
    close(x.grad, xd.grad)
    if not frozen:
        for l, d in zip(mods, (d1, d2, dn)):
            close(l.weight.grad, d.weight.grad); close(l.bias.grad, d.bias.grad)
    else:
        assert l1.weight.grad is None and ln.weight.grad is None


@pytest.mark.gpu
@pytest.mark.parametrize("N,L,masked", [(12, 96, True), (5, 40, False)])
@pytest.mark.parametrize("frozen", [False, True])
def test_attention_block_as_one_function_matches_torch(N, L, masked, frozen):
    """``ops_blocks.AttnBlockFn`` (ABI v15: ``nrl_linear3_fwd_img`` / ``nrl_linear3_dgrad_img``): LayerNorm(sdpa(x Wq^T + bq, x Wk^T + bk,
    x Wv^T + bv) Wo^T + bo + x) with ONE GEMM for the three projections each way and the residual gradient in the activation-gradient
    epilogue -- output, input gradient and all ten parameter gradients against fp64 torch (no dropout, key-padding mask)."""
    from newsreclib_amd import _lib, ops_blocks
    _lib.set_gemm_engine("bf16x3")
    D, H = 768, 12
    torch.manual_seed(N * L)
    lq, lk, lv, lo = (torch.nn.Linear(D, D).to(DEV) for _ in range(4))
    ln = torch.nn.LayerNorm(D).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.3, 0.3)
    mods = (lq, lk, lv, lo, ln)
    if frozen:
        for m in mods:
            for p in m.parameters():
                p.requires_grad_(False)
    keep = None
    if masked:
        lens = torch.randint(L // 3, L + 1, (N,), device=DEV)
        keep = (torch.arange(L, device=DEV)[None, :] < lens[:, None]).to(torch.uint8)
    x = torch.randn(N, L, D, device=DEV, requires_grad=True)
    gy = torch.randn(N, L, D, device=DEV)
    params = (lq.weight, lq.bias, lk.weight, lk.bias, lv.weight, lv.bias, lo.weight, lo.bias, ln.weight, ln.bias)
    imgs = ops_blocks.FrozenImages() if frozen else None
    for _ in range(2 if frozen else 1):          # (second pass of a frozen block: the kept images)
        x.grad = None
        y = ops_blocks.AttnBlockFn.apply(x, keep, *params, H, (D // H) ** -0.5, 0.0, 0, float(ln.eps), 0.0, 0, None, imgs, None)
        y.backward(gy)
    xd = x.detach().double().requires_grad_(True)
    dq, dk, dv, do = (torch.nn.Linear(D, D).double().to(DEV) for _ in range(4))
    dn = torch.nn.LayerNorm(D).double().to(DEV)
    dmods = (dq, dk, dv, do, dn)
    for d, l in zip(dmods, mods):
        d.load_state_dict({k: v.double() for k, v in l.state_dict().items()})

    def heads(t):
        return t.view(N, L, H, D // H).transpose(1, 2)

    am = None if keep is None else keep.bool()[:, None, None, :]
    att = torch.nn.functional.scaled_dot_product_attention(heads(dq(xd)), heads(dk(xd)), heads(dv(xd)), attn_mask=am)
    ref = dn(do(att.transpose(1, 2).reshape(N, L, D)) + xd)
    ref.backward(gy.double())

    def close(a, b, tol=2e-4):
        assert float((a.double() - b).abs().max()) <= tol * max(1e-6, float(b.abs().max()))

    close(y, ref.detach())
    close(x.grad, xd.grad)
    if not frozen:
        for l, d in zip(mods, dmods):
            close(l.weight.grad, d.weight.grad)
            if l is lk:     # softmax is shift-invariant: the key bias has NO gradient (fp64: 1e-15) -- rounding noise against the others' scale
                assert float(l.bias.grad.abs().max()) <= 2e-4 * float(dq.bias.grad.abs().max())
            else:
                close(l.bias.grad, d.bias.grad)
    else:
        assert lq.weight.grad is None and lo.weight.grad is None
