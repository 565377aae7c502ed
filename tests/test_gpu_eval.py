"""Evaluation path: device news table, batches built by index, encode-once scoring.
The cached path must reproduce the module's own forward bit-for-bit (same kernels, row-independent news
encoder) and the CPU oracle within the score contract."""
import numpy as np
import pytest
import torch

from oracle import nrms_oracle as O
from tests.helpers import build_module

pytestmark = pytest.mark.gpu


def _impressions(rng, n_imp, n_news, max_hist=12, max_cand=40):
    imps = []
    for i in range(n_imp):
        nh, nc = int(rng.integers(1, max_hist + 1)), int(rng.integers(2, max_cand + 1))
        lab = np.zeros(nc, dtype=np.float32)
        lab[rng.integers(0, nc)] = 1.0
        imps.append({"hist": torch.from_numpy(rng.integers(1, n_news, nh)), "cand": torch.from_numpy(rng.integers(1, n_news, nc)),
                     "labels": torch.from_numpy(lab), "user_idx": torch.tensor(i % 7)})
    return imps


def _table(rng, n_news, vocab, L=30, n_categ=19):
    lens = rng.integers(3, L + 1, n_news)
    ids = rng.integers(1, vocab, (n_news, L))
    ids[np.arange(L)[None, :] >= lens[:, None]] = 0
    ids[0] = 0
    return {"title": torch.from_numpy(ids), "category": torch.from_numpy(rng.integers(1, n_categ, n_news)),
            "sentiment": torch.from_numpy(rng.integers(1, 4, n_news))}


@pytest.mark.parametrize("engine_name", ["f32", "bf16x3"])
def test_cached_scores_equal_uncached_forward_and_oracle(engine_name):
    from newsreclib_amd import _lib
    from newsreclib_amd.evaluation import DeviceNewsTable, NewsVectorCache
    prev = _lib.get_gemm_engine()
    _lib.set_gemm_engine(engine_name)
    try:
        rng = np.random.default_rng(4)
        vocab, n_news = 300, 157
        params = O.make_params(vocab, seed=8)
        mod = build_module(params).eval()
        table = DeviceNewsTable(_table(rng, n_news, vocab))
        cache = NewsVectorCache(mod, table, chunk=64)        # several chunks, last one partial
        vec = cache.build()
        assert vec.shape == (n_news, 300)
        imps = _impressions(rng, 9, n_news)
        hist = torch.cat([i["hist"] for i in imps])
        cand = torch.cat([i["cand"] for i in imps])
        hs = torch.tensor([len(i["hist"]) for i in imps])
        cs = torch.tensor([len(i["cand"]) for i in imps])
        labels = torch.cat([i["labels"] for i in imps])
        got = cache.scores(hist, hs, cand, cs)
        batch = table.build_batch(hist, hs, cand, cs, labels)
        with torch.no_grad():
            ref = mod.forward(batch)
        assert got.shape == (9, int(cs.max()))
        assert torch.equal(got, ref)                          # encode-once == encode-per-batch, bit for bit
        # oracle (CPU) on the materialised batch
        cpu_batch = {"x_hist": {"title": batch["x_hist"]["title"].cpu()}, "x_cand": {"title": batch["x_cand"]["title"].cpu()},
                     "batch_hist": batch["batch_hist"].cpu(), "batch_cand": batch["batch_cand"].cpu(),
                     "labels": labels, "batch_size": 9}
        orc = O.nrms_forward(cpu_batch, params)["scores"]
        assert float((got.cpu() - orc).abs().max()) <= (2e-5 if engine_name == "f32" else 2e-4)
        # the model_step tuple from indices == the module's model_step on the built batch
        a = cache.model_step(hist, hs, cand, cs, labels)
        with torch.no_grad():
            b = mod.model_step(batch)
        assert abs(float(a[0]) - float(b[0])) <= 1e-6 and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
        assert torch.equal(a[5], b[5]) and torch.equal(a[7], b[7])
    finally:
        _lib.set_gemm_engine(prev)


def test_evaluate_impressions_metrics_and_full_impression_width():
    """Full impressions (up to 300 candidates, no negative sampling): metrics from the cached path equal the
    metrics computed from per-batch forwards."""
    from newsreclib_amd.evaluation import DeviceNewsTable, NewsVectorCache, evaluate_impressions
    from newsreclib_amd.metrics import aspect_metrics, ranking_metrics
    rng = np.random.default_rng(11)
    vocab, n_news = 400, 500
    mod = build_module(O.make_params(vocab, seed=2)).eval()
    table = DeviceNewsTable(_table(rng, n_news, vocab))
    cache = NewsVectorCache(mod, table)
    imps = _impressions(rng, 40, n_news, max_hist=50, max_cand=300)
    logs = evaluate_impressions(cache, imps, batch_size=16, num_categ_classes=19, num_sent_classes=4)
    for k in ("auc", "mrr", "ndcg@5", "ndcg@10", "categ_div@5", "categ_pers@10", "sent_div@10", "sent_pers@5", "loss"):
        assert k in logs and np.isfinite(logs[k]), k
    # reference flow: per-batch forward over the built batches, same batch composition
    outs = []
    for lo in range(0, 40, 16):
        ch = imps[lo:lo + 16]
        batch = table.build_batch(torch.cat([i["hist"] for i in ch]), torch.tensor([len(i["hist"]) for i in ch]),
                                  torch.cat([i["cand"] for i in ch]), torch.tensor([len(i["cand"]) for i in ch]),
                                  torch.cat([i["labels"] for i in ch]))
        with torch.no_grad():
            outs.append(mod.model_step(batch))
    cat = lambda j: torch.cat([o[j] for o in outs])  # noqa: E731
    ref = ranking_metrics(cat(1), cat(2), cat(3), (5, 10))
    ref.update(aspect_metrics(cat(1), cat(5), cat(7), cat(3), cat(4), 19, (5, 10), prefix="categ"))
    for k, v in ref.items():
        assert abs(logs[k] - v) <= 1e-6, (k, logs[k], v)


def test_lstur_cached_scores_equal_uncached_forward():
    from newsreclib_amd.evaluation import DeviceNewsTable, NewsVectorCache
    from oracle.lstur_oracle import make_lstur_params
    from tests.helpers import build_lstur_module
    rng = np.random.default_rng(6)
    cfg = dict(vocab=120, n_categ=7, n_users=9, D=48, F=48, W=3, Q=32, categ_dim=16, text_attrs=("title", "abstract"),
               text_order=("title", "abstract"), method="ini", p_drop=0.2, p_mask=0.5)
    params = make_lstur_params(cfg["vocab"], cfg["n_categ"], cfg["n_users"], 48, 48, 3, 32, 16, seed=5)
    mod = build_lstur_module(cfg, params).eval()
    n_news = 90
    attrs = _table(rng, n_news, cfg["vocab"], L=12, n_categ=7)
    attrs["abstract"] = _table(rng, n_news, cfg["vocab"], L=20)["title"]
    table = DeviceNewsTable(attrs)
    cache = NewsVectorCache(mod, table, chunk=32)
    imps = _impressions(rng, 6, n_news, max_hist=8, max_cand=12)
    hist, cand = torch.cat([i["hist"] for i in imps]), torch.cat([i["cand"] for i in imps])
    hs, cs = torch.tensor([len(i["hist"]) for i in imps]), torch.tensor([len(i["cand"]) for i in imps])
    uidx = torch.stack([i["user_idx"] for i in imps])
    got = cache.scores(hist, hs, cand, cs, uidx)
    batch = table.build_batch(hist, hs, cand, cs, torch.cat([i["labels"] for i in imps]), user_idx=uidx)
    with torch.no_grad():
        ref = mod.forward(batch)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("model", ["naml", "tanr", "cen", "mins"])
def test_sibling_models_cached_scores_equal_uncached_forward(model):
    """Encode-once evaluation through every sibling module mirror: scoring from the cached news vectors is
    bit-identical to the module's own forward on the same batch."""
    from newsreclib_amd.evaluation import DeviceNewsTable, NewsVectorCache
    from tests import helpers as H
    rng = np.random.default_rng(11)
    n_news, vocab = 90, 120
    if model == "naml":
        from oracle.naml_oracle import make_naml_params
        cfg = dict(vocab=vocab, n_categ=7, D=48, F=48, W=3, Q=32, categ_dim=16, text_attrs=("title", "abstract"),
                   text_order=("title", "abstract"), p_drop=0.2)
        mod = H.build_naml_module(cfg, make_naml_params(vocab, 7, 48, 48, 3, 32, 16, seed=5))
    elif model == "tanr":
        from oracle.tanr_oracle import make_tanr_params
        cfg = dict(vocab=vocab, n_categ=7, D=48, F=48, W=3, Q=32, p_drop=0.2, coef=0.2)
        mod = H.build_tanr_module(cfg, make_tanr_params(vocab, 7, 48, 48, 3, 32, seed=5))
    elif model == "cen":
        from oracle.cen_news_rec_oracle import make_cen_news_rec_params
        cfg = dict(vocab=vocab, D=40, F=48, W=3, Q=32, heads=3, recent=3, p_drop=0.2, late_fusion=False)
        mod = H.build_cen_module(cfg, make_cen_news_rec_params(vocab, 40, 48, 3, 32, seed=5))
    else:
        from oracle.mins_oracle import make_mins_params
        cfg = dict(vocab=vocab, n_categ=7, D=48, Q=32, categ_dim=16, heads=3, channels=4,
                   text_attrs=("title", "abstract"), text_order=("title", "abstract"), p_drop=0.2)
        mod = H.build_mins_module(cfg, make_mins_params(vocab, 7, 48, 32, 16, 4, seed=5))
    mod = mod.eval()
    attrs = _table(rng, n_news, vocab, L=12, n_categ=7)
    attrs["abstract"] = _table(rng, n_news, vocab, L=20)["title"]
    table = DeviceNewsTable(attrs)
    cache = NewsVectorCache(mod, table, chunk=32)
    imps = _impressions(rng, 6, n_news, max_hist=8, max_cand=12)
    hist, cand = torch.cat([i["hist"] for i in imps]), torch.cat([i["cand"] for i in imps])
    hs, cs = torch.tensor([len(i["hist"]) for i in imps]), torch.tensor([len(i["cand"]) for i in imps])
    got = cache.scores(hist, hs, cand, cs)
    batch = table.build_batch(hist, hs, cand, cs, torch.cat([i["labels"] for i in imps]))
    with torch.no_grad():
        out = mod.forward(batch)
    ref = out[0] if isinstance(out, tuple) else out
    assert torch.equal(got, ref)


# ---- evaluation forwards from the per-token q|k|v table (ABI v16) ------------------------------------
def _title_ids(rng, n_news, vocab, L, short_frac=0.5):
    """MIND-like titles: a share of them shorter than 15 tokens (pad-row sharing's "short" class), ids up to vocab - 1."""
    lens = np.where(rng.random(n_news) < short_frac, rng.integers(1, min(15, L) + 1, n_news), rng.integers(min(15, L), L + 1, n_news))
    ids = rng.integers(1, vocab, (n_news, L))
    ids[np.arange(L)[None, :] >= lens[:, None]] = 0
    ids[0, :] = 0                       # an all-padding news
    ids[-1, : min(L, 3)] = vocab - 1    # the last vocabulary row
    return torch.from_numpy(ids)


def _text_encoder(vocab, seed):
    from newsreclib_amd.news_encoder import MHSAAddAtt
    params = O.make_params(vocab, seed=seed)
    enc = MHSAAddAtt(params[O.EMB_KEY], 300, 15, 200, 0.2)
    enc.load_state_dict({k[len(O.NEWS_PREFIX):]: v for k, v in params.items() if k.startswith(O.NEWS_PREFIX)})
    return enc.cuda().eval(), params


@pytest.mark.parametrize("n_news,L,vocab", [(70, 30, 500), (9, 17, 33), (5, 32, 64), (300, 30, 70_001), (1, 16, 97), (13, 5, 40)])
@pytest.mark.parametrize("pad_share", [True, False])
def test_token_table_forward_is_bit_identical(n_news, L, vocab, pad_share, monkeypatch):
    """Row arithmetic of the in-projection is row-independent, so the q|k|v table holds, per vocabulary id, the bits the fused
    forward computes for every position with that id: news vectors from the table must be EQUAL to today's evaluation forward
    (short and long news, pad-row sharing on and off, a vocabulary that does not fill its last 32-id run, L at 16 / 17 / 32)."""
    from newsreclib_amd import _lib
    from newsreclib_amd.news_encoder import MHSAAddAtt
    _lib.set_gemm_engine("bf16x3")
    _lib.set_option("news_pad_share", pad_share)
    try:
        rng = np.random.default_rng(n_news + L)
        enc, params = _text_encoder(vocab, seed=L)
        ids = _title_ids(rng, n_news, vocab, L).cuda()
        with torch.no_grad():
            monkeypatch.setenv("NRL_TOKEN_TABLE", "0")
            plain = enc(ids)
            monkeypatch.delenv("NRL_TOKEN_TABLE")
            before = dict(MHSAAddAtt.TOKEN_TABLE_USES)
            with enc.token_table():
                tab1 = enc(ids)
                tab2 = enc(ids.flip(0))                    # a second batch from the same table
            assert MHSAAddAtt.TOKEN_TABLE_USES["built"] == before["built"] + 1
            assert MHSAAddAtt.TOKEN_TABLE_USES["forwards"] == before["forwards"] + 2
    finally:
        _lib.set_option("news_pad_share", True)
    assert torch.isfinite(plain).all()
    assert torch.equal(tab1, plain)
    assert torch.equal(tab2.flip(0), plain)
    # ... and the oracle, within the contract
    ref = O.news_encoder_fwd(ids.cpu(), {k: v for k, v in params.items() if k.startswith(O.NEWS_PREFIX)}, 15)
    assert float((tab1.cpu() - ref).abs().max()) <= 2e-4


def test_token_table_follows_the_weights_and_the_amortisation_rule(monkeypatch):
    """Automatic route (eval mode + no_grad): frozen weights qualify at once, trainable ones only under this library's optimizer;
    the table is built once the positions seen under the current weights reach the vocabulary size, rebuilt when any parameter
    changes (version counter), never used in train mode, with gradients enabled, or under NRL_TOKEN_TABLE=0."""
    from newsreclib_amd import _lib, ops_blocks
    from newsreclib_amd.news_encoder import MHSAAddAtt
    _lib.set_gemm_engine("bf16x3")
    vocab = 2000
    rng = np.random.default_rng(0)
    enc, _ = _text_encoder(vocab, seed=3)
    ids = _title_ids(rng, 20, vocab, 30).cuda()            # 600 positions per forward, V = 2000: the 4th forward builds
    uses = MHSAAddAtt.TOKEN_TABLE_USES

    def run():
        with torch.no_grad():
            return enc(ids)

    monkeypatch.setenv("NRL_TOKEN_TABLE", "0")
    plain = run()
    monkeypatch.delenv("NRL_TOKEN_TABLE")
    b0, f0 = uses["built"], uses["forwards"]
    if not ops_blocks.step_images_allowed():               # trainable weights, no optimizer of this library alive: not eligible
        for _ in range(5):
            assert torch.equal(run(), plain)
        assert (uses["built"], uses["forwards"]) == (b0, f0)
    for p in enc.parameters():
        p.requires_grad_(False)
    outs = [run() for _ in range(6)]
    assert all(torch.equal(o, plain) for o in outs)
    assert uses["built"] == b0 + 1 and uses["forwards"] == f0 + 3          # forwards 1-3 projected, 4-6 gathered
    # gradients enabled / train mode: the ordinary path
    assert torch.equal(enc(ids), plain) and uses["forwards"] == f0 + 3
    enc.train()
    with torch.no_grad():
        enc(ids, seed=1)
    enc.eval()
    assert uses["forwards"] == f0 + 3
    # an in-place update of ANY parameter moves the key: the stale table is not used, a new one is built after V more positions
    with torch.no_grad():
        enc.multihead_attention.in_proj_bias.add_(0.25)
        monkeypatch.setenv("NRL_TOKEN_TABLE", "0")
        plain2 = enc(ids)
        monkeypatch.delenv("NRL_TOKEN_TABLE")
    assert not torch.equal(plain2, plain)
    outs = [run() for _ in range(5)]
    assert all(torch.equal(o, plain2) for o in outs)
    assert uses["built"] == b0 + 2 and uses["forwards"] == f0 + 3 + 2
    # the embedding table too
    with torch.no_grad():
        enc.embedding_layer.weight[7].mul_(2.0)
        with enc.token_table():
            t3 = enc(ids)
        monkeypatch.setenv("NRL_TOKEN_TABLE", "0")
        assert torch.equal(t3, enc(ids))
    assert uses["built"] == b0 + 3


def test_token_table_c_abi_contract():
    """Raw C ABI: no table outside the fused geometry (bytes == 0, build refused), a buffer that is too small is NRL_E_WORKSPACE,
    the exact-fp32 engine is refused -- never a silent other path."""
    import ctypes

    from newsreclib_amd import _lib, ops
    lib = _lib.load()
    _lib.set_gemm_engine("bf16x3")
    assert lib.nrl_token_table_supported(30, 300, 15, 200) == 1
    assert lib.nrl_token_table_supported(33, 300, 15, 200) == 0 and lib.nrl_token_table_supported(30, 96, 6, 32) == 0
    assert lib.nrl_token_table_bytes(1000, 96, 6, 32) == 0
    nbytes = lib.nrl_token_table_bytes(1000, 300, 15, 200)
    assert nbytes >= ((1000 + 31) // 32) * 15 * 32 * 64 * 4
    enc, _ = _text_encoder(1000, seed=1)
    prm = [p.detach() for p in enc._params()]
    bp = ops._block_params(prm[1:], 15, 2, _lib.options_word())
    st = torch.cuda.current_stream().cuda_stream
    small = torch.empty(nbytes - 256, dtype=torch.uint8, device="cuda")
    assert lib.nrl_token_table_build(ctypes.byref(bp), prm[0].data_ptr(), 1000, small.data_ptr(), small.numel(), st) == -2
    bp32 = ops._block_params(prm[1:], 15, 1, _lib.options_word())
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    assert lib.nrl_token_table_build(ctypes.byref(bp32), prm[0].data_ptr(), 1000, buf.data_ptr(), buf.numel(), st) == -1
    assert b"bf16x3" in lib.nrl_last_error()
    assert lib.nrl_token_table_build(ctypes.byref(bp), prm[0].data_ptr(), 1000, buf.data_ptr(), buf.numel(), st) == 0
    ids = torch.randint(0, 1000, (7, 30), device="cuda")
    out = torch.empty(7, 300, device="cuda")
    wsb = lib.nrl_news_encoder_fwd_table_workspace_bytes(7, 30, 15)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    assert lib.nrl_news_encoder_fwd_table(ctypes.byref(bp), buf.data_ptr(), buf.numel(), 1000, ids.data_ptr(), 7, 30, out.data_ptr(),
                                          ws.data_ptr(), wsb - 256, st) == -2
    assert lib.nrl_news_encoder_fwd_table(ctypes.byref(bp), buf.data_ptr(), buf.numel(), 1000, ids.data_ptr(), 7, 30, out.data_ptr(),
                                          ws.data_ptr(), wsb, st) == 0
    with torch.no_grad():
        import os
        os.environ["NRL_TOKEN_TABLE"] = "0"
        try:
            assert torch.equal(out, enc(ids))
        finally:
            del os.environ["NRL_TOKEN_TABLE"]


def test_news_vector_cache_and_epoch_hooks_use_the_token_table():
    """`NewsVectorCache.build` (one pass over the corpus) and the module's validation / test epochs run their text encoder from
    the table: same vectors and scores, bit for bit, as with it switched off."""
    import os

    from newsreclib_amd import _lib
    from newsreclib_amd.evaluation import DeviceNewsTable, NewsVectorCache
    from newsreclib_amd.news_encoder import MHSAAddAtt
    _lib.set_gemm_engine("bf16x3")
    rng = np.random.default_rng(5)
    vocab, n_news = 900, 333
    params = O.make_params(vocab, seed=2)
    mod = build_module(params).eval()
    table = DeviceNewsTable(_table(rng, n_news, vocab))
    uses = MHSAAddAtt.TOKEN_TABLE_USES
    b0 = uses["built"]
    vec = NewsVectorCache(mod, table, chunk=100).build()
    assert uses["built"] == b0 + 1
    te = mod.news_encoder.text_encoders["title"]
    orig = te.token_table
    import contextlib
    te.token_table = lambda: contextlib.nullcontext()       # the same pass with the table out of the way
    os.environ["NRL_TOKEN_TABLE"] = "0"
    try:
        ref = NewsVectorCache(mod, table, chunk=100).build()
    finally:
        del os.environ["NRL_TOKEN_TABLE"]
        te.token_table = orig
    assert torch.equal(vec, ref)
    # epoch hooks of the drop-in module
    imps = _impressions(rng, 6, n_news)
    hist, cand = torch.cat([i["hist"] for i in imps]), torch.cat([i["cand"] for i in imps])
    hs, cs = torch.tensor([len(i["hist"]) for i in imps]), torch.tensor([len(i["cand"]) for i in imps])
    batch = table.build_batch(hist, hs, cand, cs, torch.cat([i["labels"] for i in imps]))
    f0 = uses["forwards"]
    mod.on_validation_epoch_start()
    with torch.no_grad():
        mod.validation_step(batch, 0)
        s1 = mod.forward(batch)
    mod.on_validation_epoch_end()
    assert uses["forwards"] >= f0 + 2 and not te._tt_pinned
    os.environ["NRL_TOKEN_TABLE"] = "0"
    try:
        with torch.no_grad():
            assert torch.equal(s1, mod.forward(batch))
    finally:
        del os.environ["NRL_TOKEN_TABLE"]


def test_token_table_under_inference_mode(monkeypatch):
    """Lightning runs validation / test steps under ``torch.inference_mode()`` by default: the table route (key built from the
    parameters' version counters, buffers allocated as inference tensors and reused later under ``no_grad``) must work there."""
    from newsreclib_amd import _lib
    _lib.set_gemm_engine("bf16x3")
    rng = np.random.default_rng(11)
    enc, _ = _text_encoder(777, seed=6)
    ids = _title_ids(rng, 40, 777, 30).cuda()
    monkeypatch.setenv("NRL_TOKEN_TABLE", "0")
    with torch.no_grad():
        plain = enc(ids)
    monkeypatch.delenv("NRL_TOKEN_TABLE")
    with torch.inference_mode():
        with enc.token_table():
            a = enc(ids)
    with torch.no_grad(), enc.token_table():                 # the buffer built under inference mode serves the next epoch too
        b = enc(ids)
    assert torch.equal(a, plain) and torch.equal(b, plain)
