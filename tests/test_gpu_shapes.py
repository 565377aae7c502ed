"""Shape sweep of the two shared blocks through the C ABI against the CPU oracle: token counts on both sides of
every kernel-selection threshold (32 / 64 keys, one or many row tiles), every supported head dim, feature widths
that are not multiples of the GEMM tile, single-news and single-user batches."""
import numpy as np
import pytest
import torch

from oracle import nrms_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f32", "bf16x3"])
def engine(request):
    from newsreclib_amd import _lib
    prev = _lib.get_gemm_engine()
    _lib.set_gemm_engine(request.param)
    yield request.param
    _lib.set_gemm_engine(prev)


def _params(vocab, D, Q, seed):
    return O.make_params(vocab, embed_dim=D, query_dim=Q, seed=seed)


# (n_news, tokens, embed_dim, heads, query_dim)
NEWS_SHAPES = [(1, 1, 32, 2, 8), (3, 5, 64, 4, 12), (17, 31, 96, 2, 100), (4, 32, 100, 5, 20), (5, 33, 128, 2, 36),
               (9, 47, 300, 15, 200), (2, 64, 320, 5, 64), (3, 65, 96, 6, 8), (2, 100, 64, 2, 16), (300, 7, 80, 5, 44),
               (1, 30, 300, 15, 200)]


@pytest.mark.parametrize("shape", NEWS_SHAPES)
@pytest.mark.parametrize("p_drop", [0.0, 0.3])
def test_news_encoder_shapes(shape, p_drop, engine):
    from newsreclib_amd.news_encoder import MHSAAddAtt
    N, L, D, heads, Q = shape
    V = 23
    params = _params(V, D, Q, seed=N + L)
    rng = np.random.default_rng(L)
    ids = torch.from_numpy(rng.integers(0, V, (N, L)))
    enc = MHSAAddAtt(params[O.EMB_KEY], D, heads, Q, 0.2)
    enc.load_state_dict({k[len(O.NEWS_PREFIX):]: v for k, v in params.items() if k.startswith(O.NEWS_PREFIX)})
    enc = enc.cuda()
    enc.train(p_drop > 0)
    enc.dropout.p = p_drop
    out = enc(ids.cuda(), seed=99)
    op = {k: v.clone().requires_grad_(True) for k, v in params.items() if k.startswith(O.NEWS_PREFIX)}
    m1 = m2 = None
    if p_drop > 0:
        m1 = O.dropout_multiplier(99, 0, p_drop, (N, L, D))
        m2 = O.dropout_multiplier(99, 1, p_drop, (N, L, D))
    ref = O.news_encoder_fwd(ids, op, heads, m1, m2)
    ftol, gtol = (2e-5, 2e-4) if engine == "f32" else (1e-4, 5e-4)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 5 * ftol
    d_out = torch.from_numpy(rng.standard_normal((N, D)).astype(np.float32))
    out.backward(d_out.cuda())
    ref.backward(d_out)
    for k, p in enc.named_parameters():
        want = op[O.NEWS_PREFIX + k].grad.clone()
        if k == "embedding_layer.weight":
            want[0].zero_()
        scale = max(1.0, float(want.abs().max()))
        assert float((p.grad.cpu() - want).abs().max()) <= gtol * scale, k


# (users, history slots, embed_dim, heads, query_dim): attention runs across the USERS
USER_SHAPES = [(1, 1, 32, 2, 8), (2, 50, 300, 15, 200), (31, 3, 64, 4, 12), (33, 2, 96, 2, 20), (64, 2, 80, 5, 16),
               (65, 3, 128, 2, 8), (200, 1, 48, 3, 24)]


@pytest.mark.parametrize("shape", USER_SHAPES)
def test_user_encoder_shapes(shape, engine):
    from newsreclib_amd.user_encoder import UserEncoder
    B, H, D, heads, Q = shape
    params = _params(8, D, Q, seed=B)
    rng = np.random.default_rng(B + H)
    hist = torch.from_numpy((rng.standard_normal((B, H, D)) * 0.5).astype(np.float32))
    enc = UserEncoder(D, heads, Q)
    enc.load_state_dict({k[len(O.USER_PREFIX):]: v for k, v in params.items() if k.startswith(O.USER_PREFIX)})
    enc = enc.cuda()
    hd = hist.cuda().requires_grad_(True)
    out = enc(hd)
    op = {k: v.clone().requires_grad_(True) for k, v in params.items() if k.startswith(O.USER_PREFIX)}
    hl = hist.clone().requires_grad_(True)
    ref = O.user_encoder_fwd(hl, op, heads)
    ftol, gtol = (2e-5, 2e-4) if engine == "f32" else (1e-4, 5e-4)
    assert float((out.detach().cpu() - ref.detach()).abs().max()) <= 5 * ftol
    d_out = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32))
    out.backward(d_out.cuda())
    ref.backward(d_out)
    assert float((hd.grad.cpu() - hl.grad).abs().max()) <= gtol * max(1.0, float(hl.grad.abs().max()))
    for k, p in enc.named_parameters():
        want = op[O.USER_PREFIX + k].grad
        scale = max(1.0, float(want.abs().max()))
        # (the key third of in_proj_bias has an exactly-zero true gradient: both sides are rounding noise)
        assert float((p.grad.cpu() - want).abs().max()) <= gtol * scale, k


@pytest.mark.gpu
def test_raw_stream_handle_follows_the_current_stream():
    """ops._stream() (the handle every C-ABI call is issued on) through torch's private fast path == the public
    ``torch.cuda.current_stream().cuda_stream``, on the default stream and inside ``torch.cuda.stream(side)``."""
    import torch
    from newsreclib_amd import ops
    assert ops._stream() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert ops._stream() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
    assert ops._stream() == torch.cuda.current_stream().cuda_stream
