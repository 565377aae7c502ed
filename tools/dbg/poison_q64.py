import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nrms_oracle as O
from newsreclib_amd import _lib
from newsreclib_amd.news_encoder import MHSAAddAtt
DEV = "cuda"
_lib.set_gemm_engine("bf16x3")
def run(Q, N, L, p_drop, poison):
    params = O.make_params(97, query_dim=Q, seed=Q + N)
    gen = torch.Generator().manual_seed(Q * 3 + L)
    ids = torch.randint(0, 97, (N, L), generator=gen)
    d_out = torch.randn(N, 300, generator=gen)
    enc = MHSAAddAtt(params[O.EMB_KEY], 300, 15, Q, 0.2)
    enc.load_state_dict({k[len(O.NEWS_PREFIX):]: v for k, v in params.items() if k.startswith(O.NEWS_PREFIX)})
    enc = enc.to(DEV); enc.train(p_drop > 0)
    if poison is not None:
        junk = torch.full((64 << 20,), poison, device=DEV); del junk      # the allocator hands this memory to the workspace
    out = enc(ids.to(DEV), seed=11)
    out.backward(d_out.to(DEV))
    torch.cuda.synchronize()
    bad = {k: int(torch.isnan(p.grad).sum()) for k, p in enc.named_parameters() if torch.isnan(p.grad).any()}
    return bool(torch.isnan(out).any()), bad
for Q in (64, 196, 200, 208):
    for (N, L, p) in ((9, 30, 0.2), (6, 32, 0.0), (5, 16, 0.2), (37, 30, 0.2)):
        for poison in (float("nan"), 1e30):
            print(Q, N, L, p, poison, run(Q, N, L, p, poison))
