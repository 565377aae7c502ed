#!/usr/bin/env python3
"""Times nrl_dropout_add_layernorm_bwd with and without parameter gradients at the config-4 shape (38400 x 768): tools/glue_bwd_time.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newsreclib_amd import _lib
from newsreclib_amd.ops import _stream
lib = _lib.load()
dev = "cuda"
rows, dim = 38400, 768
dy = torch.randn(rows, dim, device=dev); z = torch.randn(rows, dim, device=dev); g = torch.rand(dim, device=dev) + 0.5
mean = torch.randn(rows, device=dev) * 0.1; rstd = torch.rand(rows, device=dev) + 0.5
dx = torch.empty_like(dy); dres = torch.empty_like(dy); dg = torch.zeros(dim, device=dev); db = torch.zeros(dim, device=dev)
def run(params, p):
    _lib.check(lib.nrl_dropout_add_layernorm_bwd(dy.data_ptr(), z.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, dim, p, 7, 0,
                                                 dx.data_ptr() if p > 0 else None, dres.data_ptr(), dg.data_ptr() if params else None,
                                                 db.data_ptr() if params else None, _stream()), "bwd")
for params in (False, True):
    for p in (0.0, 0.1):
        for _ in range(3): run(params, p)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): run(params, p)
        torch.cuda.synchronize()
        print(f"params={params} p={p}: {(time.perf_counter() - t) / 20 * 1e6:.1f} us  (NRL_GLUE_RPW={os.environ.get('NRL_GLUE_RPW', 'default')})")
