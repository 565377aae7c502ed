#!/usr/bin/env python3
"""Probe: the PLM body's projection shapes on this library's row-panel GEMM (nrl_linear_fwd) against hipBLASLt bf16 GEMMs with fp32
output over K-concatenated (hi, lo) operands ([A_hi | A_lo | A_hi] x [B_hi ; B_hi ; B_lo] = the same three products per element)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newsreclib_amd import _lib, ops

def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def split3(x):          # (M, K) fp32 -> (M, 3K) bf16 [hi | lo | hi]
    hi = x.bfloat16(); lo = (x - hi.float()).bfloat16()
    return torch.cat([hi, lo, hi], dim=1)

def split3w(w):         # (N, K) fp32 -> (3K, N) bf16 [hi ; hi ; lo]
    hi = w.bfloat16(); lo = (w - hi.float()).bfloat16()
    return torch.cat([hi, hi, lo], dim=1).t().contiguous()

dev = "cuda"
for M in (38400, 3840):
    for N, K in ((768, 768), (3072, 768), (768, 3072), (2304, 768)):
        a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
        t_nrl = tm(lambda: ops.linear(a, w, b))
        # ... and as the config-4 step runs it: the weight's fragment image kept (frozen layers: across steps; trainable layers:
        # across the calls of one optimizer step) -- nrl_linear_fwd_img with image_ready = 1 after one build
        lib = _lib.load()
        ws = torch.empty(max(lib.nrl_linear_workspace_bytes(N, K), 256), dtype=torch.uint8, device=dev)
        c_img = torch.empty((M, N), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        def with_image(ready):
            _lib.check(lib.nrl_linear_fwd_img(a.data_ptr(), w.data_ptr(), b.data_ptr(), M, N, K, c_img.data_ptr(), ws.data_ptr(), ws.numel(),
                                              ready, st), "nrl_linear_fwd_img")
        with_image(0)
        t_img = tm(lambda: with_image(1))
        assert torch.equal(c_img, ops.linear(a, w, b))
        a3, w3 = split3(a), split3w(w)
        try:
            t_mm = tm(lambda: torch.mm(a3, w3, out_dtype=torch.float32))
            c = torch.mm(a3, w3, out_dtype=torch.float32) + b
            ref = (a.double() @ w.double().t() + b.double())
            err = float((c.double() - ref).abs().max() / ref.abs().max())
            err_n = float((ops.linear(a, w, b).double() - ref).abs().max() / ref.abs().max())
        except Exception as e:
            t_mm, err, err_n = float("nan"), str(e)[:80], None
        t_split = tm(lambda: split3(a))
        gf = 2.0 * M * N * K * 3 / 1e9
        print(f"M={M} N={N} K={K}: nrl_linear {t_nrl:.3f} ms ({gf / t_nrl:.0f} TF-bf16/s), image kept {t_img:.3f} ms ({gf / t_img:.0f}) | hipBLASLt 3K-concat {t_mm:.3f} ms ({gf / t_mm if t_mm == t_mm else 0:.0f}) "
              f"+ torch split {t_split:.3f} ms | rel err lt {err} nrl {err_n}", flush=True)
