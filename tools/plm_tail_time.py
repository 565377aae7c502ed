#!/usr/bin/env python3
"""Times the PLM text-encoder TAIL (BASELINE config 3: roberta-base hidden states d=768, 16 heads, L=96) on one
GPU: dropout -> seq-first multi-head attention ACROSS THE NEWS of the call (the reference's batch_first quirk:
S = number of news, one attention per token position and head) -> dropout -> additive attention, forward +
backward through the C ABI.  The transformer body itself is third-party (HF on PyTorch-ROCm) and not timed."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--news", type=int, default=7040)
    ap.add_argument("--len", type=int, default=96)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    from newsreclib_amd import ops
    torch.manual_seed(0)
    N, L, D, H, Q = args.news, args.len, args.dim, args.heads, 200
    dev = "cuda"
    hidden = (torch.randn(N, L, D, device=dev) * 0.5).requires_grad_(True)
    params = [torch.randn(3 * D, D, device=dev) * D ** -0.5, torch.zeros(3 * D, device=dev),
              torch.randn(D, D, device=dev) * D ** -0.5, torch.zeros(D, device=dev),
              torch.randn(Q, D, device=dev) * D ** -0.5, torch.zeros(Q, device=dev), torch.randn(Q, device=dev) * 0.1]
    params = [p.requires_grad_(True) for p in params]
    d_out = torch.randn(N, D, device=dev)

    def step():
        out = ops.UserEncoderFn.apply(hidden, *params, H, None, 0.2, 7)
        out.backward(d_out)
        hidden.grad = None
        for p in params:
            p.grad = None

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    attn_flops = 4.0 * N * N * D * L          # QK^T + PV, forward
    print(f"PLM tail N={N} L={L} D={D} heads={H}: {dt * 1e3:.1f} ms fwd+bwd "
          f"(seq-first attention forward alone is {attn_flops / 1e12:.1f} TFLOP: S = {N} news per (token, head))")


if __name__ == "__main__":
    main()
