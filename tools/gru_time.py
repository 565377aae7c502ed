#!/usr/bin/env python3
"""Wall time of the GRU forward / backward alone at the LSTUR (B=128, T=50, 700) and MINS (768 x 50 x 52) shapes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from newsreclib_amd import ops_lstur  # noqa: E402


def main():
    for B, T, D in ((128, 50, 700), (768, 50, 52), (128, 20, 400)):
        g = torch.Generator(device="cuda").manual_seed(0)
        r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device="cuda") * sc)  # noqa: E731
        x = r(B, T, D, sc=0.5).requires_grad_(True)
        w_ih, w_hh = r(3 * D, D, sc=D ** -0.5).requires_grad_(True), r(3 * D, D, sc=D ** -0.5).requires_grad_(True)
        b_ih, b_hh = r(3 * D, sc=0.05).requires_grad_(True), r(3 * D, sc=0.05).requires_grad_(True)
        lengths = torch.randint(1, T + 1, (B,), device="cuda")
        d_out = r(B, D)

        def fwd():
            return ops_lstur.GruFn.apply(x, lengths, None, w_ih, w_hh, b_ih, b_hh, None)

        def both():
            fwd().backward(d_out)

        for fn, name in ((fwd, "fwd"), (both, "fwd+bwd")):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            print(f"GRU B={B} T={T} D={D} {name:8s}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
