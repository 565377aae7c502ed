#!/usr/bin/env python3
"""Probe: the news encoder's forward + backward over 7040 news (B = 128) in ONE call against chunks on two streams -- the
one-news-per-wave kernels run 880 workgroups on 256 CUs = 3.44 rounds, i.e. the last of four rounds is 44 % full in every one of them.
  tools/news_chunk_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newsreclib_amd import ops
dev = torch.device("cuda", 0)
mod = bench.build_module(dev)
enc = mod.news_encoder.text_encoders["title"]
enc.train()
N, L = 7040, 30
torch.manual_seed(0)
ids = torch.randint(1, bench.VOCAB, (N, L), device=dev)
side = torch.cuda.Stream()
gout = torch.randn(N, 300, device=dev)

def one():
    v = enc(ids)
    v.backward(gout)

def chunks(sizes):
    outs, parts, n0 = [], [], 0
    main = torch.cuda.current_stream()
    for i, n in enumerate(sizes):
        part = ids[n0:n0 + n]
        if i % 2 == 1:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                outs.append(enc(part))
        else:
            outs.append(enc(part))
        n0 += n
    main.wait_stream(side)
    v = torch.cat(outs, dim=0)
    v.backward(gout)
    main.wait_stream(side)

def tm(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    for p in enc.parameters(): p.grad = None
    return best

print(f"one call over {N} news: {tm(one):.4f} ms")
for sizes in ((6144, 896), (4096, 2944), (3520, 3520), (2048, 2048, 2048, 896)):
    print(f"chunks {sizes} on two streams: {tm(lambda: chunks(sizes)):.4f} ms")
print(f"one call over 6144 news (three full rounds): ", end="")
ids6 = ids[:6144]; g6 = gout[:6144]
def one6():
    v = enc(ids6); v.backward(g6)
print(f"{tm(one6):.4f} ms;  8192 news (four full rounds): ", end="")
ids8 = torch.randint(1, bench.VOCAB, (8192, L), device=dev); g8 = torch.randn(8192, 300, device=dev)
def one8():
    v = enc(ids8); v.backward(g8)
print(f"{tm(one8):.4f} ms")
