#!/usr/bin/env python3
"""configs[0] (B = 32, V = 70k) train step, events over 200 steps: tools/b32_step_time.py [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from newsreclib_amd.nrms_module import attach_layout
from newsreclib_amd.synthetic import make_batch
from newsreclib_amd.trainer import NRMSTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
mod = bench.build_module(dev)
tr = NRMSTrainer(mod, lr=1e-4)
bs = [attach_layout(make_batch(B, 70000, "fixed", seed=1234 + i, device=dev)) for i in range(8)]
for i in range(20): tr.step(bs[i % 8], bs[(i + 1) % 8])
torch.cuda.synchronize()
best = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): tr.step(bs[i % 8], bs[(i + 1) % 8])
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 200)
print(f"B={B}: {min(best):.4f} ms/step (three passes: {', '.join(f'{b:.4f}' for b in best)})  NRL_DEFER_POSTPONE={os.environ.get('NRL_DEFER_POSTPONE', 'default')}")
