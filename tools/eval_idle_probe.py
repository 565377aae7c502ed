#!/usr/bin/env python3
"""Does the ~85 ms stall of bench.py's evaluation loop follow an IDLE period of the GPU?  Rounds of: idle for `--idle` seconds (host
sleep), three warm-up forwards + synchronize, then four groups of five back-to-back forwards (one synchronize per group), as
bench.py does.  Prints the groups in order for every round, with and without the idle period.
    python tools/eval_idle_probe.py [--rounds 6] [--idle 1.5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--idle", type=float, default=1.5)
    ap.add_argument("--bench-prelude", action="store_true", help="what bench.py runs between its timed region and the evaluation loop: "
                    "the nrl_prof pass, 20 prepare_batch calls, predict_multi_gpu")
    ap.add_argument("--parts", default="prof,prepare,predict", help="which parts of the prelude to run")
    a = ap.parse_args()
    from newsreclib_amd import _lib
    from newsreclib_amd.nrms_module import attach_layout
    from newsreclib_amd.synthetic import make_batch
    from newsreclib_amd.trainer import NRMSTrainer
    _lib.load()
    _lib.set_gemm_engine("bf16x3")
    dev = torch.device("cuda", 0)
    mod = bench.build_module(dev)
    trainer = NRMSTrainer(mod, lr=bench.LR, grad_exchange="dense")
    nb = bench.N_BATCHES
    batches = [attach_layout(make_batch(bench.B_PER_GPU, bench.VOCAB, "fixed", seed=1234 + 1000 * i, device=dev)) for i in range(nb)]
    for i in range(60):
        trainer.step(batches[i % nb], batches[(i + 1) % nb])
    torch.cuda.synchronize()
    if a.bench_prelude:
        import ctypes
        from newsreclib_amd.nrms_module import prepare_batch
        lib = _lib.load()
        parts = a.parts.split(",")
        if "prof" in parts:
            lib.nrl_prof_enable(1)
            for i in range(50):
                trainer.step(batches[i % nb], batches[(i + 1) % nb])
            torch.cuda.synchronize()
            tot_ms, launches, flops = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
            lib.nrl_prof_read(ctypes.byref(tot_ms), ctypes.byref(launches), ctypes.byref(flops))
            lib.nrl_prof_enable(0)
        if "prepare" in parts:
            for i in range(20):
                prepare_batch(batches[i % nb], bench.VOCAB)
            torch.cuda.synchronize()
        if "predict" in parts:
            bench.predict_multi_gpu(2.8)
        if "sleep" in parts:
            time.sleep(1.0)
        if "alloc" in parts:          # host-only: allocate, touch and FREE large anonymous regions (malloc -> mmap / munmap), no torch op on them
            for _ in range(8):
                x = torch.empty(32 << 20, dtype=torch.int64).fill_(1)
                del x
        if "unique" in parts:         # host-only: what predict_multi_gpu does (sort-based unique of ~1.7 M ids, 9 times)
            for _ in range(9):
                x = torch.randint(0, 70000, (1_700_000,))
                y = torch.unique(x)
                del x, y
    mod.eval()
    with torch.no_grad():
        for idle in ((0.0,) if a.bench_prelude else (0.0, a.idle, 0.0, a.idle)):
            for r in range(a.rounds):
                if idle > 0:
                    time.sleep(idle)
                for i in range(3):
                    mod.forward(batches[i % nb])
                torch.cuda.synchronize()
                groups = []
                for _ in range(4):
                    t1 = time.perf_counter()
                    for i in range(5):
                        mod.forward(batches[i % nb])
                    torch.cuda.synchronize()
                    groups.append((time.perf_counter() - t1) / 5 * 1e3)
                flag = "  <-- stall" if max(groups) > 3 else ""
                print(f"idle {idle:.1f} s, round {r}: groups (ms/forward) " + " ".join(f"{g:7.3f}" for g in groups) + flag)


if __name__ == "__main__":
    main()
