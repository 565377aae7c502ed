#!/usr/bin/env python3
"""Where does the reproducible ~85 ms stall of bench.py's evaluation loop come from (VERDICT round 5, weak item 7:
`forward_only.group_ms` = [0.6, 0.6, 0.6, 17.1] on two boxes)?  Replays the bench's sequence -- train steps, then eval-mode
forwards over the same batches -- and times EVERY forward on its own (host clock around a synchronize), printing the allocator's
counters next to each one.
    python tools/eval_stall_probe.py [--forwards 30] [--steps 12]"""
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
    ap.add_argument("--forwards", type=int, default=30)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--prof", action="store_true", help="bench.py's instrumented pass (nrl_prof_enable) before the forwards")
    ap.add_argument("--prepare", action="store_true", help="bench.py's 20 prepare_batch calls before the forwards")
    ap.add_argument("--predict", action="store_true", help="bench.py's predict_multi_gpu (host-side torch.unique) before the forwards")
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
    for i in range(a.steps):
        trainer.step(batches[i % nb], batches[(i + 1) % nb])
    torch.cuda.synchronize()
    if a.prof:
        import ctypes
        lib = _lib.load()
        lib.nrl_prof_enable(1)
        for i in range(a.steps):
            trainer.step(batches[i % nb], batches[(i + 1) % nb])
        torch.cuda.synchronize()
        tot_ms, launches, flops = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        lib.nrl_prof_read(ctypes.byref(tot_ms), ctypes.byref(launches), ctypes.byref(flops))
        lib.nrl_prof_enable(0)
    if a.prepare:
        from newsreclib_amd.nrms_module import prepare_batch
        for i in range(20):
            prepare_batch(batches[i % nb], bench.VOCAB)
        torch.cuda.synchronize()
    if a.predict:
        bench.predict_multi_gpu(2.8)
    print(f"flags: prof={a.prof} prepare={a.prepare} predict={a.predict}")

    def stats():
        s = torch.cuda.memory_stats()
        return (s.get("num_device_alloc", 0), s.get("num_device_free", 0), s.get("num_alloc_retries", 0),
                s.get("reserved_bytes.all.current", 0) >> 20)

    mod.eval()
    print("forward  batch   ms      device_allocs frees retries reserved_MiB")
    with torch.no_grad():
        for i in range(a.forwards):
            s0 = stats()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mod.forward(batches[i % nb])
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            s1 = stats()
            flag = "  <-- stall" if dt > 3e-3 else ""
            print(f"{i:7d} {i % nb:6d} {dt * 1e3:8.3f} (host {t_host * 1e3:7.3f})  +{s1[0] - s0[0]} +{s1[1] - s0[1]} +{s1[2] - s0[2]} {s1[3]}{flag}")


if __name__ == "__main__":
    main()
