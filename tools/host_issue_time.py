import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from newsreclib_amd.nrms_module import attach_layout
from newsreclib_amd.synthetic import make_batch
from newsreclib_amd.trainer import NRMSTrainer
dev = torch.device("cuda", 0)
for B in (32, 64, 128):
    mod = bench.build_module(dev)
    tr = NRMSTrainer(mod, lr=1e-4)
    bs = [attach_layout(make_batch(B, 70000, "fixed", seed=1234 + i, device=dev)) for i in range(4)]
    for i in range(10): tr.step(bs[i % 4])
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for i in range(n): tr.step(bs[i % 4])
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"B={B}: host issue {t_issue*1e3:.3f} ms/step, with sync {t_all*1e3:.3f} ms/step")
    del mod, tr, bs
