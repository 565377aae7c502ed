import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from newsreclib_amd.nrms_module import attach_layout
from newsreclib_amd.synthetic import make_batch
from newsreclib_amd.trainer import NRMSTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
mod = bench.build_module(dev)
tr = NRMSTrainer(mod, lr=1e-4)
bs = [attach_layout(make_batch(B, 70000, "fixed", seed=1234 + i, device=dev)) for i in range(4)]
for i in range(25): tr.step(bs[i % 4])
torch.cuda.synchronize()
