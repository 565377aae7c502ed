"""bench.py through the launcher the driver uses for N > 1 -- on the one GPU of the test box, so that the RCCL
initialisation (`init_process_group("nccl", device_id=...)`), the barrier, the max-over-ranks all-reduce and the JSON contract
have executed at least once on real hardware (reference multi-GPU leg: configs/trainer/ddp.yaml:4)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["dense", "owners", "auto"])
def test_bench_runs_under_torch_distributed_run_on_rccl(exchange):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29400 + os.getpid() % 500), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--no-extras", "--grad-exchange", exchange]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["rccl_ranks"] == 1 and j["n_gpus"] == 1 and j["steps"] == 3 and j["value"] > 0
    assert j["unit"] == "impressions/s" and j["scaling"] == "weak" and "roofline" in j
    assert j["grad_exchange"]["mode"] == exchange and "predicted_wire_ms" in j["grad_exchange"]
    # the self-diagnosing fields of a multi-GPU run exist and are zero at world 1 (nothing is exchanged)
    ps, meas = j["grad_exchange"]["per_step"], j["grad_exchange"]["measured"]
    for k in ("exchange_chosen", "bytes_on_the_wire_per_rank", "predicted_wire_ms", "overlap_window_ms", "predicted_exposed_ms",
              "measured_exposed_wait_ms", "steps_measured"):
        assert k in ps, k
    assert meas["world"] == 1 and meas["exposed_wait_ms_per_step"] == 0.0 and ps["measured_exposed_wait_ms"] == 0.0
    assert all(v == 0.0 for v in ps["predicted_wire_ms"].values())
