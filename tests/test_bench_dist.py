"""bench.py's RCCL leg on the one GPU there is: launched the way the driver launches N > 1 --
`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1` -- so that init_process_group("nccl"),
barrier(device_ids=...) and the MAX / SUM all-reduces have run on hardware before the first multi-GPU scaling run.
The launcher is started as a fresh child process; it spawns bench.py before anything in that child touches the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("config,extra", [("C1", ["--frames", "64"]), ("C5", [])], ids=["C1_weak", "C5_strong"])
def test_bench_under_torch_distributed_run_with_one_rank(gpu_pkg, config, extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", config, *extra,
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0
    assert d["scaling"] == ("strong" if config == "C5" else "weak")
    assert d["roofline"]["kernel"] in ("ewa_periodic_kernel", "ewa_periodic_quad_kernel") and d["roofline"]["frac"] > 0
    assert d["config"]["parallelism"].startswith("frames sharded over 1 GPU")
