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
    assert d["roofline"]["kernel"].split("<")[0] in ("ewa_periodic_kernel", "ewa_periodic_quad_kernel", "ewa_periodic_quad2_kernel") and d["roofline"]["frac"] > 0
    assert d["self_check"] == "ok", d["self_check_detail"]   # frame 0 = the Appendix-A frame, its output = the reference's crc32
    assert d["config"]["parallelism"].startswith("frames sharded over 1 GPU") and d["config"]["sync"] == "rccl"


def _one_line(r):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _plain_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "JINC_BENCH_SYNC", "JINC_BENCH_SELF_LAUNCH")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


QUICK = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-clock-sampler"]


def test_bench_started_plainly_runs_the_clip_config(gpu_pkg):
    """`python bench.py --gpus 1 --config C5 --steps 3`, no launcher, no environment."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C5", *QUICK], cwd=ROOT, env=_plain_env(),
                       capture_output=True, text=True, timeout=300)
    d = _one_line(r)
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["frames_per_rank"] == [3 * 512] and d["config"]["sync"] == "none" and d["config"]["launch"] == "single process"
    # 512 C2 frames per call: the instantiation the headline times (1024 frames) -- and the one
    # tests/test_benchmarked_instances.py::test_c2_batch_reaches_the_benchmarked_instantiation compares with the CPU checker
    from test_benchmarked_instances import BENCHMARKED
    assert d["roofline"]["kernel"] == BENCHMARKED["C2"] and d["self_check"] == "ok", (d["roofline"]["kernel"], d["self_check_detail"])


def test_bench_self_check_fails_loudly_when_the_output_is_wrong(gpu_pkg):
    """pipeline_skip has no effect on process_device; a knob that really changes pixels does not exist -- so the check is shown
    to bite through the compatibility order instead: --simd-order 2 computes the AVX2 path's result, which differs from opt=0 in
    a few pixels of C1, and bench.py (which skips the check for that switch) is asked to check anyway."""
    r = subprocess.run([sys.executable, "-c",
                        "import sys, bench; bench.FORCE_SELF_CHECK = True; bench.main(sys.argv[1:])",
                        "--config", "C1", "--frames", "8", "--simd-order", "2", *QUICK], cwd=ROOT, env=_plain_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "SELF-CHECK FAILED" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("config,frames", [("C3", "2"), ("C4", "4")])
def test_bench_checks_the_other_baseline_configs(gpu_pkg, config, frames):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--frames", frames, *QUICK], cwd=ROOT, env=_plain_env(),
                       capture_output=True, text=True, timeout=600)
    d = _one_line(r)
    assert d["self_check"] == "ok", d["self_check_detail"]


@pytest.mark.parametrize("sync", ["store", "rccl"])
def test_bench_starts_its_own_ranks(gpu_pkg, sync):
    """The way `python bench.py --gpus N` takes for N > 1, here with the one rank a one-GPU box allows: the parent touches no
    GPU, starts the rank as a fresh interpreter, passes its line through; the rank meets itself over the TCPStore (no RCCL) or,
    with --sync rccl, over RCCL."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C5", "--sync", sync, *QUICK], cwd=ROOT,
                       env=_plain_env(JINC_BENCH_SELF_LAUNCH="1"), capture_output=True, text=True, timeout=300)
    d = _one_line(r)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["launch"] == "self-launched ranks"
    assert d["config"]["sync"] == sync and d["config"]["frames_per_rank"] == [3 * 512]


def test_bench_in_one_process_over_its_devices(gpu_pkg):
    """--inproc: a filter instance and a stream per device from one host thread (one device here)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--inproc", "--config", "C1", "--frames", "64", *QUICK], cwd=ROOT,
                       env=_plain_env(), capture_output=True, text=True, timeout=300)
    d = _one_line(r)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["launch"] == "inproc" and d["config"]["frames_per_rank"] == [3 * 64]


def test_bench_under_a_launcher_with_the_store_for_the_barrier(gpu_pkg):
    env = dict(_plain_env(), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C1", "--frames", "64", "--sync", "store", *QUICK]
    d = _one_line(subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300))
    assert d["config"]["sync"] == "store" and d["config"]["launch"] == "launcher" and d["value"] > 0
