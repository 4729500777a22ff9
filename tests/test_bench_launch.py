"""`python bench.py --gpus N` started plainly: the parent starts the N ranks itself.  Here the parent
logic runs on the CPU with a stub rank (tests/_bench_stub_worker.py); tests/test_bench_dist.py runs the real thing with
one rank on the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

STUB = [sys.executable, os.path.join(ROOT, "tests", "_bench_stub_worker.py")]


def run_parent(n, argv, sync):
    """The parent in a process of its own, so that rank 0's stdout (inherited from it) can be read here."""
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(%d, %r, sync=%r, worker=%r, timeout_s=120))" % (ROOT, n, argv, sync, STUB))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)


@pytest.mark.parametrize("sync", ["store", "gloo"])
@pytest.mark.parametrize("n,config", [(2, "C5"), (3, "C5"), (2, "C2")])
def test_parent_starts_the_ranks_and_passes_one_line_through(n, config, sync):
    r = run_parent(n, ["--gpus", str(n), "--config", config, "--steps", "3"], sync)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]   # gloo's own connection notice
    assert len(lines) == 1, r.stdout          # rank 0's line only; the other ranks' stdout is dropped
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["sync"] == ("store" if sync == "store" else "rccl")
    assert d["elapsed_max"] == pytest.approx(1.0 + 0.25 * (n - 1))          # MAX over ranks
    total = bench.CONFIGS[config][6]
    if config == "C5":   # one clip: the ranks' shards add up to it
        assert d["scaling"] == "strong" and d["frames"] == 3 * total
        assert d["frames_per_rank"] == [3 * bench.shard_frames(total, k, n)[1] for k in range(n)]
    else:                # the same batch on every rank
        assert d["scaling"] == "weak" and d["frames_per_rank"] == [3 * total] * n


def test_a_failing_rank_stops_the_others_and_its_code_is_returned():
    r = run_parent(3, ["--gpus", "3", "--config", "C5", "--stub-fail-rank=1", "--stub-fail-code=7"], "store")
    assert r.returncode == 7, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 1 exited with code 7" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_plain_start_with_more_gpus_than_the_host_has_is_refused_before_any_gpu_call():
    """No launcher, no WORLD_SIZE: the parent counts devices from the KFD topology -- no HIP call, ADVICE r4 -- and says what is
    missing; where the topology cannot be read (this CPU container) it starts the ranks and the first of them says so."""
    have = bench.visible_gpu_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    n = (have if have is not None else 0) + 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    if have is not None:
        assert f"this host shows {have} HIP device(s)" in r.stderr
    else:
        assert "needs a HIP device" in r.stderr or "exited with code" in r.stderr
    text = open(os.path.join(ROOT, "bench.py")).read()
    parent = text[text.index("if world_env is None and not args.inproc"):text.index("import torch\n\n    rank = int(")]
    assert "torch" not in parent and "device_count" not in parent   # the parent's branch imports nothing that could initialise HIP


def test_devices_are_counted_from_the_kfd_topology(tmp_path):
    """CPU nodes have simd_count 0; *_VISIBLE_DEVICES narrow the list, which ends at the first entry that names no device."""
    for k, simd in enumerate((0, 0, 456, 456, 456)):
        d = tmp_path / str(k)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    base = str(tmp_path)
    assert bench.visible_gpu_count(base, {}) == 3
    assert bench.visible_gpu_count(base, {"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert bench.visible_gpu_count(base, {"ROCR_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1"}) == 1
    assert bench.visible_gpu_count(base, {"CUDA_VISIBLE_DEVICES": "0,7,1"}) == 1
    assert bench.visible_gpu_count(base, {"HIP_VISIBLE_DEVICES": ""}) == 0
    assert bench.visible_gpu_count(str(tmp_path / "missing"), {}) is None


def test_self_launch_is_refused_under_a_profiler():
    """rocprofv3's tool library has initialised the GPU in the process it is preloaded into: that process must not start ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ROCPROF_OUTPUT_PATH"] = "/tmp/x"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--inproc" in r.stderr and "rocprofv3" in r.stderr


def test_rank_count_and_gpus_must_agree_under_a_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "the two must agree" in r.stderr


def test_cpu_quota_probe_reads_this_cgroup():
    q = bench.cpu_quota()
    assert set(q) == {"quota_cores", "source", "cpuset_cpus"}
    assert q["quota_cores"] is None or q["quota_cores"] > 0
    assert isinstance(q["source"], str) and q["source"]


def test_profiler_environment_is_recognised(monkeypatch):
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.setenv("LD_PRELOAD", "")
    assert not bench.under_profiler()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()
