"""The N>1 path of bench.py on CPU: two ranks over gloo (the GPU box runs the same logic over RCCL).
Frames shard with no data-path collective, so what needs covering is the partition, the
barrier/MAX/SUM reductions and the per-rank plan replicas."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


@pytest.mark.parametrize("total,world", [(512, 1), (512, 2), (512, 8), (515, 4), (3, 8), (0, 2)])
def test_shard_frames_partitions(total, world):
    seen = []
    for r in range(world):
        start, count = bench.shard_frames(total, r, world)
        seen += list(range(start, start + count))
        assert count in (total // world, total // world + 1)
    assert seen == list(range(total))


@pytest.mark.parametrize("total,ndev", [(512, 1), (512, 8), (515, 4), (3, 8)])
def test_library_shard_is_round_robin(pkg, total, ndev):
    """jinc_shard_device: the frame -> device map of jinc_batch_process (no GPU needed to evaluate it)."""
    owners = [pkg.shard_device(n, ndev) for n in range(total)]
    assert owners == [n % ndev for n in range(total)]
    counts = [owners.count(d) for d in range(ndev)]
    assert max(counts) - min(counts) <= 1
    assert pkg.shard_device(-1, ndev) == -1 and pkg.shard_device(0, 0) == -1


def test_algorithmic_bytes_match_survey(pkg):
    """C2 10 368 000 B, C3 31 104 000 B, C4 497 664 000 B per frame."""
    want = {"C2": 10_368_000, "C3": 31_104_000, "C4": 497_664_000}
    for name in want:
        fmt, sw, sh, dw, dh, _, _ = bench.CONFIGS[name]
        assert bench.algorithmic_bytes_per_frame(pkg.FORMATS[fmt], sw, sh, dw, dh) == want[name]


def test_two_rank_gloo_run(pkg):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29613", os.path.join(ROOT, "tests", "_gloo_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "GLOO_WORKER_OK" in r.stdout
