"""Worker for test_sharding_gloo.py: one process per rank, gloo backend, CPU only.

Exercises the N>1 logic bench.py uses on the GPU box (frame sharding, MAX-over-ranks timing, SUM of
units, identical plan replica on every rank) without touching a GPU."""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 512 + 3  # C5: a batch of independent frames, deliberately not divisible by the world size
    start, count = bench.shard_frames(total, rank, world)

    # every frame is owned by exactly one rank
    owned = torch.zeros(total, dtype=torch.int32)
    owned[start:start + count] = 1
    dist.all_reduce(owned)
    assert bool((owned == 1).all()), "frames must be partitioned exactly once"

    # the library's own shard (jinc_batch_*: frame n -> device n mod G, here rank = device): also an exact partition,
    # balanced to within one frame
    pkg = entry.load_package()
    mine = torch.tensor([1 if pkg.shard_device(n, world) == rank else 0 for n in range(total)], dtype=torch.int32)
    assert abs(int(mine.sum()) - total / world) < 1.0
    dist.all_reduce(mine)
    assert bool((mine == 1).all()), "round-robin shard must own every frame exactly once"

    # MAX over ranks of the time, SUM of the units -- the only cross-rank traffic of the bench
    elapsed = 1.0 + 0.25 * rank
    t, units = bench.aggregate(elapsed, float(count), dist)
    assert abs(t - (1.0 + 0.25 * (world - 1))) < 1e-12
    assert units == float(total)

    # each rank builds its own replica of the plan; replicas must be identical (no exchange needed)
    f = pkg.Filter(pkg.FORMATS["YUV420P8"], 320, 180, 640, 360, device=-1, tap=3)
    h = hashlib.sha256()
    for tbl in range(f.num_tables):
        sx, sy, ids = f.plan_dump(tbl)
        for a in (sx, sy, ids, f.plan_sets(tbl)):
            h.update(np.ascontiguousarray(a).tobytes())
    digest = torch.frombuffer(bytearray(h.digest()), dtype=torch.uint8).clone()
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    assert all(bool((g == gathered[0]).all()) for g in gathered), "plan replicas differ between ranks"

    dist.barrier()
    if rank == 0:
        print("GLOO_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
