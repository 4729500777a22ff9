"""Stand-in for a bench.py rank, CPU only: started by bench.launch_ranks() in tests/test_bench_launch.py.

Does what a rank does around its timed region -- meets the other ranks through the sync object bench.py would use
(JINC_BENCH_SYNC: "store" = TCPStore, anything else = torch.distributed, here over gloo), takes its shard of the clip,
reports elapsed time and frames -- without a GPU.  Rank 0 prints the ONE line the parent passes through."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    args = bench.parse_args([a for a in sys.argv[1:] if not a.startswith("--stub-")])
    stub = dict(a[7:].split("=", 1) for a in sys.argv[1:] if a.startswith("--stub-"))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank and world == args.gpus
    if int(stub.get("fail-rank", -1)) == rank:
        sys.exit(int(stub.get("fail-code", 3)))     # before the first barrier: the others wait there until the parent stops them
    if os.environ.get("JINC_BENCH_SYNC") == "store":
        sync = bench.StoreSync(rank, world, timeout_s=60.0)
    else:
        import torch.distributed as dist
        sync = bench.DistSync(dist, rank, world, rank, backend="gloo")
    total = bench.CONFIGS[args.config][6]
    frames = bench.shard_frames(total, rank, world)[1] if args.config == "C5" else total
    sync.barrier()
    elapsed = 1.0 + 0.25 * rank
    sync.barrier()
    t, units, by_rank = sync.reduce(elapsed, float(frames * args.steps))
    if rank == 0:
        print(json.dumps({"n_gpus": world, "steps": args.steps, "elapsed_max": t, "frames": units, "frames_per_rank": by_rank,
                          "sync": sync.name, "scaling": "strong" if args.config == "C5" else "weak"}), flush=True)
    else:
        print("rank", rank, "must not reach the parent's stdout")
    sync.close()


if __name__ == "__main__":
    main()
