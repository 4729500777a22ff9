#!/usr/bin/env python3
"""bench.py -- headline benchmark of the EWA-Jinc hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): 1920x1080 -> 3840x2160, Y8 (u8 samples), tap=3.
A *step* is one pass of the hot path over one batch of `--frames` independent synthetic frames
that are already resident in HBM (one jinc_filter_process_device call = one periodic-interior
kernel launch + one border gather launch for the whole batch).  Frames are the sharding unit:
with N GPUs every rank (one process per GPU) holds its own batch and there is no data-path
collective -- the only communication is the barrier and the MAX over ranks of the elapsed time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames B] [--config C2|C3|C4]

How N > 1 starts (frames shard, nothing is exchanged, so any of these is the same measurement):
  * under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, the driver's form):
    one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE from the environment, barrier and MAX / SUM over RCCL;
  * started plainly (`python bench.py --gpus N`, no WORLD_SIZE): this process touches no GPU (devices are counted from the
    KFD topology, not through HIP; refused under rocprofv3, whose tool has initialised the GPU already) and starts N rank
    processes itself (one per device, fresh interpreters), which meet over a torch TCPStore on 127.0.0.1 -- no RCCL,
    the north_star's "independent per-device streams"; `--sync rccl` makes the children use RCCL instead;
  * `--inproc`: ONE process, one filter instance + one HIP stream per device, steps issued to all devices from one
    host thread.

Rank 0 prints ONE JSON line (see the task contract): value = Mpix/s over all ranks, plus
  "roofline"     -- dominant (periodic-interior) kernel: algorithmic HBM bytes / its mean launch
                    duration measured with hipEvents on the launch stream, vs the 8 TB/s peak;
                    also the un-fused fp32 VALU fraction, the roof that actually binds;
  "cpu_baseline" -- the CPU oracle (port of the reference opt=0 path) timed on this host's cores
                    on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

FORCE_SELF_CHECK = False     # tests: check frame 0 even under --simd-order (whose results differ from opt=0 by design)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_UNFUSED_PEAK = 78.6e12  # 256 CU x 128 lanes/clk x 2.4 GHz, one IEEE op per lane-clk (no FMA allowed)

CONFIGS = {
    # name: (format, src_w, src_h, dst_w, dst_h, script args, default frames per step)
    # C2 default: 1024 frames per step = twice the 512-frame clip of BASELINE.json configs[4], 10.6 GB resident in HBM;
    # one step is then ~10 ms of kernel time, so that the default --steps 100 times ~1 s (r4 weak 13)
    "C2": ("Y8", 1920, 1080, 3840, 2160, dict(tap=3), 1024),
    # BASELINE.json configs[4]: ONE clip of 512 frames sharded over the ranks (strong scaling: the per-rank batch shrinks
    # with N); the default C2 run is the weak-scaling form of the same workload
    "C5": ("Y8", 1920, 1080, 3840, 2160, dict(tap=3), 512),
    "C3": ("YUV420P16", 1920, 1080, 3840, 2160, dict(tap=8, cplace="mpeg2"), 32),
    "C4": ("RGBPS", 3840, 2160, 7680, 4320, dict(tap=4, blur=0.98), 16),
    # not a BASELINE.json config: a non-periodic ratio (1.5x, float drift => gather kernel for every pixel)
    "C1": ("Y8", 640, 360, 1280, 720, dict(tap=3), 1024),     # BASELINE configs[0] shape (the reference's CPU case), on the GPU
    "N15": ("Y8", 1280, 720, 1920, 1080, dict(tap=3), 256),
    "D23": ("Y8", 1920, 1080, 1280, 720, dict(tap=3), 256),   # 2/3 down-scale: fs = 10, period 2, source step 3
    "N3": ("Y8", 1280, 720, 3840, 2160, dict(tap=3), 128),    # 3x: drifting phases (>= 128 frames: frame-pair kernel; below: quasi-periodic kernel)
    "N15T8": ("Y8", 1280, 720, 1920, 1080, dict(tap=8), 128),  # 1.5x with Jinc256: fs 17, drifting (batches: frame-lane kernel, row-segment form)
    "U43": ("Y8", 1440, 1080, 1920, 1440, dict(tap=3), 128),  # 4/3x: exactly periodic, period 4 / source step 3
    "N480": ("YUV420P8", 720, 480, 1920, 1080, dict(tap=3), 256),  # DVD -> 1080p: 8/3 x 9/4, luma and chroma tables (>= 128 frames: frame-pair kernel)
    "N15T4": ("Y8", 1280, 720, 1920, 1080, dict(tap=4), 256),  # 1.5x with Jinc64: fs 9, drifting (batches: frame-lane kernel)
    "C2YUV": ("YUV420P8", 1920, 1080, 3840, 2160, dict(tap=3), 64),  # C2's geometry on a 4:2:0 frame (luma + two chroma planes)
    "D12Y16": ("Y16", 3840, 2160, 1920, 1080, dict(tap=3), 64),   # 4K -> 1080p, one 16-bit / float plane (D12H / D12F without the chroma planes)
    "D12Y32": ("Y32", 3840, 2160, 1920, 1080, dict(tap=3), 32),
    "S15T4": ("Y8", 640, 360, 960, 540, dict(tap=4), 128),      # small frames at 1.5x: Jinc64 / Jinc256
    "S15T8": ("Y8", 640, 360, 960, 540, dict(tap=8), 128),
    "N3T4": ("Y8", 640, 360, 1920, 1080, dict(tap=4), 128),     # 3x with Jinc64: fs 9, source step 1, drifting
    "N3T8": ("Y8", 640, 360, 1920, 1080, dict(tap=8), 128),     # 3x with Jinc256: fs 17
    "N480T4": ("YUV420P8", 720, 480, 1920, 1080, dict(tap=4), 128),  # DVD -> 1080p with Jinc64: 72 phases, source steps 3 / 4
    "N480T6": ("YUV420P8", 720, 480, 1920, 1080, dict(tap=6), 128),  # ... with Jinc144: fs 13
    "N25T6": ("Y16", 768, 432, 1920, 1080, dict(tap=6), 128),   # 5/2 with Jinc144 on 16-bit: fs 13, period 5, source step 2
    "A137": ("Y8", 1280, 720, 1754, 986, dict(tap=3), 256),    # 1.37x: no phase structure at all (>= 128 frames: frame-pair kernel)
    "A1875": ("Y8", 1024, 576, 1920, 1080, dict(tap=3), 256),  # PAL -> 1080p, 15/8: period 15, source step 8
    # what a host at look-ahead 32 hands over: groups of 16 frames (the frame-lane kernel's sub-group form, round 4), and 4 / 32 frames
    "A137L16": ("Y8", 1280, 720, 1754, 986, dict(tap=3), 16),
    "A137L32": ("Y8", 1280, 720, 1754, 986, dict(tap=3), 32),
    "A137L4": ("Y8", 1280, 720, 1754, 986, dict(tap=3), 4),
    "D169L16": ("Y8", 1920, 1080, 1600, 900, dict(tap=3), 16),
    "D169": ("Y8", 1920, 1080, 1600, 900, dict(tap=3), 256),   # 5/6 down-scale: drifting, period 5, source step 6, fs 8
    "D12": ("Y8", 3840, 2160, 1920, 1080, dict(tap=3), 128),   # 1/2 down-scale: fs = 13, period 1, source step 2
    "D12H": ("YUV420P16", 3840, 2160, 1920, 1080, dict(tap=3), 64),  # 4K 16-bit 4:2:0 -> 1080p
    "D12F": ("RGBPS", 3840, 2160, 1920, 1080, dict(tap=3), 32),       # 4K float RGB -> 1080p
    "D13": ("Y8", 3840, 2160, 1280, 720, dict(tap=3), 128),    # 1/3 down-scale: fs = 20, period 1, source step 3
    "D12T4": ("Y8", 3840, 2160, 1920, 1080, dict(tap=4), 64),   # Jinc64 at 1/2: fs = 17 (9 + 8 taps per kernel row)
    "D12T8": ("Y8", 3840, 2160, 1920, 1080, dict(tap=8), 32),    # Jinc256 at 1/2: fs = 33 (3 x 11)
    "T6": ("Y8", 1920, 1080, 3840, 2160, dict(tap=6), 64),    # Jinc144: fs = 13
    "T16": ("Y8", 1920, 1080, 3840, 2160, dict(tap=16), 16),  # tap 16: fs = 33 (1089 taps)
    "C2H": ("YUV420P16", 1920, 1080, 3840, 2160, dict(tap=3), 128),  # C2's geometry on 16-bit 4:2:0 (luma and chroma both 2x, fs 7)
    "C2F": ("RGBPS", 1920, 1080, 3840, 2160, dict(tap=3), 64),       # ... and on float RGB
    "C2T4": ("Y8", 1920, 1080, 3840, 2160, dict(tap=4), 256),        # Jinc64Resize at 2x: fs 9, support 8 x 8 on integer planes
    "C2HT4": ("YUV420P16", 1920, 1080, 3840, 2160, dict(tap=4), 64),
}


def baseline_metric():
    """The headline metric's name, verbatim from BASELINE.json."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8"))["metric"]
    except Exception:  # noqa: BLE001
        return "Mpix/s per GPU (1080p->4K tap=3 Y8); % HBM-read roofline"


def shard_frames(total_frames: int, rank: int, world: int):
    """Contiguous shard of a global batch of independent frames for `rank` (frames never interact)."""
    base, rem = divmod(total_frames, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def aggregate(elapsed_s: float, units: float, dist=None):
    """MAX over ranks of the elapsed time, SUM over ranks of the processed units."""
    if dist is None or not dist.is_initialized():
        return elapsed_s, units
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    u = torch.tensor([units], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


class NoSync:
    """One rank: nothing to meet."""
    name = "none"

    def barrier(self):
        pass

    def reduce(self, elapsed_s, units):
        return elapsed_s, units, [units]

    def close(self):
        pass


class DistSync(NoSync):
    """torch.distributed over RCCL (backend "nccl"): what a launcher-started run uses."""
    name = "rccl"

    def __init__(self, dist, rank, world, local_rank, backend="nccl"):
        self.dist, self.rank, self.world, self.local_rank, self.backend = dist, rank, world, local_rank, backend
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier(self):
        if self.backend == "nccl":
            self.dist.barrier(device_ids=[self.local_rank])
        else:
            self.dist.barrier()

    def reduce(self, elapsed_s, units):
        import torch
        t, u = aggregate(elapsed_s, units, self.dist)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        mine = torch.tensor([units], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(every, mine)
        return t, u, [float(x.item()) for x in every]

    def close(self):
        self.dist.destroy_process_group()


class StoreSync(NoSync):
    """Ranks that share nothing but a key-value store on 127.0.0.1 (torch's TCPStore, rank 0 serves): the barrier is a
    counter every rank adds to and then waits on; MAX / SUM are rank 0 reading every rank's figures.  No RCCL, no GPU
    memory, no collective: frames are independent units and the ranks only agree on when the clock runs."""
    name = "store"

    def __init__(self, rank, world, host=None, port=None, timeout_s=600.0):
        from datetime import timedelta
        from torch.distributed import TCPStore
        self.rank, self.world, self.round = rank, world, 0
        host = host or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("MASTER_PORT", "29500"))
        # under torch.distributed.run the launcher's agent already serves a store on MASTER_PORT; every rank is then a client
        serve = rank == 0 and os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") != "True"
        self.store = TCPStore(host, port, world, is_master=serve, timeout=timedelta(seconds=timeout_s), wait_for_workers=True)

    def barrier(self):
        self.round += 1
        key = f"barrier/{self.round}"
        if self.store.add(key, 1) == self.world:
            self.store.set(key + "/open", b"1")
        self.store.wait([key + "/open"])   # blocks inside the store client, no polling

    def reduce(self, elapsed_s, units):
        self.store.set(f"result/{self.rank}", json.dumps([elapsed_s, units]))
        self.barrier()
        every = [json.loads(self.store.get(f"result/{r}")) for r in range(self.world)]
        self.barrier()   # nobody leaves (and rank 0 does not close the store) before every rank has read
        return max(e for e, _ in every), sum(u for _, u in every), [u for _, u in every]

    def close(self):
        """Rank 0 serves the store: it stays until every other rank has said goodbye (a rank whose last `wait` is still being
        answered when the server goes away sees a reset connection and exits non-zero -- seen once in ~50 runs of the CPU test)."""
        if self.store is None:
            return
        try:
            if self.rank != 0:
                self.store.add("bye", 1)
            else:
                deadline = time.time() + 30.0
                while self.world > 1 and self.store.add("bye", 0) < self.world - 1 and time.time() < deadline:
                    time.sleep(0.005)
        except Exception:  # (the peers are gone already: nothing left to wait for)
            pass
        self.store = None


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv, sync="store", worker=None, timeout_s=1800.0):
    """`python bench.py --gpus N` started plainly: N rank processes, one per device, started from THIS process, which never
    touches a GPU (fresh interpreters: no exec of, and no fork from, a process that has initialised HIP).  Rank k gets
    RANK = LOCAL_RANK = k, WORLD_SIZE = n, MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT -- what torch.distributed.run
    would set -- plus JINC_BENCH_SYNC, the way the ranks meet.  Rank 0's stdout (the ONE JSON line) and every rank's stderr
    pass through; the first rank to fail ends the others.  Returns the exit code for the parent.
    `worker`: the command each rank runs (tests drive this logic on the CPU with a stub)."""
    import subprocess
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    port = free_port()
    procs = []
    for k in range(n):
        env = dict(os.environ, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), JINC_BENCH_SYNC=sync, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=None if k == 0 else subprocess.DEVNULL))
    deadline, rc = time.time() + timeout_s, 0
    live = set(range(n))
    while live:
        for k in sorted(live):
            r = procs[k].poll()
            if r is not None:
                live.discard(k)
                if r != 0 and rc == 0:
                    rc = r
                    print(f"bench.py: rank {k} exited with code {r}; stopping the other ranks", file=sys.stderr)
        if live and (rc != 0 or time.time() > deadline):
            if rc == 0:
                rc = 124
                print(f"bench.py: ranks {sorted(live)} still running after {timeout_s:.0f} s; stopping them", file=sys.stderr)
            for k in live:
                procs[k].terminate()   # the exact processes started above, nothing by pattern
            for k in list(live):
                try:
                    procs[k].wait(10)
                except subprocess.TimeoutExpired:
                    procs[k].kill()
                    procs[k].wait()
                live.discard(k)
        if live:
            time.sleep(0.05)
    return rc


def visible_gpu_count(base="/sys/class/kfd/kfd/topology/nodes", environ=None):
    """GPUs this process would see, WITHOUT initialising HIP (torch.cuda.device_count() falls back to
    hipGetDeviceCount where amdsmi cannot initialise, and the self-launch parent must not hold a HIP context when it starts
    its rank processes): the KFD topology's nodes that have SIMDs, then the *_VISIBLE_DEVICES lists applied in the order
    the runtime applies them.  None when the topology cannot be read (the ranks then find out for themselves)."""
    environ = os.environ if environ is None else environ
    try:
        nodes = sorted(os.listdir(base), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(base, node, "properties"), encoding="utf-8") if " " in line)
        except OSError:
            continue   # (a node this cgroup may not read is a device it may not use)
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = environ.get(var)
        if v is None:
            continue
        ids = [x.strip() for x in v.split(",") if x.strip() != ""]
        keep = 0
        for x in ids:   # the list ends at the first entry that names no device
            if x.isdigit() and int(x) < n:
                keep += 1
            elif x.startswith("GPU-"):
                keep += 1
            else:
                break
        n = min(n, keep)
    return n


def under_profiler():
    """rocprofv3 preloads its tool library into the benchmark: the clock-sampler pass, the instruction-pair probe, the
    host-to-host record and the CPU baseline would all land in the profile, so they are skipped there."""
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprofiler" in pre or "rocprof" in pre or any(k.startswith(("ROCPROF_", "ROCPROFILER_", "ROCP_")) for k in os.environ)


def cpu_quota():
    """What limits this process's CPU time besides the affinity mask: the cgroup's CFS quota (cpu.max on cgroup v2,
    cpu.cfs_quota_us / cpu.cfs_period_us on v1, looked up from this process's cgroup towards the root) and the cpuset.
    Returns {"quota_cores": float | None, "source": ..., "cpuset_cpus": ... | None}."""
    def read(path):
        try:
            return open(path, encoding="utf-8").read().strip()
        except OSError:
            return None

    def parents(rel):
        rel = rel.strip("/")
        parts = rel.split("/") if rel else []
        for k in range(len(parts), -1, -1):
            yield "/".join(parts[:k])

    groups = {}
    for line in (read("/proc/self/cgroup") or "").splitlines():
        f = line.split(":", 2)
        if len(f) == 3:
            for ctl in (f[1].split(",") if f[1] else [""]):
                groups[ctl] = f[2]
    best, source, cpuset = None, None, None
    for rel in parents(groups.get("", "")):            # v2
        base = os.path.join("/sys/fs/cgroup", rel)
        v = read(os.path.join(base, "cpu.max"))
        if v and v.split()[0] != "max":
            q = float(v.split()[0]) / float(v.split()[1])
            if best is None or q < best:
                best, source = q, os.path.join(base, "cpu.max") + f" = {v}"
        cpuset = cpuset or read(os.path.join(base, "cpuset.cpus.effective"))
    for ctl in ("cpu", "cpu,cpuacct"):                 # v1
        for rel in parents(groups.get("cpu", groups.get("cpuacct", ""))):
            base = os.path.join("/sys/fs/cgroup", ctl, rel)
            q, per = read(os.path.join(base, "cpu.cfs_quota_us")), read(os.path.join(base, "cpu.cfs_period_us"))
            if q and per and int(q) > 0:
                c = int(q) / int(per)
                if best is None or c < best:
                    best, source = c, os.path.join(base, "cpu.cfs_quota_us") + f" = {q} / {per}"
    for rel in parents(groups.get("cpuset", "")):
        cpuset = cpuset or read(os.path.join("/sys/fs/cgroup/cpuset", rel, "cpuset.effective_cpus")) \
            or read(os.path.join("/sys/fs/cgroup/cpuset", rel, "cpuset.cpus"))
    return {"quota_cores": round(best, 2) if best is not None else None, "source": source or "no CFS quota found in this process's cgroup path",
            "cpuset_cpus": cpuset}


def algorithmic_bytes_per_frame(fmt, sw, sh, dw, dh):
    """every source sample read once + every output sample written once."""
    b = 0
    for (w, h), (ow, oh) in zip(fmt.plane_dims(sw, sh), fmt.plane_dims(dw, dh)):
        b += (w * h + ow * oh) * fmt.sample_bytes
    return b


def cpu_model():
    try:
        for line in open("/proc/cpuinfo", encoding="utf-8", errors="replace"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg_name, scan_s=2.0, sample_s=6.0):
    """Times the CPU checker code on this host on a bounded sample of the same workload, row-parallel over OpenMP threads
    (the reference's thr==0 design) and on one core: the own AVX2+FMA code in the reference's opt=2 order (the fast CPU
    path; `value`) and the opt=0 port.  The thread count is chosen from >= scan_s seconds per candidate (a shorter probe
    picked counts that did not hold up), and `value` is the median of the scan's figure and three more runs at
    that count, all of them reported."""
    O = entry.load_oracle()
    fmt_name, sw, sh, dw, dh, kw, _ = CONFIGS[cfg_name]
    fmt = O.FORMATS[fmt_name]
    flt = O.OracleFilter(fmt, sw, sh, dw, dh, **kw)
    src = O.lcg_frame(fmt, sw, sh)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1

    def run(threads, budget, avx2=False, avx512=False):
        n, t0 = 0, time.perf_counter()
        while True:
            if avx512:
                flt.get_frame_simd(3, src, threads=threads, avx512=True)
            elif avx2:
                flt.get_frame_simd(2, src, threads=threads, avx2=True)
            else:
                flt.get_frame(src, threads=threads)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget:
                return dw * dh * n / el / 1e6, n, el

    run(1, 0.0)  # touch tables once
    quota = cpu_quota()
    q = quota["quota_cores"]
    qn = max(1, int(round(q))) if q else None
    # counts to try: the usual powers of two up to the affinity mask and, where the cgroup states a CFS quota, the points
    # around it (half, the quota, twice) -- the knee of the scan belongs to the quota, not to the mask
    cands = sorted({t for t in ((8, 16, 32, 64, 128, avail) + ((max(1, qn // 2), qn, 2 * qn) if qn else ())) if 1 <= t <= avail} or {avail})
    # the fast CPU path: own AVX2 + FMA code in the summation order of the reference's opt=2 path (oracle/simd_avx2.c; the
    # reference itself cannot be built on this box).  Not bit-equal to opt=0 -- the GPU result is; it is the CPU SPEED baseline.
    have_avx2 = bool(O.lib().oracle_avx2_available())
    # ... and own AVX-512 code in the order of the reference's opt=3 path (oracle/simd_avx512.c; north_star: "the reference
    # AVX2/AVX-512 path timed on the same box's host cores"), where the host has AVX-512 F/BW/DQ/VL
    have_avx512 = bool(O.lib().oracle_avx512_available())
    scan = {}
    for t in cands:
        run(t, 0.2, have_avx2)  # settle the thread pool at this size, untimed
        scan[t] = run(t, scan_s, have_avx2)[0]
    best = max(scan, key=scan.get)
    # `value` = median of the scan's figure and three more runs at the chosen count (sample_s seconds in total): on a shared
    # host single runs of this memory-bound loop differ by 20-30 % (r1: scan 1270 / timed 566; r2: 1600 / 1164), so one
    # longer run is no more reproducible than the scan it is compared with; the spread is reported next to it
    reps = [(scan[best], 0, scan_s)]
    for _ in range(3):
        run(best, 0.1, have_avx2)
        reps.append(run(best, sample_s / 3.0, have_avx2))
    vals = sorted(r[0] for r in reps)
    v = 0.5 * (vals[1] + vals[2])
    n, el = sum(r[1] for r in reps[1:]), sum(r[2] for r in reps[1:])
    v1, n1, el1 = run(1, 2.0, have_avx2)
    a512_rows = run(best, 2.0, avx512=True)[0] if have_avx512 else None   # rows of one frame over the same thread count
    a512_one = run(1, 2.0, avx512=True)[0] if have_avx512 else None
    o_best, _, _ = run(best, 2.0)   # the opt=0 port (strict sequential order) at the same thread count ...
    o1, on1, oel1 = run(1, 2.0)     # ... and on one core
    path = ("own AVX2+FMA and AVX-512 code in the reference's opt=2 / opt=3 summation orders (not bit-equal to opt=0)" if have_avx2 and have_avx512
            else "own AVX2+FMA code in the reference's opt=2 summation order (not bit-equal to opt=0)" if have_avx2
            else "oracle (opt=0 port)")
    # SURVEY 8(d)'s method next to it: whole frames in parallel, one single-thread instance per worker (how the reference is
    # deployed: Prefetch(P), MT_MULTI_INSTANCE).  `value` = the better of the two methods, named in `method`.
    fp_counts = sorted({p for p in ((max(1, qn // 2), qn, 2 * qn, min(avail, 4 * qn)) if qn else (16, 64, 128, avail)) if 1 <= p <= avail} or {avail})
    fp = cpu_frame_parallel(cfg_name, fp_counts)
    fp_key = "avx2_order_Mpix_s" if have_avx2 else "opt0_port_Mpix_s"
    fp_best = max(fp[fp_key], key=lambda k: fp[fp_key][k])
    row_value, row_cores = v, best
    method = "rows of one frame over OpenMP threads"
    value_path = "avx2_order" if have_avx2 else "opt0_port"
    if fp[fp_key][fp_best] > v:
        v, best = fp[fp_key][fp_best], int(fp_best)
        method = "whole frames in parallel, one single-thread filter instance per worker (MT_MULTI_INSTANCE)"
    # `value` = the best of the stated paths: the AVX-512-order code takes it where it is the faster one on this host
    a512 = fp["avx512_order_Mpix_s"]
    if a512:
        p512 = max(a512, key=lambda k: a512[k])
        if a512[p512] > v:
            v, best, value_path = a512[p512], int(p512), "avx512_order"
            method = "whole frames in parallel, one single-thread filter instance per worker (MT_MULTI_INSTANCE)"
        if a512_rows and a512_rows > v:
            v, best, value_path, method = a512_rows, row_cores, "avx512_order", "rows of one frame over OpenMP threads"
    # why more workers than `cores` do not help, from this run's own figures: CPU seconds the process was GIVEN per second
    # of wall time at each P (all threads; time.process_time()).  Where that stops growing with P the lease's CPU time --
    # a CFS quota, or neighbours on the same cores -- is the limit, whatever the affinity mask says.
    given = fp["cpu_s_per_wall_s"].get(fp_key, {})
    most = max(given.values()) if given else None
    knee = (f"the process received at most {most:.1f} CPU-seconds per second ({given}) although {avail} CPUs are in its affinity mask"
            + (f"; cgroup quota {q} cores ({quota['source']})" if q else "; no CFS quota is visible from inside this cgroup")) if most else None
    return {"value": round(v, 2), "unit": "Mpix/s", "cores": best, "kind": "port", "method": method,
            "cpu_quota_cores": q, "cpu_quota_source": quota["source"], "cpuset_cpus": quota["cpuset_cpus"],
            "cpu_seconds_per_wall_second_at_most": round(most, 1) if most else None, "knee": knee,
            "sample": f"{path}; best of two methods on a host with {avail} CPUs in the affinity mask ({cpu_model()}): (a) frame-parallel, "
                      f"P single-thread instances for P in {fp_counts}, {fp['seconds_per_point']:.0f}s each; (b) {n} frames of {cfg_name} "
                      f"in {el:.1f}s with rows over {row_cores} OpenMP threads, the fastest of {cands} at {scan_s:.0f}s each",
            "path": value_path,
            "avx2_order_Mpix_s": fp["avx2_order_Mpix_s"] or None, "avx512_order_Mpix_s": fp["avx512_order_Mpix_s"] or None,
            "avx512_order_row_parallel_value": round(a512_rows, 2) if a512_rows else None,
            "avx512_order_single_core_value": round(a512_one, 2) if a512_one else None,
            "avx512_available": have_avx512,
            "frame_parallel": fp, "row_parallel_value": round(row_value, 2), "row_parallel_cores": row_cores,
            "cpu_model": cpu_model(), "affinity_cpus": avail,
            "single_core_value": round(v1, 2), "single_core_sample": f"{n1} frames in {el1:.1f}s",
            "runs_at_chosen_count_Mpix_s": [round(x, 1) for x in vals],
            "opt0_port_value": round(o_best, 2), "opt0_port_single_core_value": round(o1, 2),
            "thread_scan_Mpix_s": {str(k): round(x, 1) for k, x in scan.items()},
            "port_vs_reference_build_container": port_vs_reference_record()}


def port_vs_reference_record():
    """How the timed port compares with the reference's own AVX2 / AVX-512 code where both have run: the build container, one
    thread, C2 (profiles/cpu_port_vs_reference.py wrote the record; the reference's figures in it are SURVEY.md section 6's
    and the round-5 judge's -- this repository cannot build the reference).  A static record, not measured by this run."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "cpu_port_build_container.json")))
    except (OSError, ValueError):
        return None
    rec.pop("per_round_Mpix_s", None)
    return rec


def cpu_frame_parallel(cfg_name, counts, seconds=2.0):
    """SURVEY 8(d)'s method: P single-thread workers, each with its OWN filter instance (tables of its own, like the P instances
    AviSynth creates under Prefetch(P) for an MT_MULTI_INSTANCE filter, ref JincResize.cpp:649-652) and its own frames, all
    running whole frames at once; aggregate Mpix/s for each P in `counts`, for the AVX2-order code and the opt=0 port."""
    import threading
    O = entry.load_oracle()
    fmt_name, sw, sh, dw, dh, kw, _ = CONFIGS[cfg_name]
    fmt = O.FORMATS[fmt_name]
    have_avx2 = bool(O.lib().oracle_avx2_available())
    have_avx512 = bool(O.lib().oracle_avx512_available())
    pmax = max(counts)
    workers = [None] * pmax

    def build(k):  # instance k: plan build included here, untimed (the reference pays it per instance at script load)
        flt = O.OracleFilter(fmt, sw, sh, dw, dh, **kw)
        src = O.lcg_frame(fmt, sw, sh, seed=12345 + k)
        dst = [O.alloc_plane(w, h, fmt.dtype) for (w, h) in flt.out_dims()]
        workers[k] = (flt, src, dst)

    t_build = time.perf_counter()
    ts = [threading.Thread(target=build, args=(k,)) for k in range(pmax)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    t_build = time.perf_counter() - t_build

    def run(p, path):
        avx2, avx512 = path == "avx2", path == "avx512"
        done = [0] * p
        go, stop = threading.Event(), threading.Event()

        def work(k):
            flt, src, dst = workers[k]
            go.wait()
            while not stop.is_set():
                for i, d in enumerate(dst):   # one frame = every plane, one thread (the library call releases the GIL)
                    t = flt.table_for_plane(i)
                    if avx512:
                        t.resize_simd(3, src[i], d, -0.5 if (i and not fmt.rgb) else 0.0, 1, False, True)
                    elif avx2:
                        t.resize_simd(2, src[i], d, -0.5 if (i and not fmt.rgb) else 0.0, 1, True)
                    else:
                        t.resize(src[i], d, flt.peak, 1)
                done[k] += 1

        ts = [threading.Thread(target=work, args=(k,)) for k in range(p)]
        [t.start() for t in ts]
        t0, c0 = time.perf_counter(), time.process_time()
        go.set()
        time.sleep(seconds)
        stop.set()
        [t.join() for t in ts]   # frames in progress are finished and counted: elapsed includes them
        el, cpu = time.perf_counter() - t0, time.process_time() - c0
        return sum(done) * dw * dh / el / 1e6, cpu / el

    out = {"instances_built_in_s": round(t_build, 2), "seconds_per_point": seconds, "avx2_order_Mpix_s": {}, "avx512_order_Mpix_s": {},
           "opt0_port_Mpix_s": {},
           # CPU seconds (all threads of this process) per second of wall time while P workers ran: the cores it really got
           "cpu_s_per_wall_s": {"avx2_order_Mpix_s": {}, "avx512_order_Mpix_s": {}, "opt0_port_Mpix_s": {}}}
    for p in counts:
        for key, path in (("avx2_order_Mpix_s", "avx2"), ("avx512_order_Mpix_s", "avx512"), ("opt0_port_Mpix_s", "opt0")):
            if (path == "avx2" and not have_avx2) or (path == "avx512" and not have_avx512):
                continue
            rate, cores = run(p, path)
            out[key][str(p)] = round(rate, 1)
            out["cpu_s_per_wall_s"][key][str(p)] = round(cores, 1)
    return out


def host_mode_name(pin_mode, batch=False):
    return {0: "pageable, copied by the CPU through the library's pinned buffers (the library's default)",
            3: "pageable, handed to the HIP runtime as they are"}.get(
        int(pin_mode), "pinned once, kept until jinc_batch_free" if batch else "pinned once, cached by address (frame pool)")


# How the host-to-host records treat the caller's planes (register_host_buffers): the library's default unless the environment says
# otherwise (2: registered once and cached -- round 3 - 5's records and profiles/round6/host_modes.log; 3: handed to the runtime).
E2E_MODE = int(os.environ.get("JINC_BENCH_E2E_MODE", "0"))


def e2e_record(pkg, config, depth=128, seconds=1.5, pin_mode=E2E_MODE):
    """Host planes in, host planes out through the look-ahead pipeline (jinc_filter_submit / _wait, one host thread,
    caller buffers treated as `pin_mode` says): frames/s and host GB/s.  PCIe-inclusive, therefore NOT `value`; recorded next to it.
    pin_mode 2 (any non-zero value but 3): registrations cached by address (this function's buffers live as long as the instance: a
    frame pool); 0: pageable planes, copied by the CPU through the library's own pinned buffers (the library's default); 3: pageable
    planes handed to the HIP runtime as they are."""
    import numpy as np
    fmt_name, sw, sh, dw, dh, kw, _ = CONFIGS[config]
    fmt = pkg.FORMATS[fmt_name]
    frame_bytes = algorithmic_bytes_per_frame(fmt, sw, sh, dw, dh)
    depth = max(1, min(depth, int((2 << 30) // max(1, frame_bytes))))   # at most ~2 GiB of host frames (C2: 128, C4: 4)
    f = pkg.Filter(fmt, sw, sh, dw, dh, device=0, **kw)
    f.set_pipeline(depth, pin_mode)
    rng = np.random.default_rng(3)
    nbuf = depth + 1
    srcs, dsts = [], []
    for _ in range(nbuf):
        planes = []
        for (w, h) in fmt.plane_dims(sw, sh):
            p = pkg.alloc_plane(w, h, fmt.dtype)
            p[:] = (rng.random(p.shape) * (((1 << fmt.bits) - 1) if fmt.sample_bytes < 4 else 1)).astype(fmt.dtype)
            planes.append(p)
        srcs.append(planes)
        dsts.append([pkg.alloc_plane(w, h, fmt.dtype) for (w, h) in fmt.plane_dims(dw, dh)])
    tickets = []

    def pump(n):
        tickets.append(f.submit(srcs[n % nbuf], dsts[n % nbuf]))
        if len(tickets) >= depth:
            f.wait(tickets.pop(0))

    for k in range(2 * nbuf):  # warm-up: allocations, registration
        pump(k)
    while tickets:
        f.wait(tickets.pop(0))
    n, t0, kernel = 0, time.perf_counter(), ""
    while time.perf_counter() - t0 < seconds:
        pump(n)
        n += 1
        if n == 2 * depth:
            kernel = f.last_kernel(0)   # steady state, not the final partial group
    while tickets:
        f.wait(tickets.pop(0))
    el = time.perf_counter() - t0
    rec = {"what": "host planes -> jinc_filter_submit/_wait -> host planes, one host thread; not `value`",
           "pin_mode": host_mode_name(pin_mode),
           "frames_per_s": round(n / el, 1), "Mpix_per_s": round(n / el * dw * dh / 1e6, 1),
           "host_GB_per_s": round(n / el * frame_bytes / 1e9, 2), "frames_in_flight": depth, "frames_per_launch": f.pipeline_group,
           "kernel": kernel, "seconds": round(el, 2)}
    f.close()
    return rec


def e2e_batch_record(pkg, config, ndevices, seconds=2.0, streams=32, pin_mode=0):
    """The path a plugin or batch tool takes on a node: host planes -> jinc_batch_process (frame n -> device n mod G, one worker
    and one registrar thread per device on the CPUs of the device's NUMA node, `streams` frames in flight per device, no
    collective) -> host planes.  Total and per-device frames/s, GB/s per link.  PCIe-inclusive, NOT `value`.  Run by rank 0
    from ONE process over the first `ndevices` visible devices after the timed region (VERDICT r5 Next 6a: the 1 -> 8 curve
    of the device-resident kernels scales trivially; this is the leg with a serial stage in it)."""
    import numpy as np
    fmt_name, sw, sh, dw, dh, kw, _ = CONFIGS[config]
    fmt = pkg.FORMATS[fmt_name]
    frame_bytes = algorithmic_bytes_per_frame(fmt, sw, sh, dw, dh)
    ndevices = max(1, min(int(ndevices), pkg.device_count()))
    in_flight = 2 * streams * ndevices
    nbuf = max(2, min(in_flight + 16, int((6 << 30) // max(1, frame_bytes))))   # distinct host frames: at most ~6 GiB
    streams = max(1, min(streams, nbuf // (2 * ndevices)))                       # a buffer is never in flight twice
    per_call = max(nbuf, (256 * ndevices) // nbuf * nbuf)                         # frames per jinc_batch_process call (buffers cycle)
    rng = np.random.default_rng(5)
    srcs, dsts = [], []
    for _ in range(nbuf):
        planes = []
        for (w, h) in fmt.plane_dims(sw, sh):
            p = pkg.alloc_plane(w, h, fmt.dtype)
            p[:] = (rng.random(p.shape) * (((1 << fmt.bits) - 1) if fmt.sample_bytes < 4 else 1)).astype(fmt.dtype)
            planes.append(p)
        srcs.append(planes)
        dsts.append([pkg.alloc_plane(w, h, fmt.dtype) for (w, h) in fmt.plane_dims(dw, dh)])
    b = pkg.Batch(fmt, sw, sh, dw, dh, ndevices=ndevices, streams=streams, register_host_buffers=int(pin_mode), **kw)
    frames = [srcs[k % nbuf] for k in range(per_call)]
    outs = [dsts[k % nbuf] for k in range(per_call)]
    b.process(frames, outs)   # warm-up: group buffers, registration of the host frames
    n, calls, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        b.process(frames, outs)
        n += per_call
        calls += 1
    el = time.perf_counter() - t0
    G = b.devices
    rec = {"what": "host planes -> jinc_batch_process (frame n -> device n mod G; a worker and a registrar thread per device, NUMA-bound) -> host planes; not `value`",
           "devices": G, "frames_per_s": round(n / el, 1), "frames_per_s_per_device": round(n / el / G, 1),
           "Mpix_per_s": round(n / el * dw * dh / 1e6, 1), "host_GB_per_s": round(n / el * frame_bytes / 1e9, 2),
           "GB_per_s_per_link": round(n / el * frame_bytes / 1e9 / G, 2), "frames_in_flight_per_device": streams,
           "frames_per_call": per_call, "calls": calls, "distinct_host_frames": nbuf, "pin_mode": host_mode_name(pin_mode, batch=True),
           "cpus_of_device": {str(d): (lambda c: f"{len(c)} CPUs ({c[0]}..{c[-1]})" if c else "unknown: not bound")(b.device_cpus(d)) for d in range(G)},
           "seconds": round(el, 2)}
    b.close()
    return rec


def through_pinned_to_device(torch, t):
    """A torch CPU tensor on the device through pinned memory of torch's own.  Plain `.to("cuda")` / `.cpu()` hand heap pages of this
    process to the HIP runtime, which maps them into the device behind the copy: the path round 6's unexplained GPU memory access
    faults came from (once inside torch's own `tensor.cpu()`; profiles/round6/README.md).  The bench keeps its own copies off it."""
    pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pinned.copy_(t.contiguous())
    return pinned.to("cuda")


def through_pinned_to_host(torch, t):
    pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pinned.copy_(t)
    return pinned.clone()


def lcg_planes(fmt, w, h, seed=12345):
    """SURVEY.md Appendix A item 4, the synthetic frame every known answer of tests/golden/kat.json was recorded on: one 32-bit
    LCG stream (s = s * 1664525 + 1013904223, r = s >> 8) over the planes in processing order, rows without padding.  Own
    numpy code (the oracle is not imported on this path): s_k = A_k * s_0 + C_k with A, C built by doubling, all mod 2^32."""
    import numpy as np
    dims = fmt.plane_dims(w, h)
    total = sum(pw * ph for pw, ph in dims)
    A = np.empty(total, dtype=np.uint32)
    Cc = np.empty(total, dtype=np.uint32)
    A[0], Cc[0] = 1664525, 1013904223
    n = 1
    with np.errstate(over="ignore"):
        while n < total:
            m = min(n, total - n)
            A[n:n + m] = A[:m] * A[n - 1]
            Cc[n:n + m] = A[:m] * Cc[n - 1] + Cc[:m]
            n += m
        state = A * np.uint32(seed & 0xFFFFFFFF) + Cc
    r = state >> np.uint32(8)
    if fmt.sample_bytes == 1:
        vals = (r & np.uint32(0xFF)).astype(np.uint8)
    elif fmt.sample_bytes == 2:
        vals = (r & np.uint32((1 << fmt.bits) - 1)).astype(np.uint16)
    else:
        vals = (r & np.uint32(0xFFFFFF)).astype(np.float32) / np.float32(16777215.0)
    out, at = [], 0
    for pw, ph in dims:
        out.append(vals[at:at + pw * ph].reshape(ph, pw))
        at += pw * ph
    return out


def known_answer(config):
    """crc32 of the reference's opt=0 output on the Appendix-A frame for `config` (tests/golden/kat.json: recorded by the survey
    from executing the reference), or None where no known answer exists.  A constant compare: no oracle on this path."""
    fmt_name, sw, sh, dw, dh, kw, _ = CONFIGS[config]
    try:
        kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json"), encoding="utf-8"))["outputs"]
    except Exception:  # noqa: BLE001
        return None
    for rec in kat:
        if (rec["format"] == fmt_name and rec["src"] == [sw, sh] and rec["dst"] == [dw, dh]
                and all(rec["args"].get(k) == v for k, v in kw.items()) and set(rec["args"]) == set(kw)):
            return rec
    return None


def check_positions(frames):
    """Batch positions that carry the Appendix-A frame: the first frame, one in the middle of a 64-frame group and the last one
    (kernels whose lanes are frames -- frame-lane, sub-group and frame-pair forms, border forms that depend on the
    batch size -- can be wrong at other batch positions than 0)."""
    return sorted({0, min(frames - 1, 37), frames - 1})


def self_check(torch, config, fmt, dst_t, ddims):
    """crc32 over the batch's OUTPUT at every position whose input is the Appendix-A frame (check_positions; planes in
    processing order, rows truncated to the row size) against the known answer: the kernels the timed region ran -- at the
    batch size it ran them -- have reproduced the reference's bytes at each of them, or the benchmark fails."""
    import zlib
    rec = known_answer(config)
    if rec is None:
        return {"status": "no known answer for this config", "crc32": None}
    frames = int(dst_t[0].shape[0])
    got, ok, nbytes = {}, True, 0
    for pos in check_positions(frames):
        c, nbytes = 0, 0
        for t, (w, h) in zip(dst_t, ddims):
            if t.dtype == torch.uint16:
                t = t.view(torch.int16)
            host = through_pinned_to_host(torch, t[pos, :h, :w].contiguous()).numpy()
            c = zlib.crc32(host.tobytes(), c)
            nbytes += host.nbytes
        got[pos] = f"{c & 0xFFFFFFFF:08x}"
        ok = ok and got[pos] == rec["crc32"] and nbytes == rec["bytes"]
    return {"status": "ok" if ok else "MISMATCH", "crc32": got[0], "crc32_by_batch_position": {str(k): v for k, v in got.items()},
            "expected": rec["crc32"], "bytes": nbytes, "frames_checked": sorted(got),
            "source": f"tests/golden/kat.json [{rec['name']}] (reference opt=0 on the Appendix-A LCG frame, seed 12345)"}


def make_workload(pkg, torch, config, frames, device, seed, lcg_first=False):
    """Creates the filter and a batch of `frames` synthetic frames resident in HBM (random samples, not
    zeros: DVFS differs); returns (filter, step(), stream, format, output plane dims).  One step() =
    one jinc_filter_process_device call over the whole batch on torch's current stream."""
    fmt_name, sw, sh, dw, dh, kw, _ = CONFIGS[config]
    fmt = pkg.FORMATS[fmt_name]
    flt = pkg.Filter(fmt, sw, sh, dw, dh, device=device, **kw)
    tdtype = {1: torch.uint8, 2: torch.uint16, 4: torch.float32}[fmt.sample_bytes]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    sdims, ddims = fmt.plane_dims(sw, sh), fmt.plane_dims(dw, dh)
    sb = fmt.sample_bytes

    def pitch_elems(w):
        return ((w * sb + 255) // 256 * 256) // sb

    src_t, dst_t = [], []
    for (w, h) in sdims:
        shape = (frames, h, pitch_elems(w))
        t = torch.empty(shape, device="cuda", dtype=torch.float32 if sb == 4 else (torch.uint8 if sb == 1 else torch.int16))
        for f0 in range(0, frames, 64):  # in chunks: randint produces int32 first
            part = (min(64, frames - f0), h, pitch_elems(w))
            if sb == 4:
                t[f0:f0 + part[0]] = torch.rand(part, device="cuda", generator=gen, dtype=torch.float32)
            else:
                t[f0:f0 + part[0]] = torch.randint(0, 1 << fmt.bits, part, device="cuda", generator=gen, dtype=torch.int32).to(t.dtype)
        if sb == 2:
            t = t.view(torch.uint16)
        src_t.append(t)
    if lcg_first:   # the Appendix-A frame the known answers were recorded on, at the batch positions self_check reads
        import numpy as np
        for t, plane, (w, h) in zip(src_t, lcg_planes(fmt, sw, sh), sdims):
            host = np.ascontiguousarray(plane)
            dev_plane = through_pinned_to_device(torch, torch.from_numpy(host.view(np.int16) if sb == 2 else host))
            for pos in check_positions(frames):
                if sb == 2:
                    t.view(torch.int16)[pos, :h, :w] = dev_plane
                else:
                    t[pos, :h, :w] = dev_plane
    # A/B knob (measurements only): JINC_BENCH_DST_SHIFT = bytes the destination planes start beyond a 256-byte boundary (one spare
    # row per frame holds the overhang) -- which store alignment do the kernels meet?
    dst_shift = int(os.environ.get("JINC_BENCH_DST_SHIFT", "0"))
    for (w, h) in ddims:
        dst_t.append(torch.zeros((frames, h + (1 if dst_shift else 0), pitch_elems(w)), device="cuda", dtype=tdtype))
    if dst_shift:
        lcg_first = False   # (shifted destinations: the output's rows do not start where the tensors' do)
    sp = [t.data_ptr() for t in src_t]
    spitch = [t.stride(1) * sb for t in src_t]
    sstride = [t.stride(0) * sb for t in src_t]
    dp = [t.data_ptr() + dst_shift for t in dst_t]
    dpitch = [t.stride(1) * sb for t in dst_t]
    dstride = [t.stride(0) * sb for t in dst_t]
    stream = torch.cuda.current_stream()

    def step():
        flt.process_device(sp, spitch, sstride, dp, dpitch, dstride, frames, stream=stream.cuda_stream)

    step.keepalive = (src_t, dst_t)
    step.checkable = lcg_first
    return flt, step, stream, fmt, ddims


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default 100: about 1 s of GPU time on C2; VERDICT r4: 20 steps were 0.19 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=0, help="frames per step per GPU (default per config)")
    ap.add_argument("--config", default="C2", help="one of CONFIGS (scripts may add entries before calling main())")
    ap.add_argument("--sync", choices=("auto", "rccl", "store"), default="auto",
                    help="how ranks meet for the barrier and the MAX / SUM: rccl = torch.distributed (default under a launcher), "
                         "store = a TCPStore on 127.0.0.1, no RCCL (default when bench.py starts the ranks itself)")
    ap.add_argument("--inproc", action="store_true", help="N devices from ONE process: a filter instance and a stream per device, one host thread")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-host record (untimed, after the timed region)")
    ap.add_argument("--no-clock-sampler", action="store_true")
    ap.add_argument("--kernel-mode", type=int, default=0, help="0 auto, 1 force gather kernel")
    ap.add_argument("--simd-order", type=int, default=0, help="1 / 2 / 3: the compatibility kernel in the reference's SSE4.1 / AVX2 / AVX-512 summation order")
    ap.add_argument("--border-overlap", type=int, default=-1, help="-1 automatic, 0 serial, 1 border kernel on a side stream")
    ap.add_argument("--border-strips", type=int, default=-1, help="-1 default, 1 strip kernels, 2 row strips only, 0 gather kernel over the border frame")
    ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B / tuning knob of the library (include/jincresize_hip_test.h enum jinc_knob, lower-case name, e.g. quad_rg=8); "
                         "JINC_<NAME> environment variables are translated the same way -- the library itself reads no environment")
    ap.add_argument("--no-self-check", action="store_true", help="skip the crc32 of frame 0's output against tests/golden/kat.json")
    args = ap.parse_args(argv)
    if args.config not in CONFIGS:
        ap.error(f"unknown --config {args.config}; choose from {', '.join(sorted(CONFIGS))}")
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    return args


def main(argv=None):
    args = parse_args(argv)
    world_env = os.environ.get("WORLD_SIZE")
    # started plainly with --gpus N > 1: this process starts the N ranks itself and only waits for them (it must not touch
    # the GPU: its children are fresh interpreters, and nothing that has initialised HIP is forked or exec'ed)
    # (JINC_BENCH_SELF_LAUNCH=1 takes this way for N = 1 as well: the one-GPU box's test of it)
    if world_env is None and not args.inproc and (args.gpus > 1 or os.environ.get("JINC_BENCH_SELF_LAUNCH") == "1"):
        if under_profiler():
            # the profiler's preloaded tool has initialised the GPU in THIS process already: starting rank processes from it
            # would be the exec from a GPU-initialised process this pool forbids
            raise SystemExit("bench.py --gpus N under rocprofv3: the self-launch path starts processes from a profiled (GPU-initialised) "
                             "parent; profile one rank (`--gpus 1`) or all devices from one process (`--gpus N --inproc`) instead")
        have = visible_gpu_count()   # KFD topology + *_VISIBLE_DEVICES: no HIP call in this process
        if have is not None and have < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this host shows {have} HIP device(s)")
        sync = "store" if args.sync == "auto" else args.sync
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv), sync=sync))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.inproc:
        if world != 1:
            raise SystemExit("bench.py --inproc runs in one process: start it without a launcher")
    elif world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} under a launcher that started {world} rank(s): the two must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists in the product path)")
    devices = list(range(args.gpus)) if args.inproc else [local_rank]
    if args.inproc and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py --inproc --gpus {args.gpus}: this host shows {torch.cuda.device_count()} HIP device(s)")
    torch.cuda.set_device(devices[0])
    # under torch.distributed.run (RANK set) the process group is always created, also for one rank, so that
    # the N = 1 launch exercises the same RCCL barrier / reductions as N = 8
    how = os.environ.get("JINC_BENCH_SYNC", "") if args.sync == "auto" else args.sync
    if world > 1 or "RANK" in os.environ:
        if how == "store":
            sync = StoreSync(rank, world)
        else:
            import torch.distributed as dist
            sync = DistSync(dist, rank, world, local_rank, "nccl")  # nccl == RCCL on ROCm
    else:
        sync = NoSync()
    n_gpus = args.gpus if args.inproc else world

    pkg = entry.load_package()
    knobs_applied = pkg.apply_env_knobs()
    for kv in args.knob:
        name, _, val = kv.partition("=")
        pkg.set_knob(name.strip(), float(val))
        knobs_applied[name.strip().lower()] = float(val)
    fmt_name, sw, sh, dw, dh, kw, default_frames = CONFIGS[args.config]
    strong = args.config == "C5"
    profiled = under_profiler()
    quiet = profiled or args.no_clock_sampler   # no second dispatch, no probe kernels in a profile

    # one workload per device this process drives (one, unless --inproc): a plan replica, a batch resident in that device's
    # HBM, a stream.  Frames are the sharding unit; with C5 (one clip, strong scaling) shard k owns a contiguous run of them.
    loads = []
    for k, dev in enumerate(devices):
        shard, nshards = (k, len(devices)) if args.inproc else (rank, world)
        B = args.frames or default_frames
        if strong:
            B = shard_frames(B, shard, nshards)[1]
            if B < 1:
                raise SystemExit("C5: more ranks than frames")
        torch.cuda.set_device(dev)
        flt, step, stream, fmt, ddims = make_workload(pkg, torch, args.config, B, dev, 12345 + shard * B,
                                                      lcg_first=not args.no_self_check)   # every rank / device checks its own batch
        flt.set_kernel_mode(args.kernel_mode)
        if args.simd_order:
            flt.set_simd_order(args.simd_order)
        if args.border_overlap >= 0:
            flt.set_border_overlap(bool(args.border_overlap))
        if args.border_strips >= 0:
            flt.set_border_strips(args.border_strips)
        loads.append({"device": dev, "filter": flt, "step": step, "stream": stream, "frames": B})
    torch.cuda.set_device(devices[0])
    flt, stream, B = loads[0]["filter"], loads[0]["stream"], loads[0]["frames"]
    info = flt.plan_info(0)
    sb = fmt.sample_bytes

    def step_all():   # one step on every device of this process: launches only, nothing waits in here
        for w in loads:
            w["step"]()

    def sync_all():
        for w in loads:
            torch.cuda.synchronize(w["device"])

    # Untimed device spin-up before the W warm-up steps: a C2 step is ~1 ms, so a handful of warm-up steps ends before the
    # shader clock and the caches have settled (measured: 482 vs 541 Gpix/s for --steps 5 --warmup 2 without / with it).
    # JINC_BENCH_SPINUP_MS=0 switches it off.
    spin_ms = float(os.environ.get("JINC_BENCH_SPINUP_MS", "300"))
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < spin_ms:
        step_all()
        sync_all()
    for _ in range(args.warmup):
        step_all()
    sync_all()
    flt.set_profiling(True)
    flt.kernel_times()  # reset
    sync.barrier()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_all()
    sync_all()
    sync.barrier()
    elapsed = time.perf_counter() - t0
    per_ms, per_n, gat_ms, gat_n = flt.kernel_times()
    flt.set_profiling(False)
    dom_instance = flt.last_instance(0)   # the timed steps' interior kernel with its template arguments
    # every rank checks the batch of every device it drives; rank 0 reports its own detail and how many batches of the job failed
    check, bad_here = None, 0
    if not args.simd_order or FORCE_SELF_CHECK:
        for w in loads:
            if not getattr(w["step"], "checkable", False):
                continue
            torch.cuda.set_device(w["device"])
            c = self_check(torch, args.config, fmt, w["step"].keepalive[1], ddims)
            bad_here += c["status"] == "MISMATCH"
            if check is None or c["status"] == "MISMATCH":
                check = c
        torch.cuda.set_device(devices[0])
    _, bad_total, bad_by_rank = sync.reduce(0.0, float(bad_here))
    if check is not None:
        check["batches_failed_in_the_job"] = int(bad_total)
        check["batches_failed_by_rank"] = [int(x) for x in bad_by_rank]
        if bad_total:
            check["status"] = "MISMATCH"
    # Shader clock under this load, in a SECOND, untimed pass of the same steps with eight single-lane samplers (one per XCD)
    # beside the kernels (kernel_probe.hip).  Not during the timed region: any second dispatch that stays active, however
    # small, costs kernels with short-lived workgroups 10-15 % (1080p -> 720p 253 -> 222 Gpix/s, C2 1 %;
    # profiles/round3/clock_sampler_priority.log), so the value is taken without it and the clock right after.
    clock_ghz = None
    if rank == 0 and not quiet:
        sampler = pkg.ClockSampler(devices[0], 60.0)
        for _ in range(args.steps):
            loads[0]["step"]()
        stream.synchronize()   # the steps' stream only: a device-wide synchronize would wait for the samplers themselves
        clock_ghz = sampler.stop()
    # untimed, right after the timed region (clocks warm): what the kernels' instruction pair sustains on THIS part
    pair_probe = None
    if rank == 0 and not quiet:
        try:
            pair_probe = {str(w): pkg.valu_pair_probe(devices[0], w) for w in (4, 6, 8)}
        except Exception:  # noqa: BLE001
            pair_probe = None
    # what the border kernels cost when nothing runs beside them: a few untimed steps with the border launches queued
    # BEHIND the interior on the same stream (their event times beside the interior only say how long they sat there)
    border_alone_ms = None
    if rank == 0 and not quiet and per_n > 0 and gat_n > 0 and args.border_overlap < 0:
        try:
            flt.set_border_overlap(False)
            flt.set_profiling(True)
            flt.kernel_times()
            for _ in range(3):
                loads[0]["step"]()
            torch.cuda.synchronize(devices[0])
            border_alone_ms = flt.kernel_times()[2] / 3.0
            flt.set_profiling(False)
            flt.set_border_overlap(None)
        except Exception:  # noqa: BLE001
            border_alone_ms = None

    frames_done = float(sum(w["frames"] for w in loads) * args.steps)
    elapsed_max, frames_all, frames_by_rank = sync.reduce(elapsed, frames_done)
    if args.inproc:
        frames_by_rank = [float(w["frames"] * args.steps) for w in loads]
    mpix = frames_all * dw * dh / elapsed_max / 1e6

    if rank == 0:
        bytes_frame = algorithmic_bytes_per_frame(fmt, sw, sh, dw, dh)
        fs = info.filter_size
        samples_frame = sum(w * h for (w, h) in ddims)
        src_bytes_frame = sum(w * h for (w, h) in fmt.plane_dims(sw, sh)) * sb
        if per_n > 0:
            dom_name, dom_ms, dom_n = flt.last_kernel(0), per_ms, per_n
            if dom_instance.startswith(dom_name):
                dom_name = dom_instance   # as rocprofv3 names it: "ewa_periodic_quad2_kernel<unsigned char, 8, 1026u, 6>"
        else:   # whole planes on the gather kernel (or, with --simd-order, on the compatibility kernel)
            dom_name, dom_ms, dom_n = (flt.last_kernel(0) or "ewa_gather_kernel"), gat_ms, gat_n
        # one launch per plane per step; algorithmic bytes of a launch = the batch's bytes for that plane,
        # so summed over the planes of a step it is bytes_frame * B
        launches_per_step = max(1, dom_n // max(1, args.steps))
        kernel_ms_per_step = dom_ms / args.steps
        achieved_gbs = bytes_frame * B / (kernel_ms_per_step * 1e-3) / 1e9
        # operations the reference's chain holds per sample (2 fs^2) and operations the kernels EXECUTE: the periodic
        # kernels leave out taps whose coefficient is exactly 0.0f (trimmed support; exact), so the
        # VALU's utilisation is counted on what it really does, and the algorithmic figure is reported beside it
        valu_ops_algorithmic = 2.0 * fs * fs * samples_frame * B / (kernel_ms_per_step * 1e-3)
        taps_exec, taps_ref = 0.0, 0.0
        for i, (w, h) in enumerate(ddims):
            tbl = 1 if (flt.num_tables > 1 and i in (1, 2)) else 0
            fs_t = flt.plan_info(tbl).filter_size
            kname = flt.last_kernel(tbl)
            taps = (flt.periodic_taps(tbl, rows_kernel=3) if kname in ("ewa_periodic_quad2_kernel", "ewa_periodic_quad8_kernel", "ewa_periodic_quad2x8_kernel")
                    else flt.periodic_taps(tbl, rows_kernel=4) if kname == "ewa_periodic_rowpair_kernel"
                    else flt.periodic_taps(tbl, rows_kernel="rows" in kname) if kname.startswith("ewa_periodic")
                    else flt.periodic_taps(tbl, rows_kernel=2) if kname == "ewa_direct_kernel" else 0.0)
            taps_exec += w * h * (taps or fs_t * fs_t)
            taps_ref += w * h * fs_t * fs_t
        valu_ops = 2.0 * taps_exec * B / (kernel_ms_per_step * 1e-3)
        valu_ops_algorithmic = 2.0 * taps_ref * B / (kernel_ms_per_step * 1e-3)
        traffic = traffic_raw = None
        pmc_clock = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath)).get(args.config, {})
                if "effective_clock_ghz" in rec:   # GRBM_GUI_ACTIVE / 8 / kernel time of the committed PMC pass
                    pmc_clock = {"ghz": rec["effective_clock_ghz"], "kernel_ms_in_that_pass": rec.get("clock_pass_kernel_ms"),
                                 "source": "profiles/traffic.json (rocprofv3 --pmc GRBM_GUI_ACTIVE pass of this command)"}
                # the PMC figure belongs to ONE kernel: when knobs or the kernel mode route the interior elsewhere it is not this run's
                recorded = rec.get("kernel")
                same_kernel = not recorded or dom_instance == recorded
                if "hbm_bytes_per_launch" in rec and not same_kernel:
                    traffic = traffic_raw = None
                elif "hbm_bytes_per_launch" in rec:  # PMC figure (2 x FETCH_SIZE + WRITE_SIZE), scaled to this run's frames per launch
                    scale = B / (rec.get("frames_per_launch") or B)
                    traffic = int(rec["hbm_bytes_per_launch"] * scale)
                    traffic_raw = int(rec.get("hbm_bytes_per_launch_raw", 0) * scale) or None
            except Exception:  # noqa: BLE001
                traffic = traffic_raw = None
        how_parallel = (f"frames sharded over {n_gpus} GPU(s), no collective; "
                        + ("one process, a filter instance and a stream per device" if args.inproc else
                           f"one process per GPU, ranks meet over {sync.name}" if n_gpus > 1 or sync.name != "none" else "one process"))
        line = {
            "metric": baseline_metric() if args.config == "C2" else f"Mpix/s ({args.config})",
            "value": round(mpix, 1), "unit": "Mpix/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32",  # arithmetic type of the path (un-fused fp32 accumulate over u8/u16/f32 samples)
            "data": "synthetic",
            "config": {"workload": f"{args.config}: {sw}x{sh}->{dw}x{dh} {fmt_name} tap={kw['tap']}"
                                   + (f" blur={kw['blur']}" if 'blur' in kw else ""),
                       "sample_type": {1: "u8", 2: "u16", 4: "f32"}[sb], "frames_per_step_per_gpu": B,
                       "frames_per_rank": [int(x) for x in frames_by_rank], "sync": sync.name, "launch": "inproc" if args.inproc else
                       ("self-launched ranks" if os.environ.get("JINC_BENCH_SYNC") else ("launcher" if "RANK" in os.environ else "single process")),
                       "timed_region_s": round(elapsed_max, 4), "resident_bytes_per_gpu": (bytes_frame * B),
                       "untimed_spinup_ms_before_warmup": spin_ms, "parallelism": how_parallel,
                       "kernel": dom_name, "direct_kernel_premise": flt.direct_premise, "filter_size": fs, "plan_sets": info.num_sets,
                       "plan_bytes": int(info.plan_bytes), "under_profiler": profiled, "knobs": knobs_applied or None},
            # frame 0 of the batch is the Appendix-A frame; its OUTPUT after the timed steps against the reference's crc32
            "self_check": check["status"] if check else None, "self_check_detail": check,
            "roofline": {"bound": "hbm",   # the metric's wording (% of the HBM roofline); what BINDS the kernel is binding_roof below
                         "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_note": "2 x FETCH_SIZE + WRITE_SIZE from profiles/traffic.json (gfx950 FETCH correction); raw sum in traffic_raw",
                         "traffic_raw": traffic_raw,
                         "kernel": dom_name, "kernel_ms_per_launch": round(dom_ms / max(1, dom_n), 4),
                         "launches": dom_n, "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": bytes_frame * B // max(1, launches_per_step),
                         "binding_roof": "un-fused fp32 VALU (v_mul_f32+v_add_f32 per tap; FMA/MFMA would break bit-exactness)",
                         "valu_achieved_Tops": round(valu_ops / 1e12, 2), "valu_peak_Tops": VALU_UNFUSED_PEAK / 1e12,
                         "valu_frac": round(valu_ops / VALU_UNFUSED_PEAK, 4),
                         # taps per sample: the reference's chain (fs^2) and what the kernels execute (zero-coefficient taps left out);
                         # valu_* above and below count EXECUTED multiplies and adds, *_algorithmic the reference's 2 fs^2 per sample
                         "taps_per_sample_reference": round(taps_ref / samples_frame, 2), "taps_per_sample_executed": round(taps_exec / samples_frame, 2),
                         # NOT a utilisation: the rate a kernel that executed every one of the reference's 2 fs^2 operations would need
                         # for this throughput, as a multiple of the peak -- above 1 where zero-coefficient taps are elided
                         "reference_equivalent_Tops": round(valu_ops_algorithmic / 1e12, 2),
                         "reference_equivalent_rate_vs_peak_with_zero_taps_elided": round(valu_ops_algorithmic / VALU_UNFUSED_PEAK, 4),
                         # what the part sustains under this load: shader clock sampled beside the same steps in a second, untimed pass
                         # right after the timed one (median / min / max over 8 samplers = XCDs), the VALU peak at that clock and
                         # the fraction of it
                         "shader_clock_ghz": round(clock_ghz[1], 3) if clock_ghz else None,
                         "shader_clock_ghz_min_max": [round(clock_ghz[0], 3), round(clock_ghz[2], 3)] if clock_ghz else None,
                         "valu_frac_at_sampled_clock": round(valu_ops / (256 * 128 * clock_ghz[1] * 1e9), 4) if clock_ghz else None,
                         # the micro-architecture guide's recipe: GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the committed PMC pass of
                         # this command (profiles/traffic.json); it reads ~12 % below the sampled shader clock
                         "effective_clock_ghz": pmc_clock["ghz"] if pmc_clock else None, "effective_clock_pmc": pmc_clock,
                         "valu_peak_at_clock_Tops": round(256 * 128 * pmc_clock["ghz"] * 1e9 / 1e12, 2) if pmc_clock else None,
                         "valu_frac_at_clock": round(valu_ops / (256 * 128 * pmc_clock["ghz"] * 1e9), 4) if pmc_clock else None,
                         # plain v_mul_f32 (SGPR coefficient) + v_add_f32 with nothing else in the loop, chip filled at 4 / 6 / 8
                         # waves per SIMD, measured on this device right after the timed region: [Tops, shader clock GHz]
                         "valu_pair_sustained_Tops": {w: [round(t, 2), round(g, 3)] for w, (t, g) in pair_probe.items()} if pair_probe else None,
                         "valu_frac_of_pair_sustained": round(valu_ops / 1e12 / max(t for t, _ in pair_probe.values()), 4) if pair_probe else None,
                         # the north_star's "HBM-read" reading: source bytes only (each source sample once)
                         "hbm_read_frac": round(src_bytes_frame * B / (kernel_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         # what the border frame costs: the step minus its interior kernel (the border launches run beside the
                         # interior on a side stream), and the border kernels' own duration with nothing beside them (untimed
                         # serial pass after the timed region)
                         "step_minus_interior_ms": round(elapsed_max / args.steps * 1e3 - kernel_ms_per_step, 4) if per_n > 0 else None,
                         "border_ms_alone": round(border_alone_ms, 4) if border_alone_ms is not None else None},
        }
        line["e2e"] = None
        if n_gpus == 1 and not args.no_e2e and not profiled:
            try:
                line["e2e"] = e2e_record(pkg, args.config)
            except Exception as exc:  # noqa: BLE001  (a record next to the value, never a reason to lose the line)
                line["e2e"] = {"error": str(exc)}
        # the host-to-host leg over ALL the job's devices, from this one process (rank 0), for every N -- the curve the
        # driver draws from `value` is the device-resident one; this record is the path with host threads and the links in it
        line["e2e_batch"] = None
        if not args.no_e2e and not profiled:
            try:
                line["e2e_batch"] = e2e_batch_record(pkg, args.config, n_gpus, pin_mode=E2E_MODE)
            except Exception as exc:  # noqa: BLE001
                line["e2e_batch"] = {"error": str(exc)}
        if n_gpus == 1 and not args.no_cpu_baseline and not profiled:
            line["cpu_baseline"] = cpu_baseline(args.config)
        else:
            line["cpu_baseline"] = None
        if check and check["status"] == "MISMATCH":
            print("bench.py: SELF-CHECK FAILED -- frame 0's output does not reproduce the reference's crc32; no result line is printed.\n"
                  + json.dumps(line), file=sys.stderr, flush=True)
            for w in loads:
                w["filter"].close()
            sync.close()
            raise SystemExit(1)
        print(json.dumps(line), flush=True)

    for w in loads:
        w["filter"].close()
    sync.close()


if __name__ == "__main__":
    main()
