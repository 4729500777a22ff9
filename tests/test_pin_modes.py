"""How the library treats the caller's host buffers (jinc_filter_set_pipeline's / jinc_batch_create's register_host_buffers).

Round 6 ended with three modes: 0 (the default) pageable planes copied by the CPU through pinned buffers of the library's own -- the
device never maps the caller's pages; any other value but 3: registered once, the registrations cached by address (the promise of a
host whose frame memory stays mapped -- the `pooling_host` fixture models such a host for the test's duration); 3: pageable planes
handed to the HIP runtime as they are (the default of rounds 1 - 5).  A fourth, "registered while the frame is in flight", was built
and withdrawn (profiles/round6/README.md tells why, and why the default changed).

What is tested here: the default at every pipeline shape, with pitches that are not the row size, out-of-order waits and frames
nobody waits for; the helper threads of the plane copies (and threads = 1, which keeps them off); mode 3; pageable sources with results
into memory the host pinned; cached registrations are kept and given back; jinc_batch_process pins with one registrar per device
(several on this one-device box through the test header) in exact byte ranges that cover every plane whole; the NUMA lookup of
batch.cpp against a fake sysfs tree (CPU); host_copy.cpp's plane copy without a device (CPU).  Every test of the suite ends with the
registry empty and as many hipHostUnregister as hipHostRegister calls (tests/conftest.py)."""
import os

import numpy as np
import pytest

from conftest import assert_planes_equal, fresh_copies, fresh_mapping, fresh_planes


def _case(O):
    fmt, sw, sh, tw, th = "YUV420P8", 320, 180, 640, 360
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    return fmt, sw, sh, tw, th, of


@pytest.mark.gpu
def test_pool_mode_keeps_registrations_and_leaving_it_gives_them_back(gpu_pkg, O, pooling_host):
    """Cached registrations: what a buffer cost to register is paid once; set_pipeline(…, 0) and close() give everything back."""
    fmt, sw, sh, tw, th, of = _case(O)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_pipeline(8, gpu_pkg.PIN_POOL, 4)
    base_ranges = gpu_pkg.transport_counts(reset=True)[2]
    srcs = [fresh_copies(O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=70 + k)) for k in range(8)]
    dsts = [fresh_planes(f.out_dims(), np.uint8) for _ in range(8)]
    for rnd in range(3):
        tickets = [f.submit(s, d) for s, d in zip(srcs, dsts)]
        for t in tickets:
            f.wait(t)
        assert gpu_pkg.transport_counts()[2] == base_ranges + 8 * 6   # 3 source + 3 destination planes per frame, cached
    for k in (0, 7):
        assert_planes_equal(dsts[k], of.get_frame(srcs[k], threads=4), f.out_dims(), what=f"frame {k}")
    f.set_pipeline(8, gpu_pkg.PIN_NONE, 4)
    assert gpu_pkg.transport_counts()[2] == base_ranges and gpu_pkg.host_registrations() == 0
    by_shader, by_dma, _ = gpu_pkg.transport_counts()
    assert (by_shader, by_dma) == (24, 0)
    f.close()


def _padded(planes, pad, fill):
    """Copies of `planes` inside rows `pad` bytes longer, the padding filled with `fill`: (the views, the arrays that own them)."""
    views, owners = [], []
    for p in planes:
        h, w = p.shape
        big = np.full((h, w + pad // p.itemsize), fill, p.dtype)
        big[:, :w] = p
        views.append(big[:, :w])
        owners.append(big)
    return views, owners


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,sw,sh,tw,th", [("YUV420P8", 320, 180, 640, 360), ("Y16", 160, 90, 219, 123), ("YUV444PS", 96, 54, 144, 81)])
def test_default_mode_keeps_the_callers_pages_off_the_device(gpu_pkg, O, fmt, sw, sh, tw, th):
    """register_host_buffers = 0 (every new instance): pageable planes are copied by the CPU through pinned buffers of the
    library's own -- no registration, no plane handed to the runtime -- at one frame in flight and at eight, waits out of order,
    pitches that are not the row size (the bytes between the rows stay what they were), frames nobody waits for."""
    ofmt = O.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=900 + k) for k in range(12)]
    want = [of.get_frame(s, threads=4) for s in srcs]
    dims = f.out_dims()
    gpu_pkg.transport_counts(reset=True)
    assert_planes_equal(f.get_frame(srcs[0]), want[0], dims, what="one synchronous frame")
    assert gpu_pkg.staged_frames() == 1
    for depth, group, defer_kb in ((1, 0, -1), (8, 0, -1), (8, 3, -1), (16, 0, -1), (16, 0, 0)):
        gpu_pkg.set_knob("stage_defer_kb", defer_kb)   # 0: source planes copied at submit also in groups of four or more (default: with the group)
        f.set_pipeline(depth, gpu_pkg.PIN_NONE, group)
        padded = [_padded(s, 64, 7) for s in srcs]
        dsts = []
        for k in range(len(srcs)):
            planes = [np.full((h, w + 48 // np.dtype(gpu_pkg.FORMATS[fmt].dtype).itemsize), 0x5A if fmt != "YUV444PS" else 0.25,
                              gpu_pkg.FORMATS[fmt].dtype) for (w, h) in dims]
            dsts.append(planes)
        tickets = [f.submit([v for v in padded[k][0]], [d[:, :w] for d, (w, h) in zip(dsts[k], dims)]) for k in range(len(srcs))]
        for k in (5, 0, 11, 3):
            f.wait(tickets[k])
            assert_planes_equal([d[:, :w] for d, (w, h) in zip(dsts[k], dims)], want[k], dims, what=f"depth {depth} group {group} frame {k}")
        f.flush()
        f.set_pipeline(1, gpu_pkg.PIN_NONE)      # drains: the frames nobody waited for have arrived as well
        for k in range(len(srcs)):
            assert_planes_equal([d[:, :w] for d, (w, h) in zip(dsts[k], dims)], want[k], dims, what=f"depth {depth} group {group} frame {k} after the drain")
            for d, (w, h) in zip(dsts[k], dims):
                assert (d[:, w:] == (0x5A if fmt != "YUV444PS" else 0.25)).all(), "bytes between the rows were written"
    by_shader, by_dma, ranges = gpu_pkg.transport_counts()
    assert (by_shader, by_dma, ranges) == (0, 0, 0) and gpu_pkg.host_registrations() == 0
    assert gpu_pkg.staged_frames() == 1 + 5 * len(srcs)
    f.close()


@pytest.mark.gpu
def test_threads_1_keeps_plane_copies_on_the_callers_thread_and_the_result_the_same(gpu_pkg, O):
    """The reference's threads argument (1: this thread only) decides whether pageable planes may be copied by the helper threads;
    planes of 0.9 / 3.7 MB here, large enough for them.  Same frames either way, and with the lanes knob at 2 and 7."""
    fmt, sw, sh, tw, th = "Y8", 1280, 720, 2560, 1440
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    srcs = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=7 + k) for k in range(3)]
    want = [of.get_frame(s, threads=8) for s in srcs]
    for kw, lanes, bands, group in ((dict(threads=1), -1, -1, 2), ({}, -1, -1, 2), ({}, 2, -1, 2), ({}, 7, 3, 1), (dict(threads=0), 1, 1, 1),
                                    (dict(threads=1), -1, 2, 1)):
        gpu_pkg.set_knob("copy_threads", lanes)
        gpu_pkg.set_knob("stage_bands", bands)
        try:
            f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
            f.set_pipeline(4, gpu_pkg.PIN_NONE, group)     # group 1: frames travel alone, their results in row bands
            dsts = [[gpu_pkg.alloc_plane(tw, th, np.uint8)] for _ in srcs]
            for t in [f.submit(s, d) for s, d in zip(srcs, dsts)]:
                f.wait(t)
            for k in range(3):
                assert_planes_equal(dsts[k], want[k], f.out_dims(), what=f"{kw} lanes {lanes} bands {bands} group {group} frame {k}")
            assert_planes_equal(f.get_frame(srcs[1]), want[1], f.out_dims(), what=f"{kw} lanes {lanes} bands {bands}: synchronous frame")
            f.close()
        finally:
            gpu_pkg.set_knob("copy_threads", -1)
            gpu_pkg.set_knob("stage_bands", -1)


@pytest.mark.gpu
def test_frames_in_flight_when_the_instance_is_freed_still_arrive(gpu_pkg, O):
    fmt, sw, sh, tw, th, of = _case(O)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_pipeline(4, gpu_pkg.PIN_NONE, 2)
    srcs = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=40 + k) for k in range(4)]
    dsts = [[gpu_pkg.alloc_plane(w, h, np.uint8) for (w, h) in f.out_dims()] for _ in range(4)]
    dims = f.out_dims()
    for s, d in zip(srcs, dsts):
        f.submit(s, d)          # two full groups, launched; nobody waits
    f.close()
    for k in range(4):
        assert_planes_equal(dsts[k], of.get_frame(srcs[k], threads=4), dims, what=f"frame {k}")


@pytest.mark.gpu
def test_runtime_mode_hands_pageable_planes_to_the_runtime(gpu_pkg, O, pooling_host):
    """register_host_buffers = 3, the default of rounds 1 - 5: hipMemcpy2DAsync on the caller's planes as they are."""
    fmt, sw, sh, tw, th, of = _case(O)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_pipeline(8, gpu_pkg.PIN_RUNTIME, 4)
    srcs = [fresh_copies(O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=60 + k)) for k in range(8)]
    dsts = [fresh_planes(f.out_dims(), np.uint8) for _ in range(8)]
    gpu_pkg.transport_counts(reset=True)
    tickets = [f.submit(s, d) for s, d in zip(srcs, dsts)]
    for t in tickets:
        f.wait(t)
    assert gpu_pkg.transport_counts()[:2] == (0, 8) and gpu_pkg.staged_frames() == 0 and gpu_pkg.host_registrations() == 0
    for k in (0, 3, 7):
        assert_planes_equal(dsts[k], of.get_frame(srcs[k], threads=4), f.out_dims(), what=f"frame {k}")
    f.close()


@pytest.mark.gpu
def test_pageable_sources_with_results_into_memory_the_host_pinned(gpu_pkg, O):
    """A host that pins only its OUTPUT pool (adopt_host_range): sources go through the library's buffers, results leave by the
    shader straight into the pinned planes."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th = "Y8", 320, 180, 438, 246
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    n, dpitch = 8, 448
    pool = torch.empty(n * dpitch * th, dtype=torch.uint8).pin_memory()
    host = pool.numpy()
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_pipeline(8, gpu_pkg.PIN_NONE, 4)
    f.adopt_host_range(pool.data_ptr(), pool.numel())
    srcs = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=500 + k) for k in range(n)]
    dsts = [[host[k * dpitch * th:(k + 1) * dpitch * th].reshape(th, dpitch)[:, :tw]] for k in range(n)]
    gpu_pkg.transport_counts(reset=True)
    tickets = [f.submit(s, d) for s, d in zip(srcs, dsts)]
    for t in tickets:
        f.wait(t)
    assert gpu_pkg.transport_counts()[:2] == (n, 0) and gpu_pkg.staged_frames() == 0
    for k in range(n):
        assert_planes_equal(dsts[k], of.get_frame(srcs[k], threads=4), f.out_dims(), what=f"frame {k}")
    f.close()
    del host, pool


def _batch_planes(gpu_pkg, O, fmt, sw, sh, tw, th, n, seed):
    ofmt = O.FORMATS[fmt]
    frames = [O.lcg_frame(ofmt, sw, sh, seed=seed + k) for k in range(n)]
    # source planes one after the other with odd gaps of a few hundred bytes (allocator headers, small objects in between): they
    # share pages with their neighbours and merge into few registrations
    room = sum(p.nbytes + 512 for fr in frames for p in fr) + 4096
    src_pool = fresh_mapping(room)
    srcs, off = [], 37
    for fr in frames:
        planes = []
        for p in fr:
            v = src_pool[off:off + p.nbytes].view(p.dtype).reshape(p.shape)
            v[...] = p
            planes.append(v)
            off += p.nbytes + 16 + (off * 7) % 400 // p.itemsize * p.itemsize
        srcs.append(planes)
    ddims = ofmt.plane_dims(tw, th)
    # destination planes in ONE allocation, frame after frame (what a batch tool does): neighbouring chunks share pages
    pitches = [(w + 63) // 64 * 64 for (w, h) in ddims]
    per_frame = sum(p * h for p, (w, h) in zip(pitches, ddims))
    pool = fresh_mapping(per_frame * n + 4096)[64:]
    dsts, off = [], 0
    for k in range(n):
        planes = []
        for p, (w, h) in zip(pitches, ddims):
            planes.append(pool[off:off + p * h].reshape(h, p))
            off += p * h
        dsts.append(planes)
    return srcs, dsts, (pool, src_pool)


@pytest.mark.gpu
def test_batch_registrars_pin_whole_planes_side_by_side(gpu_pkg, O, pooling_host):
    """jinc_batch_process with four registrar threads on this one device (test header; one per device on a node): 100 frames
    whose destination planes lie back to back in one allocation, the source planes in 300 small ones.  Every plane must lie
    inside ONE registered range (the runtime refuses a copy that starts in one registered object and runs past its end), no
    registration may be refused, every frame must leave by the shader; the registrations stay until the batch is freed and a
    second call finds them."""
    fmt, sw, sh, tw, th = "YUV420P8", 320, 180, 640, 360
    n = 100
    srcs, dsts, pool = _batch_planes(gpu_pkg, O, fmt, sw, sh, tw, th, n, 4000)
    b = gpu_pkg.Batch(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, ndevices=1, streams=32, register_host_buffers=gpu_pkg.PIN_POOL)
    b.set_registrars(4)
    gpu_pkg.transport_counts(reset=True)
    try:
        b.process(srcs, dsts)
    except gpu_pkg.JincError as e:
        raise AssertionError(f"{e}; refused registrations: {b.refused()}")
    assert b.refused()[0] == 0, b.refused()
    by_shader, by_dma, _ = gpu_pkg.transport_counts()
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    for k in (0, 15, 16, 17, 47, 48, 99):
        assert_planes_equal(dsts[k], of.get_frame(srcs[k], threads=4), O.FORMATS[fmt].plane_dims(tw, th), what=f"frame {k}")
    assert (by_shader, by_dma) == (n, 0)
    live = gpu_pkg.host_registrations()
    assert live > 0
    probe = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    probe.adopt_host_range(dsts[40][0].ctypes.data, dsts[40][0].nbytes)       # still pinned: another instance may adopt it
    probe.release_host_range(dsts[40][0].ctypes.data, dsts[40][0].nbytes)
    with pytest.raises(gpu_pkg.JincError):
        probe.adopt_host_range(np.zeros(1 << 16, np.uint8).ctypes.data, 1 << 16)   # pageable memory is not
    probe.close()
    b.process(srcs, dsts)                                                      # a second call registers nothing new
    assert gpu_pkg.host_registrations() == live and b.refused()[0] == 0
    assert_planes_equal(dsts[63], of.get_frame(srcs[63], threads=4), O.FORMATS[fmt].plane_dims(tw, th), what="second call")
    b.close()
    assert gpu_pkg.host_registrations() == 0
    del pool


@pytest.mark.gpu
def test_batch_reports_the_cpus_of_its_device_and_runs_with_and_without_binding(gpu_pkg, O, pooling_host):
    fmt, sw, sh, tw, th = "Y8", 192, 108, 384, 216
    srcs, dsts, pool = _batch_planes(gpu_pkg, O, fmt, sw, sh, tw, th, 24, 10)
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    b = gpu_pkg.Batch(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, ndevices=1, streams=8, register_host_buffers=gpu_pkg.PIN_POOL)
    cpus = b.device_cpus(0)
    assert cpus == sorted(set(cpus))   # may be empty (no NUMA information): then nothing is bound
    mine = os.sched_getaffinity(0)
    for on in (True, False):
        b.set_affinity(on)
        b.process(srcs, dsts)
        assert_planes_equal(dsts[23], of.get_frame(srcs[23]), [(tw, th)], what=f"affinity {on}")
        assert os.sched_getaffinity(0) == mine   # the caller's thread is never re-bound
    b.close()


def test_numa_lookup_reads_the_node_of_the_pci_function_and_its_cpulist(pkg, tmp_path):
    """batch.cpp's lookup against a fake sysfs tree: /sys/bus/pci/devices/<bdf>/numa_node -> /sys/devices/system/node/nodeN/cpulist."""
    root = tmp_path / "sys"
    for bdf, node in (("0000:05:00.0", "0\n"), ("0000:c5:00.0", "1\n"), ("0000:e5:00.0", "-1\n")):
        d = root / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(node)
    for node, cpus in ((0, "0-3,64-67\n"), (1, "8,10-11\n")):
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus)
    assert pkg.numa_cpus(str(root), "0000:05:00.0") == [0, 1, 2, 3, 64, 65, 66, 67]
    assert pkg.numa_cpus(str(root), "0000:C5:00.0") == [8, 10, 11]      # hipDeviceGetPCIBusId prints upper-case hex
    assert pkg.numa_cpus(str(root), "0000:e5:00.0") == []               # numa_node = -1: a machine without NUMA
    assert pkg.numa_cpus(str(root), "0000:aa:00.0") == []               # unknown device
    assert pkg.numa_cpus(str(tmp_path / "nothing"), "0000:05:00.0") == []


def test_plane_copies_on_the_helper_threads_move_every_row_and_nothing_else(pkg):
    """host_copy.cpp without a device: planes large enough for the helper threads, pitches that are not the row size, several
    callers at once (one pool for the process), the one-thread path of threads = 1."""
    import threading
    rng = np.random.default_rng(11)

    def one(rows, row_bytes, spitch, dpitch, helpers, seed):
        src = rng.integers(0, 256, (rows, spitch), dtype=np.uint8)
        dst = np.full((rows, dpitch), seed & 0xFF, np.uint8)
        pkg.copy_rows(dst, src, row_bytes, rows, helpers)
        assert np.array_equal(dst[:, :row_bytes], src[:, :row_bytes]), (rows, row_bytes, helpers)
        assert (dst[:, row_bytes:] == (seed & 0xFF)).all(), "bytes between the rows were written"

    for rows, row_bytes, spitch, dpitch in ((2160, 3840, 3840, 3840), (1080, 1920, 1984, 2048), (7, 1 << 20, 1 << 20, (1 << 20) + 64),
                                            (2161, 1001, 1024, 1003), (1, 4 << 20, 4 << 20, 4 << 20), (300, 100, 128, 100)):
        for helpers in (True, False):
            one(rows, row_bytes, spitch, dpitch, helpers, rows)
    errors = []

    def many(seed):
        try:
            for k in range(6):
                one(1080 + seed, 3840, 3840 + 64 * (k % 2), 3904, True, seed + k)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=many, args=(s,)) for s in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:1]
    with pytest.raises(pkg.JincError):
        pkg.copy_rows(np.zeros((4, 8), np.uint8), np.zeros((4, 16), np.uint8), 12, 4)   # rows longer than the destination's pitch


def test_the_helper_pool_is_sized_by_the_cpus_the_process_may_use(pkg):
    """host_copy.cpp counts the affinity mask cut down to the cgroup's CFS quota (a container with 256 CPUs in its mask and a
    quota of 16 cores is throttled as a whole beyond 16 busy threads); bench.py's cpu_quota() reads the same files independently."""
    import bench
    mask = len(os.sched_getaffinity(0))
    quota = bench.cpu_quota()["quota_cores"]
    want = mask if not quota else max(1, min(mask, int(quota + 0.5)))
    assert pkg.usable_cpus() == want, (pkg.usable_cpus(), mask, quota)


@pytest.mark.gpu
def test_the_helper_pool_is_sized_by_the_cpus_the_process_may_use_on_the_gpu_box(gpu_pkg):
    """The same check where the cgroup has a quota (the pool's boxes: 256 CPUs in the mask, 16 cores of quota)."""
    test_the_helper_pool_is_sized_by_the_cpus_the_process_may_use(gpu_pkg)


@pytest.mark.gpu
def test_four_client_threads_share_the_helper_pool(gpu_pkg, O):
    """Prefetch(4) at look-ahead 1: four instances on four threads, synchronous frames with planes large enough for the helper
    threads -- one pool for the process, every caller takes part in its own copies."""
    import threading
    fmt, sw, sh, tw, th = "Y8", 1280, 720, 2560, 1440
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    srcs = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=300 + k) for k in range(4)]
    want = [of.get_frame(s, threads=8) for s in srcs]
    errors = []

    def client(k):
        try:
            f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
            for rnd in range(12):
                j = (k + rnd) % 4
                assert_planes_equal(f.get_frame(srcs[j]), want[j], f.out_dims(), what=f"thread {k} round {rnd}")
            f.close()
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=client, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:1]
