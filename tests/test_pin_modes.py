"""How the library treats the caller's host buffers (jinc_filter_set_pipeline's / jinc_batch_create's register_host_buffers).

Round 6 (Next 6 / 7) built three modes and kept two: buffers go to the HIP runtime as
they are (0, the default), or they are registered once and the registrations cached by address (non-zero: the promise of a host
whose frame memory stays mapped -- the `pooling_host` fixture marks those tests).  The third, "registered while the frame is in
flight", passed its own tests -- planes unmapped and mapped again at the same addresses between frames included -- but the
registration churn it brings ended full test runs in GPU memory access faults inside the RUNTIME's copies from pageable memory
(4 of 9 runs, no registration of the library alive; profiles/round6/README.md), so it was withdrawn and its tests with it.

What is tested here: cached registrations are kept and given back; jinc_batch_process pins with one registrar per device
(several on this one-device box through the test header) in exact byte ranges that cover every plane whole; the NUMA lookup of
batch.cpp against a fake sysfs tree (CPU).  Every test of the suite ends with the registry empty and as many hipHostUnregister as
hipHostRegister calls (tests/conftest.py)."""
import os

import numpy as np
import pytest

from conftest import assert_planes_equal


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
    srcs = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=70 + k) for k in range(8)]
    dsts = [[gpu_pkg.alloc_plane(w, h, np.uint8) for (w, h) in f.out_dims()] for _ in range(8)]
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


def _batch_planes(gpu_pkg, O, fmt, sw, sh, tw, th, n, seed):
    ofmt = O.FORMATS[fmt]
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=seed + k) for k in range(n)]
    ddims = ofmt.plane_dims(tw, th)
    # destination planes in ONE allocation, frame after frame (what a batch tool does): neighbouring chunks share pages
    pitches = [(w + 63) // 64 * 64 for (w, h) in ddims]
    per_frame = sum(p * h for p, (w, h) in zip(pitches, ddims))
    pool = np.zeros(per_frame * n + 64, np.uint8)
    dsts, off = [], 0
    for k in range(n):
        planes = []
        for p, (w, h) in zip(pitches, ddims):
            planes.append(pool[off:off + p * h].reshape(h, p))
            off += p * h
        dsts.append(planes)
    return srcs, dsts, pool


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
