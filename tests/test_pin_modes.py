"""How the library treats the caller's host buffers (jinc_filter_set_pipeline's register_host_buffers; VERDICT r5 weak 5 /
Next 7):

  mode 1 (PIN_IN_FLIGHT)  a plane is registered with hipHostRegister when its frame is submitted and given back when the
                          frame's wait returns: no registration outlives a buffer the host is free to release -- tested here
                          under the allocator AS IT IS (no mallopt), with buffers that are unmapped and mapped again at the
                          same addresses between frames: correct pixels, no fault, the registry empty afterwards;
  mode 2 (PIN_POOL)       registrations cached by address, for a host whose frame memory stays mapped -- tested under the
                          `pooling_host` fixture, which models exactly that host.

Also: jinc_batch_process pins per call (mode 1) or until jinc_batch_free (mode 2), with one registrar per device (several on
this one-device box through the test header), and the NUMA lookup of batch.cpp against a fake sysfs tree (CPU)."""
import mmap
import os

import numpy as np
import pytest

from conftest import assert_planes_equal


class MappedPlane:
    """A plane in an anonymous mapping of its own: close() really unmaps it (numpy arrays come and go through malloc, which
    may or may not give pages back; here the pages are gone for certain)."""

    def __init__(self, w, h, dtype=np.uint8, pad_pages=0):
        isz = np.dtype(dtype).itemsize
        self.pitch = (w * isz + 63) // 64 * 64
        self.nbytes = self.pitch * h
        self.map = mmap.mmap(-1, (self.nbytes + 4095) // 4096 * 4096 + 4096 * pad_pages)
        self.array = np.frombuffer(self.map, dtype=dtype, count=self.nbytes // isz).reshape(h, self.pitch // isz)
        self.address = self.array.ctypes.data

    def close(self):
        self.array = None
        self.map.close()


def _case(O):
    fmt, sw, sh, tw, th = "YUV420P8", 320, 180, 640, 360
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    return fmt, sw, sh, tw, th, of


@pytest.mark.gpu
@pytest.mark.parametrize("depth,group", [(1, 0), (4, 2), (16, 8)], ids=["depth1", "depth4", "depth16"])
def test_buffers_that_are_unmapped_and_mapped_again_between_frames_give_correct_pixels(gpu_pkg, O, depth, group):
    """Mode 1 under the default allocator.  Every round maps fresh source and destination planes (the kernel hands the same
    addresses out again: counted), runs `depth` frames through submit / wait and unmaps everything.  A registration cached by
    address would now point at pages that are gone -- round 5's fault; here nothing is cached: the registry is back at its
    starting size after every round's waits, and every frame is bit-exact."""
    fmt, sw, sh, tw, th, of = _case(O)
    F = gpu_pkg.FORMATS[fmt]
    f = gpu_pkg.Filter(F, sw, sh, tw, th, device=0)
    f.set_pipeline(depth, gpu_pkg.PIN_IN_FLIGHT, group)
    base_ranges = gpu_pkg.transport_counts(reset=True)[2]
    seen, came_back = set(), 0
    sdims, ddims = O.FORMATS[fmt].plane_dims(sw, sh), f.out_dims()
    for rnd in range(6):
        frames = []
        for k in range(depth):
            src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=900 + 16 * rnd + k)
            sp = [MappedPlane(w, h, pad_pages=rnd % 2) for (w, h) in sdims]   # sizes alternate: a range comes back LONGER or shorter
            dp = [MappedPlane(w, h, pad_pages=rnd % 2) for (w, h) in ddims]
            for m, s, (w, h) in zip(sp, src, sdims):
                m.array[:h, :w] = s[:h, :w]
            for m in sp + dp:
                came_back += m.address in seen
                seen.add(m.address)
            frames.append((src, sp, dp))
        tickets = [f.submit([m.array for m in sp], [m.array for m in dp]) for (_, sp, dp) in frames]
        for t, (src, sp, dp) in zip(tickets, frames):
            f.wait(t)
            assert_planes_equal([m.array for m in dp], of.get_frame(src, threads=4), ddims, what=f"round {rnd}")
        assert gpu_pkg.transport_counts()[2] == base_ranges, "a registration outlived its frame's wait"
        for (_, sp, dp) in frames:
            for m in sp + dp:
                m.close()   # the pages go back to the kernel NOW
    by_shader, by_dma, _ = gpu_pkg.transport_counts()
    assert by_dma == 0 and by_shader == 6 * depth, (by_shader, by_dma)   # pinned in flight: every frame left by the shader
    f.close()
    assert came_back > 0, "the kernel never handed an address out again: the test did not reach its case"


@pytest.mark.gpu
def test_one_buffer_under_many_frames_in_flight_is_held_until_the_last_of_them(gpu_pkg, O):
    """Mode 1: the same source planes submitted 12 times (a host that repeats a frame); the registration is shared by the
    frames in flight and leaves with the last one, whatever the order of the waits."""
    fmt, sw, sh, tw, th, of = _case(O)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_pipeline(12, gpu_pkg.PIN_IN_FLIGHT, 4)
    base_ranges = gpu_pkg.transport_counts(reset=True)[2]
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=31)
    want = of.get_frame(src, threads=4)
    dsts = [[gpu_pkg.alloc_plane(w, h, np.uint8) for (w, h) in f.out_dims()] for _ in range(12)]
    tickets = [f.submit(src, d) for d in dsts]
    for k in (5, 0, 11, 3, 7, 1, 2, 4, 6, 8, 9):
        f.wait(tickets[k])
        assert_planes_equal(dsts[k], want, f.out_dims(), what=f"frame {k}")
        assert gpu_pkg.transport_counts()[2] > base_ranges   # frame 10 is still in flight and holds the source planes
    f.wait(tickets[10])
    assert_planes_equal(dsts[10], want, f.out_dims(), what="frame 10")
    assert gpu_pkg.transport_counts()[2] == base_ranges
    f.close()


@pytest.mark.gpu
def test_pool_mode_keeps_registrations_and_leaving_it_gives_them_back(gpu_pkg, O, pooling_host):
    """Mode 2: what a buffer cost to register is paid once; set_pipeline(…, 1 or 0) and close() give everything back."""
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
    f.set_pipeline(8, gpu_pkg.PIN_IN_FLIGHT, 4)
    assert gpu_pkg.transport_counts()[2] == base_ranges
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
@pytest.mark.parametrize("mode", [1, 2], ids=["per_call", "until_free"])
def test_batch_registrars_pin_disjoint_ranges_side_by_side(gpu_pkg, O, mode, request):
    """jinc_batch_process with four registrar threads on this one device (test header; one per device on a node): 100 frames
    whose destination planes lie back to back in one allocation, so neighbouring 16-frame chunks share pages -- every page
    must end up in exactly one registration and every frame must leave by the shader.  Mode 1: the registrations end with
    the call (adopting a plane afterwards fails: it is not pinned); mode 2: they stay until the batch is freed."""
    if mode == 2:
        request.getfixturevalue("pooling_host")
    fmt, sw, sh, tw, th = "YUV420P8", 320, 180, 640, 360
    n = 100
    srcs, dsts, pool = _batch_planes(gpu_pkg, O, fmt, sw, sh, tw, th, n, 4000)
    b = gpu_pkg.Batch(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, ndevices=1, streams=32, register_host_buffers=mode)
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
    probe = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    if mode == 1:
        with pytest.raises(gpu_pkg.JincError):
            probe.adopt_host_range(dsts[40][0].ctypes.data, dsts[40][0].nbytes)   # not pinned any more
        b.process(srcs, dsts)   # and a second call pins again
        assert_planes_equal(dsts[63], of.get_frame(srcs[63], threads=4), O.FORMATS[fmt].plane_dims(tw, th), what="second call")
    else:
        probe.adopt_host_range(dsts[40][0].ctypes.data, dsts[40][0].nbytes)       # still pinned
        probe.release_host_range(dsts[40][0].ctypes.data, dsts[40][0].nbytes)
    probe.close()
    b.close()
    del pool


@pytest.mark.gpu
def test_batch_reports_the_cpus_of_its_device_and_runs_with_and_without_binding(gpu_pkg, O):
    fmt, sw, sh, tw, th = "Y8", 192, 108, 384, 216
    srcs, dsts, pool = _batch_planes(gpu_pkg, O, fmt, sw, sh, tw, th, 24, 10)
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    b = gpu_pkg.Batch(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, ndevices=1, streams=8, register_host_buffers=1)
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
