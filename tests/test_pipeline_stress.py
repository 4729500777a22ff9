"""Look-ahead pipeline under load: hundreds of distinct multi-plane frames per pipeline shape, waits in random order, shapes changed
on the live instance -- every frame must be exactly what the synchronous single-frame call of a second instance returns.  The
frames are small, so launches, copies and events dominate: what this is after is ordering (side stream, plane pairs, group reuse),
not arithmetic."""
import numpy as np
import pytest

from conftest import fresh_copies, fresh_planes

pytestmark = pytest.mark.gpu

CASES = [
    ("YUV420P8", 200, 120, 274, 164, {}),               # no phase structure: gather kernel / frame-lane kernels by group size
    ("YUV420P8", 160, 96, 320, 192, {}),                # 2x: window kernel
    ("YUV420P16", 192, 108, 288, 162, dict(tap=4)),     # 1.5x with tap 4: runs form
    ("RGBP8", 128, 80, 192, 120, {}),                   # three planes of one table
]
SHAPES = [(1, 0, False), (3, 0, True), (6, 2, False), (16, 0, True), (24, 12, False), (64, 0, True)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_many_frames_through_changing_pipeline_shapes(gpu_pkg, O, case, pooling_host):
    fmt, sw, sh, tw, th, kw = case
    ofmt = O.FORMATS[fmt]
    n = 160
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=40000 + k) for k in range(n)]
    ref_filter = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    want = [ref_filter.get_frame(s) for s in srcs]
    ref_filter.close()
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    dims = f.out_dims()
    np_dtype = srcs[0][0].dtype
    for depth, group, register in SHAPES:
        f.set_pipeline(depth, register, group)
        rng = np.random.default_rng(depth * 1000 + group)
        # Registered shapes: the frames go through a POOL of depth + 2 buffer sets (source planes copied in, as a host's frame pool
        # would hand them out), so that a shape registers a few dozen planes once instead of 960 planes through a cache of 64 --
        # this test is after ordering, and every registration is a fresh mapping of heap pages into the device (tests/conftest.py).
        nb = depth + 2 if register else n
        if register:   # (planes the device maps: mappings of their own, conftest.fresh_mapping)
            dsts = [fresh_planes(dims, np_dtype) for _ in range(nb)]
            pool_srcs = [fresh_copies(srcs[0]) for _ in range(nb)]
        else:
            dsts = [[gpu_pkg.alloc_plane(w, h, np_dtype) for (w, h) in dims] for _ in range(nb)]
            pool_srcs = None
        tickets, waiting, bad = {}, [], []

        def collect(j):
            f.wait(tickets[j])
            got = dsts[j % nb]
            for i, (w, h) in enumerate(dims):
                if not np.array_equal(got[i][:h, :w], want[j][i][:h, :w]):
                    rows = np.nonzero(np.any(got[i][:h, :w] != want[j][i][:h, :w], axis=1))[0]
                    bad.append((j, i, int(rows[0]), int(rows[-1]), len(rows)))

        for k in range(n):
            if k - nb in waiting:      # the buffer set this frame takes is still out: its frame is collected first
                waiting.remove(k - nb)
                collect(k - nb)
            src = srcs[k]
            if register:
                src = pool_srcs[k % nb]
                for a, b in zip(src, srcs[k]):
                    a[...] = b
            tickets[k] = f.submit(src, dsts[k % nb])
            waiting.append(k)
            if k % 37 == 36:
                f.flush()
            while len(waiting) >= depth or (waiting and rng.random() < 0.15):
                collect(waiting.pop(int(rng.integers(0, len(waiting)))))
        for j in waiting:
            collect(j)
        assert not bad, f"depth {depth} group {group} registered {register}: (frame, plane, first row, last row, rows) {bad[:8]} of {len(bad)}"
    f.close()
