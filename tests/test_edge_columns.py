"""Border columns inside the interior kernel (round 5): on integer planes at 2x with tap 3 / tap 4 the first and the last tile column
of ewa_periodic_quad2_kernel / ewa_periodic_quad2x8_kernel compute the plane's border columns of their rows from the tile they have staged (PeriodicArgs::EdgeColumns,
device_plan.cpp plan_edge_columns).  Forced through jinc_filter_set_border_strips(4) with kernel mode QUAD on small planes, compared
with the oracle and with the border kernels' bytes; `last_border` bit 64 says the form ran.  The automatic rule's case (C2 batches)
is in test_strip_kernel.py / test_benchmarked_instances.py."""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs, to_device, to_host

pytestmark = pytest.mark.gpu

CASES = [
    ("Y8", 192, 108, 384, 216, dict(tap=3)),                      # one tile column: both sides in the same workgroups
    ("Y8", 500, 70, 1000, 140, dict(tap=3)),                      # four tile columns, the last one partial
    ("Y8", 263, 301, 526, 602, dict(tap=3)),                      # three tile columns, seven tile rows (the last partial), odd sizes
    ("Y8", 135, 50, 270, 100, dict(tap=3)),                       # the last tile column holds four periods
    ("Y16", 333, 211, 666, 422, dict(tap=3)),
    ("Y10", 150, 100, 300, 200, dict(tap=3)),
    ("Y8", 300, 120, 600, 240, dict(tap=3, blur=0.95)),
    ("YUV420P8", 256, 144, 512, 288, dict(tap=3)),                # chroma sited as MPEG-2: the 6-row x 7-column support
    ("YUV420P16", 400, 144, 800, 288, dict(tap=3)),
    ("YUV444P8", 256, 144, 512, 288, dict(tap=3)),
    ("YUV420P8", 256, 144, 512, 288, dict(tap=3, cplace="topleft")),
    ("Y8", 192, 108, 384, 216, dict(tap=4)),                      # filter size 9 on the 8 x 8 support: ewa_periodic_quad2x8_kernel
    ("Y8", 500, 90, 1000, 180, dict(tap=4)),
    ("Y8", 263, 301, 526, 602, dict(tap=4, blur=0.98)),
    ("Y16", 333, 211, 666, 422, dict(tap=4)),
    ("YUV420P8", 400, 144, 800, 288, dict(tap=4)),
    ("YUV420P10", 256, 144, 512, 288, dict(tap=4, cplace="topleft")),
    ("Y8", 200, 120, 400, 240, dict(tap=3, src_left=2.5, src_top=-1.25, src_width=190.5, src_height=118.0)),  # cropped: uneven borders (may not configure)
]


def _id(c):
    return f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}_" + "_".join(f"{k}{v}" for k, v in c[5].items())


@pytest.mark.parametrize("frames", [1, 3])
@pytest.mark.parametrize("case", CASES, ids=_id)
def test_edge_columns_match_the_oracle_and_the_border_kernels(gpu_pkg, O, case, frames):
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th, kw = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=9100 + k) for k in range(frames)]
    want = [of.get_frame(s, threads=8) for s in srcs]

    def run(strips):
        f.set_border_strips(strips)
        f.set_kernel_mode(gpu_pkg.KernelMode.QUAD)
        with gpu_pkg.knobs(quad2x8=1):   # (tap 4: two periods per lane whatever the call's size)
            return [f.get_frame(srcs[0])] if frames == 1 else _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, gpu_pkg.KernelMode.QUAD)

    got = run(4)
    fused = [t for t in range(f.num_tables) if f.last_border(t) & 64]
    if not fused:
        f.close()
        pytest.skip("no table of this plan puts its border columns into the interior kernel")
    for t in fused:
        assert f.last_instance(t).startswith(("ewa_periodic_quad2_kernel<", "ewa_periodic_quad2x8_kernel<")), f.last_instance(t)
        assert f.last_border(t) & (256 | 32 | 8 | 4 | 1) == 0, f.last_border(t)   # no column kernel beside it
    with gpu_pkg.knobs(edge_cols=0):
        plain = run(4)
        assert all(f.last_border(t) & 64 == 0 for t in range(f.num_tables))
    gathered = run(0)
    for k in range(frames):
        assert_planes_equal(got[k], want[k], f.out_dims(), what=f"{_id(case)} frame {k}: edge columns vs oracle")
        assert_planes_equal(got[k], plain[k], f.out_dims(), what=f"{_id(case)} frame {k}: edge columns vs the column kernel's")
        assert_planes_equal(got[k], gathered[k], f.out_dims(), what=f"{_id(case)} frame {k}: edge columns vs the gather kernel's border")
    f.close()


def test_edge_columns_write_nothing_outside_their_plane(gpu_pkg, O):
    """Destination pitch wider than the plane, guard bytes between rows and around the frames: untouched."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, n = "Y8", 263, 150, 526, 300, 3
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th)
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0)
    f.set_border_strips(4)
    f.set_kernel_mode(gpu_pkg.KernelMode.QUAD)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=60 + k) for k in range(n)]
    src_t = to_device(torch.stack([torch.from_numpy(s[0]) for s in srcs]))
    pitch, rows = 640, th + 5
    dst_t = torch.full((n, rows, pitch), 0xA5, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    f.process_device([src_t.data_ptr()], [src_t.stride(1)], [src_t.stride(0)], [dst_t[:, 2:, 7:].data_ptr()], [pitch], [rows * pitch], n,
                     stream=stream.cuda_stream)
    stream.synchronize()
    assert f.last_border(0) & 64, f.last_border(0)
    out = to_host(dst_t).numpy()
    f.close()
    for k in range(n):
        want = of.get_frame(srcs[k], threads=8)[0][:th, :tw]
        bad = np.argwhere(out[k, 2:2 + th, 7:7 + tw] != want)
        assert bad.size == 0, f"frame {k}: {len(bad)} samples differ, rows {bad[:, 0].min()}..{bad[:, 0].max()}, columns {sorted(set(bad[:, 1].tolist()))[:24]}"
        guard = out[k].copy()
        guard[2:2 + th, 7:7 + tw] = 0xA5
        assert (guard == 0xA5).all(), f"frame {k}: bytes outside the plane were written"


def test_the_knob_that_reroutes_the_interior_launch_takes_the_columns_back(gpu_pkg, O):
    """ROWPAIR_SMALL = 1 sends a tap-3 interior to the row-pair kernel, which has no edge tiles: the border columns must then come
    from the column kernel again (found by bench.py's self-check under profiles/recheck_rules.py: no line, exit code 1)."""
    fmt, sw, sh, tw, th = "Y8", 263, 150, 526, 300
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=5)
    want = of.get_frame(src, threads=8)
    f.set_border_strips(4)
    f.set_kernel_mode(gpu_pkg.KernelMode.QUAD)
    assert_planes_equal(f.get_frame(src), want, f.out_dims(), what="quad form with edge columns")
    assert f.last_border(0) & 64
    for mode in (gpu_pkg.KernelMode.QUAD, 0):
        f.set_kernel_mode(mode)
        with gpu_pkg.knobs(rowpair_small=1):
            assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=f"kernel mode {int(mode)}, interior rerouted to the row-pair kernel")
            assert f.last_border(0) & 64 == 0, (int(mode), f.last_border(0), f.last_instance(0))
    f.close()
