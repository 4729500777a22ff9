"""ewa_periodic_rowpair_kernel (round 5): the row-streamed periodic kernel in packed phase-pair form -- 2x up-scales with 12 .. 17 taps per
kernel row (taps 6, 7, 8; trimmed supports 12 / 14 / 16 on integer planes, 17 columns x 16 rows for chroma sited as MPEG-2, the full
13 / 15 / 17 windows on float planes and under kernel mode 15).  Every chain is the reference's (ref JincResize.cpp:570-579): bit-exact
against the oracle and against ewa_periodic_rows_kernel (kernel mode 3), in each of its three tile shapes, on planes whose last
lanes hold 1 / 2 / 3 periods and whose last chunk holds one period-row, single frames and batches, with non-finite float samples."""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs

pytestmark = pytest.mark.gpu

CASES = [
    # (format, src, dst, args, taps per kernel row the interior runs on [table 0, table 1])
    ("Y8", 150, 100, 300, 200, dict(tap=6), (12,)),
    ("Y16", 151, 101, 302, 202, dict(tap=7), (14,)),
    ("Y16", 161, 97, 322, 194, dict(tap=8), (16,)),
    ("Y10", 163, 99, 326, 198, dict(tap=8), (16,)),
    ("Y32", 150, 100, 300, 200, dict(tap=6), (13,)),          # float planes below the trim threshold: the full window, every tap
    ("Y32", 154, 90, 308, 180, dict(tap=7), (15,)),
    ("Y32", 160, 95, 320, 190, dict(tap=8), (17,)),
    ("YUV420P16", 322, 194, 644, 388, dict(tap=8), (16, 17)),  # C3 in small: chroma 17 columns x 16 rows
    ("YUV420P8", 300, 200, 600, 400, dict(tap=6, cplace="mpeg1"), (12, 12)),
    ("YUV422P10", 302, 100, 604, 200, dict(tap=7), (14, None)),
    ("RGBP8", 149, 111, 298, 222, dict(tap=8, blur=0.97), (16,)),
    ("Y8", 700, 40, 1400, 80, dict(tap=8), (16,)),            # several tiles of every shape in x
    ("Y16", 70, 300, 140, 600, dict(tap=6), (12,)),           # ... and in y
]


def _id(c):
    return f"{c[0]}_{c[1]}x{c[2]}_tap{c[5]['tap']}"


@pytest.mark.parametrize("lw", [0, 64, 32, 16], ids=["auto", "256x16", "128x32", "64x64"])
@pytest.mark.parametrize("case", CASES, ids=_id)
def test_rowpair_kernel_matches_oracle_and_rows_kernel(gpu_pkg, O, case, lw):
    fmt, sw, sh, tw, th, kw, taps = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(ofmt, sw, sh, seed=515)
    if ofmt.bits == 32:
        rng = np.random.default_rng(9)
        for p in src:
            p[:] = (rng.standard_normal(p.shape) * 0.8).astype(np.float32)
    want = of.get_frame(src, threads=8)
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    with gpu_pkg.knobs(**({"rows_pair": lw} if lw else {})):
        got = f.get_frame(src)
        for t in range(f.num_tables):
            if taps[t] is None:
                continue
            inst = f.last_instance(t)
            assert inst.startswith("ewa_periodic_rowpair_kernel<"), inst
            n, shape, rows = [int(x) for x in inst.rstrip(">").split(",")[1:]]   # taps per kernel row, lanes along x, period-rows per lane
            assert n == taps[t] and (lw == 0 or shape == lw) and rows == 2, inst
    assert_planes_equal(got, want, f.out_dims(), what=_id(case))
    f.set_kernel_mode(gpu_pkg.KernelMode.ROWS)       # the un-packed kernel on the same support
    rows = f.get_frame(src)
    assert f.last_kernel(0) == "ewa_periodic_rows_kernel"
    assert_planes_equal(got, rows, f.out_dims(), what=_id(case) + " vs rows kernel")
    f.set_kernel_mode(gpu_pkg.KernelMode.FULL_WINDOW)  # the reference's full window: the pair form with no tap left out
    full = f.get_frame(src)
    assert f.last_instance(0).startswith("ewa_periodic_rowpair_kernel<")
    assert_planes_equal(full, want, f.out_dims(), what=_id(case) + " full window")
    with gpu_pkg.knobs(rows_pair=0):                   # knob: the rows kernel under the automatic choice
        f.set_kernel_mode(0)
        f.get_frame(src)
        assert f.last_kernel(0) == "ewa_periodic_rows_kernel"
    f.close()


@pytest.mark.parametrize("fmt,tap", [("Y8", 8), ("Y16", 6), ("Y32", 8), ("YUV420P16", 8)])
@pytest.mark.parametrize("frames", [3, 17])
def test_rowpair_kernel_in_batches(gpu_pkg, O, fmt, tap, frames):
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    sw, sh, tw, th = 266, 74, 532, 148
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, tap=tap)
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, tap=tap)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=2200 + k) for k in range(frames)]
    got = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
    assert f.last_kernel(0) == "ewa_periodic_rowpair_kernel", f.last_kernel(0)
    for k in range(frames):
        assert_planes_equal(got[k], of.get_frame(srcs[k], threads=8), f.out_dims(), what=f"{fmt} tap {tap} frame {k}")
    f.close()


@pytest.mark.parametrize("tap", [6, 8])
@pytest.mark.parametrize("lw", [64, 16])
def test_rowpair_kernel_with_non_finite_float_samples(gpu_pkg, O, tap, lw):
    """Float planes on the trimmed support, frame by frame (forced kernel modes take the flag-and-redo path at any call size): the
    trimmed launch flags the frames in which it stages an infinity or a NaN, the full-window launch behind it computes them again
    with every tap in place -- both on the pair form.  NaN footprints and bits as the oracle's."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th, frames = "Y32", 150, 70, 300, 140, 9
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, tap=tap)
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, tap=tap)
    rng = np.random.default_rng(5)
    srcs = []
    for k in range(frames):
        src = [(rng.standard_normal((sh, sw)) * 0.7).astype(np.float32)]
        if k % 3 == 1:
            src[0][(k * 7) % sh, (k * 31) % sw] = (np.inf, -np.inf, np.nan)[k % 3]
            src[0][sh - 1, 0] = np.nan
        srcs.append(src)
    with gpu_pkg.knobs(rows_pair=lw):
        got = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, gpu_pkg.KernelMode.PERIODIC)
        assert f.last_kernel(0) == "ewa_periodic_rowpair_kernel", f.last_kernel(0)
    w, h = f.out_dims()[0]
    for k in range(frames):
        a, b = got[k][0][:h, :w], of.get_frame(srcs[k], threads=8)[0][:h, :w]
        na, nb = np.isnan(a), np.isnan(b)
        assert np.array_equal(na, nb), f"frame {k}: NaN footprint differs ({int(na.sum())} vs {int(nb.sum())})"
        assert np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32)), f"frame {k}: bits differ"
    f.close()


SHORT_ROW_CASES = [
    ("Y8", 301, 99, 602, 198, dict(tap=3), (6,)),                       # C2's geometry in small: 6 x 6 with chord rows
    ("Y16", 150, 101, 300, 202, dict(tap=3), (6,)),
    ("Y32", 150, 100, 300, 200, dict(tap=3), (7,)),                      # float below the trim threshold: the full 7 x 7 window
    ("Y8", 150, 100, 300, 200, dict(tap=4), (8,)),
    ("Y12", 301, 99, 602, 198, dict(tap=4, blur=0.98), (8,)),
    ("RGBPS", 160, 100, 320, 200, dict(tap=4, blur=0.98), (9,)),         # C4's arguments in small: full 9 x 9 window
    ("Y8", 150, 100, 300, 200, dict(tap=5), (10,)),
    ("Y32", 150, 100, 300, 200, dict(tap=5), (11,)),
    ("YUV420P8", 302, 200, 604, 400, dict(tap=3), (6, 7)),               # chroma sited as MPEG-2: 6 rows x 7 columns
    ("YUV420P16", 300, 200, 600, 400, dict(tap=4, cplace="topleft"), (8, None)),
    ("YUV444P8", 150, 100, 300, 200, dict(tap=3), (6,)),
]


@pytest.mark.parametrize("lw", [0, 64, 16], ids=["auto", "256", "64"])
@pytest.mark.parametrize("case", SHORT_ROW_CASES, ids=_id)
def test_rowpair_kernel_on_short_kernel_rows(gpu_pkg, O, case, lw):
    """Taps 3 .. 5 at 2x (6 .. 11 taps per kernel row; one assembly statement per chain row up to 9 taps) through the knob that puts
    them on the pair form wherever the plan carries the coefficient pairs; single frames and a 5-frame batch."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th, kw, taps = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=640 + k) for k in range(5)]
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    with gpu_pkg.knobs(rowpair_small=1, **({"rows_pair": lw} if lw else {})):
        got = f.get_frame(srcs[0])
        for t in range(f.num_tables):
            if taps[t] is not None:
                inst = f.last_instance(t)
                assert inst.startswith(f"ewa_periodic_rowpair_kernel<") and int(inst.split(",")[1]) == taps[t], inst
        assert_planes_equal(got, of.get_frame(srcs[0], threads=8), f.out_dims(), what=_id(case))
        batch = _run_batch(torch, gpu_pkg, f, gfmt, srcs, 5, 0)
        assert f.last_kernel(0) == "ewa_periodic_rowpair_kernel"
        for k in range(5):
            assert_planes_equal(batch[k], of.get_frame(srcs[k], threads=8), f.out_dims(), what=_id(case) + f" frame {k}")
        f.set_kernel_mode(gpu_pkg.KernelMode.FULL_WINDOW)
        assert_planes_equal(f.get_frame(srcs[1]), of.get_frame(srcs[1], threads=8), f.out_dims(), what=_id(case) + " full window")
    f.close()


def test_c3_batch_reaches_the_benchmarked_instantiation(gpu_pkg, O):
    """Three C3 frames per call (bench.py: 32): the instantiation `bench.py --config C3` reports as roofline.kernel, every frame
    against the oracle."""
    from test_benchmarked_instances import _batch_against_oracle
    inst = _batch_against_oracle(gpu_pkg, O, "YUV420P16", 1920, 1080, 3840, 2160, dict(tap=8, cplace="mpeg2"), 3, 12345)
    assert inst[0].startswith("ewa_periodic_rowpair_kernel<unsigned short, 16, ") and inst[1].startswith("ewa_periodic_rowpair_kernel<unsigned short, 17, "), inst
