"""ewa_strip_kernel (round 5): border rows and border columns of exactly periodic plans with filter sizes 5 / 7 / 9 at source step 1 --
one register window per lane for the strip's whole thickness, a coefficient set per (border line, phase) in SGPRs.  Forced through
jinc_filter_set_border_strips(3) (small calls take one gather launch over the border frame by themselves) and compared with the
oracle and with the round-4 border forms; `last_border` says which kernels ran."""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs

pytestmark = pytest.mark.gpu

CASES = [
    ("Y8", 192, 108, 384, 216, dict(tap=3)),
    ("Y8", 1000, 70, 2000, 140, dict(tap=3)),                    # several 256-period workgroups along the rows
    ("Y8", 70, 700, 140, 1400, dict(tap=3)),                     # ... and along the columns
    ("Y16", 333, 211, 666, 422, dict(tap=3)),
    ("Y10", 150, 100, 300, 200, dict(tap=4)),                    # fs 9
    ("Y32", 150, 100, 300, 200, dict(tap=2)),                    # fs 5
    ("Y32", 131, 77, 262, 154, dict(tap=4, blur=0.98)),
    ("Y8", 97, 61, 291, 183, dict(tap=3, quant_x=1, quant_y=1)),  # 3x, exactly periodic with one phase per residue: three phases per axis
    ("Y8", 96, 64, 384, 256, dict(tap=3)),                       # 4x
    ("YUV420P8", 256, 144, 512, 288, dict(tap=3)),               # chroma sited as MPEG-2
    ("YUV420P16", 256, 144, 512, 288, dict(tap=4, cplace="topleft")),
    ("RGBPS", 160, 100, 320, 200, dict(tap=3, blur=0.95)),
    ("Y8", 200, 120, 400, 240, dict(tap=3, src_left=2.5, src_top=-1.25, src_width=190.5, src_height=118.0)),  # cropped: uneven borders
]


def _id(c):
    return f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}_tap{c[5]['tap']}"


@pytest.mark.parametrize("frames", [1, 5])
@pytest.mark.parametrize("case", CASES, ids=_id)
def test_strip_kernel_matches_oracle_and_the_other_border_forms(gpu_pkg, O, case, frames):
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th, kw = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    if not f.plan_info(0).periodic:
        f.close()
        pytest.skip("this geometry is not exactly periodic on this build")
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=7100 + k) for k in range(frames)]
    if ofmt.bits == 32:
        rng = np.random.default_rng(3)
        for s in srcs:
            for p in s:
                p[:] = (rng.standard_normal(p.shape) * 0.8).astype(np.float32)
        srcs[-1][0][0, 0] = np.inf          # a non-finite sample in the corner every border kernel reads
        srcs[-1][0][sh - 1, sw // 2] = np.nan
    want = [of.get_frame(s, threads=8) for s in srcs]

    def run(strips):
        f.set_border_strips(strips)
        return [f.get_frame(srcs[0])] if frames == 1 else _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)

    got = run(3)
    if f.last_border(0) == 1:   # (a periodic interior whose border sets do not repeat along the strips: the gather kernel's)
        f.close()
        pytest.skip("this plan has no strip border")
    assert f.last_border(0) & 48 == 48, f.last_border(0)      # rows and columns on ewa_strip_kernel
    others = {1: run(1), 0: run(0)}
    assert others and f.last_border(0) == 1
    for k in range(frames):
        for i, (w, h) in enumerate(f.out_dims()):
            a, b = got[k][i][:h, :w], want[k][i][:h, :w]
            if a.dtype == np.float32:
                na, nb = np.isnan(a), np.isnan(b)
                assert np.array_equal(na, nb), f"{_id(case)} frame {k} plane {i}: NaN footprint differs"
                assert np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32)), f"{_id(case)} frame {k} plane {i}: bits differ"
            else:
                assert np.array_equal(a, b), f"{_id(case)} frame {k} plane {i} differs from the oracle at {int((a != b).sum())} samples"
        for strips, o in others.items():
            for i, (w, h) in enumerate(f.out_dims()):
                assert np.array_equal(got[k][i][:h, :w].view(np.uint8), o[k][i][:h, :w].view(np.uint8)), f"frame {k} plane {i}: strip kernel vs border form {strips}"
    f.close()


def test_strip_kernel_is_the_automatic_choice_in_large_calls(gpu_pkg, O):
    """24 frames of 1080p -> 4K Y8 per call (>= 5e9 taps: the strip border): rows on ewa_strip_kernel and columns inside the interior
    kernel by the rule; the knobs edge_cols = 0 / strip_lds = 0 put the earlier kernels back; every frame of all forms the same bytes,
    three against the oracle."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th, frames = "Y8", 1920, 1080, 3840, 2160, 24
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th)
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=4400 + k) for k in range(frames)]
    fused = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
    assert f.last_border(0) == (16 | 64), f.last_border(0)       # rows on ewa_strip_kernel, columns inside ewa_periodic_quad2_kernel's edge tiles
    assert f.last_instance(0).startswith("ewa_periodic_quad2_kernel<unsigned char"), f.last_instance(0)
    with gpu_pkg.knobs(edge_cols=0):
        pairs = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
        assert f.last_border(0) == (16 | 256), f.last_border(0)  # ... columns on ewa_colpair_kernel
    with gpu_pkg.knobs(edge_cols=0, colpair=0):
        auto = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
        assert f.last_border(0) == (16 | 8), f.last_border(0)    # ... columns (24 frames) on the frame-lane kernel
    with gpu_pkg.knobs(strip_lds=2, edge_cols=0, colpair=0):
        both = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
        assert f.last_border(0) == 48, f.last_border(0)
    with gpu_pkg.knobs(strip_lds=0, edge_cols=0, colpair=0):
        old = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
        assert f.last_border(0) == (2 | 8), f.last_border(0)     # direct row strips + frame-lane columns
    for k in range(frames):
        assert_planes_equal(fused[k], old[k], f.out_dims(), what=f"frame {k}: border columns in the interior kernel vs the round-4 border kernels")
        assert_planes_equal(auto[k], old[k], f.out_dims(), what=f"frame {k}: ewa_strip_kernel vs the round-4 border kernels")
        assert_planes_equal(pairs[k], old[k], f.out_dims(), what=f"frame {k}: ewa_colpair_kernel's columns vs the round-4 border kernels")
        assert_planes_equal(both[k], old[k], f.out_dims(), what=f"frame {k}: ewa_strip_kernel on rows and columns vs the round-4 border kernels")
        if k in (0, 11, frames - 1):
            assert_planes_equal(fused[k], of.get_frame(srcs[k], threads=16), f.out_dims(), what=f"frame {k} vs oracle")
    f.close()


def test_single_plane_calls_take_the_strip_border_earlier_where_the_interior_takes_the_columns(gpu_pkg, O):
    """Rules::kStripBorderMinTapsEdgeCols (round 5): 8 frames of 1080p -> 4K Y8 per call are 3.3e9 taps -- below the 5e9 from which every
    plan takes the strip border, above the 2.4e9 from which single-plane calls do whose border columns the interior kernel computes;
    4 frames stay on the gather launch over the border frame.  Both bit-exact."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th = "Y8", 1920, 1080, 3840, 2160
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th)
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=8800 + k) for k in range(8)]
    want = [of.get_frame(s, threads=16) for s in srcs]
    for frames, border in ((8, 16 | 64), (4, 1)):
        got = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
        assert f.last_border(0) == border, (frames, f.last_border(0))
        for k in range(frames):
            assert_planes_equal(got[k], want[k], f.out_dims(), what=f"{frames} frames per call, frame {k}")
    f.close()
