"""Host logic of the product (no GPU): the compact plan must expand to exactly the reference's
per-pixel tables as restated by the oracle -- start_x/start_y and every coefficient bit."""
import json
import os

import numpy as np
import pytest

from conftest import oracle_kwargs

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))

CASES = [
    ("Y8", 64, 48, 160, 120, {}),
    ("Y8", 640, 360, 1280, 720, {}),
    ("Y8", 37, 23, 91, 50, dict(tap=3, blur=0.9, src_left=1.3, src_top=0.7, src_width=33.1, src_height=20.2)),
    ("Y8", 64, 48, 40, 30, {}),                      # downscale, filter_size 11
    ("Y8", 320, 180, 480, 270, {}),                  # 1.5x: float drift, many phases
    ("Y8", 50, 40, 120, 96, dict(tap=4, blur=0.98, src_left=-2.5, src_top=1.25, src_width=55, src_height=41.5,
                                 quant_x=7, quant_y=13)),
    ("Y8", 40, 30, 80, 60, dict(tap=2, src_left=0.125, src_top=0, src_width=20, src_height=15, quant_x=1, quant_y=1)),
    ("Y8", 100, 80, 300, 240, dict(tap=1)),
    ("Y8", 90, 70, 360, 280, dict(tap=5)),           # libstdc++ cyl_bessel_j branch of the LUT
    ("Y8", 120, 90, 240, 180, dict(tap=16)),
    ("Y8", 64, 64, 64, 64, dict(src_left=0.5, src_top=-0.25)),   # pure shift
    ("Y8", 200, 120, 100, 300, dict(src_width=-10, src_height=-6.5)),  # relative crop, mixed up/down
    ("YUV420P16", 320, 180, 640, 360, dict(tap=8, cplace="mpeg2")),
    ("YUV420P8", 320, 180, 640, 360, dict(tap=3, cplace="topleft")),
    ("YUV422P8", 320, 180, 500, 300, dict(tap=3, cplace="mpeg1")),
    ("YUV411P8", 320, 180, 640, 360, dict(tap=3)),
    ("YUVA420P8", 128, 96, 256, 192, dict(tap=3)),
    ("RGBPS", 200, 100, 400, 200, dict(tap=4, blur=0.98)),
    ("Y8", 1280, 720, 1920, 1080, {}),
    ("Y8", 1920, 1080, 1280, 720, {}),
]


def _compare(pkg, O, fmt, sw, sh, tw, th, kw):
    f = pkg.Filter(pkg.FORMATS[fmt], sw, sh, tw, th, device=-1, **kw)
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    assert np.array_equal(f.lut(), of.lut), "LUT differs"
    assert f.num_tables == len(of.tables)
    infos = []
    for t in range(f.num_tables):
        info = f.plan_info(t)
        sx, sy, ids = f.plan_dump(t)
        sets = f.plan_sets(t)
        ot = of.tables[t]
        m = ot.meta()
        fs, cs = ot.filter_size, ot.coeff_stride
        assert info.filter_size == fs and (info.dst_width, info.dst_height) == (ot.dst_w, ot.dst_h)
        assert np.array_equal(m[:, :, 0], np.broadcast_to(sx[None, :], m.shape[:2])), "start_x"
        assert np.array_equal(m[:, :, 1], np.broadcast_to(sy[:, None], m.shape[:2])), "start_y"
        fac = ot.factor().reshape(-1, fs, cs)
        assert not fac[:, :, fs:].any(), "stride padding must stay zero (ref :476)"
        oidx = m[:, :, 2] // (fs * cs)
        pairs = np.unique(np.stack([ids.ravel(), oidx.ravel()], 1), axis=0)
        a = sets[pairs[:, 0]].view(np.uint32)
        b = np.ascontiguousarray(fac[pairs[:, 1]][:, :, :fs]).view(np.uint32)
        assert np.array_equal(a, b), "coefficient bits differ"
        assert info.num_sets <= ot.num_sets
        infos.append(info)
    f.close()
    return infos


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_plan_matches_reference_tables(pkg, O, case):
    _compare(pkg, O, *case)


def test_c2_plan_is_compact_and_periodic(pkg, O):
    """Headline config: 196 distinct sets, 2x2 phase period, interior found."""
    (info,) = _compare(pkg, O, "Y8", 1920, 1080, 3840, 2160, {})
    want = {tuple(d["src"]) + (d["tap"],): d["distinct"] for d in KAT["distinct_sets_by_content"]}
    assert info.num_sets == want[(1920, 1080, 3)]
    assert info.periodic == 1 and (info.period_x, info.period_y, info.step_x, info.step_y) == (2, 2, 1, 1)
    assert info.plan_bytes < 1 << 20


def test_tiny_source_is_rejected(pkg):
    """Source smaller than the filter footprint: the reference reads out of bounds."""
    with pytest.raises(pkg.JincError) as e:
        pkg.Filter(pkg.FORMATS["Y8"], 5, 5, 20, 20, device=-1)
    assert "smaller than the filter footprint" in str(e.value)


def test_host_only_instance_refuses_frames(pkg):
    """No CPU fallback: a filter without a device must fail loudly on frame calls."""
    f = pkg.Filter(pkg.FORMATS["Y8"], 64, 48, 128, 96, device=-1)
    src = [pkg.alloc_plane(64, 48, np.uint8)]
    with pytest.raises(pkg.JincError) as e:
        f.get_frame(src)
    assert e.value.code == -2


@pytest.mark.parametrize("sw,sh,tw,th,exact,q", [
    (1280, 720, 1920, 1080, False, (3, 3, 2, 2)),     # 1.5x: phases drift, origins affine
    (640, 360, 1920, 1080, False, (3, 3, 1, 1)),      # 3x
    (720, 480, 1920, 1080, False, (8, 9, 3, 4)),      # 8/3 x 9/4
    (1440, 1080, 1920, 1440, True, (4, 4, 3, 3)),     # 4/3x: exact arithmetic (step 0.75), periodic with source step 3
    (1920, 1080, 3840, 2160, True, (2, 2, 1, 1)),     # 2x
    (1920, 1080, 1280, 720, True, (2, 2, 3, 3)),      # 2/3 down-scale
])
def test_quasi_periodic_structure(pkg, sw, sh, tw, th, exact, q):
    """Plan analysis used to pick the interior kernel: exact periodicity vs affine window origins only."""
    f = pkg.Filter(pkg.FORMATS["Y8"], sw, sh, tw, th, device=-1)
    info = f.plan_info()
    assert bool(info.periodic) == exact
    assert info.quasi == 1
    assert (info.quasi_period_x, info.quasi_period_y, info.quasi_step_x, info.quasi_step_y) == q
    if exact:
        assert (info.period_x, info.period_y, info.step_x, info.step_y) == q
    # the affine property itself, re-checked from the dumped plan
    sx, sy, _ = f.plan_dump()
    for start, a0, a1, P, S in ((sx, info.interior_x0, info.interior_x1, q[0], q[2]),
                                (sy, info.interior_y0, info.interior_y1, q[1], q[3])):
        seg = start[a0:a1]
        assert np.all(seg[P:] - seg[:-P] == S)


def test_create_on_missing_device_fails_loudly(pkg):
    """Without a usable HIP device the filter cannot be created for frame work (no silent CPU fallback)."""
    n = pkg.device_count()
    with pytest.raises(pkg.JincError) as e:
        pkg.Filter(pkg.FORMATS["Y8"], 64, 48, 128, 96, device=n + 3)   # never a valid index
    assert e.value.code in (-2, -3)
    assert "HIP" in str(e.value) or "device" in str(e.value)


@pytest.mark.parametrize("seed", range(40))
def test_plan_matches_reference_tables_random_geometry(pkg, O, seed):
    """Seeded random geometry (sizes, ratios incl. down-scales, taps 1..16, quantisation, blur, crops, formats):
    the compact plan expands to the oracle's reference-layout tables bit for bit."""
    rng = np.random.default_rng(7000 + seed)
    fmt = ["Y8", "YUV420P8", "YUV422P16", "YUV411P8", "RGBPS"][rng.integers(5)]
    sw = int(rng.integers(48, 200)) & ~3
    sh = int(rng.integers(48, 160)) & ~1
    tw = max(16, int(sw * rng.uniform(0.4, 3.5)) & ~3)
    th = max(16, int(sh * rng.uniform(0.4, 3.5)) & ~1)
    kw = dict(tap=int(rng.integers(1, 17)))
    if rng.random() < 0.5:
        kw.update(quant_x=int(rng.integers(1, 257)), quant_y=int(rng.integers(1, 257)))
    if rng.random() < 0.5:
        kw["blur"] = float(np.round(rng.uniform(0.7, 1.3), 3))
    if rng.random() < 0.5:
        kw.update(src_left=float(np.round(rng.uniform(-4, 8), 3)), src_top=float(np.round(rng.uniform(-4, 8), 3)),
                  src_width=float(np.round(sw - rng.uniform(0, 12), 3)), src_height=float(np.round(sh - rng.uniform(0, 12), 3)))
    if fmt == "YUV420P8":
        kw["cplace"] = ["mpeg2", "mpeg1", "topleft"][rng.integers(3)]
    try:
        _compare(pkg, O, fmt, sw, sh, tw, th, kw)
    except pkg.JincError as e:
        assert "smaller than the filter footprint" in str(e)   # heavy down-scale with many taps: reference UB


def test_lut_matches_oracle_over_taps_and_blurs(pkg, O):
    """The 1024-entry double LUT (ref JincResize.cpp:255-275) of product and oracle, every tap 1..16 x 25 blurs:
    covers the Taylor, libstdc++ cyl_bessel_j and asymptotic branches on many arguments.  (The asymptotic branch
    is where sin()/cos() vs sincos() differ by an ulp in glibc -- both sides call sincos(), like the reference's
    GCC Release build does after optimisation)"""
    blurs = [1.0] + [float(b) for b in np.round(np.linspace(0.55, 1.45, 24), 3)]
    for tap in range(1, 17):
        for blur in blurs:
            f = pkg.Filter(pkg.FORMATS["Y8"], 256, 256, 512, 512, device=-1, tap=tap, blur=blur)
            a = f.lut()
            f.close()
            b = O.make_lut(tap, blur)
            assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), (tap, blur)
