"""Host side of the runs form of the direct kernel (csrc/plan.cpp build_plan_runs): the rectangles a drifting plan is cut into.
No device needed.  Properties: the rectangles tile the interior block exactly once; inside a rectangle every pixel has the
rectangle's coefficient set and its window origin is affine in the period index -- the direct kernel's premise; the item numbering
is what the kernel's lookup expects."""
import numpy as np
import pytest

CASES = [
    ("Y8", 1280, 720, 1920, 1080, dict(tap=8)),     # 1.5x with Jinc256
    ("Y8", 1280, 720, 1920, 1080, dict(tap=4)),
    ("Y8", 640, 360, 1920, 1080, dict(tap=6)),      # 3x
    ("YUV420P8", 720, 480, 1920, 1080, dict(tap=4)),  # DVD -> 1080p: 72 phases, luma and chroma tables
    ("Y16", 768, 432, 1920, 1080, dict(tap=6)),     # 5/2
    ("Y8", 322, 182, 483, 273, dict(tap=8, blur=0.9, quant_x=128, quant_y=64)),   # (with quant 97 x 31 this small plan is exactly periodic)
    ("Y8", 320, 180, 480, 270, dict(tap=3)),        # fs 7: the list exists whatever kernel then uses it
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}_tap{c[5].get('tap')}")
def test_rectangles_tile_the_interior_and_are_exactly_periodic_inside(pkg, case):
    fmt, sw, sh, tw, th, kw = case
    f = pkg.Filter(pkg.FORMATS[fmt], sw, sh, tw, th, device=-1, **kw)
    for t in range(f.num_tables):
        info = f.plan_info(t)
        assert info.quasi == 1 and info.periodic == 0
        px, py, sx, sy = info.quasi_period_x, info.quasi_period_y, info.quasi_step_x, info.quasi_step_y
        start_x, start_y, ids = f.plan_dump(t)
        runs, n_items = f.plan_runs(t)
        assert len(runs) > 0
        ni, nj = (info.interior_x1 - info.interior_x0) // px, (info.interior_y1 - info.interior_y0) // py
        x_end, y_end = info.interior_x0 + px * ni, info.interior_y0 + py * nj
        seen = np.zeros(ids.shape, np.int32)
        items = 0
        for k, (set_id, x0, y0, sx0, sy0, rni, rnj, first) in enumerate(runs):
            assert first == items, k
            items += ((rni + 3) // 4 * ((rnj + 3) // 4) + 63) // 64
            xs = x0 + px * np.arange(rni)
            ys = y0 + py * np.arange(rnj)
            assert xs[-1] < x_end and ys[-1] < y_end and x0 >= info.interior_x0 and y0 >= info.interior_y0
            block = ids[np.ix_(ys, xs)]
            assert np.all(block == set_id), k
            assert np.array_equal(start_x[xs], sx0 + sx * np.arange(rni)), k
            assert np.array_equal(start_y[ys], sy0 + sy * np.arange(rnj)), k
            seen[np.ix_(ys, xs)] += 1
        assert items == n_items
        inside = np.zeros(ids.shape, bool)
        inside[info.interior_y0:y_end, info.interior_x0:x_end] = True
        assert np.all(seen[inside] == 1) and np.all(seen[~inside] == 0)
        # ordered by position: consecutive items read neighbouring source rows
        key = runs[:, 2].astype(np.int64) * (1 << 20) + runs[:, 1]
        assert np.all(np.diff(key) > 0)
    f.close()


@pytest.mark.parametrize("case", [("Y8", 1920, 1080, 3840, 2160, {}), ("Y8", 1280, 720, 1754, 986, {}), ("Y8", 3840, 2160, 1920, 1080, {})],
                         ids=["2x_periodic", "1.37x_no_affine_origins", "half_periodic"])
def test_plans_without_runs(pkg, case):
    fmt, sw, sh, tw, th, kw = case
    f = pkg.Filter(pkg.FORMATS[fmt], sw, sh, tw, th, device=-1, **kw)
    runs, n_items = f.plan_runs(0)
    assert len(runs) == 0 and n_items == 0
    f.close()
