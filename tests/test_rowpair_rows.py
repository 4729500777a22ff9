"""Border rows on ewa_periodic_rowpair_kernel (round 5): at 2x with taps 5 .. 8 (filter sizes 11 .. 17) the border rows of each end
of a plane are one launch of the interior's packed kernel with the rows as its row phases (device_plan.cpp plan_rowpair_rows).
Forced through jinc_filter_set_border_strips(4) on small planes (small calls take one gather launch over the border frame by
themselves), compared with the oracle, with ewa_direct_kernel's row strips (knob rowpair_rows = 0) and with the gather kernel's
border; `last_border` bit 128 says the launches ran."""
import numpy as np
import pytest

from conftest import oracle_kwargs

pytestmark = pytest.mark.gpu

CASES = [
    ("Y8", 192, 108, 384, 216, dict(tap=6)),
    ("Y8", 700, 60, 1400, 120, dict(tap=5)),                      # several tiles of 256 periods along the rows
    ("Y8", 263, 151, 526, 302, dict(tap=7)),
    ("Y16", 333, 111, 666, 222, dict(tap=8)),                     # filter size 17: 17 + 15 border rows
    ("Y10", 150, 100, 300, 200, dict(tap=6, blur=0.95)),
    ("Y32", 160, 100, 320, 200, dict(tap=6)),                     # float planes: every tap executed
    ("RGBPS", 131, 77, 262, 154, dict(tap=5)),
    ("YUV420P16", 320, 180, 640, 360, dict(tap=8)),               # C3's format: chroma sited as MPEG-2
    ("YUV420P8", 256, 144, 512, 288, dict(tap=6, cplace="topleft")),
    ("Y8", 200, 120, 400, 240, dict(tap=6, src_left=2.5, src_top=-1.25, src_width=190.5, src_height=118.0)),  # cropped: uneven borders
]


def _id(c):
    return f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}_" + "_".join(f"{k}{v}" for k, v in c[5].items())


def _same(a, b):
    if a.dtype != np.float32:
        return np.array_equal(a, b)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32))


@pytest.mark.parametrize("frames", [1, 3])
@pytest.mark.parametrize("case", CASES, ids=_id)
def test_border_rows_on_the_pair_kernel_match_the_oracle_and_the_row_strips(gpu_pkg, O, case, frames):
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th, kw = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=5100 + k) for k in range(frames)]
    if ofmt.bits == 32:
        rng = np.random.default_rng(5)
        for s in srcs:
            for p in s:
                p[:] = (rng.standard_normal(p.shape) * 0.8).astype(np.float32)
        srcs[-1][0][1, sw // 3] = np.inf          # non-finite samples inside the first / last fs source rows
        srcs[-1][0][sh - 2, sw // 2] = np.nan
    want = [of.get_frame(s, threads=8) for s in srcs]

    def run(strips):
        f.set_border_strips(strips)
        return [f.get_frame(srcs[0])] if frames == 1 else _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)

    got = run(4)
    paired = [t for t in range(f.num_tables) if f.last_border(t) & 128]
    if not paired:
        f.close()
        pytest.skip("no table of this plan runs its border rows on the pair kernel")
    for t in paired:
        assert f.last_border(t) & (16 | 2) == 0, f.last_border(t)   # no other row kernel beside it
    with gpu_pkg.knobs(rowpair_rows=0):
        strips = run(4)
        assert all(f.last_border(t) & 128 == 0 for t in range(f.num_tables))
    gathered = run(0)
    for k in range(frames):
        for i, (w, h) in enumerate(f.out_dims()):
            a = got[k][i][:h, :w]
            assert _same(a, want[k][i][:h, :w]), f"{_id(case)} frame {k} plane {i}: pair-kernel border rows vs oracle"
            assert _same(a, strips[k][i][:h, :w]), f"{_id(case)} frame {k} plane {i}: pair-kernel border rows vs ewa_direct_kernel's row strips"
            assert _same(a, gathered[k][i][:h, :w]), f"{_id(case)} frame {k} plane {i}: pair-kernel border rows vs the gather kernel's border"
    f.close()
