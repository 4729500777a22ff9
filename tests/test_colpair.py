"""ewa_colpair_kernel (round 5): the border columns of exactly periodic plans at source step 1 on packed column pairs -- two adjacent
border columns are the halves of a register pair (same samples, different coefficient sets), lanes along the column.  The automatic
choice wherever configured (odd filter sizes 7 .. 17) unless the interior kernel computes the columns itself (test_edge_columns.py).
Forced through jinc_filter_set_border_strips(4) on small planes, compared with the oracle, with the kernels it replaces (knob colpair
= 0) and with the gather kernel's border; `last_border` bit 256 says it ran."""
import numpy as np
import pytest

from conftest import oracle_kwargs

pytestmark = pytest.mark.gpu

CASES = [
    ("Y8", 192, 108, 384, 216, dict(tap=6)),                      # filter size 13: 13 + 11 border columns
    ("Y8", 60, 700, 120, 1400, dict(tap=5)),                      # eleven blocks of 64 period-rows
    ("Y8", 263, 151, 526, 302, dict(tap=7)),
    ("Y16", 333, 211, 666, 422, dict(tap=8)),                     # filter size 17
    ("Y10", 150, 100, 300, 200, dict(tap=6, blur=0.95)),
    ("Y32", 160, 100, 320, 200, dict(tap=6)),                     # float planes, non-finite samples in the columns' windows
    ("RGBPS", 131, 77, 262, 154, dict(tap=5)),
    ("YUV420P16", 320, 180, 640, 360, dict(tap=8)),               # C3's format: chroma sited as MPEG-2
    ("YUV420P8", 256, 144, 512, 288, dict(tap=6, cplace="topleft")),
    ("Y8", 97, 61, 291, 183, dict(tap=5, quant_x=1, quant_y=1)),  # 3x: three row phases
    ("Y8", 96, 64, 384, 256, dict(tap=5)),                        # 4x: four
    ("Y8", 192, 108, 384, 216, dict(tap=3)),                      # filter size 7
    ("Y16", 200, 120, 400, 240, dict(tap=4)),                     # filter size 9
    ("Y32", 150, 100, 300, 200, dict(tap=4, blur=0.98)),
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
def test_border_columns_on_column_pairs_match_the_oracle_and_the_other_column_kernels(gpu_pkg, O, case, frames):
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th, kw = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=6100 + k) for k in range(frames)]
    if ofmt.bits == 32:
        rng = np.random.default_rng(6)
        for s in srcs:
            for p in s:
                p[:] = (rng.standard_normal(p.shape) * 0.8).astype(np.float32)
        srcs[-1][0][sh // 3, 1] = np.inf          # non-finite samples inside the first / last fs source columns
        srcs[-1][0][sh // 2, sw - 2] = np.nan
    want = [of.get_frame(s, threads=8) for s in srcs]

    def run(strips, knob):
        f.set_border_strips(strips)
        with gpu_pkg.knobs(colpair=knob):
            return [f.get_frame(srcs[0])] if frames == 1 else _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)

    got = run(4, 1)
    paired = [t for t in range(f.num_tables) if f.last_border(t) & 256]
    if not paired:
        f.close()
        pytest.skip("no table of this plan runs its border columns on ewa_colpair_kernel")
    for t in paired:
        assert f.last_border(t) & (64 | 32 | 8 | 4 | 1) == 0, f.last_border(t)   # no other column kernel beside it
    others = run(4, 0)
    assert all(f.last_border(t) & 256 == 0 for t in range(f.num_tables))
    gathered = run(0, 1)
    for k in range(frames):
        for i, (w, h) in enumerate(f.out_dims()):
            a = got[k][i][:h, :w]
            assert _same(a, want[k][i][:h, :w]), f"{_id(case)} frame {k} plane {i}: column pairs vs oracle"
            assert _same(a, others[k][i][:h, :w]), f"{_id(case)} frame {k} plane {i}: column pairs vs the kernels they replace"
            assert _same(a, gathered[k][i][:h, :w]), f"{_id(case)} frame {k} plane {i}: column pairs vs the gather kernel's border"
    f.close()
