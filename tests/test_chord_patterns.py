"""The compile-time chord patterns of the chroma planes' (fs - 1)-row x fs-column supports (ewa_periodic_quad2_kernel<.., pattern, 7>,
ewa_periodic_quad2x8_kernel<.., 0u, 9, pattern>; csrc/kernels.h) against the coefficient tables themselves -- no GPU: the product's host
plan gives the four phase sets of a 2x up-scale of 4:2:0 material, the spans of their non-zero taps are derived here independently of
csrc/device_plan.cpp, and the library says whether its pattern (or the row-phase-swapped twin) leaves out no more than those spans allow.
Leaving out a tap whose coefficient is not zero would change results; the GPU tests would catch it, this one says why."""
import numpy as np
import pytest


def _phase_sets(f, table):
    """The interior's four phase sets (q, p) of a 2x plan and its filter size."""
    info = f.plan_info(table)
    assert info.periodic and info.period_x == 2 and info.period_y == 2
    _, _, ids = f.plan_dump(table)
    sets = f.plan_sets(table)
    x0, y0 = info.interior_x0, info.interior_y0
    return [[sets[ids[y0 + q, x0 + p]] for p in range(2)] for q in range(2)], info.filter_size


def _spans(phase_sets, fs):
    """Rows of the joint non-zero box, and per (ly, q) lead / trail zeros common to both column phases, encoded as the library reads them."""
    nz_rows = [r for r in range(fs) if any(phase_sets[q][p][r].any() for q in range(2) for p in range(2))]
    r0, nr = nz_rows[0], nz_rows[-1] - nz_rows[0] + 1
    spans, taps = 0, 0
    for q in range(2):
        for ly in range(nr):
            lead = trail = 3
            union = np.zeros(fs, bool)
            for p in range(2):
                row = phase_sets[q][p][r0 + ly]
                nz = np.nonzero(row)[0]
                a = int(nz[0]) if len(nz) else fs
                b = int(fs - 1 - nz[-1]) if len(nz) else 0
                lead, trail = min(lead, a), min(trail, b)
                union |= row != 0
            spans |= lead << (4 * (2 * ly + q)) | trail << (4 * (2 * ly + q) + 2)
            taps += int(union.sum())
    return nr, spans, taps / 2.0


@pytest.mark.parametrize("fmt", ["YUV420P8", "YUV420P16", "YUV422P10"])
@pytest.mark.parametrize("tap,fs,support_rows,pattern_taps", [(3, 7, 6, 36.0), (4, 9, 8, 60.0)])
def test_chroma_chord_patterns_fit_the_tables(pkg, fmt, tap, fs, support_rows, pattern_taps):
    f = pkg.Filter(pkg.FORMATS[fmt], 1920, 1080, 3840, 2160, device=-1, tap=tap)
    sets, got_fs = _phase_sets(f, 1)   # (4:2:2: sub-sampled horizontally only -- the same column siting)
    assert got_fs == fs
    nr, spans, union_taps = _spans(sets, fs)
    assert nr == support_rows, nr
    which = pkg.lib().jinc_debug_chord_pattern(fs, spans)
    assert which in (1, 2), f"{fmt} tap {tap}: neither pattern fits the table's spans {spans:#x}"
    # ... and the pattern is as tight as spans can be: the union of the phase pair's non-zero taps
    assert union_taps == pattern_taps, union_taps
    # a plan whose rows are all full takes no pattern; one tap more in a row the pattern trims does not fit
    assert pkg.lib().jinc_debug_chord_pattern(fs, 0) == 0
    assert pkg.lib().jinc_debug_chord_pattern(5, spans) == -1
    f.close()


def test_luma_tables_have_no_such_support(pkg):
    """Luma at 2x: a square (fs - 1) x (fs - 1) box -- the 6 x 6 / 8 x 8 forms, not these."""
    for fmt, table, kw in (("YUV420P8", 0, dict(tap=3)), ("Y8", 0, dict(tap=4))):
        f = pkg.Filter(pkg.FORMATS[fmt], 1920, 1080, 3840, 2160, device=-1, **kw)
        sets, fs = _phase_sets(f, table)
        nz_cols = [c for c in range(fs) if any(sets[q][p][:, c].any() for q in range(2) for p in range(2))]
        assert len(nz_cols) == fs - 1, (fmt, table, nz_cols)
        f.close()
