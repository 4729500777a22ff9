"""SIMD-order compatibility modes: the summation order of the reference's opt = 1 / 2 / 3 paths.

CPU part: the scalar restatement oracle/simd_order.c is pinned by the only reference-derived facts that exist for these
paths -- on C1 (640x360 -> 1280x720 Y8 tap 3, Appendix-A frame) the reference's opt = 1 / 2 / 3 outputs differ from
opt = 0 in 3 / 7 / 7 of 921 600 pixels, by one code value -- and the own-written AVX2
code (oracle/simd_avx2.c, the CPU speed baseline of bench.py) is held bit-equal to the restatement of order 2.
GPU part: kernel_simdorder.hip through the C ABI (jinc_filter_set_simd_order) against the restatement, bit-exact."""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs


def test_restated_simd_orders_reproduce_the_reference_pixel_differences(O):
    fmt = O.FORMATS["Y8"]
    f = O.OracleFilter(fmt, 640, 360, 1280, 720, tap=3)
    src = O.lcg_frame(fmt, 640, 360)
    base = f.get_frame(src, threads=8)[0][:720, :1280].astype(np.int32)
    assert O.crc32_planes(f.get_frame(src, threads=8), f.out_dims()) == "1266b444"  # the opt = 0 known answer
    for order, want in ((1, 3), (2, 7), (3, 7)):
        out = f.get_frame_simd(order, src, threads=8)[0][:720, :1280].astype(np.int32)
        diff = out - base
        assert int((diff != 0).sum()) == want, f"opt={order}: {int((diff != 0).sum())} pixels differ, reference: {want}"
        assert int(np.abs(diff).max()) == 1


CASES = [
    ("Y8", 97, 61, 291, 183, {}),
    ("Y8", 160, 90, 219, 123, {}),                      # no phase structure
    ("Y8", 192, 108, 128, 72, {}),                      # fs 10: two 8-lane groups per row, three 4-lane groups
    ("Y10", 128, 96, 256, 192, {}),                     # saturates to 65535, not to the clip's peak 1023
    ("Y16", 150, 100, 300, 200, dict(tap=6)),           # fs 13
    ("Y32", 128, 96, 256, 192, {}),
    ("YUV420P8", 128, 96, 256, 192, dict(cplace="mpeg2")),
    ("YUV444PS", 96, 64, 192, 128, dict(tap=4)),        # float chroma: lower clamp -0.5
    ("RGBPS", 96, 64, 150, 100, dict(tap=3, blur=0.98)),
    ("Y8", 160, 120, 320, 240, dict(tap=8)),            # fs 17: more than one 16-lane group
]


def _frame(O, fmt, sw, sh, seed):
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=seed)
    if O.FORMATS[fmt].bits == 32:  # negative and > 1 samples so that the lower clamp and its -0.5 / 0 choice matter
        rng = np.random.default_rng(seed)
        for p in src:
            p[:] = (rng.standard_normal(p.shape) * 0.6).astype(np.float32)
    return src


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_own_avx2_code_equals_the_restated_avx2_order(O, case):
    if not O.lib().oracle_avx2_available():
        pytest.skip("host CPU without AVX2 + FMA")
    fmt, sw, sh, tw, th, kw = case
    f = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = _frame(O, fmt, sw, sh, 31)
    assert_planes_equal(f.get_frame_simd(2, src, threads=4, avx2=True), f.get_frame_simd(2, src, threads=1), f.out_dims(), what="avx2")


@pytest.mark.parametrize("case", CASES + [("Y8", 120, 90, 240, 180, dict(tap=16)), ("Y16", 300, 200, 100, 50, dict(tap=4)), ("Y32", 64, 48, 40, 30, {})],
                         ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_own_avx512_code_equals_the_restated_avx512_order(O, case):
    """oracle/simd_avx512.c (16-lane partial sums over all kernel rows, FMA, 512 -> 256 -> 128 fold; the second CPU speed baseline of
    bench.py) against the scalar restatement of the reference's opt = 3 order, bit for bit -- filter sizes with one, two (fs 17, 33,
    34) and three 16-lane groups per kernel row, windows that end on the plane's last sample (masked loads, no over-read)."""
    if not O.lib().oracle_avx512_available():
        pytest.skip("host CPU without AVX-512 F/BW/DQ/VL")
    fmt, sw, sh, tw, th, kw = case
    f = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = _frame(O, fmt, sw, sh, 37)
    assert_planes_equal(f.get_frame_simd(3, src, threads=4, avx512=True), f.get_frame_simd(3, src, threads=1), f.out_dims(), what="avx512")


def test_own_avx512_code_reproduces_the_reference_pixel_difference_count_on_c1(O):
    """The one reference-derived fact about opt = 3: on C1 it differs from opt = 0 in 7 of 921 600 pixels, by one code value."""
    if not O.lib().oracle_avx512_available():
        pytest.skip("host CPU without AVX-512 F/BW/DQ/VL")
    fmt = O.FORMATS["Y8"]
    f = O.OracleFilter(fmt, 640, 360, 1280, 720, tap=3)
    src = O.lcg_frame(fmt, 640, 360)
    base = f.get_frame(src, threads=8)[0][:720, :1280].astype(np.int32)
    out = f.get_frame_simd(3, src, threads=8, avx512=True)[0][:720, :1280].astype(np.int32)
    assert int((out != base).sum()) == 7 and int(np.abs(out - base).max()) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("order", [1, 2, 3], ids=["sse41_order", "avx2_order", "avx512_order"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_gpu_simd_order_modes_match_the_restatement(gpu_pkg, O, case, order):
    fmt, sw, sh, tw, th, kw = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = _frame(O, fmt, sw, sh, 77)
    want = of.get_frame_simd(order, src, threads=4)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    f.set_simd_order(order)
    got = f.get_frame(src)
    assert f.last_kernel(0) == "ewa_simd_order_kernel"
    assert_planes_equal(got, want, f.out_dims(), what=f"order {order}")
    f.set_simd_order(0)  # back to the parity target
    assert_planes_equal(f.get_frame(src), of.get_frame(src, threads=4), f.out_dims(), what="opt=0 after switching back")
    f.close()


@pytest.mark.gpu
def test_gpu_simd_orders_reproduce_the_reference_pixel_differences_on_c1(gpu_pkg, O):
    """The reference-derived counts (3 / 7 / 7 pixels of C1 differ from opt = 0) from the GPU output directly."""
    fmt = O.FORMATS["Y8"]
    src = O.lcg_frame(fmt, 640, 360)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS["Y8"], 640, 360, 1280, 720, device=0, tap=3)
    base = f.get_frame(src)[0][:720, :1280].astype(np.int32)
    for order, want in ((1, 3), (2, 7), (3, 7)):
        f.set_simd_order(order)
        out = f.get_frame(src)[0][:720, :1280].astype(np.int32)
        assert int((out != base).sum()) == want
    f.close()
