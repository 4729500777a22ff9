"""GPU parity tests (run on a real MI355X with `-m gpu`): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs -- bit-exact for u8/u16 AND for float planes
(strict sequential un-fused fp32, so "<= 1 ULP" is met with 0 ULP) -- and against the reference's
own opt=0 crc32 known answers at BASELINE.json's full sizes."""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

from conftest import assert_planes_equal, fresh_copies, fresh_planes, oracle_kwargs, to_device, to_host

pytestmark = pytest.mark.gpu

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))

# (format, src_w, src_h, dst_w, dst_h, script args)
SMALL_CASES = [
    ("Y8", 64, 48, 160, 120, {}),                                   # the SURVEY tiny KAT shape (non-periodic)
    ("Y8", 640, 360, 1280, 720, {}),                                # C1 shape: periodic kernel + border frame
    ("Y8", 200, 120, 400, 240, dict(tap=4)),                        # fs 9 periodic
    ("Y8", 131, 77, 262, 154, {}),                                  # ragged tiles (not multiples of 64 / 28)
    ("Y8", 97, 61, 291, 183, {}),                                   # 3x: float-drift phases
    ("Y8", 96, 64, 384, 256, {}),                                   # 4x
    ("Y8", 320, 180, 480, 270, {}),                                 # 1.5x: gather kernel everywhere
    ("Y8", 1920 // 4, 1080 // 4, 1280 // 4, 720 // 4, {}),          # downscale, fs 10
    ("Y8", 64, 48, 40, 30, {}),                                     # downscale, fs 11
    ("Y8", 640, 480, 64, 48, {}),                                   # 10x downscale: fs 65, footprint beyond the LDS tile
    ("Y16", 300, 200, 100, 50, dict(tap=4)),                        # anisotropic downscale (3x / 4x), fs 34
    ("Y32", 256, 256, 128, 128, {}),                                # 2:1 downscale, fs 13, period 1 / source step 2
    ("Y8", 37, 23, 91, 50, dict(tap=3, blur=0.9, src_left=1.3, src_top=0.7, src_width=33.1, src_height=20.2)),
    ("Y8", 50, 40, 120, 96, dict(tap=4, blur=0.98, src_left=-2.5, src_top=1.25, src_width=55, src_height=41.5,
                                 quant_x=7, quant_y=13)),
    ("Y8", 40, 30, 80, 60, dict(tap=2, src_left=0.125, src_top=0, src_width=20, src_height=15, quant_x=1, quant_y=1)),
    ("Y8", 100, 80, 300, 240, dict(tap=1)),
    ("Y8", 90, 70, 180, 140, dict(tap=5)),
    ("Y8", 160, 120, 320, 240, dict(tap=8)),                        # fs 17
    ("Y8", 120, 90, 240, 180, dict(tap=16)),                        # fs 33 (runtime-size loop)
    ("Y8", 150, 100, 300, 200, dict(tap=2)),                        # fs 5  row-streamed periodic kernel
    ("Y16", 150, 100, 300, 200, dict(tap=6)),                       # fs 13
    ("Y32", 150, 100, 300, 200, dict(tap=7)),                       # fs 15
    ("Y8", 300, 40, 1200, 160, dict(tap=1)),                        # 4x, fs 3, several 256-column tiles
    ("Y16", 531, 70, 1062, 140, dict(tap=8)),                       # fs 17, ragged tile edges in x and y
    ("Y10", 128, 96, 256, 192, {}),                                 # peak 1023 clamp
    ("Y12", 128, 96, 256, 192, dict(tap=4)),
    ("Y14", 128, 96, 200, 150, {}),
    ("Y16", 128, 96, 256, 192, {}),
    ("Y32", 128, 96, 256, 192, {}),
    ("YUV420P8", 128, 96, 256, 192, dict(cplace="mpeg2")),
    ("YUV420P8", 128, 96, 256, 192, dict(cplace="mpeg1")),
    ("YUV420P8", 128, 96, 256, 192, dict(cplace="topleft")),
    ("YUV420P16", 160, 96, 320, 192, dict(tap=8, cplace="mpeg2")),  # C3 in miniature
    ("YUV422P10", 128, 96, 300, 200, {}),
    ("YUV411P8", 128, 96, 256, 192, {}),
    ("YUV444P16", 96, 64, 192, 128, {}),
    ("YUVA420P8", 128, 96, 256, 192, {}),                           # alpha uses the luma table (ref :555)
    ("YUV420PS", 128, 96, 256, 192, {}),
    ("RGBP8", 96, 64, 192, 128, {}),
    ("RGBAP16", 96, 64, 200, 100, {}),
    ("RGBPS", 200, 100, 400, 200, dict(tap=4, blur=0.98)),          # C4 in miniature
    # 3/2 on a small frame: the interior classes are still exactly periodic (source step 2 -> direct kernel) while every
    # border pixel owns a coefficient set -- the strip kernels must not be used (found by the widened random sweep)
    ("YUV420P8", 88, 108, 132, 162, dict(tap=2, quant_x=255, quant_y=67, cplace="mpeg2")),
]


def _id(c):
    extra = "_".join(f"{k}{v}" for k, v in c[5].items() if k in ("tap", "cplace"))
    return f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}" + (f"_{extra}" if extra else "")


@pytest.mark.parametrize("mode", [0, 1, 15], ids=["auto", "gather", "full_window"])
@pytest.mark.parametrize("case", SMALL_CASES, ids=_id)
def test_get_frame_matches_oracle(gpu_pkg, O, case, mode):
    fmt, sw, sh, tw, th, kw = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=4242)
    want = of.get_frame(src, threads=4)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    f.set_kernel_mode(mode)
    got = f.get_frame(src)
    assert_planes_equal(got, want, f.out_dims(), what=_id(case))
    f.close()


@pytest.mark.parametrize("k", [k for k in KAT["outputs"]], ids=lambda k: k["name"])
def test_reference_known_answers_on_gpu(gpu_pkg, O, k):
    """crc32 of the GPU output == crc32 of the reference's own opt=0 output, full sizes."""
    fmt = gpu_pkg.FORMATS[k["format"]]
    src = O.lcg_frame(O.FORMATS[k["format"]], *k["src"])  # the Appendix-A synthetic frame (generator only)
    f = gpu_pkg.Filter(fmt, k["src"][0], k["src"][1], k["dst"][0], k["dst"][1], device=0, **k["args"])
    got = f.get_frame(src)
    crc, sha = 0, hashlib.sha256()
    for p, (w, h) in zip(got, f.out_dims()):
        b = np.ascontiguousarray(p[:h, :w]).tobytes()
        crc = zlib.crc32(b, crc)
        sha.update(b)
    assert f"{crc & 0xFFFFFFFF:08x}" == k["crc32"]
    if "sha256_prefix" in k:
        assert sha.hexdigest().startswith(k["sha256_prefix"])
    f.close()


@pytest.mark.parametrize("mode", [0, 1, 15], ids=["auto", "gather", "full_window"])
@pytest.mark.parametrize("k", KAT["outputs_r5"], ids=lambda k: k["name"])
def test_known_answers_recorded_by_the_round5_judge_on_gpu(gpu_pkg, O, k, mode):
    """crc32 of the GPU output == the reference's own opt=0 output, 22 cases the round-5 judge recorded from its run of the
    reference (seed-777 LCG frame; crops, sitings, every chroma layout, 10- / 14-bit, float, quant, down-scales, taps
    5 ... 16): no oracle on this path, the frame generator aside."""
    fmt = gpu_pkg.FORMATS[k["format"]]
    src = O.lcg_frame(O.FORMATS[k["format"]], *k["src"], seed=k["seed"])
    f = gpu_pkg.Filter(fmt, k["src"][0], k["src"][1], k["dst"][0], k["dst"][1], device=0, **k["args"])
    f.set_kernel_mode(mode)
    got = f.get_frame(src)
    crc, n = 0, 0
    for p, (w, h) in zip(got, f.out_dims()):
        b = np.ascontiguousarray(p[:h, :w]).tobytes()
        crc = zlib.crc32(b, crc)
        n += len(b)
    f.close()
    assert n == k["bytes"]
    assert f"{crc & 0xFFFFFFFF:08x}" == k["crc32"]


@pytest.mark.parametrize("tw,th,modes", [(192, 128, (0, 1, 9)), (48, 32, (0, 1))], ids=["2x", "half_direct_kernel"])
def test_float_special_values(gpu_pkg, O, tw, th, modes):
    """opt=0 neither clamps nor NaN-guards float sources; denormals must survive."""
    fmt = "Y32"
    sw, sh = 96, 64
    rng = np.random.default_rng(5)
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=1)
    p = src[0]
    p[:sh, :sw] = (rng.standard_normal((sh, sw)) * 4).astype(np.float32)          # negative and > 1 values
    p[8:36, 4:44] = np.float32(1e-41)                                              # denormal inputs (block > window)
    p[12:20, 10:30] = np.float32(-3e-39)
    p[28, 50] = np.inf
    p[40, 20] = -np.inf
    p[50, 60] = np.nan
    p[36:50, 60:90] = 0.0
    p[36:50, 60:90] *= -1.0                                                         # negative zeros
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    want = of.get_frame(src)[0][:th, :tw]
    for mode in modes:
        f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
        f.set_kernel_mode(mode)
        got = f.get_frame(src)[0][:th, :tw]
        nan_w, nan_g = np.isnan(want), np.isnan(got)
        assert np.array_equal(nan_w, nan_g), "NaN footprint differs"
        assert np.array_equal(got[~nan_g].view(np.uint32), want[~nan_w].view(np.uint32)), "finite/inf/denormal bits differ"
        assert nan_w.any() and np.isinf(want).any()
        sub = np.abs(want[~nan_w]) < np.float32(1.1754944e-38)
        assert (sub & (want[~nan_w] != 0)).any(), "test must exercise denormal outputs"
        f.close()


def test_integer_extremes(gpu_pkg, O):
    """All-zero, all-peak, checkerboard and hard-edged bars: the bars ring below 0 and above peak, which
    exercises both clamp bounds (ref :582)."""
    for fmt, peak in (("Y8", 255), ("Y10", 1023), ("Y16", 65535)):
        sw, sh, tw, th = 128, 96, 256, 192
        of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
        f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
        yy, xx = np.mgrid[0:sh, 0:sw]
        for name, img in (("zero", np.zeros((sh, sw))), ("peak", np.full((sh, sw), peak)),
                          ("checker", ((xx + yy) % 2) * peak), ("stripes", (xx % 2) * peak),
                          ("bars", ((xx // 8 + yy // 8) % 2) * peak)):
            p = gpu_pkg.alloc_plane(sw, sh, O.FORMATS[fmt].dtype)
            p[:sh, :sw] = img
            want = of.get_frame([p])
            got = f.get_frame([p])
            assert_planes_equal(got, want, f.out_dims(), what=f"{fmt} {name}")
            if name == "bars":
                v = want[0][:th, :tw]
                assert v.min() == 0 and v.max() == peak
                # without the clamp the sums leave [0, peak]: check via the float path on the same pattern
                fo = O.OracleFilter(O.FORMATS["Y32"], sw, sh, tw, th)
                pf = gpu_pkg.alloc_plane(sw, sh, np.float32)
                pf[:sh, :sw] = img
                vf = fo.get_frame([pf])[0][:th, :tw]
                assert vf.min() < -0.5 and vf.max() > peak + 0.5, "bars must overshoot both clamp bounds"
        f.close()


def test_pitch_and_padding_are_respected(gpu_pkg, O):
    """Odd pitches on both sides; bytes outside row_size must be neither read into results nor written."""
    sw, sh, tw, th = 100, 60, 200, 120
    of = O.OracleFilter(O.FORMATS["Y16"], sw, sh, tw, th)
    rng = np.random.default_rng(3)
    src = np.full((sh, 173), 0xABCD, np.uint16)
    src[:, :sw] = rng.integers(0, 65536, (sh, sw), dtype=np.uint16)
    want = of.get_frame([src])[0]
    f = gpu_pkg.Filter(gpu_pkg.FORMATS["Y16"], sw, sh, tw, th, device=0)
    import ctypes as C
    dst = np.full((th, 311), 0x5A5A, np.uint16)
    P4, I4 = C.c_void_p * 4, C.c_int * 4
    sp, spi, dp, dpi = P4(), I4(), P4(), I4()
    sp[0], spi[0], dp[0], dpi[0] = src.ctypes.data, src.strides[0], dst.ctypes.data, dst.strides[0]
    assert gpu_pkg.lib().jinc_filter_get_frame(f._h, sp, spi, dp, dpi) == 0
    assert np.array_equal(dst[:, :tw], want[:th, :tw])
    assert (dst[:, tw:] == 0x5A5A).all(), "padding was overwritten"
    f.close()


def test_device_batch_path(gpu_pkg, O):
    """jinc_filter_process_device: device-resident planes, batch of frames in one call (frames = shard unit)."""
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    fmt, sw, sh, tw, th, n = "YUV420P8", 192, 108, 384, 216, 5
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    sdims, ddims = gpu_pkg.FORMATS[fmt].plane_dims(sw, sh), f.out_dims()
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=100 + i) for i in range(n)]
    src_t = [to_device(torch.stack([torch.from_numpy(np.ascontiguousarray(fr[i])) for fr in frames])) for i in range(3)]
    dst_t = [torch.zeros((n, h, (w + 63) // 64 * 64), dtype=torch.uint8, device="cuda") for (w, h) in ddims]
    stream = torch.cuda.current_stream()
    f.process_device([t.data_ptr() for t in src_t], [t.stride(1) for t in src_t], [t.stride(0) for t in src_t],
                     [t.data_ptr() for t in dst_t], [t.stride(1) for t in dst_t], [t.stride(0) for t in dst_t],
                     n, stream=stream.cuda_stream)
    stream.synchronize()
    for k in range(n):
        want = of.get_frame(frames[k], threads=4)
        got = [to_host(dst_t[i][k]).numpy() for i in range(3)]
        assert_planes_equal(got, want, ddims, what=f"frame {k}")
    f.close()


def test_full_size_properties(gpu_pkg, O):
    """Size-independent properties at the headline size (1080p -> 4K Y8 tap 3):
    constant frames map to the same constant (coefficient sets are normalised, ref :505-514),
    and auto (periodic) and forced-gather kernels agree bit for bit on random data."""
    sw, sh, tw, th = 1920, 1080, 3840, 2160
    f = gpu_pkg.Filter(gpu_pkg.FORMATS["Y8"], sw, sh, tw, th, device=0)
    info = f.plan_info()
    assert info.periodic == 1 and info.num_sets == 196
    for v in (0, 1, 128, 255):
        p = gpu_pkg.alloc_plane(sw, sh, np.uint8)
        p[:] = v
        out = f.get_frame([p])[0][:th, :tw]
        assert (out == v).all()
    src = O.lcg_frame(O.FORMATS["Y8"], sw, sh, seed=777)
    a = f.get_frame(src)[0][:th, :tw].copy()
    f.set_kernel_mode(1)
    b = f.get_frame(src)[0][:th, :tw]
    assert np.array_equal(a, b)
    f.close()


FULL_SIZE = [
    ("Y8", 3840, 2160, 1920, 1080, {}, "ewa_direct_kernel"),             # 4K -> 1080p: direct interior, strips
    ("YUV420P8", 1280, 720, 1920, 1080, {}, "ewa_quasi_kernel"),         # 720p -> 1080p: drifting, per-lane coefficients
    ("Y16", 1920, 1080, 3840, 2160, dict(tap=12), "ewa_direct_kernel"),  # fs 25 at full size
    ("Y8", 1280, 720, 1754, 986, {}, "ewa_gather_kernel"),               # no phase structure
]


@pytest.mark.parametrize("case", FULL_SIZE, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_full_size_frames_of_the_other_kernels_match_the_oracle(gpu_pkg, O, case):
    """Real frame sizes (many tiles, 32-bit offset ranges, all border structures) through the kernels the small cases
    reach only with a few tiles: bit-exact against the oracle, plus agreement with the forced gather kernel."""
    fmt, sw, sh, tw, th, kw, kernel = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=2024)
    want = of.get_frame(src, threads=16)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    assert f.interior_kernel(0) == kernel
    got = f.get_frame(src)
    assert_planes_equal(got, want, f.out_dims(), what=f"{fmt} {sw}x{sh}->{tw}x{th}")
    if kernel == "ewa_quasi_kernel":
        # one frame of this size goes to the runs form of the direct kernel by itself (fs 7, source step 2, small call:
        # csrc/dispatch.cpp Rules); the quasi-periodic kernel, which larger calls take, is forced here
        assert f.last_kernel(0) == "ewa_direct_runs_kernel"
        f.set_kernel_mode(10)
        assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=f"{fmt} {sw}x{sh}->{tw}x{th} quasi-periodic kernel")
        assert f.last_kernel(0) == "ewa_quasi_kernel"
    f.close()


def test_many_instances_share_a_device(gpu_pkg, O):
    """MT_MULTI_INSTANCE (ref :649-652): several instances on several host threads, one device."""
    import threading
    fmt, sw, sh, tw, th = "Y8", 160, 90, 320, 180
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=9)
    want = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th).get_frame(src)
    errs = []

    def work():
        try:
            f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
            for _ in range(5):
                assert_planes_equal(f.get_frame(src), want, f.out_dims())
            f.close()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_integer_conversion_ties(gpu_pkg):
    """The device conversion (v_cvt_pk_u8_f32 for 8-bit; med3 + rndne + cvt for 9..16-bit and, in the packed-pair kernels, med3 +
    the 2^23 sum + v_perm: the hook runs every fourth element pair through that form) against
    clamp(r, 0, peak) + lrintf (ref :581-582): every tie k + 0.5, both clamp bounds, values just around
    ties, huge values, infinities, NaN (-> 0 on both sides) and negative zero."""
    for dtype, peak in ((np.uint8, 255.0), (np.uint16, 1023.0), (np.uint16, 4095.0), (np.uint16, 65535.0)):
        ties = np.arange(-3, int(peak) + 3, dtype=np.float64) + 0.5
        if len(ties) > 6000:
            ties = np.concatenate([ties[:3000], ties[-3000:]])
        vals = np.concatenate([
            ties, np.nextafter(ties.astype(np.float32), np.float32(np.inf)),
            np.nextafter(ties.astype(np.float32), np.float32(-np.inf)),
            np.array([0.0, -0.0, 0.49999997, 0.50000006, peak, peak - 0.5, peak + 0.4999, peak + 0.5, peak + 1e6, 3e9, 1e30,
                      -1e-30, -0.4, -0.5, -0.51, -1e9, np.inf, -np.inf, 1e-45, -1e-45]),
            np.random.default_rng(0).uniform(-5, peak + 5, 5000),
        ]).astype(np.float32)
        want = np.rint(np.clip(vals, np.float32(0), np.float32(peak))).astype(dtype)
        for shift in range(4):   # every value through every form of the hook (which form an element takes goes by its index mod 4)
            v, w = np.roll(vals, shift), np.roll(want, shift)
            got = gpu_pkg.debug_convert(v, dtype, peak)
            assert np.array_equal(got, w), (dtype, peak, shift, v[got != w][:8], got[got != w][:8], w[got != w][:8])
        assert gpu_pkg.debug_convert(np.array([np.nan] * 8, np.float32), dtype, peak).tolist() == [0] * 8
    f = np.array([1.5, -2.25, np.inf, 1e-41, -0.0], np.float32)
    assert np.array_equal(gpu_pkg.debug_convert(f, np.float32, 0.0).view(np.uint32), f.view(np.uint32))


@pytest.mark.parametrize("mode", [2, 3, 4, 5, 6, 7, 9, 13, 15], ids=["window", "rows", "window_rg4", "packed_rg4", "packed_rg8", "quasi", "direct", "quad", "full_window"])
@pytest.mark.parametrize("fmt,sw,sh,tw,th", [("Y8", 640, 360, 1280, 720), ("Y16", 333, 211, 666, 422),
                                             ("Y32", 200, 150, 400, 300), ("YUV420P8", 258, 130, 516, 260),
                                             ("Y8", 100, 80, 400, 320)])
def test_kernel_variants_match_oracle(gpu_pkg, O, fmt, sw, sh, tw, th, mode):
    """The A/B kernel variants (row-streamed, other tile heights, packed v_pk_mul/v_pk_add) are bit-exact too."""
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=31337)
    want = of.get_frame(src, threads=4)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    assert f.plan_info().periodic == 1 and f.plan_info().filter_size == 7
    f.set_kernel_mode(mode)
    got = f.get_frame(src)
    assert_planes_equal(got, want, f.out_dims(), what=f"{fmt} mode {mode}")
    f.close()


QUASI_CASES = [
    ("Y8", 320, 180, 480, 270, {}, (3, 3, 2, 2)),                    # 1.5x: drifting phases, affine origins (period 3, step 2)
    ("Y8", 1280 // 2, 720 // 2, 1920 // 2, 1080 // 2, {}, (3, 3, 2, 2)),
    ("Y16", 211, 97, 633, 291, {}, (3, 3, 1, 1)),                    # 3x
    ("Y32", 150, 120, 225, 180, dict(tap=4), (3, 3, 2, 2)),          # fs 9
    ("YUV420P8", 256, 144, 384, 216, {}, (3, 3, 2, 2)),              # chroma table too
    ("Y8", 240, 160, 640, 360, {}, None),                            # 8/3 x 9/4: whatever the plan finds
    ("Y8", 300, 200, 400, 250, dict(tap=2), None),                   # 4/3 x 5/4
    ("Y8", 360, 270, 480, 360, {}, None),                            # 4/3x: exactly periodic with source step 3
    ("Y16", 240, 160, 640, 360, dict(tap=4), None),                  # 8/3 x 9/4, fs 9
]


@pytest.mark.parametrize("mode", [0, 1, 8, 10], ids=["auto", "gather", "quasi_waterfall", "quasi_lane_coefficients"])
@pytest.mark.parametrize("case", QUASI_CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_quasi_periodic_plans(gpu_pkg, O, case, mode):
    """Ratios whose phase classes drift (the reference accumulates positions in float): the plan is not
    periodic, but the window origins are affine, so the quasi-periodic kernel takes the interior."""
    fmt, sw, sh, tw, th, kw, want_q = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=2718)
    want = of.get_frame(src, threads=4)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    info = f.plan_info()
    if want_q is not None:
        assert info.periodic == 0 and info.quasi == 1
        assert (info.quasi_period_x, info.quasi_period_y, info.quasi_step_x, info.quasi_step_y) == want_q
    f.set_kernel_mode(mode)
    got = f.get_frame(src)
    assert_planes_equal(got, want, f.out_dims(), what=f"{fmt} {sw}x{sh}->{tw}x{th} mode {mode}")
    f.close()


# exactly periodic plans outside the register/LDS kernels: (case, expected (px, py, sx, sy) or None)
DIRECT_CASES = [
    ("Y8", 384, 216, 192, 108, {}, (1, 1, 2, 2)),                    # 1/2: fs 13
    ("Y16", 384, 216, 128, 72, {}, (1, 1, 3, 3)),                    # 1/3: fs 20
    ("Y32", 384, 216, 256, 144, {}, (2, 2, 3, 3)),                   # 2/3: fs 10, period 2
    ("Y8", 480, 270, 320, 180, {}, (2, 2, 3, 3)),
    ("Y8", 400, 300, 200, 100, {}, (1, 1, 2, 3)),                    # anisotropic 1/2 x 1/3: fs 20
    ("Y16", 300, 200, 100, 50, dict(tap=4), (1, 1, 3, 4)),           # 1/3 x 1/4: fs 34
    ("Y10", 320, 200, 160, 100, dict(tap=2), (1, 1, 2, 2)),          # fs 9 with a source step, peak 1023
    ("Y8", 1100, 100, 550, 50, {}, (1, 1, 2, 2)),                    # several 256-column tiles, ragged right edge
    ("YUV420P8", 512, 288, 256, 144, {}, (1, 1, 2, 2)),              # chroma table as well
    ("RGBPS", 256, 144, 128, 72, dict(tap=4), (1, 1, 2, 2)),         # fs 17, float planes
    ("Y8", 300, 200, 600, 400, dict(tap=12), (2, 2, 1, 1)),          # fs 25
    ("Y16", 200, 120, 400, 240, dict(tap=16), (2, 2, 1, 1)),         # fs 33
    ("Y32", 160, 100, 640, 400, dict(tap=10), (4, 4, 1, 1)),         # 4x tap 10: fs 21, 16 phases
    ("Y8", 128, 128, 128, 128, dict(src_left=0.5, src_top=0.25), (1, 1, 1, 1)),   # pure shift
    ("Y8", 320, 180, 256, 144, {}, (4, 4, 5, 5)),                    # 4/5: source step 5 -> not covered, gather kernel
]


@pytest.mark.parametrize("mode", [0, 9, 1], ids=["auto", "direct", "gather"])
@pytest.mark.parametrize("case", DIRECT_CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_direct_periodic_kernel(gpu_pkg, O, case, mode):
    """Down-scales and large taps whose plans are exactly periodic (any filter size, source step <= 4) run on
    ewa_direct_kernel (no LDS, row segments fetched in the source format): bit-exact like the rest."""
    fmt, sw, sh, tw, th, kw, want_p = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=1618)
    want = of.get_frame(src, threads=4)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    info = f.plan_info()
    assert info.periodic == 1
    assert (info.period_x, info.period_y, info.step_x, info.step_y) == want_p
    f.set_kernel_mode(mode)
    got = f.get_frame(src)
    assert_planes_equal(got, want, f.out_dims(), what=f"{fmt} {sw}x{sh}->{tw}x{th} mode {mode}")
    f.close()


# Interior forms of ewa_direct_kernel (kernel_direct_impl.inc DirectShape): every filter size the row walk's two loops
# meet -- single-step rows (fs 9..16) at source steps 2..4, equal steps with a last step of nta, nta-1 and nta-2 taps --
# for the three sample sizes, with ragged last chunks (rows % 4 != 0) and phases that start at every byte shift.
WALK_CASES = [
    ("Y8", 388, 218, 194, 109, dict(tap=2)),              # 1/2: fs 9
    ("Y16", 388, 218, 194, 109, {}),                      # fs 13
    ("Y32", 300, 210, 200, 140, {}),                      # 2/3: fs 10, period 2
    ("Y8", 303, 213, 202, 142, dict(tap=4)),              # 2/3: fs 13
    ("Y16", 306, 216, 204, 144, dict(tap=5)),             # 2/3: fs 16
    ("Y8", 402, 222, 134, 74, dict(tap=2)),               # 1/3: fs 14
    ("Y32", 402, 222, 134, 74, dict(tap=2)),
    ("Y8", 404, 220, 101, 55, dict(tap=1)),               # 1/4 tap 1: fs 10 at source step 4
    ("Y16", 400, 300, 100, 100, dict(tap=1)),             # 1/4 x 1/3
    ("Y32", 400, 300, 200, 75, dict(tap=1)),              # 1/2 x 1/4: single-step rows, row step 4
    ("Y8", 388, 218, 194, 109, dict(tap=4)),              # 1/2: fs 17 = 9 + 8
    ("Y32", 388, 218, 194, 109, dict(tap=5)),             # fs 21 = 11 + 10
    ("Y16", 402, 222, 134, 74, dict(tap=4)),              # 1/3: fs 26 = 13 + 13
    ("Y8", 402, 222, 134, 74, dict(tap=6)),               # fs 39 = 13 * 3
    ("Y8", 404, 220, 101, 55, dict(tap=4)),               # 1/4: fs 34 = 12 + 12 + 10
    ("Y16", 404, 220, 101, 55, dict(tap=5)),              # fs 42 = 14 * 3
    ("Y8", 404, 220, 101, 55, dict(tap=6)),               # fs 50: outside the walk, shape 0
    ("Y8", 150, 110, 300, 220, dict(tap=9)),              # 2x tap 9: fs 19 = 10 + 9, source step 1
    ("Y32", 150, 110, 300, 220, dict(tap=11)),            # fs 23 = 12 + 11
    ("YUV420P8", 516, 292, 258, 146, {}),                 # luma and chroma tables
    ("Y8", 303, 213, 202, 142, dict(tap=2)),              # 2/3 with tap 2: fs 7 -- the ladders' lowest rungs (forced shape only)
    ("Y16", 363, 273, 484, 364, {}),                      # 4/3x: fs 7 at source step 3
]


@pytest.mark.parametrize("shape", [0, 2, 3], ids=["per_chain", "walk", "walk_wide"])
@pytest.mark.parametrize("case", WALK_CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}_{c[5].get('tap', 3)}")
def test_direct_kernel_interior_forms(gpu_pkg, O, case, shape):
    fmt, sw, sh, tw, th, kw = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    n = 3
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=4242 + i) for i in range(n)]
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    assert f.plan_info().periodic == 1
    gpu_pkg.set_direct_shape(shape)
    try:
        f.set_kernel_mode(9)
        got = [f.get_frame(fr) for fr in frames]
    finally:
        gpu_pkg.set_direct_shape(-1)
    assert f.last_kernel(0) == "ewa_direct_kernel"
    assert gpu_pkg.last_direct_shape() in ((0,) if shape == 0 else (0, 2) if shape == 2 else (0, 2, 3))
    for i in range(n):
        assert_planes_equal(got[i], of.get_frame(frames[i], threads=4), f.out_dims(), what=f"{fmt} {sw}x{sh}->{tw}x{th} shape {shape} frame {i}")
    f.close()


def test_direct_kernel_wide_walk_on_a_batch(gpu_pkg, O):
    """The automatic choice takes the 8-column walk for 8-bit batches that fill the device: 4K -> 1080p, 6 frames; every frame
    against the oracle's crc of the same frame."""
    torch = pytest.importorskip("torch")
    import zlib
    fmt, sw, sh, tw, th, n = "Y8", 3840, 2160, 1920, 1080, 8
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=77 + i) for i in range(n)]
    src = to_device(torch.from_numpy(np.stack([np.ascontiguousarray(fr[0][:sh, :sw]) for fr in frames]))).contiguous()
    dst = torch.zeros((n, th, tw), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    f.process_device([src.data_ptr()], [sw], [sw * sh], [dst.data_ptr()], [tw], [tw * th], n, stream=stream.cuda_stream)
    stream.synchronize()
    assert gpu_pkg.last_direct_shape() == 3
    out = to_host(dst).numpy()
    for i in (0, n - 1):
        want = of.get_frame(frames[i], threads=8)
        assert zlib.crc32(out[i].tobytes()) == zlib.crc32(np.ascontiguousarray(want[0][:th, :tw]).tobytes()), f"frame {i}"
    f.close()


@pytest.mark.parametrize("sw", [200, 202], ids=["pitch200", "pitch202_not_multiple_of_4"])
def test_direct_kernel_tight_pitch_device_batch(gpu_pkg, O, sw):
    """Device entry with pitch == row size (no padding at all) and a batch of frames: the direct kernel fetches
    aligned dwords inside the plane only; a pitch that is not a multiple of 4 falls back to the gather kernel."""
    torch = pytest.importorskip("torch")
    fmt, sh, tw, th = "Y8", 120, sw // 2, 60
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    assert f.plan_info().periodic == 1 and f.plan_info().step_x == 2
    n = 5
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=900 + i) for i in range(n)]
    src = to_device(torch.from_numpy(np.stack([np.ascontiguousarray(fr[0][:sh, :sw]) for fr in frames]))).contiguous()
    dst = torch.zeros((n, th, tw), dtype=torch.uint8, device="cuda")
    assert src.stride(1) == sw and dst.stride(1) == tw                    # pitch == row size
    stream = torch.cuda.current_stream()
    f.process_device([src.data_ptr()], [sw], [sw * sh], [dst.data_ptr()], [tw], [tw * th], n, stream=stream.cuda_stream)
    stream.synchronize()
    out = to_host(dst).numpy()
    for i in range(n):
        want = of.get_frame(frames[i], threads=4)
        assert np.array_equal(out[i], want[0][:th, :tw]), f"frame {i}"
    f.close()


STRIP_CASES = [
    ("Y8", 640, 360, 1280, 720, {}),                  # C1 shape, window kernel interior
    ("YUV420P16", 160, 96, 320, 192, dict(tap=8, cplace="mpeg2")),   # C3 in miniature, row-streamed interior
    ("RGBPS", 200, 100, 400, 200, dict(tap=4, blur=0.98)),            # C4 in miniature
    ("Y8", 360, 270, 480, 360, {}),                   # 4/3x: exact quasi kernel interior, source step 3
    ("Y16", 384, 216, 128, 72, {}),                   # 1/3 down-scale, direct interior
    ("Y8", 300, 200, 600, 400, dict(tap=12)),         # fs 25
    ("Y8", 131, 77, 262, 154, {}),                    # ragged sizes
    ("Y8", 133, 79, 399, 237, dict(tap=2)),           # 3x, may drift -> whatever the plan finds
]


@pytest.mark.parametrize("strips", [1, 2, 3, 4, 0], ids=["strips", "row_strips_only", "strip_kernel", "round5_forms", "gather_border"])
@pytest.mark.parametrize("case", STRIP_CASES, ids=_id)
def test_border_strips_and_gather_border_agree_with_oracle(gpu_pkg, O, case, strips):
    """The border frame of exactly periodic plans runs as row/column strips on ewa_direct_kernel (+ corners on the
    gather kernel) by default; the all-gather border stays available.  Both are bit-exact."""
    fmt, sw, sh, tw, th, kw = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=777)
    want = of.get_frame(src, threads=4)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    f.set_border_strips(strips)
    got = f.get_frame(src)
    assert_planes_equal(got, want, f.out_dims(), what=f"{_id(case)} strips={strips}")
    f.close()


def test_buffer_range_check_premise(gpu_pkg):
    """kernel_direct.hip bounds its whole-segment fetches with the buffer descriptor while the row offset travels in the
    instruction's scalar offset: that is only safe if the hardware range check covers the scalar offset (it does on
    gfx950 -- profiles/probes/soffset_probe.hip -- although LLVM documents the opposite).  The library probes this per
    device and drops the direct kernel otherwise; this test pins the premise itself."""
    assert gpu_pkg.lib().jinc_debug_buffer_range_check(0) == 1


@pytest.mark.parametrize("fmt,slacks", [("Y8", (0, 1, 2, 3, 4, 2048)), ("Y16", (0, 2, 4, 6, 2050))], ids=["u8", "u16"])
def test_plane_ending_on_a_page_boundary(gpu_pkg, O, fmt, slacks):
    """ewa_direct_kernel fetches naturally aligned dwords only, bounded by the dword that holds the plane's last
    sample: planes that end on (or within 3 bytes of) a 4 KiB page boundary, at every base misalignment the sample
    size allows, are served without touching the next page and without losing the last samples."""
    torch = pytest.importorskip("torch")
    sw, sh, tw, th = 200, 120, 100, 60
    ofmt = O.FORMATS[fmt]
    sb = ofmt.sample_bytes
    of = O.OracleFilter(ofmt, sw, sh, tw, th)
    frame = O.lcg_frame(ofmt, sw, sh, seed=4096)
    want = of.get_frame(frame, threads=4)[0][:th, :tw]
    plane = np.ascontiguousarray(frame[0][:sh, :sw])
    nbytes = sw * sh * sb
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    assert f.interior_kernel(0) == "ewa_direct_kernel"
    pool = torch.zeros(4 * 4096 + nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.zeros((th, tw * sb), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    for slack in slacks:   # bytes between the plane's end and the next page boundary
        off = (-(pool.data_ptr() + nbytes + slack)) % 4096
        view = pool[off:off + nbytes]
        assert (view.data_ptr() + nbytes + slack) % 4096 == 0 and view.data_ptr() % sb == 0
        view.copy_(to_device(torch.from_numpy(plane.view(np.uint8).reshape(-1))))
        dst.zero_()
        f.process_device([view.data_ptr()], [sw * sb], [0], [dst.data_ptr()], [tw * sb], [0], 1, stream=stream.cuda_stream)
        stream.synchronize()
        got = to_host(dst).numpy().view(ofmt.dtype)
        assert np.array_equal(got, want), f"slack {slack} (base % 4 = {view.data_ptr() % 4})"
    f.close()


def _random_case(rng):
    fmts = ["Y8", "Y10", "Y16", "Y32", "YUV420P8", "YUV422P16", "YUV444P8", "YUV411P8", "YUV420PS", "RGBP8", "RGBAP16",
            "YUVA420P8", "RGBPS"]
    fmt = fmts[rng.integers(len(fmts))]
    sw = int(rng.integers(40, 140)) & ~3
    sh = int(rng.integers(40, 110)) & ~1
    kind = rng.integers(5)
    if kind == 0:      # integer up-scale
        r = int(rng.integers(2, 5)); tw, th = sw * r, sh * r
    elif kind == 1:    # 1.5x / 3x style
        tw, th = sw * 3 // 2 // 4 * 4, sh * 3 // 2 // 2 * 2
    elif kind == 2:    # arbitrary up-scale
        tw = int(sw * rng.uniform(1.0, 3.0)) // 4 * 4; th = int(sh * rng.uniform(1.0, 3.0)) // 2 * 2
    elif kind == 3:    # mild down-scale
        tw = max(16, int(sw * rng.uniform(0.5, 1.0)) // 4 * 4); th = max(16, int(sh * rng.uniform(0.5, 1.0)) // 2 * 2)
    else:              # anisotropic
        tw = int(sw * rng.uniform(0.7, 2.5)) // 4 * 4; th = int(sh * rng.uniform(0.7, 2.5)) // 2 * 2
    kw = dict(tap=int(rng.integers(1, 9)))
    if rng.random() < 0.4:
        kw.update(quant_x=int(rng.integers(1, 257)), quant_y=int(rng.integers(1, 257)))
    if rng.random() < 0.4:
        kw["blur"] = float(np.round(rng.uniform(0.8, 1.25), 3))
    if rng.random() < 0.5:
        kw.update(src_left=float(np.round(rng.uniform(-3, 6), 2)), src_top=float(np.round(rng.uniform(-3, 6), 2)),
                  src_width=float(np.round(sw - rng.uniform(0, 10), 2)), src_height=float(np.round(sh - rng.uniform(0, 10), 2)))
    if "420" in fmt:
        kw["cplace"] = ["mpeg2", "mpeg1", "topleft"][rng.integers(3)]
    elif fmt.startswith("YUV4") and ("422" in fmt or "411" in fmt):
        kw["cplace"] = ["mpeg2", "mpeg1"][rng.integers(2)]
    return fmt, sw, sh, tw, th, kw


def _random_case_v2(rng):
    """Second generation of the sweep: larger frames (several tiles), exact down-scales (direct kernel interior),
    taps up to 16, crops mostly absent so that the plans keep their structure."""
    fmts = ["Y8", "Y10", "Y16", "Y32", "YUV420P8", "YUV420P16", "YUV422P16", "YUV444P8", "RGBP8", "RGBPS", "YUVA420P8"]
    fmt = fmts[rng.integers(len(fmts))]
    kind = rng.integers(6)
    sw = int(rng.integers(24, 120)) * 4
    sh = int(rng.integers(20, 70)) * 4
    kw = dict(tap=int(rng.integers(1, 9)))
    if kind == 0:      # exact down-scale 1/2, 1/3, 1/4, 2/3, 3/4
        num, den = [(1, 2), (1, 3), (1, 4), (2, 3), (3, 4)][rng.integers(5)]
        sw, sh = sw // (4 * den) * 4 * den, sh // (4 * den) * 4 * den
        tw, th = sw * num // den, sh * num // den
        kw["tap"] = int(rng.integers(1, 5))
    elif kind == 1:    # integer up-scale with a large tap
        r = int(rng.integers(2, 4)); tw, th = sw * r, sh * r
        kw["tap"] = int(rng.integers(9, 17))
        sw, sh, tw, th = sw // 2 // 4 * 4, sh // 2 // 4 * 4, sw // 2 // 4 * 4 * r, sh // 2 // 4 * 4 * r
    elif kind == 2:    # drifting ratios 3/2, 3, 9/4 x 8/3, 5/4
        num, den = [(3, 2), (3, 1), (5, 4), (5, 2)][rng.integers(4)]
        sw, sh = sw // (4 * den) * 4 * den, sh // (4 * den) * 4 * den
        tw, th = sw * num // den, sh * num // den
        kw["tap"] = int(rng.integers(2, 5))
    elif kind == 3:    # anisotropic mix of exact ratios
        rx = [(1, 2), (2, 1), (3, 2), (4, 3), (1, 1)][rng.integers(5)]
        ry = [(1, 3), (2, 1), (3, 1), (2, 3), (1, 1)][rng.integers(5)]
        sw, sh = sw // (4 * rx[1]) * 4 * rx[1], sh // (4 * ry[1]) * 4 * ry[1]
        tw, th = sw * rx[0] // rx[1], sh * ry[0] // ry[1]
        kw["tap"] = int(rng.integers(1, 6))
    elif kind == 4:    # arbitrary ratio (no structure)
        tw = int(sw * rng.uniform(0.6, 2.2)) // 4 * 4; th = int(sh * rng.uniform(0.6, 2.2)) // 4 * 4
    else:              # 2x with every tap
        tw, th = sw * 2, sh * 2
        kw["tap"] = int(rng.integers(1, 17))
    tw, th = max(16, tw), max(16, th)
    if rng.random() < 0.3:
        kw.update(quant_x=int(rng.integers(1, 257)), quant_y=int(rng.integers(1, 257)))
    if rng.random() < 0.3:
        kw["blur"] = float(np.round(rng.uniform(0.8, 1.25), 3))
    if rng.random() < 0.15:
        kw.update(src_left=float(np.round(rng.uniform(-3, 6), 2)), src_top=float(np.round(rng.uniform(-3, 6), 2)))
    if "420" in fmt:
        kw["cplace"] = ["mpeg2", "mpeg1", "topleft"][rng.integers(3)]
    elif "422" in fmt:
        kw["cplace"] = ["mpeg2", "mpeg1"][rng.integers(2)]
    return fmt, sw, sh, tw, th, kw


def _random_case_v3(rng):
    """Third generation: extreme geometry -- tiny frames, thin strips, 8x ratios, big taps on small sources (many are
    rejected like the reference's out-of-bounds cases), odd chroma sizes."""
    fmts = ["Y8", "Y16", "Y32", "YUV420P8", "YUV444P16", "RGBPS"]
    fmt = fmts[rng.integers(len(fmts))]
    even = 2 if "420" in fmt else 1
    sw = int(rng.integers(8, 72)) // even * even
    sh = int(rng.integers(8, 72)) // even * even
    shape = rng.integers(4)
    if shape == 0:      # thin strip
        sh = [8, 10, 12, 16][rng.integers(4)]
    elif shape == 1:    # tall strip
        sw = [8, 10, 12, 16][rng.integers(4)]
    tw = max(4, int(sw * [0.25, 0.5, 1.0, 2.0, 4.0, 8.0, rng.uniform(0.3, 6.0)][rng.integers(7)])) // even * even
    th = max(4, int(sh * [0.25, 0.5, 1.0, 2.0, 4.0, 8.0, rng.uniform(0.3, 6.0)][rng.integers(7)])) // even * even
    kw = dict(tap=int([1, 2, 3, 3, 4, 8, 16][rng.integers(7)]))
    if rng.random() < 0.3:
        kw.update(quant_x=int([1, 2, 16, 256][rng.integers(4)]), quant_y=int([1, 3, 64, 256][rng.integers(4)]))
    if rng.random() < 0.2:
        kw["blur"] = float(np.round(rng.uniform(0.6, 1.6), 3))
    if "420" in fmt:
        kw["cplace"] = ["mpeg2", "mpeg1", "topleft"][rng.integers(3)]
    return fmt, sw, sh, tw, th, kw


# JINC_SWEEP_SEEDS=N widens the sweeps for soak runs (default: 3 x 48 cases, a few seconds)
_SWEEP = int(os.environ.get("JINC_SWEEP_SEEDS", "48"))


@pytest.mark.parametrize("seed", range(_SWEEP))
@pytest.mark.parametrize("gen", [1, 2, 3], ids=["small", "structured", "extreme"])
def test_randomised_arguments(gpu_pkg, O, seed, gen):
    """Seeded sweep over formats, ratios, taps, quantisation, blur, crops and chroma siting: the HIP path
    (automatic kernel choice) must equal the oracle bit for bit whatever structure the plan has."""
    rng = np.random.default_rng(1000 * gen + seed)
    fmt, sw, sh, tw, th, kw = {1: _random_case, 2: _random_case_v2, 3: _random_case_v3}[gen](rng)
    try:
        of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    except Exception:
        pytest.skip("oracle rejects this geometry")
    try:
        f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    except gpu_pkg.JincError as e:
        assert "smaller than the filter footprint" in str(e)   # the reference reads out of bounds there
        return
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=seed)
    want = of.get_frame(src, threads=4)
    got = f.get_frame(src)
    what = f"gen {gen} seed {seed}: {fmt} {sw}x{sh}->{tw}x{th} {kw}"
    assert_planes_equal(got, want, f.out_dims(), what=what)
    # Plans with affine window origins: the quasi-periodic kernel in each of its forms, whatever the automatic choice was
    # (calls of fewer than 3e6 samples -- every frame of this sweep -- go to the gather kernel since round 2): 7 = per-lane
    # coefficient registers / exact, 8 = waterfall over sets in SGPRs, 10 = per-row look-up + per-lane registers; each also
    # with a tile's phases split over several workgroups (what small calls do in automatic mode).
    # exactly periodic plans: the strip kernels over the border frame, which calls of this size no longer take by themselves
    if any(f.plan_info(t).periodic for t in range(f.num_tables)):
        for strips in (1, 3, 4):   # the round-4 strip kernels; ewa_strip_kernel (round 5) where the plan has it; 4: round 5's other
            f.set_border_strips(strips)   # forms where configured (column pairs, rows on the pair kernel; edge columns under mode 13 below)
            assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=what + f" border strips {strips}")
        f.set_border_strips(4)
        f.set_kernel_mode(13)         # the quad forms with the border columns in their edge tiles (integer planes at 2x with tap 3 / 4)
        with gpu_pkg.knobs(quad2x8=1):
            assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=what + " border strips 4, quad forms")
        f.set_kernel_mode(0)
        f.set_border_strips(-1)
        # the periodic family on the trimmed support (integer planes) in each of its forms -- window, rows, quad -- and on the
        # reference's full window (15)
        for mode in (2, 3, 13, 15):
            f.set_kernel_mode(mode)
            assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=what + f" kernel mode {mode}")
        f.set_kernel_mode(0)
    if any(f.plan_info(t).quasi for t in range(f.num_tables)):
        for mode in (7, 8, 10, 14):  # 14: the direct kernel's runs form wherever the plan has runs (fs >= 9)
            f.set_kernel_mode(mode)
            assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=what + f" kernel mode {mode}")
    f.close()


@pytest.mark.parametrize("register", [False, True], ids=["pageable", "registered"])
def test_frame_pipeline(gpu_pkg, O, register, pooling_host):
    """Look-ahead pipeline: several frames in flight per instance, collected out of order; every frame must
    equal the synchronous GetFrame result (and the oracle)."""
    fmt, sw, sh, tw, th = "YUV420P8", 200, 120, 400, 240
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_pipeline(3, register)
    n = 10
    srcs = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=500 + k) for k in range(n)]
    dsts = [[gpu_pkg.alloc_plane(w, h, np.uint8) for (w, h) in f.out_dims()] for _ in range(n)]
    if register:   # planes the device maps come from mappings of their own (conftest.fresh_mapping)
        srcs = [fresh_copies(s) for s in srcs]
        dsts = [fresh_planes(f.out_dims(), np.uint8) for _ in range(n)]
    tickets = [None] * n
    for k in range(n):
        tickets[k] = f.submit(srcs[k], dsts[k])
        if k >= 2:
            f.wait(tickets[k - 2])
    for k in (n - 1, n - 2):      # out of order on purpose
        f.wait(tickets[k])
    f.wait(tickets[0])            # waiting twice is harmless
    for k in range(n):
        assert_planes_equal(dsts[k], of.get_frame(srcs[k], threads=2), f.out_dims(), what=f"frame {k}")
    # the synchronous entry point still works afterwards and drains the pipeline first
    t = f.submit(srcs[0], dsts[0])
    got = f.get_frame(srcs[1])
    assert_planes_equal(got, of.get_frame(srcs[1]), f.out_dims())
    f.wait(t)
    f.close()


def test_device_entry_rejects_bad_layouts(gpu_pkg):
    """Misaligned pointers / pitches are refused with a message instead of faulting on the device."""
    torch = pytest.importorskip("torch")
    f = gpu_pkg.Filter(gpu_pkg.FORMATS["Y16"], 64, 48, 128, 96, device=0)
    src = torch.zeros((48, 64), dtype=torch.int16, device="cuda")
    dst = torch.zeros((96, 128), dtype=torch.int16, device="cuda")
    ok = dict(src_ptrs=[src.data_ptr()], src_pitches=[128], src_strides=[0], dst_ptrs=[dst.data_ptr()], dst_pitches=[256],
              dst_strides=[0], nframes=1)
    f.process_device(**ok)
    for bad in (dict(src_ptrs=[src.data_ptr() + 1]), dict(dst_pitches=[255]), dict(src_pitches=[126]), dict(nframes=0),
                dict(dst_ptrs=[0])):
        with pytest.raises(gpu_pkg.JincError):
            f.process_device(**{**ok, **bad})
    torch.cuda.synchronize()
    f.close()


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 15],
                         ids=["auto", "gather", "window", "rows", "window_rg4", "packed_rg4", "packed_rg8", "quasi_exact", "quasi_waterfall", "direct",
                              "quasi_lane_coefficients", "quad", "full_window"])
def test_every_kernel_reproduces_the_reference_crc_at_full_size(gpu_pkg, O, mode):
    """C2 at full size through every kernel family: the crc32 of the reference's own opt=0 output."""
    k = next(x for x in KAT["outputs"] if x["name"] == "C2")
    src = O.lcg_frame(O.FORMATS[k["format"]], *k["src"])
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[k["format"]], k["src"][0], k["src"][1], k["dst"][0], k["dst"][1], device=0, **k["args"])
    f.set_kernel_mode(mode)
    # both forms of the border frame: one call of this size takes the gather kernel over the frame by itself (csrc/dispatch.cpp
    # Rules::kStripBorderMinTaps); the strip kernels are what batches run
    for strips in (-1, 1):
        f.set_border_strips(strips)
        got = f.get_frame(src)
        crc = 0
        for p, (w, h) in zip(got, f.out_dims()):
            crc = zlib.crc32(np.ascontiguousarray(p[:h, :w]).tobytes(), crc)
        assert f"{crc & 0xFFFFFFFF:08x}" == k["crc32"], strips
    f.close()


def test_create_free_cycles_do_not_leak_device_memory(gpu_pkg, O):
    """Plans, lane-major coefficient copies, pipeline slots, streams and events are all released by jinc_filter_free:
    free device memory returns to where it was after 60 create / use / free cycles over every kind of plan."""
    torch = pytest.importorskip("torch")
    kinds = [("Y8", 192, 108, 384, 216, {}), ("Y8", 192, 108, 288, 162, {}), ("YUV420P8", 256, 144, 128, 72, {}),
             ("Y16", 160, 90, 219, 123, {}), ("Y8", 96, 64, 192, 128, dict(tap=12))]
    srcs = [O.lcg_frame(O.FORMATS[k[0]], k[1], k[2]) for k in kinds]

    def cycle(n):
        for i in range(n):
            fmt, sw, sh, tw, th, kw = kinds[i % len(kinds)]
            f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
            f.get_frame(srcs[i % len(kinds)])
            f.close()

    cycle(10)                                  # warm up allocator pools of the runtime
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cycle(60)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 60 cycles"


QUAD_CASES = [
    ("Y8", 192, 108, 384, 216, dict(tap=3)),
    ("Y8", 1000, 300, 2000, 600, dict(tap=3)),          # several tiles in x, partial last tile
    ("Y16", 320, 180, 640, 360, dict(tap=3)),
    ("Y10", 320, 180, 640, 360, dict(tap=3)),
    ("Y32", 320, 180, 640, 360, dict(tap=3)),
    ("YUV420P8", 256, 144, 512, 288, dict(tap=3, cplace="mpeg1")),   # luma and chroma tables both 2x
    ("YUV420P8", 256, 144, 512, 288, dict(tap=3, cplace="mpeg2")),   # chroma sited as MPEG-2: 6 rows x 7 columns, seven taps per kernel row
    ("YUV420P16", 320, 180, 640, 360, dict(tap=3)),                  # (the default siting)
    ("YUV422P10", 256, 144, 512, 288, dict(tap=3)),                  # 4:2:2: the chroma table is 2x in both axes here too
    ("YUV420PS", 256, 144, 512, 288, dict(tap=3)),                   # float chroma on the 6 x 7 support behind the finite scan
    ("RGBPS", 160, 100, 320, 200, dict(tap=3, blur=0.95)),
    # fs 9 (tap 4): ewa_periodic_quad9_kernel
    ("Y8", 192, 108, 384, 216, dict(tap=4)),
    ("Y16", 700, 200, 1400, 400, dict(tap=4)),
    ("RGBPS", 320, 180, 640, 360, dict(tap=4, blur=0.98)),       # C4's arguments
    ("YUV420P10", 256, 144, 512, 288, dict(tap=4, cplace="topleft")),
]


@pytest.mark.parametrize("case", QUAD_CASES, ids=_id)
@pytest.mark.parametrize("frames", [1, 5])
def test_quad_form_of_the_periodic_kernel(gpu_pkg, O, case, frames):
    """ewa_periodic_quad_kernel (kernel mode 13): 2x up-scales, a lane computes the 2 x 2 pixels of a period from one window
    with packed multiplies / adds; bit-exact against the oracle, single frames (half-height tiles) and batches."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, kw = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    f.set_kernel_mode(13)
    # integer planes whose support trims to 6 x 6 take the two-periods-per-lane form; the others the one-period forms
    quad_names = {6: ("ewa_periodic_quad2_kernel",), 8: ("ewa_periodic_quad8_kernel",)}.get(f.periodic_support(0), ("ewa_periodic_quad_kernel",))
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=6100 + k) for k in range(frames)]
    if frames == 1:
        got = f.get_frame(srcs[0])
        assert f.last_kernel(0) in quad_names, f.last_kernel(0)
        assert_planes_equal(got, of.get_frame(srcs[0], threads=4), f.out_dims(), what=_id(case))
    else:
        from test_framelane_pair import _run_batch
        got = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 13)
        assert f.last_kernel(0) in quad_names, f.last_kernel(0)
        for k in range(frames):
            assert_planes_equal(got[k], of.get_frame(srcs[k], threads=4), f.out_dims(), what=_id(case) + f" frame {k}")
    f.close()


@pytest.mark.parametrize("fmt,sw,sh,tw,th,kw,full,trimmed", [
    ("Y8", 1920, 1080, 3840, 2160, dict(tap=3), 7, 6),            # C2: the first kernel row and column of all four phase sets are 0.0f
    ("YUV420P16", 640, 360, 1280, 720, dict(tap=8), 17, 16),      # C3's geometry in small
    ("Y16", 320, 180, 640, 360, dict(tap=4), 9, 8),
    ("Y32", 320, 180, 640, 360, dict(tap=3), 7, 6),               # float samples: frame by frame, where every sample is finite
    ("RGBPS", 320, 180, 640, 360, dict(tap=4, blur=0.98), 9, 8),
], ids=["C2_u8", "tap8_u16", "tap4_u16", "tap3_f32", "C4_f32_small"])
def test_trimmed_support_of_the_periodic_kernels(gpu_pkg, O, fmt, sw, sh, tw, th, kw, full, trimmed):
    """The periodic kernels run on the bounding box of the phase sets' non-zero coefficients (float planes: the frames a scan found
    finite, in calls large enough to pay for the scan -- forced kernel modes take it at any size).  Kernel mode 15 switches the
    trimming off.  All are the oracle's result."""
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    assert f.plan_info(0).filter_size == full
    assert f.periodic_support(0) == trimmed
    if fmt == "YUV420P16":  # chroma sited as MPEG-2 (an eighth of a sample to the left): all 17 columns, but 16 kernel rows
        assert f.periodic_support(1) == 17
        assert f.periodic_taps(1, rows_kernel=True) <= 17 * 16 and f.periodic_taps(0, rows_kernel=True) <= 16 * 16
    f.set_kernel_mode(15)
    assert f.periodic_support(0) == full
    f.set_kernel_mode(0)
    if sw <= 640:
        src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=99)
        want = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw)).get_frame(src, threads=8)
        for mode in (0, 3, 13, 15):
            f.set_kernel_mode(mode)
            assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=f"{fmt} mode {mode}")
    f.close()


def test_zero_coefficient_taps_meet_extreme_integer_samples(gpu_pkg, O):
    """The trimmed support leaves out taps whose coefficient is 0.0f.  All-zero, all-peak and alternating planes (sums that
    are exactly 0, exactly peak, and cancel) must come out as the oracle computes them with every tap in place."""
    for fmt, peak in (("Y8", 255), ("Y16", 65535), ("Y10", 1023)):
        sw, sh, tw, th = 96, 64, 192, 128
        of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
        f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
        assert f.periodic_support(0) == 6
        base = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=5)
        for kind in ("zeros", "peak", "checker", "columns"):
            src = [p.copy() for p in base]
            a = src[0]
            if kind == "zeros":
                a[:] = 0
            elif kind == "peak":
                a[:] = peak
            elif kind == "checker":
                a[:] = 0
                a[::2, ::2] = peak
                a[1::2, 1::2] = peak
            else:
                a[:] = 0
                a[:, ::3] = peak
            want = of.get_frame(src, threads=4)
            for mode in (0, 2, 13, 15):
                f.set_kernel_mode(mode)
                assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=f"{fmt} {kind} mode {mode}")
        f.close()


@pytest.mark.parametrize("fmt,sw,sh,tw,th,kw", [("Y32", 320, 180, 640, 360, dict(tap=3)), ("RGBPS", 200, 120, 400, 240, dict(tap=4, blur=0.98)),
                                                ("YUV420PS", 256, 144, 512, 288, dict(tap=3))], ids=["Y32_tap3", "RGBPS_tap4", "YUV420PS_tap3"])
@pytest.mark.parametrize("mode", [0, 2, 3, 13], ids=["auto", "window", "rows", "quad"])
def test_float_planes_take_the_trimmed_support_only_where_every_sample_is_finite(gpu_pkg, O, fmt, sw, sh, tw, th, kw, mode):
    """A tap with coefficient 0.0f may be left out only if its sample is finite (0 x inf and 0 x NaN are NaN, and the reference
    multiplies every tap, ref :570-579).  A batch in which some frames hold an infinity or a NaN -- placed where only
    zero-coefficient taps of some outputs meet it -- and the others are finite: frame by frame the oracle's bits, NaN
    footprints included.  (Forced kernel modes take the scan + two-launch path at any call size; mode 0 takes it from 1e9 taps per
    plane and call on -- bench.py's float configurations -- and the plain full window below, as here.)"""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    frames = 24 if fmt == "Y32" else 6
    rng = np.random.default_rng(17)
    srcs = []
    for k in range(frames):
        src = O.lcg_frame(ofmt, sw, sh, seed=7300 + k)
        for p in src:
            p[:] = (rng.standard_normal(p.shape) * 0.7).astype(np.float32)
        if k % 3 == 1:      # a frame with non-finite samples in plane 0 (and, every other time, in the last plane)
            src[0][sh // 2, sw // 3] = np.inf
            src[0][5, 7] = np.nan
            src[0][sh - 3, sw - 2] = -np.inf
            if k % 2 and len(src) > 1:
                src[-1][3, 3] = np.nan
        srcs.append(src)
    f.set_kernel_mode(mode)
    got = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, mode)
    for k in range(frames):
        want = of.get_frame(srcs[k], threads=8)
        for i, (w, h) in enumerate(f.out_dims()):
            a, b = got[k][i][:h, :w], want[i][:h, :w]
            na, nb = np.isnan(a), np.isnan(b)
            assert np.array_equal(na, nb), f"frame {k} plane {i}: NaN footprint differs ({int(na.sum())} vs {int(nb.sum())})"
            assert np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32)), f"frame {k} plane {i}: bits differ"
        if k % 3 == 1:
            assert np.isnan(want[0]).any()
    f.close()


@pytest.mark.parametrize("tap", [3, 4, 8])
@pytest.mark.parametrize("mode", [2, 3, 13], ids=["window", "rows", "quad"])
def test_one_non_finite_sample_anywhere_in_a_float_plane(gpu_pkg, O, tap, mode):
    """The trimmed launch of a float plane is its own finite-sample scan: it flags a frame in whose tiles it stages an infinity or
    a NaN, a small scan covers the source samples no tile stages (the rim only zero-coefficient taps and border pixels reach), and
    flagged frames are computed again on the full window.  One frame per position of a single infinity -- every corner, the
    first and last rows and columns and their neighbours up to a filter size in, tile seams, the middle -- between finite
    frames: NaN footprints and bits as the oracle's, frame by frame."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th = "Y32", 150, 70, 300, 140
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, tap=tap)
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, tap=tap)
    fs = f.plan_info(0).filter_size
    edge = list(range(0, fs + 2))
    xs = sorted(set(edge + [sw - 1 - e for e in edge] + [63, 64, 65, 127, 128, 129, sw // 2]))
    ys = sorted(set(edge + [sh - 1 - e for e in edge] + [sh // 2]))
    spots = [(ys[i % len(ys)], x) for i, x in enumerate(xs)] + [(y, xs[(3 * i) % len(xs)]) for i, y in enumerate(ys)] + \
            [(0, 0), (0, sw - 1), (sh - 1, 0), (sh - 1, sw - 1)]
    rng = np.random.default_rng(23)
    srcs, marks = [], []
    for k, (y, x) in enumerate(spots):
        src = [(rng.standard_normal((sh, sw)) * 0.7).astype(np.float32)]
        if k % 4 != 3:     # (every fourth frame stays finite)
            src[0][y, x] = (np.inf, -np.inf, np.nan)[k % 3]
            marks.append(k)
        srcs.append(src)
    got = _run_batch(torch, gpu_pkg, f, gfmt, srcs, len(srcs), mode)
    for k in range(len(srcs)):
        want = of.get_frame(srcs[k], threads=8)
        w, h = f.out_dims()[0]
        a, b = got[k][0][:h, :w], want[0][:h, :w]
        na, nb = np.isnan(a), np.isnan(b)
        assert np.array_equal(na, nb), f"frame {k}, sample {spots[k]}: NaN footprint differs ({int(na.sum())} vs {int(nb.sum())})"
        assert np.array_equal(a[~na].view(np.uint32), b[~nb].view(np.uint32)), f"frame {k}, sample {spots[k]}: bits differ"
    assert any(np.isnan(of.get_frame(srcs[k], threads=8)[0]).any() for k in marks[:3])
    f.close()


@pytest.mark.parametrize("fmt,sw,sh,tw,th,kw", [("Y8", 160, 90, 320, 180, {}), ("Y16", 100, 60, 200, 120, dict(tap=4)), ("Y32", 96, 64, 192, 128, {}),
                                                ("YUV420P8", 128, 96, 256, 192, dict(cplace="mpeg1")), ("Y8", 192, 108, 96, 54, {}),
                                                ("Y8", 120, 90, 160, 120, {})],
                         ids=["Y8_2x", "Y16_2x_tap4", "Y32_2x", "YUV420P8_2x", "Y8_half", "Y8_4to3"])
@pytest.mark.parametrize("frames", [16, 21, 64, 131])
def test_border_columns_of_periodic_plans_in_batches(gpu_pkg, O, fmt, sw, sh, tw, th, kw, frames):
    """Batches of >= 16 frames run the border columns (and corners) of exactly periodic plans on the frame-lane kernel -- lanes
    = frames, a border pixel's private coefficient set a scalar load (up to 32 frames: its sub-group form, frames x output rows)
    -- instead of the column-strip + corner kernels (which kernel mode 3 keeps).  Every frame of the batch against the oracle; the
    two forms against each other on all of them."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    assert f.plan_info(0).periodic == 1
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=8800 + k) for k in range(frames)]
    f.set_border_strips(1)          # small planes would take one gather launch over the border frame by themselves
    auto = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
    strips = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 3)
    for k in range(frames):
        assert_planes_equal(auto[k], strips[k], f.out_dims(), what=f"{fmt} frame {k}: frame-lane border columns vs column strips")
        if k in (0, min(63, frames // 2), frames - 1):
            assert_planes_equal(auto[k], of.get_frame(srcs[k], threads=4), f.out_dims(), what=f"{fmt} frame {k} vs oracle")
    f.close()


@pytest.mark.parametrize("fmt", ["Y8", "Y16"])
def test_jinc64_at_2x_in_batches_takes_two_periods_per_lane(gpu_pkg, O, fmt):
    """1080p -> 4K with tap 4 on integer planes, nine frames per call: enough 128 x 32 tiles for ewa_periodic_quad2x8_kernel (two
    periods per lane on the 8 x 8 support) to be the automatic choice.  Every frame against the full-window result (kernel mode
    15), two of them against the oracle."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    sw, sh, tw, th, frames = 1920, 1080, 3840, 2160, 9
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, tap=4)
    assert f.plan_info(0).filter_size == 9 and f.periodic_support(0) == 8
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=9900 + k) for k in range(frames)]
    auto = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
    full = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 15)
    of = O.OracleFilter(ofmt, sw, sh, tw, th, tap=4)
    for k in range(frames):
        assert_planes_equal(auto[k], full[k], f.out_dims(), what=f"{fmt} frame {k}: trimmed two-period form vs full window")
        if k in (0, frames - 1):
            assert_planes_equal(auto[k], of.get_frame(srcs[k], threads=8), f.out_dims(), what=f"{fmt} frame {k} vs oracle")
    f.close()


def test_chroma_sited_as_mpeg2_at_tap_4_runs_on_eight_rows_of_nine_taps(gpu_pkg, O):
    """1080p -> 4K 4:2:0 with tap 4 (Jinc64Resize) and the default siting, 32 frames per call: the chroma table's disc spans eight source
    rows and nine columns; ewa_periodic_quad2x8_kernel takes that support with nine taps per kernel row and the chords by a compile-time
    pattern -- 60 of the window's 81 taps (round 5; the full window on ewa_periodic_kernel before).  Every frame against the full window
    (kernel mode 15) and against all 72 taps of the support (knob quad_inner = 0), two against the oracle; 8- and 16-bit."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    for fmt in ("YUV420P8", "YUV420P16"):
        sw, sh, tw, th, frames = 1920, 1080, 3840, 2160, 32   # (32 frames: the chroma planes' launches fill the chip with 128 x 32 tiles)
        ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
        f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, tap=4)
        assert f.periodic_support(0) == 8 and f.periodic_support(1) == 8
        assert f.periodic_taps(1, rows_kernel=3) == 60.0 and f.periodic_taps(0, rows_kernel=3) == 56.0
        srcs = [O.lcg_frame(ofmt, sw, sh, seed=7900 + k) for k in range(frames)]
        auto = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
        inst = f.last_instance(1)
        assert inst.startswith("ewa_periodic_quad2x8_kernel<") and ", 9, " in inst and not inst.endswith(", 9, 0ul>"), inst
        assert f.last_border(1) & 64, f.last_border(1)     # ... and the plane's border columns in that kernel's edge tiles
        full = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 15)
        with gpu_pkg.knobs(quad_inner=0):   # (read when the plan is built): all 72 taps of the 8 x 9 support
            g = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, tap=4)
            plain = _run_batch(torch, gpu_pkg, g, gfmt, srcs, frames, 0)
            assert g.periodic_taps(1, rows_kernel=3) == 72.0 and g.last_instance(1).endswith(", 9, 0ul>"), g.last_instance(1)
            g.close()
        of = O.OracleFilter(ofmt, sw, sh, tw, th, tap=4)
        for k in range(frames):
            assert_planes_equal(auto[k], full[k], f.out_dims(), what=f"{fmt} frame {k}: 8 x 9 support vs full window")
            assert_planes_equal(auto[k], plain[k], f.out_dims(), what=f"{fmt} frame {k}: chord pattern vs all 72 taps")
            if k in (0, frames - 1):
                assert_planes_equal(auto[k], of.get_frame(srcs[k], threads=16), f.out_dims(), what=f"{fmt} frame {k} vs oracle")
        f.close()


def test_chroma_sited_as_mpeg2_runs_on_six_rows_of_seven_taps(gpu_pkg, O):
    """1080p -> 4K 4:2:0 with the default siting, 16 frames per call: the chroma table's disc spans six source rows and (shifted by an
    eighth of a sample) seven columns; ewa_periodic_quad2_kernel takes that support with seven taps per kernel row.  Every frame
    against the full window (kernel mode 15), two against the oracle."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    fmt, sw, sh, tw, th, frames = "YUV420P8", 1920, 1080, 3840, 2160, 16
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0)
    assert f.periodic_support(0) == 6 and f.periodic_support(1) == 6
    # (round 5: the chords of that support by a compile-time pattern -- per (kernel row, row phase) the taps that are zero for both
    # column phases are not executed: 36 of the 42)
    assert f.periodic_taps(1, rows_kernel=3) == 36.0 and f.periodic_taps(0, rows_kernel=3) == 34.0
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=7700 + k) for k in range(frames)]
    auto = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
    assert f.last_kernel(1) == "ewa_periodic_quad2_kernel", f.last_kernel(1)
    inst = f.last_instance(1)
    assert inst.endswith(", 7>") and not inst.endswith(", 0u, 7>"), inst   # seven taps per kernel row, a chord pattern
    full = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 15)
    with gpu_pkg.knobs(quad_inner=0):   # (read when the plan is built): all 42 taps of the 6 x 7 support
        g = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0)
        plain = _run_batch(torch, gpu_pkg, g, gfmt, srcs, frames, 0)
        assert g.periodic_taps(1, rows_kernel=3) == 42.0 and g.last_instance(1).endswith(", 0u, 7>"), g.last_instance(1)
        g.close()
    of = O.OracleFilter(ofmt, sw, sh, tw, th)
    for k in range(frames):
        assert_planes_equal(auto[k], full[k], f.out_dims(), what=f"frame {k}: 6 x 7 support vs full window")
        assert_planes_equal(auto[k], plain[k], f.out_dims(), what=f"frame {k}: chord pattern vs all 42 taps")
        if k in (0, frames - 1):
            assert_planes_equal(auto[k], of.get_frame(srcs[k], threads=8), f.out_dims(), what=f"frame {k} vs oracle")
    f.close()
