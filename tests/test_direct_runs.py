"""Runs form of the direct kernel (csrc/device_plan.cpp plan_runs, kernels.h DirectRun): drifting ratios (1.5x, 3x, 8/3 x 9/4,
5/2 ...) with filter sizes above the quasi-periodic kernel's -- the aliases Jinc144Resize / Jinc256Resize at those ratios (ref
/root/reference/src/JincResize.cpp:1085-1108).  The interior is cut into rectangles of one coefficient set each; inside a
rectangle the windows are exactly periodic, which is the direct kernel's premise.  Bit-exact against the oracle, frame by frame
and in device batches; the gather kernel stays the fallback where the direct kernel's fetches are not safe."""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs, to_device, to_host

pytestmark = pytest.mark.gpu

RUNS = "ewa_direct_runs_kernel"

# (format, src_w, src_h, dst_w, dst_h, script args)
RUN_CASES = [
    ("Y8", 320, 180, 480, 270, dict(tap=8)),            # 1.5x, fs 17: two 9-tap steps per kernel row
    ("Y8", 640, 360, 960, 540, dict(tap=6)),            # 1.5x, fs 13: single-step rows at source step 2
    ("Y16", 211, 97, 633, 291, dict(tap=5)),            # 3x, fs 11, source step 1
    ("Y32", 150, 120, 225, 180, dict(tap=4)),           # fs 9: the quasi-periodic kernel's plan, runs when forced
    ("YUV420P8", 256, 144, 384, 216, dict(tap=8)),      # chroma table too
    ("Y8", 240, 160, 640, 360, dict(tap=6)),            # 8/3 x 9/4: 72 phases, source steps 3 and 4
    ("RGBPS", 160, 90, 240, 135, dict(tap=7)),          # float planes, fs 15
    ("Y10", 200, 120, 500, 300, dict(tap=5)),           # 5/2, peak 1023
    ("Y16", 320, 180, 480, 270, dict(tap=12)),          # fs 25
    ("Y8", 300, 200, 450, 300, dict(tap=16)),           # fs 33: three 11-tap steps
    ("Y8", 322, 182, 483, 273, dict(tap=8, blur=0.9, quant_x=97, quant_y=31)),  # ragged sizes, other class structure
    ("Y8", 1280, 720, 1920, 1080, dict(tap=8)),         # 1.5x with Jinc256Resize at full size
]


def _id(c):
    return f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}_tap{c[5].get('tap', 3)}"


@pytest.mark.parametrize("mode", [0, 14, 1], ids=["auto", "runs", "gather"])
@pytest.mark.parametrize("case", RUN_CASES, ids=_id)
def test_drifting_plans_with_large_taps(gpu_pkg, O, case, mode):
    fmt, sw, sh, tw, th, kw = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=31337)
    want = of.get_frame(src, threads=8)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    info = f.plan_info()
    f.set_kernel_mode(mode)
    got = f.get_frame(src)
    assert_planes_equal(got, want, f.out_dims(), what=f"{fmt} {sw}x{sh}->{tw}x{th} {kw} mode {mode}")
    drifting = info.quasi == 1 and info.periodic == 0
    big_call = tw * th * info.filter_size ** 2 >= 1e8   # calls below 1e8 taps stay with the gather kernel (csrc/dispatch.cpp Rules)
    if drifting and (mode == 14 or (mode == 0 and big_call)):
        assert f.last_kernel(0) == RUNS
    if drifting and mode in (0, 14):
        assert f.interior_kernel(0) == RUNS
    if mode == 1:
        assert f.last_kernel(0) == "ewa_gather_kernel"
    f.close()


def test_full_size_case_is_a_drifting_plan(gpu_pkg):
    """The premise of the full-size case above: 1280 x 720 -> 1920 x 1080 drifts (no exact period), origins affine."""
    f = gpu_pkg.Filter(gpu_pkg.FORMATS["Y8"], 1280, 720, 1920, 1080, device=0, tap=8)
    info = f.plan_info()
    assert (info.periodic, info.quasi, info.filter_size) == (0, 1, 17)
    assert (info.quasi_period_x, info.quasi_period_y, info.quasi_step_x, info.quasi_step_y) == (3, 3, 2, 2)
    assert f.interior_kernel(0) == RUNS
    f.close()


@pytest.mark.parametrize("fmt", ["Y8", "Y16", "Y32"])
@pytest.mark.parametrize("sw,expect", [(200, RUNS), (202, "ewa_gather_kernel")], ids=["pitch200", "pitch202_not_multiple_of_4"])
def test_device_batch_with_tight_pitch(gpu_pkg, O, fmt, sw, expect):
    """Device entry, pitch == row size, five frames per call: the runs form fetches aligned dwords inside the plane only; where
    the pitch in bytes is not a multiple of 4 (8-bit, 202 samples) the call falls back to the gather kernel."""
    torch = pytest.importorskip("torch")
    sh, tw, th, n = 120, sw * 3 // 2, 180, 5
    kw = dict(tap=6)
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    assert f.plan_info().quasi == 1 and f.plan_info().periodic == 0
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=1200 + i) for i in range(n)]
    host = np.stack([np.ascontiguousarray(fr[0][:sh, :sw]) for fr in frames])
    src = to_device(torch.from_numpy(host)).contiguous()
    dst = torch.zeros((n, th, tw), dtype=src.dtype, device="cuda")
    sb = host.dtype.itemsize
    stream = torch.cuda.current_stream()
    f.process_device([src.data_ptr()], [sw * sb], [sw * sh * sb], [dst.data_ptr()], [tw * sb], [tw * th * sb], n, stream=stream.cuda_stream)
    stream.synchronize()
    f.set_kernel_mode(14)  # (five small frames are below the automatic rule's 1e8 taps)
    f.process_device([src.data_ptr()], [sw * sb], [sw * sh * sb], [dst.data_ptr()], [tw * sb], [tw * th * sb], n, stream=stream.cuda_stream)
    stream.synchronize()
    assert f.last_kernel(0) == (expect if sb == 1 else RUNS)
    out = to_host(dst).numpy()
    for i in range(n):
        want = of.get_frame(frames[i], threads=4)
        assert np.array_equal(out[i].view(np.uint32) if sb == 4 else out[i], want[0][:th, :tw].view(np.uint32) if sb == 4 else want[0][:th, :tw]), f"frame {i}"
    f.close()


def _random_case(rng):
    fmts = ["Y8", "Y10", "Y16", "Y32", "YUV420P8", "YUV444P16", "RGBPS"]
    fmt = fmts[rng.integers(len(fmts))]
    (nx, dx), (ny, dy) = [[(3, 2), (3, 1), (5, 2), (5, 4), (8, 3), (9, 4), (5, 3), (7, 4)][k] for k in rng.integers(8, size=2)]
    if rng.random() < 0.6:
        ny, dy = nx, dx
    sw = int(rng.integers(6, 40)) * 4 * dx
    sh = int(rng.integers(5, 24)) * 4 * dy
    kw = dict(tap=int(rng.integers(4, 13)))
    if rng.random() < 0.3:
        kw.update(quant_x=int(rng.integers(1, 257)), quant_y=int(rng.integers(1, 257)))
    if rng.random() < 0.3:
        kw["blur"] = float(np.round(rng.uniform(0.8, 1.25), 3))
    if "420" in fmt:
        kw["cplace"] = ["mpeg2", "mpeg1", "topleft"][rng.integers(3)]
    return fmt, sw, sh, sw * nx // dx, sh * ny // dy, kw


# JINC_RUNS_SWEEP_SEEDS=N widens the sweep for soak runs
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("JINC_RUNS_SWEEP_SEEDS", "32"))))
def test_randomised_drifting_ratios(gpu_pkg, O, seed):
    """Seeded sweep over drifting ratios with taps 4..12: automatic choice and the forced runs form, bit for bit."""
    rng = np.random.default_rng(4000 + seed)
    fmt, sw, sh, tw, th, kw = _random_case(rng)
    try:
        of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    except Exception:
        pytest.skip("oracle rejects this geometry")
    try:
        f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    except gpu_pkg.JincError as e:
        assert "smaller than the filter footprint" in str(e)
        return
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=seed)
    want = of.get_frame(src, threads=4)
    what = f"seed {seed}: {fmt} {sw}x{sh}->{tw}x{th} {kw}"
    for mode in (0, 14):
        f.set_kernel_mode(mode)
        assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=what + f" mode {mode}")
    f.close()


@pytest.mark.parametrize("fmt,tap,n", [("Y8", 6, 70), ("Y16", 4, 64), ("Y32", 8, 65), ("Y8", 4, 9), ("Y16", 4, 20), ("Y32", 4, 33)],
                         ids=["Y8_tap6_70", "Y16_tap4_64", "Y32_tap8_65", "Y8_tap4_9", "Y16_tap4_20", "Y32_tap4_33"])
def test_batches_take_the_border_frame_to_the_framelane_kernel(gpu_pkg, O, fmt, tap, n):
    """From 32 frames per call on -- from 8 with tap 4, where the frame-lane kernel's sub-group form applies -- the border frame of
    a runs-form plan (every border pixel owns a coefficient set) runs on the frame-lane kernel (lanes = frames, or frames x output
    rows) beside the interior's runs: every frame equals the forced gather kernel's result, some are checked against the oracle."""
    torch = pytest.importorskip("torch")
    sw, sh, tw, th = 160, 92, 240, 138
    kw = dict(tap=tap)
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=5200 + i) for i in range(n)]
    host = np.stack([np.ascontiguousarray(fr[0][:sh, :sw]) for fr in frames])
    sb = host.dtype.itemsize
    src = to_device(torch.from_numpy(host.view(np.int16) if sb == 2 else host)).contiguous()
    outs = {}
    for mode in (14, 1):
        dst = torch.zeros((n, th, tw), dtype=src.dtype, device="cuda")
        f.set_kernel_mode(mode)
        stream = torch.cuda.current_stream()
        f.process_device([src.data_ptr()], [sw * sb], [sw * sh * sb], [dst.data_ptr()], [tw * sb], [tw * th * sb], n, stream=stream.cuda_stream)
        stream.synchronize()
        outs[mode] = to_host(dst).numpy().view(host.dtype if sb != 4 else np.uint32)
        assert f.last_kernel(0) == (RUNS if mode == 14 else "ewa_gather_kernel")
    assert np.array_equal(outs[14], outs[1])
    for i in (0, min(63, n // 2), n - 1):
        want = np.ascontiguousarray(of.get_frame(frames[i], threads=4)[0][:th, :tw])
        assert np.array_equal(outs[14][i], want.view(np.uint32) if sb == 4 else want), f"frame {i}"
    f.close()
