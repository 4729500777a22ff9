"""The kernel instantiations the benchmark times, under the oracle (VERDICT r4 "weak" 1).

The launchers of the periodic family choose between template instantiations by call size: the quad forms take full-height
tiles (8 row groups) only when a launch holds >= 12 000 (6 x 6 support) / >= 6 144 (8 x 8) full-tile workgroups -- bench.py's 1024
C2 frames and 16 C4 frames, the look-ahead pipeline's 64-frame groups -- and every other parity test stays below that, on the
4-row-group siblings of the same templates.  Here:
  * the same code paths forced onto full-height tiles through the test header's knob (quad_rg = 8): the quad-form, trimmed-support
    and non-finite-sample tests and the three seeded sweeps once more;
  * one real-size batch per benchmarked configuration that reaches its instantiation by the AUTOMATIC rule, every frame against
    the oracle, with jinc_filter_last_instance asserted -- the very string bench.py prints as roofline.kernel.
"""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs
import test_gpu_parity as P

pytestmark = pytest.mark.gpu

# roofline.kernel of `python bench.py --config <name>` at its default batch size -> the test below that compares exactly this
# instantiation with the oracle (tests/test_bench_dist.py asserts that bench.py's line names one of them)
BENCHMARKED = {
    "C2": "ewa_periodic_quad2_kernel<unsigned char, 8, 1026u, 6>",
    "C2H": "ewa_periodic_quad2_kernel<unsigned short, 8, 1026u, 6>",
    "C4": "ewa_periodic_quad8_kernel<float, 8, %uu>",      # (the tap-4 chord pattern's value is filled in from a small call)
    "C2T4": "ewa_periodic_quad2x8_kernel<unsigned char, 4, %uu, 8, 0ul>",
}


@pytest.fixture()
def full_height_tiles(gpu_pkg):
    with gpu_pkg.knobs(quad_rg=8):
        yield
    assert gpu_pkg.get_knob("quad_rg") is None


def _rg_of(instance):
    """Row groups per tile from 'kernel<type, RG, ...>'."""
    return int(instance.split("<", 1)[1].rstrip(">").split(",")[1])


@pytest.mark.parametrize("case", P.QUAD_CASES, ids=P._id)
@pytest.mark.parametrize("frames", [1, 5])
def test_quad_forms_on_full_height_tiles(gpu_pkg, O, full_height_tiles, case, frames):
    P.test_quad_form_of_the_periodic_kernel(gpu_pkg, O, case, frames)
    inst = gpu_pkg.last_instance()
    assert inst.startswith("ewa_periodic_quad"), inst
    if not inst.startswith("ewa_periodic_quad9_kernel"):   # (the fs-9 full-window form has one tile height)
        assert _rg_of(inst) == 8, inst


@pytest.mark.parametrize("fmt,sw,sh,tw,th,kw,full,trimmed", [
    ("YUV420P16", 640, 360, 1280, 720, dict(tap=8), 17, 16), ("Y16", 320, 180, 640, 360, dict(tap=4), 9, 8),
    ("Y32", 320, 180, 640, 360, dict(tap=3), 7, 6), ("RGBPS", 320, 180, 640, 360, dict(tap=4, blur=0.98), 9, 8)],
    ids=["tap8_u16", "tap4_u16", "tap3_f32", "C4_f32_small"])
def test_trimmed_support_on_full_height_tiles(gpu_pkg, O, full_height_tiles, fmt, sw, sh, tw, th, kw, full, trimmed):
    P.test_trimmed_support_of_the_periodic_kernels(gpu_pkg, O, fmt, sw, sh, tw, th, kw, full, trimmed)


@pytest.mark.parametrize("fmt,sw,sh,tw,th,kw", [("Y32", 320, 180, 640, 360, dict(tap=3)), ("RGBPS", 200, 120, 400, 240, dict(tap=4, blur=0.98)),
                                                ("YUV420PS", 256, 144, 512, 288, dict(tap=3))], ids=["Y32_tap3", "RGBPS_tap4", "YUV420PS_tap3"])
def test_non_finite_frames_in_batches_on_full_height_tiles(gpu_pkg, O, full_height_tiles, fmt, sw, sh, tw, th, kw):
    P.test_float_planes_take_the_trimmed_support_only_where_every_sample_is_finite(gpu_pkg, O, fmt, sw, sh, tw, th, kw, 13)
    inst = gpu_pkg.last_instance()
    assert inst.startswith("ewa_periodic_quad") and _rg_of(inst) == 8, inst


@pytest.mark.parametrize("tap", [3, 4])
def test_one_non_finite_sample_anywhere_on_full_height_tiles(gpu_pkg, O, full_height_tiles, tap):
    P.test_one_non_finite_sample_anywhere_in_a_float_plane(gpu_pkg, O, tap, 13)
    inst = gpu_pkg.last_instance()
    assert inst.startswith("ewa_periodic_quad") and _rg_of(inst) == 8, inst


@pytest.mark.parametrize("seed", range(P._SWEEP))
@pytest.mark.parametrize("gen", [1, 2, 3], ids=["small", "structured", "extreme"])
def test_seeded_sweeps_on_full_height_tiles(gpu_pkg, O, full_height_tiles, seed, gen):
    """The three seeded sweeps' periodic legs (kernel mode 13 among them) with the quad forms on 8 row groups per tile."""
    rng = np.random.default_rng(1000 * gen + seed)
    fmt, sw, sh, tw, th, kw = {1: P._random_case, 2: P._random_case_v2, 3: P._random_case_v3}[gen](rng)
    try:
        of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
        f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    except Exception:
        pytest.skip("geometry rejected (covered by test_randomised_arguments)")
    if not any(f.plan_info(t).periodic for t in range(f.num_tables)):
        f.close()
        pytest.skip("no periodic table: no quad form")
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=seed)
    want = of.get_frame(src, threads=4)
    f.set_kernel_mode(gpu_pkg.KernelMode.QUAD)
    assert_planes_equal(f.get_frame(src), want, f.out_dims(), what=f"gen {gen} seed {seed}: {fmt} {sw}x{sh}->{tw}x{th} {kw} quad, 8 row groups")
    for t in range(f.num_tables):
        inst = f.last_instance(t)
        if inst.startswith("ewa_periodic_quad") and not inst.startswith("ewa_periodic_quad9"):
            assert _rg_of(inst) == 8, inst
    f.close()


def _batch_against_oracle(gpu_pkg, O, fmt, sw, sh, tw, th, kw, frames, seed0, threads=16):
    """`frames` Appendix-A frames (seeds seed0 ...) through jinc_filter_process_device in ONE call under the automatic rule; every
    frame against the oracle.  Returns the filter's last_instance per table."""
    torch = pytest.importorskip("torch")
    from test_framelane_pair import _run_batch
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    srcs = [O.lcg_frame(ofmt, sw, sh, seed=seed0 + k) for k in range(frames)]
    got = _run_batch(torch, gpu_pkg, f, gfmt, srcs, frames, 0)
    inst = [f.last_instance(t) for t in range(f.num_tables)]
    dims = f.out_dims()
    f.close()
    for k in range(frames):
        assert_planes_equal(got[k], of.get_frame(srcs[k], threads=threads), dims, what=f"{fmt} {sw}x{sh}->{tw}x{th} {kw} frame {k} of {frames}")
        got[k] = None
    return inst


def test_c2_batch_reaches_the_benchmarked_instantiation(gpu_pkg, O):
    """40 C2 frames per call (>= 12 000 full-tile workgroups): the automatic rule launches the instantiation BENCH times."""
    inst = _batch_against_oracle(gpu_pkg, O, "Y8", 1920, 1080, 3840, 2160, dict(tap=3), 40, 12345)
    assert inst[0] == BENCHMARKED["C2"], inst


def test_c2_geometry_on_16_bit_reaches_full_height_tiles(gpu_pkg, O):
    inst = _batch_against_oracle(gpu_pkg, O, "Y16", 1920, 1080, 3840, 2160, dict(tap=3), 40, 4321)
    assert inst[0] == BENCHMARKED["C2H"], inst


def _tap4_pattern(gpu_pkg):
    """The tap-4 chord pattern's template argument, read from a small forced call (a constant of the build, not of the frame)."""
    f = gpu_pkg.Filter(gpu_pkg.FORMATS["Y8"], 192, 108, 384, 216, device=0, tap=4)
    f.set_kernel_mode(gpu_pkg.KernelMode.QUAD)
    f.get_frame([np.zeros((108, 192), dtype=np.uint8)])
    inst = f.last_instance(0)
    f.close()
    assert inst.startswith("ewa_periodic_quad8_kernel<unsigned char, 4, "), inst
    return int(inst.rsplit(",", 1)[1].strip(" u>"))


def test_c4_batch_reaches_the_benchmarked_instantiation(gpu_pkg, O):
    """4 C4 frames per call: >= 6 144 full-tile workgroups per plane and >= 1e9 taps -- float planes on the trimmed 8 x 8 support,
    quad form, 8 row groups per tile: what `bench.py --config C4` times (16 frames)."""
    inst = _batch_against_oracle(gpu_pkg, O, "RGBPS", 3840, 2160, 7680, 4320, dict(tap=4, blur=0.98), 4, 12345)
    assert inst[0] == BENCHMARKED["C4"] % _tap4_pattern(gpu_pkg), inst


def test_jinc64_batch_reaches_the_two_period_form(gpu_pkg, O):
    """9 frames of 1080p -> 4K Y8 with tap 4: ewa_periodic_quad2x8_kernel by the automatic rule, every frame against the oracle."""
    inst = _batch_against_oracle(gpu_pkg, O, "Y8", 1920, 1080, 3840, 2160, dict(tap=4), 9, 777)
    assert inst[0] == BENCHMARKED["C2T4"] % _tap4_pattern(gpu_pkg), inst
