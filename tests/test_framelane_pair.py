"""GPU parity tests of the frame-pair form of the frame-lane kernel (kernel_framelane_pair.hip: 128 frames per workgroup, a
lane owns two adjacent frames, packed multiplies / adds).  Through the C ABI, bit-exact against the CPU oracle: batches
that fill whole groups of 128 frames (automatic choice; the rest of the batch runs on the 64-frame form), and -- forced
with kernel mode 12 -- every lane-fill state (1 frame, odd counts, 127, 129 ...), single frames of every small geometry
case and the seeded argument sweeps."""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs, to_device, to_host
from test_gpu_parity import SMALL_CASES, _id, _random_case, _random_case_v2, _random_case_v3, _SWEEP

pytestmark = pytest.mark.gpu

PAIR = "ewa_framelane_pair_kernel"


def _run_batch(torch, gpu_pkg, f, gfmt, frames, n, mode, pad=64):
    """Frames 0..n-1 through jinc_filter_process_device on device-resident planes; returns per-frame plane lists."""
    np_dtype = frames[0][0].dtype
    sb = np.dtype(np_dtype).itemsize
    tdtype = {1: torch.uint8, 2: torch.int16, 4: torch.float32}[sb]

    def to_t(a):
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a)

    src_t = [to_device(torch.stack([to_t(fr[i]) for fr in frames[:n]])) for i in range(gfmt.planes)]
    dst_t = [torch.zeros((n, h, (w * sb + pad - 1) // pad * pad // sb), dtype=tdtype, device="cuda") for (w, h) in f.out_dims()]
    f.set_kernel_mode(mode)
    stream = torch.cuda.current_stream()
    f.process_device([t.data_ptr() for t in src_t], [t.stride(1) * sb for t in src_t], [t.stride(0) * sb for t in src_t],
                     [t.data_ptr() for t in dst_t], [t.stride(1) * sb for t in dst_t], [t.stride(0) * sb for t in dst_t],
                     n, stream=stream.cuda_stream)
    stream.synchronize()
    return [[to_host(dst_t[i][k]).numpy().view(np_dtype) for i in range(gfmt.planes)] for k in range(n)]


BATCH_CASES = [
    # (format, src, dst, args, batch sizes in automatic mode, batch sizes forced)
    ("Y8", 160, 90, 219, 123, {}, (128, 129, 200, 259), (1, 2, 3, 64, 127)),      # 1.37x: no phase structure, fs 7
    ("Y8", 300, 200, 411, 274, {}, (128,), ()),                                      # many tiles per launch
    ("Y16", 160, 90, 219, 123, {}, (130,), (5, 65)),
    ("Y32", 160, 90, 219, 123, {}, (128,), (7, 66)),
    ("Y10", 160, 90, 219, 123, dict(tap=2), (131,), (9,)),                           # fs 5, peak 1023
    ("Y8", 128, 72, 192, 108, {}, (192,), (33,)),                                    # 1.5x: drifting phases (the quasi-periodic kernel's plan below 128 frames)
    ("YUV420P8", 160, 96, 222, 130, dict(cplace="topleft"), (128,), (19,)),          # luma + chroma tables
    ("RGBPS", 96, 64, 131, 90, dict(blur=0.98), (128,), (3,)),
    ("Y8", 64, 48, 397, 301, dict(src_left=1.5, src_top=-2.25, src_width=50.5, src_height=40.125), (128,), (2,)),  # 7.9x, crop
    ("Y8", 200, 120, 111, 67, dict(tap=1), (), (4,)),                                # down-scale with tap 1: fs 4 -> no pair form
]


@pytest.mark.parametrize("case", BATCH_CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_batches_of_frames(gpu_pkg, O, case):
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, kw, auto_sizes, forced_sizes = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    fs = of.tables[0].filter_size
    nmax = max(auto_sizes + forced_sizes)
    frames = [O.lcg_frame(ofmt, sw, sh, seed=1900 + i) for i in range(nmax)]
    wants = [of.get_frame(fr, threads=8) for fr in frames]
    for n, mode in [(n, 0) for n in auto_sizes] + [(n, 12) for n in forced_sizes]:
        got = _run_batch(torch, gpu_pkg, f, gfmt, frames, n, mode)
        if mode == 12:  # forced: the pair form wherever it is configured (filter sizes 5 and 7)
            assert (f.last_kernel(0) == PAIR) == (fs in (5, 7)), f.last_kernel(0)
        elif n % 128 == 0:   # (otherwise last_kernel names the kernel of the remainder, a call of its own)
            assert f.last_kernel(0) == PAIR, f.last_kernel(0)
        for k in range(n):
            assert_planes_equal(got[k], wants[k], f.out_dims(), what=f"batch {n} (mode {mode}) frame {k}")
    f.close()


@pytest.mark.parametrize("case", SMALL_CASES, ids=_id)
def test_single_frame_through_the_pair_form(gpu_pkg, O, case):
    """Lane 0's first frame only; plans the pair form is not configured for (filter sizes other than 5 and 7, footprints
    beyond the LDS tile) take their usual kernels under mode 12."""
    fmt, sw, sh, tw, th, kw = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=4343)
    want = of.get_frame(src, threads=4)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    f.set_kernel_mode(12)
    got = f.get_frame(src)
    assert_planes_equal(got, want, f.out_dims(), what=_id(case))
    f.close()


def test_full_size_batch(gpu_pkg, O):
    """1280x720 -> 1754x986 (no phase structure), 128 + 3 frames: the pair form for the first 128, the rest under the normal
    rules (from 3 frames: lanes = 4 frames x 16 output rows); all frames against the forced gather kernel, four of them against the oracle."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, n = "Y8", 1280, 720, 1754, 986, 131
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(199)
    src = torch.randint(0, 256, (n, sh, 1280), device="cuda", generator=gen, dtype=torch.int32).to(torch.uint8)
    outs = []
    for mode in (0, 1):
        dst = torch.zeros((n, th, 1792), dtype=torch.uint8, device="cuda")
        f.set_kernel_mode(mode)
        stream = torch.cuda.current_stream()
        f.process_device([src.data_ptr()], [src.stride(1)], [src.stride(0)], [dst.data_ptr()], [dst.stride(1)], [dst.stride(0)], n,
                         stream=stream.cuda_stream)
        stream.synchronize()
        if mode == 0:  # 128 frames on the pair form, the remaining 3 as a call of their own (the frame-lane kernel's sub-group form)
            assert f.last_kernel(0) == "ewa_framelane_sub_kernel", f.last_kernel(0)
        outs.append(to_host(dst[:, :, :tw]).numpy())
    assert np.array_equal(outs[0], outs[1])
    for k in (0, 63, 127, 130):
        frame = [np.ascontiguousarray(to_host(src[k]).numpy())]
        want = of.get_frame(frame, threads=16)[0][:th, :tw]
        assert np.array_equal(outs[0][k], want), f"frame {k}"
    f.close()


def test_unaligned_destination_takes_the_sample_stores(gpu_pkg, O):
    """A destination whose base / pitch is not a multiple of 4 samples cannot take the packed 4-sample stores."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, n = "Y8", 100, 60, 137, 83, 5
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_kernel_mode(12)
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=150 + i) for i in range(n)]
    src_t = to_device(torch.stack([torch.from_numpy(np.ascontiguousarray(fr[0])) for fr in frames]))
    for pitch, offset in ((139, 0), (140, 1), (141, 3)):
        buf = torch.full((n * th * pitch + 8,), 0xAB, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream()
        f.process_device([src_t.data_ptr()], [src_t.stride(1)], [src_t.stride(0)], [buf.data_ptr() + offset], [pitch],
                         [th * pitch], n, stream=stream.cuda_stream)
        stream.synchronize()
        assert f.last_kernel(0) == PAIR
        out = to_host(buf).numpy()
        body = out[offset:offset + n * th * pitch].reshape(n, th, pitch)
        for k in range(n):
            want = of.get_frame(frames[k], threads=4)[0][:th, :tw]
            assert np.array_equal(body[k, :, :tw], want), f"pitch {pitch} offset {offset} frame {k}"
        assert (body[:, :, tw:] == 0xAB).all(), "padding between rows was written"
        assert (out[:offset] == 0xAB).all() and (out[offset + n * th * pitch:] == 0xAB).all()
    f.close()


@pytest.mark.parametrize("seed", range(_SWEEP))
@pytest.mark.parametrize("gen", [1, 2, 3], ids=["small", "structured", "extreme"])
def test_randomised_arguments_through_the_pair_form(gpu_pkg, O, seed, gen):
    """The seeded sweeps of test_gpu_parity.py with the pair form forced where it is configured: a single frame through
    jinc_filter_get_frame and, for every second seed, a device-resident batch of three distinct frames (lane 0 holds two
    frames, lane 1 one)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(1000 * gen + seed)
    fmt, sw, sh, tw, th, kw = {1: _random_case, 2: _random_case_v2, 3: _random_case_v3}[gen](rng)
    try:
        of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    except Exception:
        pytest.skip("oracle rejects this geometry")
    try:
        f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    except gpu_pkg.JincError as e:
        assert "smaller than the filter footprint" in str(e)
        return
    f.set_kernel_mode(12)
    what = f"gen {gen} seed {seed}: {fmt} {sw}x{sh}->{tw}x{th} {kw}"
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=seed + 11 * k) for k in range(3 if seed % 2 == 0 else 1)]
    wants = [of.get_frame(fr, threads=4) for fr in frames]
    assert_planes_equal(f.get_frame(frames[0]), wants[0], f.out_dims(), what=what)
    if len(frames) > 1:
        got = _run_batch(torch, gpu_pkg, f, gpu_pkg.FORMATS[fmt], frames, len(frames), 12)
        for k in range(len(frames)):
            assert_planes_equal(got[k], wants[k], f.out_dims(), what=what + f" batch frame {k}")
    f.close()
