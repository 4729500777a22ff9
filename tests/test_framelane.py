"""GPU parity tests of the frame-lane kernel (kernel_framelane.hip: the 64 lanes of a wave are one output pixel of
64 different frames).  Through the C ABI, bit-exact against the CPU oracle: every small geometry case as a single frame
(kernel mode 11 = forced), and device-resident batches of distinct frames in every lane-fill state (1, partial wave,
64, 64 + partial, 2 x 64 + partial)."""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs, to_device, to_host
from test_gpu_parity import SMALL_CASES, _id, _random_case, _random_case_v2, _random_case_v3, _SWEEP

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", SMALL_CASES, ids=_id)
def test_single_frame_through_the_framelane_kernel(gpu_pkg, O, case):
    fmt, sw, sh, tw, th, kw = case
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th, **oracle_kwargs(kw))
    src = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=4242)
    want = of.get_frame(src, threads=4)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0, **kw)
    f.set_kernel_mode(11)
    got = f.get_frame(src)
    if not f.last_kernel(0).startswith("ewa_framelane"):  # footprints beyond the 64 KB LDS tile fall back to the other kernels
        fs, sb = of.tables[0].filter_size, np.dtype(O.FORMATS[fmt].dtype).itemsize
        assert (fs + 3) ** 2 * (64 * sb + 4) > 48 * 1024, f"frame-lane kernel not used for fs {fs}, {sb}-byte samples"
    assert_planes_equal(got, want, f.out_dims(), what=_id(case))
    f.close()


BATCH_CASES = [
    # (format, src, dst, args, batch sizes)
    ("Y8", 160, 90, 219, 123, {}, (1, 3, 16, 64, 70, 130)),                  # 1.37x: no phase structure, fs 7
    ("Y8", 300, 200, 411, 274, {}, (64, 96)),                                # many tiles per launch
    ("Y16", 300, 200, 411, 274, {}, (40,)),
    ("Y32", 300, 200, 380, 250, {}, (64,)),
    ("Y8", 400, 300, 333, 250, {}, (70,)),                                   # 5/6 down-scale (fs 8)
    ("Y16", 160, 90, 219, 123, {}, (17, 65)),
    ("Y32", 160, 90, 219, 123, {}, (17, 64)),
    ("Y8", 192, 108, 160, 90, {}, (33, 64)),                                 # 5/6 down-scale: fs 8
    ("Y8", 128, 72, 192, 108, dict(tap=8), (20, 64)),                        # 1.5x with tap 8: fs 17, drifting
    ("Y8", 128, 72, 240, 135, dict(tap=4), (40,)),                           # 15/8 with tap 4: fs 9
    ("Y8", 150, 100, 330, 190, dict(tap=6), (24,)),                          # fs 13: run-time filter size path
    ("Y10", 160, 90, 219, 123, dict(tap=2), (19,)),                          # fs 5 (run-time path), peak 1023
    ("YUV420P8", 160, 96, 222, 130, dict(cplace="topleft"), (18,)),          # luma + chroma tables
    ("RGBPS", 96, 64, 131, 90, dict(tap=4, blur=0.98), (16,)),
    ("Y8", 64, 48, 397, 301, dict(src_left=1.5, src_top=-2.25, src_width=50.5, src_height=40.125), (16,)),  # 7.9x, crop
]


@pytest.mark.parametrize("case", BATCH_CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_batches_of_frames(gpu_pkg, O, case):
    """jinc_filter_process_device on distinct frames: every frame of the batch against the oracle, for batch sizes that
    leave lanes idle, fill a wave exactly and spill into further frame groups.  Batches of >= 16 frames of plans
    without phase structure take the frame-lane kernel by themselves (from 24 for filter sizes above 9); smaller ones are
    forced (kernel mode 11)."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, kw, sizes = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    ddims = f.out_dims()
    nmax = max(sizes)
    frames = [O.lcg_frame(ofmt, sw, sh, seed=900 + i) for i in range(nmax)]
    wants = [of.get_frame(fr, threads=8) for fr in frames]
    np_dtype = frames[0][0].dtype
    tdtype = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.int16, np.dtype(np.float32): torch.float32}[np.dtype(np_dtype)]
    sb = np.dtype(np_dtype).itemsize

    def to_t(a):
        a = np.ascontiguousarray(a)
        return torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a)

    for n in sizes:
        src_t = [to_device(torch.stack([to_t(fr[i]) for fr in frames[:n]])) for i in range(gfmt.planes)]
        dst_t = [torch.zeros((n, h, (w * sb + 63) // 64 * 64 // sb), dtype=tdtype, device="cuda") for (w, h) in ddims]
        # automatic from 16 frames, from 24 for filter sizes above 9 -- except drifting plans the direct kernel's runs form takes
        # (1.5x with tap 8: since round 3 the frame-lane kernel is never their automatic choice below 17 phases)
        f.set_kernel_mode(0)
        has_runs = f.interior_kernel(0) == "ewa_direct_runs_kernel"
        f.set_kernel_mode(0 if n >= 24 and not has_runs else 11)
        stream = torch.cuda.current_stream()
        f.process_device([t.data_ptr() for t in src_t], [t.stride(1) * sb for t in src_t], [t.stride(0) * sb for t in src_t],
                         [t.data_ptr() for t in dst_t], [t.stride(1) * sb for t in dst_t], [t.stride(0) * sb for t in dst_t],
                         n, stream=stream.cuda_stream)
        stream.synchronize()
        # (last_kernel names the kernel of the batch's last part: beyond whole groups of 128 frames a remainder of fewer than 16
        # frames is a call of its own under the normal rules -- for these plans the gather kernel)
        small_rest = n > 128 and 0 < n % 128 < 2   # (automatic mode: n >= 24; from 2 frames the sub-group form of the frame-lane kernel)
        assert f.last_kernel(0).startswith("ewa_gather" if small_rest else "ewa_framelane"), (n, f.last_kernel(0))
        for k in range(n):
            got = [to_host(dst_t[i][k]).numpy().view(np_dtype) for i in range(gfmt.planes)]
            assert_planes_equal(got, wants[k], ddims, what=f"batch {n} frame {k}")
    f.close()


def test_full_size_batch(gpu_pkg, O):
    """1280x720 -> 1754x986 (no phase structure), 64 + 5 frames (a full frame group and a partial one): all frames against
    the forced gather kernel, four of them against the oracle."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, n = "Y8", 1280, 720, 1754, 986, 69
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    src = torch.randint(0, 256, (n, sh, 1280), device="cuda", generator=gen, dtype=torch.int32).to(torch.uint8)
    outs = []
    for mode in (0, 1):
        dst = torch.zeros((n, th, 1792), dtype=torch.uint8, device="cuda")
        f.set_kernel_mode(mode)
        stream = torch.cuda.current_stream()
        f.process_device([src.data_ptr()], [src.stride(1)], [src.stride(0)], [dst.data_ptr()], [dst.stride(1)], [dst.stride(0)], n,
                         stream=stream.cuda_stream)
        stream.synchronize()
        if mode == 0:
            assert f.last_kernel(0) == "ewa_framelane_win1k_kernel", f.last_kernel(0)  # 1024 threads, 32 x 32 tiles
        outs.append(to_host(dst[:, :, :tw]).numpy())
    assert np.array_equal(outs[0], outs[1])
    for k in (0, 31, 64, 68):
        frame = [np.ascontiguousarray(to_host(src[k]).numpy())]
        want = of.get_frame(frame, threads=16)[0][:th, :tw]
        assert np.array_equal(outs[0][k], want), f"frame {k}"
    f.close()


def test_unaligned_destination_takes_the_sample_stores(gpu_pkg, O):
    """A destination whose base / pitch is not a multiple of 4 samples cannot take the packed 4-sample stores."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, n = "Y8", 100, 60, 137, 83, 20
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=50 + i) for i in range(n)]
    src_t = to_device(torch.stack([torch.from_numpy(np.ascontiguousarray(fr[0])) for fr in frames]))
    for pitch, offset in ((139, 0), (140, 1), (141, 3)):
        buf = torch.full((n * th * pitch + 8,), 0xAB, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream()
        f.process_device([src_t.data_ptr()], [src_t.stride(1)], [src_t.stride(0)], [buf.data_ptr() + offset], [pitch],
                         [th * pitch], n, stream=stream.cuda_stream)
        stream.synchronize()
        assert f.last_kernel(0).startswith("ewa_framelane")
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
def test_randomised_arguments_through_the_framelane_kernel(gpu_pkg, O, seed, gen):
    """The seeded sweeps of test_gpu_parity.py (formats, ratios, taps, quantisation, blur, crops, chroma siting, extreme
    geometry) with the frame-lane kernel forced: a single frame through jinc_filter_get_frame, and for every fourth seed a
    device-resident batch of three distinct frames (lanes 0..2 of a wave)."""
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
    f.set_kernel_mode(11)
    what = f"gen {gen} seed {seed}: {fmt} {sw}x{sh}->{tw}x{th} {kw}"
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=seed + 7 * k) for k in range(3 if seed % 4 == 0 else 1)]
    wants = [of.get_frame(fr, threads=4) for fr in frames]
    assert_planes_equal(f.get_frame(frames[0]), wants[0], f.out_dims(), what=what)
    f.set_kernel_mode(16)  # the sub-group form (filter sizes 5, 7, 8, 9; the others stay on the 64-frame form)
    assert_planes_equal(f.get_frame(frames[0]), wants[0], f.out_dims(), what=what + " sub-group form")
    if len(frames) > 1:
        f.set_kernel_mode(11 if seed % 8 == 0 else 16)
        gfmt = gpu_pkg.FORMATS[fmt]
        np_dtype = frames[0][0].dtype
        sb = np.dtype(np_dtype).itemsize
        tdtype = {1: torch.uint8, 2: torch.int16, 4: torch.float32}[sb]
        n = len(frames)

        def to_t(a):
            a = np.ascontiguousarray(a)
            return torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a)

        src_t = [to_device(torch.stack([to_t(fr[i]) for fr in frames])) for i in range(gfmt.planes)]
        dst_t = [torch.zeros((n, h, (w * sb + 63) // 64 * 64 // sb), dtype=tdtype, device="cuda") for (w, h) in f.out_dims()]
        stream = torch.cuda.current_stream()
        f.process_device([t.data_ptr() for t in src_t], [t.stride(1) * sb for t in src_t], [t.stride(0) * sb for t in src_t],
                         [t.data_ptr() for t in dst_t], [t.stride(1) * sb for t in dst_t], [t.stride(0) * sb for t in dst_t],
                         n, stream=stream.cuda_stream)
        stream.synchronize()
        for k in range(n):
            got = [to_host(dst_t[i][k]).numpy().view(np_dtype) for i in range(gfmt.planes)]
            assert_planes_equal(got, wants[k], f.out_dims(), what=what + f" batch frame {k}")
    f.close()
