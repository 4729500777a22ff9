"""GPU parity tests of the frame-lane kernel's sub-group form (kernel_framelane_sub.hip: groups of fewer than 64 frames -- a
wave is 4 / 8 / 16 / 32 frames x 16 / 8 / 4 / 2 output rows of the tile; the lanes of a sub-group share ONE copy of the pixel's
coefficient set, read through DPP operands).  Through the C ABI, bit-exact against the CPU oracle: every frame of device-resident
batches in every fill state of the sub-groups (kernel mode 16 = forced), the automatic choice for batches of 2 .. 48 frames and
for what a batch leaves beyond whole groups of 64, unaligned destinations."""
import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs, to_device, to_host
from test_framelane_pair import _run_batch

pytestmark = pytest.mark.gpu

CASES = [
    # (format, src, dst, args, batch sizes)
    ("Y8", 160, 90, 219, 123, {}, (1, 3, 8, 9, 16, 17, 31, 32, 33, 50, 63)),     # 1.37x: no phase structure, fs 7
    ("Y16", 160, 90, 219, 123, {}, (5, 16, 24)),
    ("Y32", 160, 90, 219, 123, {}, (7, 16, 40)),
    ("Y8", 300, 200, 411, 274, {}, (16, 30)),                                    # many tiles per launch
    ("Y10", 160, 90, 219, 123, dict(tap=2), (4, 16, 19)),                        # fs 5, peak 1023
    ("Y8", 192, 108, 160, 90, {}, (6, 16, 33)),                                  # 5/6 down-scale: fs 8
    ("Y16", 400, 300, 333, 250, {}, (12, 20)),                                   # fs 8, 16-bit
    ("Y8", 128, 72, 240, 135, dict(tap=4), (8, 16, 28)),                         # 15/8 with tap 4: fs 9
    ("Y32", 128, 72, 240, 135, dict(tap=4), (3, 18)),                            # fs 9, float
    ("YUV420P8", 160, 96, 222, 130, dict(cplace="topleft"), (16, 18)),           # luma + chroma tables
    ("RGBPS", 96, 64, 131, 90, dict(blur=0.98), (16,)),
    ("Y8", 64, 48, 397, 301, dict(src_left=1.5, src_top=-2.25, src_width=50.5, src_height=40.125), (16,)),  # 7.9x, crop
    ("Y8", 37, 29, 51, 40, {}, (2, 16)),                                         # tiles narrower and shorter than a strip group
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}{'_tap%d' % c[5]['tap'] if 'tap' in c[5] else ''}")
def test_groups_of_fewer_than_64_frames(gpu_pkg, O, case):
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, kw, sizes = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    frames = [O.lcg_frame(ofmt, sw, sh, seed=1700 + i) for i in range(max(sizes))]
    wants = [of.get_frame(fr, threads=8) for fr in frames]
    for n in sizes:
        got = _run_batch(torch, gpu_pkg, f, gfmt, frames, n, 16)
        assert f.last_kernel(0) == "ewa_framelane_sub_kernel", (n, f.last_kernel(0))
        for k in range(n):
            assert_planes_equal(got[k], wants[k], f.out_dims(), what=f"batch {n} frame {k}")
    f.close()


@pytest.mark.parametrize("n,kernel", [(1, "ewa_gather_kernel"), (2, "ewa_framelane_sub_kernel"), (3, "ewa_framelane_sub_kernel"), (7, "ewa_framelane_sub_kernel"),
                                      (16, "ewa_framelane_sub_kernel"), (24, "ewa_framelane_sub_kernel"), (33, "ewa_framelane_sub_kernel"),
                                      (48, "ewa_framelane_sub_kernel"), (49, "ewa_framelane_win"), (64, "ewa_framelane_win"),
                                      (70, "ewa_framelane_win")])
def test_automatic_choice_by_batch_size(gpu_pkg, O, n, kernel):
    """What a host at look-ahead 32 hands over (groups of 16) and everything else from 2 to 48 frames of a plan without phase
    structure takes the sub-group form by itself (groups of 16 frames x 4 output rows from 9 frames on); so does what a batch
    leaves beyond whole groups of 64 (70 = 64 + 6: last_kernel names the bulk)."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th = "Y8", 160, 90, 219, 123
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=2900 + i) for i in range(n)]
    got = _run_batch(torch, gpu_pkg, f, gpu_pkg.FORMATS[fmt], frames, n, 0)
    assert f.last_kernel(0).startswith(kernel), f.last_kernel(0)  # (ewa_framelane_win_kernel or its 1024-thread shape)
    for k in range(n):
        assert_planes_equal(got[k], of.get_frame(frames[k], threads=4), f.out_dims(), what=f"frame {k}")
    f.close()


def test_unaligned_destination(gpu_pkg, O):
    """Destinations that cannot take the packed 4-sample stores; bytes between the rows stay untouched."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, n = "Y8", 100, 60, 137, 83, 13
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_kernel_mode(16)
    frames = [O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=50 + i) for i in range(n)]
    src_t = to_device(torch.stack([torch.from_numpy(np.ascontiguousarray(fr[0])) for fr in frames]))
    for pitch, offset in ((139, 0), (140, 1), (141, 3), (140, 0)):
        buf = torch.full((n * th * pitch + 8,), 0xAB, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream()
        f.process_device([src_t.data_ptr()], [src_t.stride(1)], [src_t.stride(0)], [buf.data_ptr() + offset], [pitch],
                         [th * pitch], n, stream=stream.cuda_stream)
        stream.synchronize()
        assert f.last_kernel(0) == "ewa_framelane_sub_kernel"
        out = to_host(buf).numpy()
        body = out[offset:offset + n * th * pitch].reshape(n, th, pitch)
        for k in range(n):
            want = of.get_frame(frames[k], threads=4)[0][:th, :tw]
            assert np.array_equal(body[k, :, :tw], want), f"pitch {pitch} offset {offset} frame {k}"
        assert (body[:, :, tw:] == 0xAB).all(), "padding between rows was written"
        assert (out[:offset] == 0xAB).all() and (out[offset + n * th * pitch:] == 0xAB).all()
    f.close()
