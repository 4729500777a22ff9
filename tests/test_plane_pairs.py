"""One frame per call: two planes that share a table and lie at the same distance in source and destination (U and V of a frame)
go out as ONE two-frame launch per kernel (csrc/dispatch.cpp plane_pair).  Frames whose planes sit in one buffer -- the layout of an
AviSynth / VapourSynth frame -- take that path for certain; planes in reverse order or at different distances do not, and every
layout must give the oracle's bits."""
import numpy as np
import pytest

from conftest import oracle_kwargs, to_device, to_host

pytestmark = pytest.mark.gpu

CASES = [
    ("YUV420P8", 256, 144, 512, 288, {}),               # periodic window kernel, gather border
    ("YUV420P16", 256, 144, 384, 216, dict(tap=6)),     # drifting: runs form on the larger plane, gather on the small ones
    ("YUV444P8", 200, 120, 274, 164, {}),               # no phase structure: gather kernel
    ("YUV422P10", 256, 144, 128, 72, {}),               # down-scale: direct kernel
    ("RGBPS", 160, 90, 320, 180, dict(tap=4)),          # three planes of one table: the first two pair
    ("YUVA420P8", 128, 96, 256, 192, {}),               # alpha shares the luma table but is not its neighbour
]


@pytest.mark.parametrize("layout", ["ascending", "descending", "uneven"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_planes_of_one_buffer(gpu_pkg, O, case, layout):
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, kw = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th, **oracle_kwargs(kw))
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    frame = O.lcg_frame(ofmt, sw, sh, seed=8080)
    want = of.get_frame(frame, threads=4)
    sdims, ddims = ofmt.plane_dims(sw, sh), f.out_dims()
    np_dtype = frame[0].dtype
    sb = np_dtype.itemsize

    def lay_out(dims):
        """byte offsets of the planes in one buffer: pitch = row bytes rounded up to 64, planes 256-byte aligned"""
        pitches = [(w * sb + 63) // 64 * 64 for (w, h) in dims]
        sizes = [(p * h + 255) // 256 * 256 for p, (w, h) in zip(pitches, dims)]
        order = list(range(len(dims)))
        if layout == "descending":
            order.reverse()
        offs, at = [0] * len(dims), 0
        for k, i in enumerate(order):
            if layout == "uneven" and k == 2:
                at += 4096 + 256          # a gap in front of the third plane: no common distance
            offs[i] = at
            at += sizes[i]
        return pitches, offs, at

    sp, so, stotal = lay_out(sdims)
    dp, do, dtotal = lay_out(ddims)
    src = torch.zeros(stotal, dtype=torch.uint8, device="cuda")
    dst = torch.zeros(dtotal, dtype=torch.uint8, device="cuda")
    for i, (w, h) in enumerate(sdims):
        plane = np.zeros((h, sp[i]), np.uint8)
        plane[:, :w * sb] = np.ascontiguousarray(frame[i][:h, :w]).view(np.uint8).reshape(h, w * sb)
        src[so[i]:so[i] + h * sp[i]] = to_device(torch.from_numpy(plane.reshape(-1)))
    stream = torch.cuda.current_stream()
    f.process_device([src.data_ptr() + o for o in so], sp, [0] * len(sp), [dst.data_ptr() + o for o in do], dp, [0] * len(dp), 1,
                     stream=stream.cuda_stream)
    stream.synchronize()
    out = to_host(dst).numpy()
    for i, (w, h) in enumerate(ddims):
        got = out[do[i]:do[i] + h * dp[i]].reshape(h, dp[i])[:, :w * sb]
        exp = np.ascontiguousarray(want[i][:h, :w]).view(np.uint8).reshape(h, w * sb)
        assert np.array_equal(got, exp), f"plane {i} ({layout})"
    f.close()
