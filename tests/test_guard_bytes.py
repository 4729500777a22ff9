"""No kernel writes outside the rows it was given: destination planes of a batch sit in ONE buffer between sentinel-filled guard
zones (in front of, between and behind the planes; the padding bytes of every row belong to the caller too and must survive), for
every interior kernel family, single frames and batches, tight and padded pitches."""
import numpy as np
import pytest

from conftest import to_device, to_host

pytestmark = pytest.mark.gpu

CASES = [
    ("YUV420P8", 200, 120, 274, 164, {}, (1, 2, 16, 33)),            # no structure: gather kernel, frame-lane kernels from 16 frames
    ("YUV420P8", 256, 144, 512, 288, {}, (1, 5, 64)),                 # 2x: window / quad kernels, strips in batches
    ("YUV420P16", 256, 144, 384, 216, dict(tap=6), (1, 3, 40)),       # 1.5x tap 6: runs form, frame-lane border from 32 frames
    ("YUV444P8", 192, 108, 288, 162, {}, (1, 4, 70)),                 # 1.5x tap 3: quasi-periodic kernel / runs form / frame-lane
    ("RGBPS", 160, 90, 320, 180, dict(tap=4), (1, 6)),                # float planes
    ("YUV422P10", 256, 144, 128, 72, {}, (1, 7)),                     # down-scale: direct kernel, strips
    ("Y8", 300, 200, 600, 400, dict(tap=8), (1, 9)),                  # row-streamed kernel
    ("YUV420P8", 128, 96, 397, 301, {}, (1, 130)),                    # 3.1x: frame-pair form for the whole group of 128 + remainder
]

GUARD = 4096
SENTINEL = 0xA5


@pytest.mark.parametrize("pad", [0, 64, -256], ids=["tight", "padded", "pitch_and_frames_aligned_to_256"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_{c[1]}x{c[2]}to{c[3]}x{c[4]}")
def test_nothing_outside_the_rows_is_written(gpu_pkg, O, case, pad):
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th, kw, sizes = case
    ofmt, gfmt = O.FORMATS[fmt], gpu_pkg.FORMATS[fmt]
    f = gpu_pkg.Filter(gfmt, sw, sh, tw, th, device=0, **kw)
    sdims, ddims = ofmt.plane_dims(sw, sh), f.out_dims()
    frame = O.lcg_frame(ofmt, sw, sh, seed=99)
    sb = frame[0].dtype.itemsize
    # every batch size under the automatic border form and, for exactly periodic plans, with ewa_strip_kernel (3) and round 5's other border forms (4) forced
    for n, strips in [(n, st) for n in sizes for st in ((-1, 3, 4) if f.plan_info(0).periodic else (-1,))]:
        f.set_border_strips(strips)
        # source: frames of a plane back to back, pitch = row bytes rounded up to 4
        sp = [(w * sb + 3) // 4 * 4 for (w, h) in sdims]
        sfs = [p * h for p, (w, h) in zip(sp, sdims)]
        srcs = []
        for i, (w, h) in enumerate(sdims):
            plane = np.zeros((h, sp[i]), np.uint8)
            plane[:, :w * sb] = np.ascontiguousarray(frame[i][:h, :w]).view(np.uint8).reshape(h, w * sb)
            srcs.append(to_device(torch.from_numpy(np.tile(plane.reshape(-1), n))))
        # destination: [guard] plane 0 frames [guard] plane 1 frames [guard] ...
        if pad >= 0:
            dp = [(w * sb + 3) // 4 * 4 + pad for (w, h) in ddims]
            dfs = [p * h for p, (w, h) in zip(dp, ddims)]
        else:   # the look-ahead pipeline's own layout: vector stores of every width are possible
            dp = [(w * sb + 255) // 256 * 256 for (w, h) in ddims]
            dfs = [(p * h + 255) // 256 * 256 for p, (w, h) in zip(dp, ddims)]
        offs, at = [], GUARD
        for i in range(len(ddims)):
            offs.append(at)
            at += dfs[i] * n + GUARD
        dst = torch.full((at,), SENTINEL, dtype=torch.uint8, device="cuda")
        stream = torch.cuda.current_stream()
        f.process_device([s.data_ptr() for s in srcs], sp, sfs, [dst.data_ptr() + o for o in offs], dp, dfs, n, stream=stream.cuda_stream)
        stream.synchronize()
        out = to_host(dst).numpy()
        first = None
        for i, (w, h) in enumerate(ddims):
            region = np.stack([out[offs[i] + k * dfs[i]:offs[i] + k * dfs[i] + h * dp[i]].reshape(h, dp[i]) for k in range(n)])
            for k in range(n):
                assert np.all(out[offs[i] + k * dfs[i] + h * dp[i]:offs[i] + (k + 1) * dfs[i]] == SENTINEL), f"{n} frames, plane {i}: bytes behind frame {k}"
            if first is None:
                first = [None] * len(ddims)
            assert np.all(region[:, :, w * sb:] == SENTINEL), f"{n} frames, plane {i}: row padding overwritten ({f.last_kernel(0)})"
            for k in range(1, n):   # every frame of the batch had the same source
                assert np.array_equal(region[k, :, :w * sb], region[0, :, :w * sb]), f"{n} frames, plane {i}, frame {k}"
        guards = [(0, GUARD)] + [(offs[i] + dfs[i] * n, offs[i] + dfs[i] * n + GUARD) for i in range(len(ddims))]
        for a, b in guards:
            assert np.all(out[a:b] == SENTINEL), f"{n} frames: guard bytes [{a}, {b}) overwritten ({f.last_kernel(0)})"
    f.close()
