"""BASELINE.json configs[4] ("C5"): a batch of 512 independent 1080p -> 4K Y8 tap=3 frames through the library's
multi-device sharder (jinc_batch_*: frame n -> device n mod G, per-device streams, no collective), on however many
devices are visible.  Every frame's output is checked against the CPU oracle by crc32; frame 0 is the Appendix-A frame
whose crc32 is one of the reference's known answers."""
import json
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def _crc(plane, w, h):
    return zlib.crc32(np.ascontiguousarray(plane[:h, :w]).tobytes()) & 0xFFFFFFFF


def test_512_frames_sharded_over_the_visible_devices(gpu_pkg, O):
    sw, sh, tw, th, n = 1920, 1080, 3840, 2160, 512
    ofmt = O.FORMATS["Y8"]
    frames = [O.lcg_frame(ofmt, sw, sh, seed=12345 + k) for k in range(n)]   # seeds 12345 .. 12856
    b = gpu_pkg.Batch(gpu_pkg.FORMATS["Y8"], sw, sh, tw, th, ndevices=0, streams=3, tap=3)
    ndev = b.devices
    assert ndev == gpu_pkg.device_count() >= 1
    assert [b.device_of_frame(k) for k in range(2 * ndev)] == [k % ndev for k in range(2 * ndev)]
    outs = b.process(frames)
    b.close()
    of = O.OracleFilter(ofmt, sw, sh, tw, th, tap=3)
    bad = []
    for k in range(n):
        want = of.get_frame(frames[k], threads=16)[0]
        if _crc(outs[k][0], tw, th) != _crc(want, tw, th):
            bad.append(k)
    assert not bad, f"frames differ from the oracle: {bad[:10]} ({len(bad)} of {n})"
    c2 = next(k for k in KAT["outputs"] if k["name"].startswith("C2"))
    assert f"{_crc(outs[0][0], tw, th):08x}" == c2["crc32"]   # the reference's own opt=0 crc32 for seed 12345
    assert len({_crc(o[0], tw, th) for o in outs}) == n        # 512 distinct inputs -> 512 distinct outputs


def test_small_batches_and_stream_counts(gpu_pkg, O):
    """Batch sizes around the number of frames in flight (empty, fewer than streams, not a multiple), 4:2:0 planes."""
    fmt, sw, sh, tw, th = "YUV420P8", 160, 96, 320, 192
    ofmt = O.FORMATS[fmt]
    of = O.OracleFilter(ofmt, sw, sh, tw, th)
    for streams, n in ((1, 3), (2, 0), (2, 1), (3, 7), (16, 5)):
        b = gpu_pkg.Batch(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, streams=streams)   # (pageable frames, the default: registering modes are tests/test_pin_modes.py's subject)
        frames = [O.lcg_frame(ofmt, sw, sh, seed=7 + k) for k in range(n)]
        outs = b.process(frames)
        for k in range(n):
            want = of.get_frame(frames[k], threads=4)
            for i, (w, h) in enumerate(b.out_dims()):
                assert np.array_equal(outs[k][i][:h, :w], want[i][:h, :w]), (streams, n, k, i)
        b.close()
