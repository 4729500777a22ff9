"""Pins the CPU oracle to the reference: every known-answer vector SURVEY.md 8(c) recorded from
executing the reference's opt=0 code, and the 22 outputs + 16 LUT hashes the round-5 judge recorded from its own run of
the reference (Next 3: crops, sitings, 4:2:2 / 4:1:1 / 4:4:4 / alpha / RGBA, 10- and 14-bit, float, quant,
down-scales, taps 5, 6, 7, 12, 16) -- tests/golden/kat.json.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


@pytest.mark.parametrize("k", KAT["lut_samples"], ids=lambda k: f"tap{k['tap']}_blur{k['blur']}")
def test_lut_samples(O, k):
    lut = O.make_lut(k["tap"], k["blur"])
    assert lut[0] == k["lut0"] and lut[1023] == k["lut1023"]
    assert lut[512] == k["lut512"]  # exact double


def test_lut_is_deterministic_and_windowed(O):
    for tap in range(1, 17):
        lut = O.make_lut(tap, 1.0)
        assert lut[0] == 1.0 and lut[1023] == 0.0
        assert np.all(np.isfinite(lut)) and np.array_equal(lut, O.make_lut(tap, 1.0))


@pytest.mark.parametrize("k", KAT["table_stats"], ids=lambda k: f"{k['src'][0]}x{k['src'][1]}to{k['dst'][0]}x{k['dst'][1]}_tap{k['tap']}")
def test_table_stats(O, k):
    lut = O.make_lut(k["tap"], k.get("blur", 1.0))
    t = O.Table(lut, quant_x=256, quant_y=256, src_w=k["src"][0], src_h=k["src"][1], dst_w=k["dst"][0],
                dst_h=k["dst"][1], radius=O.jinc_zero(k["tap"]), crop_left=0.0, crop_top=0.0,
                crop_width=float(k["src"][0]), crop_height=float(k["src"][1]))
    assert (t.filter_size, t.num_sets, t.cached_phases) == (k["filter_size"], k["sets"], k["cached_phases"])
    assert t.coeff_stride == (t.filter_size + 15) // 16 * 16


@pytest.mark.parametrize("k", KAT["outputs"], ids=lambda k: k["name"])
def test_output_kat(O, k):
    """crc32 (and sha256 prefix) of the oracle's output == the reference's own opt=0 output."""
    fmt = O.FORMATS[k["format"]]
    flt = O.OracleFilter(fmt, k["src"][0], k["src"][1], k["dst"][0], k["dst"][1], **k["args"])
    src = O.lcg_frame(fmt, *k["src"])
    out = flt.get_frame(src, threads=4)  # row-parallel like the reference's PSTL fan-out; same arithmetic
    dims = flt.out_dims()
    assert sum(w * h for w, h in dims) * fmt.sample_bytes == k["bytes"]
    assert O.crc32_planes(out, dims) == k["crc32"]
    if "sha256_prefix" in k:
        h = hashlib.sha256()
        for p, (w, hh) in zip(out, dims):
            h.update(np.ascontiguousarray(p[:hh, :w]).tobytes())
        assert h.hexdigest().startswith(k["sha256_prefix"])
    # table statistics of the same run
    stats = {(tuple(s["src"]), tuple(s["dst"]), s["tap"]): s for s in KAT["table_stats"]}
    s = stats.get((tuple(k["src"]), tuple(k["dst"]), k["args"]["tap"]))
    if s:
        t = flt.tables[0]
        assert (t.filter_size, t.num_sets, t.cached_phases) == (s["filter_size"], s["sets"], s["cached_phases"])


def fnv1a64(data: bytes) -> str:
    h = 0xCBF29CE484222325
    for b in data:
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


@pytest.mark.parametrize("k", KAT["lut_fnv1a64"], ids=lambda k: f"tap{k['tap']}")
def test_lut_bytes_are_the_references(O, pkg, k):
    """FNV-1a-64 over the 8192 LUT bytes == the reference's Lut::InitLut (judge-side run), every tap 1..16 -- the oracle's
    LUT and the product's (csrc/jinc_lut.cpp, read back through the test header; no device needed)."""
    assert fnv1a64(np.asarray(O.make_lut(k["tap"], k["blur"]), dtype=np.float64).tobytes()) == k["fnv1a64"]
    f = pkg.Filter(pkg.FORMATS["Y8"], 256, 256, 512, 512, device=-1, tap=k["tap"], blur=k["blur"])
    try:
        assert fnv1a64(np.asarray(f.lut(), dtype=np.float64).tobytes()) == k["fnv1a64"]
    finally:
        f.close()


@pytest.mark.parametrize("k", KAT["outputs_r5"], ids=lambda k: k["name"])
def test_output_kat_recorded_by_the_round5_judge(O, k):
    """crc32 of the oracle's output == the reference's opt=0 output on the seed-777 LCG frame (Next 3)."""
    from conftest import oracle_kwargs
    fmt = O.FORMATS[k["format"]]
    flt = O.OracleFilter(fmt, k["src"][0], k["src"][1], k["dst"][0], k["dst"][1], **oracle_kwargs(k["args"]))
    out = flt.get_frame(O.lcg_frame(fmt, *k["src"], seed=k["seed"]), threads=4)
    dims = flt.out_dims()
    assert sum(w * h for w, h in dims) * fmt.sample_bytes == k["bytes"]
    assert O.crc32_planes(out, dims) == k["crc32"]


def test_threads_do_not_change_results(O):
    fmt = O.FORMATS["YUV420P16"]
    flt = O.OracleFilter(fmt, 96, 64, 200, 130, tap=4)
    src = O.lcg_frame(fmt, 96, 64, seed=7)
    a = flt.get_frame(src, threads=1)
    b = flt.get_frame(src, threads=3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_numpy_restatement_of_frame_loop(O):
    """Independent check of oracle_resize_plane: a numpy float32 restatement of ref :570-584 on a tiny case."""
    fmt = O.FORMATS["Y8"]
    flt = O.OracleFilter(fmt, 24, 20, 50, 44, tap=3)
    src = O.lcg_frame(fmt, 24, 20, seed=99)
    out = flt.get_frame(src)[0]
    t = flt.tables[0]
    meta, fs = t.meta(), t.filter_size
    for (x, y) in [(0, 0), (49, 43), (25, 20), (3, 40), (48, 1), (17, 17)]:
        sx, sy, _ = meta[y, x]
        c = t.coeff_set(x, y)
        r = np.float32(0)
        for ly in range(fs):
            for lx in range(fs):
                r = np.float32(r + np.float32(np.float32(src[0][sy + ly, sx + lx]) * c[ly, lx]))
        r = min(max(r, np.float32(0)), np.float32(255))
        assert int(np.rint(r)) == int(out[y, x])
