"""The AviSynth+ plugin shell (plugin/jincresize_avs.cpp) driven by a miniature host
(tests/mock_avs/): registration surface, argument parsing by position and by name, alias functions re-entering
JincResize through avs_invoke, error strings, _ChromaLocation handling, MT mode, frame ownership, look-ahead.
The mock API header is self-written (the image has no AviSynth SDK): these tests prove the shell's logic, not binary
compatibility with a real host."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_avs")
LIBDIR = os.path.join(ROOT, "avisynth-jincresize_amd", "lib")


@pytest.fixture(scope="session")
def host(pkg):
    """Builds plugin + mock host into one shared library (g++, seconds) and loads it."""
    out_dir = os.path.join(MOCK, "build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libmock_avs_plugin.so")
    srcs = [os.path.join(ROOT, "plugin", "jincresize_avs.cpp"), os.path.join(MOCK, "mock_host.cpp")]
    COMPAT = os.path.join(ROOT, "plugin", "compat")
    deps = srcs + [os.path.join(COMPAT, "avisynth_c.h"), os.path.join(ROOT, "include", "jincresize_hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        cmd = ["g++", "-std=c++17", "-shared", "-fPIC", "-O1", "-Wall", "-Wextra", "-Wno-unused-parameter", "-fvisibility=hidden",
               "-I" + COMPAT, "-I" + os.path.join(ROOT, "include"), *srcs, "-L" + LIBDIR, "-ljincresize_hip",
               "-Wl,-rpath," + LIBDIR, "-o", so]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    L = C.CDLL(so)
    L.mock_env_new.restype = C.c_void_p
    L.mock_env_new.argtypes = [C.c_int, C.c_int, C.c_int]
    L.mock_env_free.argtypes = [C.c_void_p]
    L.mock_load_plugin.restype = C.c_char_p
    L.mock_load_plugin.argtypes = [C.c_void_p]
    L.mock_function_count.argtypes = [C.c_void_p]
    L.mock_function_name.restype = C.c_char_p
    L.mock_function_name.argtypes = [C.c_void_p, C.c_int]
    L.mock_function_params.restype = C.c_char_p
    L.mock_function_params.argtypes = [C.c_void_p, C.c_int]
    L.mock_live_frames.restype = C.c_long
    L.mock_live_frames.argtypes = [C.c_void_p]
    L.mock_live_clips.restype = C.c_long
    L.mock_live_clips.argtypes = [C.c_void_p]
    L.mock_source_new.restype = C.c_void_p
    L.mock_source_new.argtypes = [C.c_void_p] + [C.c_int] * 12
    L.mock_source_frame.restype = C.c_void_p
    L.mock_source_frame.argtypes = [C.c_void_p, C.c_int]
    L.mock_source_get_frame_calls.argtypes = [C.c_void_p]
    L.mock_source_calls_of_frame.argtypes = [C.c_void_p, C.c_int]
    L.mock_env_set_frame_pool.argtypes = [C.c_void_p, C.c_int]
    L.mock_env_pool_reuses.restype = C.c_long
    L.mock_env_pool_reuses.argtypes = [C.c_void_p]
    L.mock_frame_plane.restype = C.c_void_p
    L.mock_frame_plane.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mock_invoke.restype = C.c_void_p
    L.mock_invoke.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_char_p,
                              C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
    L.mock_value_error.restype = C.c_char_p
    L.mock_value_error.argtypes = [C.c_void_p]
    L.mock_value_clip.restype = C.c_void_p
    L.mock_value_clip.argtypes = [C.c_void_p]
    L.mock_value_free.argtypes = [C.c_void_p]
    L.mock_clip_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mock_clip_get_frame.restype = C.c_void_p
    L.mock_clip_get_frame.argtypes = [C.c_void_p, C.c_int]
    L.mock_clip_error.restype = C.c_char_p
    L.mock_clip_error.argtypes = [C.c_void_p]
    L.mock_clip_mt_mode.argtypes = [C.c_void_p]
    L.mock_clip_release.argtypes = [C.c_void_p]
    L.mock_source_release.argtypes = [C.c_void_p]
    L.mock_frame_prop_int.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
    L.mock_frame_release.argtypes = [C.c_void_p]
    return L


class Host:
    """One script environment with the plugin loaded."""

    def __init__(self, L, version=10, bugfix=0, cpu_flags=-1):
        self.L = L
        self.env = L.mock_env_new(version, bugfix, cpu_flags)
        self.description = L.mock_load_plugin(self.env).decode()

    def functions(self):
        return {self.L.mock_function_name(self.env, i).decode(): self.L.mock_function_params(self.env, i).decode()
                for i in range(self.L.mock_function_count(self.env))}

    def source(self, fmt, w, h, frames, chroma_location=-1, pitch_align=64):
        """Source clip filled with the oracle's LCG frames (seed 12345 + n); `frames` = list of plane lists."""
        clip = self.L.mock_source_new(self.env, w, h, fmt.bits, fmt.sample_bytes, fmt.planes, 1, int(fmt.rgb), fmt.sub_w, fmt.sub_h,
                                      len(frames), chroma_location, pitch_align)
        for n, planes in enumerate(frames):
            fr = self.L.mock_source_frame(clip, n)
            for i, p in enumerate(planes):
                self.write_plane(fr, i, p)
        return clip

    def write_plane(self, frame, index, arr):
        pitch, row, hh = C.c_int(), C.c_int(), C.c_int()
        ptr = self.L.mock_frame_plane(frame, index, C.byref(pitch), C.byref(row), C.byref(hh))
        view = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_ubyte)), shape=(hh.value, pitch.value))
        raw = np.ascontiguousarray(arr[:hh.value]).view(np.uint8).reshape(hh.value, -1)
        view[:, :row.value] = raw[:, :row.value]

    def read_plane(self, frame, index, dtype):
        pitch, row, hh = C.c_int(), C.c_int(), C.c_int()
        ptr = self.L.mock_frame_plane(frame, index, C.byref(pitch), C.byref(row), C.byref(hh))
        view = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_ubyte)), shape=(hh.value, pitch.value))
        return np.ascontiguousarray(view[:, :row.value]).view(dtype).copy()

    def invoke(self, name, clip, tw, th, **named):
        keys = list(named)
        n = len(keys)
        names = (C.c_char_p * max(1, n))(*[k.encode() for k in keys])
        kinds = bytearray()
        ivals, fvals, svals = (C.c_int * max(1, n))(), (C.c_double * max(1, n))(), (C.c_char_p * max(1, n))()
        for i, k in enumerate(keys):
            v = named[k]
            if isinstance(v, str):
                kinds += b"s"
                svals[i] = v.encode()
            elif isinstance(v, float):
                kinds += b"f"
                fvals[i] = v
            else:
                kinds += b"i"
                ivals[i] = int(v)
        self._keep = (names, svals)
        val = self.L.mock_invoke(self.env, name.encode(), clip, tw, th, n, names, bytes(kinds), ivals, fvals, svals)
        err = self.L.mock_value_error(val)
        out = (None, err.decode()) if err is not None else (self.L.mock_value_clip(val), None)
        self.L.mock_value_free(val)
        return out

    def prop(self, frame, key):
        v = C.c_longlong()
        return v.value if self.L.mock_frame_prop_int(frame, key.encode(), C.byref(v)) else None

    def close(self):
        self.L.mock_env_free(self.env)


def test_registration_surface(host):
    """The five script functions with the reference's parameter strings (ref :1044-1108) and description (:1110)."""
    h = Host(host)
    assert h.description == "JincResize"
    f = h.functions()
    assert f["JincResize"] == ("cii[src_left]f[src_top]f[src_width]f[src_height]f[quant_x]i[quant_y]i[tap]i[blur]f[cplace]s"
                               "[threads]i[opt]i[initial_capacity]i[initial_factor]f")
    alias = "cii[src_left]f[src_top]f[src_width]f[src_height]f[quant_x]i[quant_y]i[cplace]s[threads]i"
    assert {k: v for k, v in f.items() if k != "JincResize"} == {n: alias for n in
                                                                  ("Jinc36Resize", "Jinc64Resize", "Jinc144Resize", "Jinc256Resize")}
    h.close()


ERRORS = [
    (dict(tap=0), "JincResize: tap must be between 1..16."),
    (dict(tap=17), "JincResize: tap must be between 1..16."),
    (dict(quant_x=0), "JincResize: quant_x must be between 1..256."),
    (dict(quant_y=257), "JincResize: quant_y must be between 1..256."),
    (dict(cplace="left"), "JincResize: cplace must be MPEG2, MPEG1 or topleft."),
    (dict(opt=4), "JincResize: opt higher than 3 is not allowed."),
    (dict(threads=2), "JincResize: threads must be either 0 or 1."),
    (dict(initial_factor=0.5), "JincResize: initial_factor must be eqaul to or greater than 1.0."),
    (dict(initial_capacity=0), "JincResize: initial_capacity must be greater than 0."),
]


@pytest.mark.parametrize("named,message", ERRORS, ids=["_".join(f"{k}{v}" for k, v in n.items()) for n, _ in ERRORS])
def test_create_errors_reach_the_host_with_the_reference_text(host, O, named, message):
    h = Host(host)
    fmt = O.FORMATS["YUV420P8"]
    src = h.source(fmt, 64, 48, [O.lcg_frame(fmt, 64, 48)])
    clip, err = h.invoke("JincResize", src, 128, 96, **named)
    assert clip is None and err == message
    assert host.mock_live_clips(h.env) == 1   # the failed filter released its clip (ref :682-687); the source remains
    host.mock_source_release(src)
    h.close()


def test_interface_version_gate(host, O):
    """ref :689-698: interface 9 needs bug-fix level 2 (r3688); 8 is refused; 10 passes."""
    fmt = O.FORMATS["Y8"]
    for version, bugfix, ok in ((8, 9, False), (9, 1, False), (9, 2, True), (10, 0, True)):
        h = Host(host, version, bugfix)
        src = h.source(fmt, 64, 48, [O.lcg_frame(fmt, 64, 48)])
        clip, err = h.invoke("JincResize", src, 128, 96)
        if ok:
            assert err is None
            host.mock_clip_release(clip)
        else:
            assert err == "JincResize: AviSynth+ version must be r3688 or later."
        host.mock_source_release(src)
        h.close()


def test_opt_needs_the_cpu_flag_of_the_host(host, O):
    fmt = O.FORMATS["Y8"]
    h = Host(host, cpu_flags=0)
    src = h.source(fmt, 64, 48, [O.lcg_frame(fmt, 64, 48)])
    for opt, text in ((3, "JincResize: opt=3 requires AVX-512F."), (2, "JincResize: opt=2 requires AVX2."),
                      (1, "JincResize: opt=1 requires SSE4.1.")):
        clip, err = h.invoke("JincResize", src, 128, 96, opt=opt)
        assert clip is None and err == text
    host.mock_source_release(src)
    h.close()


def test_non_planar_clip_is_refused(host):
    h = Host(host)
    clip = host.mock_source_new(h.env, 64, 48, 8, 1, 1, 0, 0, 0, 0, 1, -1, 64)   # planar = 0
    out, err = h.invoke("JincResize", clip, 128, 96)
    assert out is None and err == "JincResize: clip must be in planar format."
    host.mock_source_release(clip)
    h.close()


def test_filter_object_without_a_gpu_or_with_one(host, O, pkg):
    """Creation succeeds, the output clip has the target size, MT mode is MULTI_INSTANCE (ref :649-652); frame calls
    either work (GPU box) or fail loudly through fi->error -- there is no CPU path to fall back to."""
    h = Host(host)
    fmt = O.FORMATS["YUV420P8"]
    src = h.source(fmt, 64, 48, [O.lcg_frame(fmt, 64, 48)])
    clip, err = h.invoke("JincResize", src, 160, 120, tap=4)
    assert err is None
    w, hh, n = C.c_int(), C.c_int(), C.c_int()
    host.mock_clip_info(clip, C.byref(w), C.byref(hh), C.byref(n))
    assert (w.value, hh.value, n.value) == (160, 120, 1)
    assert host.mock_clip_mt_mode(clip) == 2
    frame = host.mock_clip_get_frame(clip, 0)
    if pkg.device_count() == 0:
        assert b"HIP device" in host.mock_clip_error(clip)
        assert not frame          # the error AND no frame: AviSynth+ throws on fi->error and would never release one
    else:
        assert host.mock_clip_error(clip) is None
        host.mock_frame_release(frame)
    assert host.mock_clip_get_frame(clip, 5) is None          # child has no such frame -> null (ref :610-611)
    host.mock_clip_release(clip)
    host.mock_source_release(src)
    assert host.mock_live_clips(h.env) == 0 and host.mock_live_frames(h.env) == 0
    h.close()


GPU_CASES = [
    ("Y8", 96, 64, 192, 128, "JincResize", {}, None),
    ("YUV420P8", 128, 96, 256, 192, "JincResize", dict(tap=4, cplace="topleft"), 2),
    ("YUV420P16", 128, 96, 200, 150, "JincResize", dict(src_left=1.5, src_top=-0.5, src_width=120.0, src_height=90.0, quant_x=64), 2),
    ("YUV422P10", 128, 96, 256, 192, "Jinc36Resize", dict(cplace="MPEG1"), 2),
    ("RGBPS", 96, 64, 192, 128, "Jinc64Resize", {}, None),
    ("YUV444P8", 96, 64, 48, 32, "Jinc144Resize", dict(quant_y=32), None),
    ("YUVA420P8", 128, 96, 256, 192, "Jinc256Resize", {}, 2),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=lambda c: f"{c[5]}_{c[0]}")
def test_frames_through_the_plugin_match_the_oracle(host, O, case):
    """Script call -> plugin -> C ABI -> GPU -> frame in the host's buffers, against the oracle; the alias functions
    arrive in JincResize with tap 3/4/6/8; _ChromaLocation = 2 is written for sub-sampled formats only, whatever the siting -- what the reference binary does (ref :617-625; d->cplace is never assigned)."""
    fmt_name, sw, sh, tw, th, fn, named, want_loc = case
    fmt = O.FORMATS[fmt_name]
    frames = [O.lcg_frame(fmt, sw, sh, seed=12345 + n) for n in range(2)]
    h = Host(host)
    src = h.source(fmt, sw, sh, frames)
    clip, err = h.invoke(fn, src, tw, th, **named)
    assert err is None, err
    kw = dict(named)
    kw.update({"Jinc36Resize": dict(tap=3), "Jinc64Resize": dict(tap=4), "Jinc144Resize": dict(tap=6), "Jinc256Resize": dict(tap=8)}.get(fn, {}))
    of = O.OracleFilter(fmt, sw, sh, tw, th, **oracle_kwargs(kw))
    dtype = {1: np.uint8, 2: np.uint16, 4: np.float32}[fmt.sample_bytes]
    for n in range(2):
        fr = host.mock_clip_get_frame(clip, n)
        assert host.mock_clip_error(clip) is None
        got = [h.read_plane(fr, i, dtype) for i in range(fmt.planes)]
        want = of.get_frame(frames[n], threads=4)
        assert_planes_equal(got, want, fmt.plane_dims(tw, th), what=f"{fn} {fmt_name} frame {n}")
        assert h.prop(fr, "_ChromaLocation") == want_loc
        host.mock_frame_release(fr)
    host.mock_clip_release(clip)
    host.mock_source_release(src)
    assert host.mock_live_clips(h.env) == 0 and host.mock_live_frames(h.env) == 0
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("by_siting", [False, True], ids=["as_reference", "by_siting"])
def test_chroma_location_of_frame_zero_decides_when_cplace_is_not_given(host, O, monkeypatch, by_siting):
    """ref :727-742: _ChromaLocation 0/1/2 of the first frame selects mpeg2/mpeg1/topleft (the pixels show it); other values
    are an error.  The property written is 2 in every case, as the reference binary does (ref :617-625, d->cplace never
    assigned); JINCRESIZE_CHROMALOC=siting writes the siting in use instead."""
    if by_siting:
        monkeypatch.setenv("JINCRESIZE_CHROMALOC", "siting")
    else:
        monkeypatch.delenv("JINCRESIZE_CHROMALOC", raising=False)
    fmt = O.FORMATS["YUV420P8"]
    frames = [O.lcg_frame(fmt, 128, 96)]
    for loc, cplace in ((0, "mpeg2"), (1, "mpeg1"), (2, "topleft")):
        h = Host(host)
        src = h.source(fmt, 128, 96, frames, chroma_location=loc)
        clip, err = h.invoke("JincResize", src, 256, 192)
        assert err is None
        fr = host.mock_clip_get_frame(clip, 0)
        got = [h.read_plane(fr, i, np.uint8) for i in range(3)]
        want = O.OracleFilter(fmt, 128, 96, 256, 192, cplace=cplace).get_frame(frames[0], threads=4)
        assert_planes_equal(got, want, fmt.plane_dims(256, 192), what=f"_ChromaLocation {loc}")
        assert h.prop(fr, "_ChromaLocation") == (loc if by_siting else 2)
        host.mock_frame_release(fr)
        host.mock_clip_release(clip)
        host.mock_source_release(src)
        h.close()
    h = Host(host)
    src = h.source(fmt, 128, 96, frames, chroma_location=4)
    clip, err = h.invoke("JincResize", src, 256, 192)
    assert clip is None and err == "JincResize: invalid _ChromaLocation"
    host.mock_source_release(src)
    h.close()


@pytest.mark.gpu
def test_lookahead_ring_with_skipped_frames_does_not_leak(host, O, monkeypatch):
    """A client that skips frames (SelectEven: n, n+2, n+4, ...) leaves look-ahead frames unconsumed; they must be waited
    for and released before their ring slot is reused (two frame references leaked per skipped frame)."""
    monkeypatch.setenv("JINCRESIZE_LOOKAHEAD", "3")
    fmt = O.FORMATS["Y8"]
    nframes = 12
    frames = [O.lcg_frame(fmt, 64, 48, seed=900 + n) for n in range(nframes)]
    of = O.OracleFilter(fmt, 64, 48, 128, 96)
    h = Host(host)
    src = h.source(fmt, 64, 48, frames)
    clip, err = h.invoke("JincResize", src, 128, 96)
    assert err is None
    for n in (0, 2, 4, 5, 9, 11, 3):
        fr = host.mock_clip_get_frame(clip, n)
        assert host.mock_clip_error(clip) is None
        got = [h.read_plane(fr, 0, np.uint8)]
        assert_planes_equal(got, of.get_frame(frames[n], threads=2), fmt.plane_dims(128, 96), what=f"frame {n}")
        host.mock_frame_release(fr)
    host.mock_clip_release(clip)
    host.mock_source_release(src)
    assert host.mock_live_clips(h.env) == 0 and host.mock_live_frames(h.env) == 0
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [2, 3])
def test_lookahead_ring_sequential_and_seek(host, O, depth, monkeypatch):
    """JINCRESIZE_LOOKAHEAD=k keeps k child frames in flight; every frame is still the oracle's, child
    frames are requested once each when access is sequential, a jump moves the window (frames the new window does not
    cover are dropped), nothing leaks; with look-ahead on the filter answers MT_SERIALIZED (one instance sees the clip)."""
    monkeypatch.setenv("JINCRESIZE_LOOKAHEAD", str(depth))
    fmt = O.FORMATS["YUV420P8"]
    nframes = 7
    frames = [O.lcg_frame(fmt, 96, 64, seed=500 + n) for n in range(nframes)]
    of = O.OracleFilter(fmt, 96, 64, 192, 128)
    h = Host(host)
    src = h.source(fmt, 96, 64, frames)
    clip, err = h.invoke("JincResize", src, 192, 128)
    assert err is None
    assert host.mock_clip_mt_mode(clip) == 3
    order = list(range(nframes)) + [2, 5, 6, 0]
    for step, n in enumerate(order):
        fr = host.mock_clip_get_frame(clip, n)
        assert host.mock_clip_error(clip) is None
        got = [h.read_plane(fr, i, np.uint8) for i in range(3)]
        assert_planes_equal(got, of.get_frame(frames[n], threads=4), fmt.plane_dims(192, 128), what=f"frame {n} depth {depth}")
        assert h.prop(fr, "_ChromaLocation") == 2   # what the reference binary writes for 4:2:0 (ref :617-625)
        host.mock_frame_release(fr)
        if step == nframes - 1:
            assert host.mock_source_get_frame_calls(src) == nframes + 1   # + the frame-0 property probe at create time
    host.mock_clip_release(clip)
    host.mock_source_release(src)
    assert host.mock_live_clips(h.env) == 0 and host.mock_live_frames(h.env) == 0
    h.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case,group,kernel", [
    (("Y8", 1280, 720, 1754, 986, "JincResize", {}), None, "ewa_framelane_sub_kernel"),   # no phase structure, fs 7: groups of 16 = four sub-groups per wave
    (("Y8", 1280, 720, 1920, 1080, "Jinc256Resize", {}), 32, "ewa_direct_runs_kernel"),   # 1.5x tap 8 (fs 17): one launch per group
], ids=["A137_lookahead32", "N15T8_lookahead32_group32"])
def test_lookahead_32_reaches_the_batch_kernels_through_get_frame(host, O, pkg, case, group, kernel, monkeypatch):
    """with JINCRESIZE_LOOKAHEAD=32 the plugin's per-frame GetFrame (ref :603-630) is served by coalesced
    launches -- the frame-lane kernels -- and every frame is still exactly that frame's result."""
    monkeypatch.setenv("JINCRESIZE_LOOKAHEAD", "32")
    monkeypatch.delenv("JINCRESIZE_PIN_FRAMES", raising=False)   # pageable frames, the default (pinned in place: tests/test_plugin_prefetch.py)
    if group:
        monkeypatch.setenv("JINCRESIZE_GROUP", str(group))
    fmt_name, sw, sh, tw, th, fn, named = case
    fmt = O.FORMATS[fmt_name]
    nframes = 64
    frames = [O.lcg_frame(fmt, sw, sh, seed=31000 + n) for n in range(nframes)]
    kw = dict(named)
    kw.update({"Jinc256Resize": dict(tap=8)}.get(fn, {}))
    single = pkg.Filter(pkg.FORMATS[fmt_name], sw, sh, tw, th, device=0, **kw)   # the same frames, one synchronous call each
    want = [single.get_frame(fr) for fr in frames]
    single.close()
    of = O.OracleFilter(fmt, sw, sh, tw, th, **oracle_kwargs(kw))
    h = Host(host)
    host.mock_env_set_frame_pool(h.env, 4096)   # the host recycles frame buffers
    src = h.source(fmt, sw, sh, frames)
    clip, err = h.invoke(fn, src, tw, th, **named)
    assert err is None, err
    seen = set()
    for n in range(nframes):
        fr = host.mock_clip_get_frame(clip, n)
        assert host.mock_clip_error(clip) is None
        seen.add(pkg.last_call())
        got = [h.read_plane(fr, 0, np.uint8)]
        assert_planes_equal(got, want[n], fmt.plane_dims(tw, th), what=f"{fn} frame {n} grouped vs single")
        if n in (0, 41):
            assert_planes_equal(got, of.get_frame(frames[n], threads=8), fmt.plane_dims(tw, th), what=f"{fn} frame {n} vs oracle")
        host.mock_frame_release(fr)
    assert host.mock_source_get_frame_calls(src) == nframes + 1   # each child frame once (+ the frame-0 property probe)
    assert all(name.startswith(kernel) and k == (group or 16) for name, k in seen), seen
    host.mock_clip_release(clip)
    host.mock_source_release(src)
    assert host.mock_live_clips(h.env) == 0 and host.mock_live_frames(h.env) == 0
    h.close()


@pytest.mark.gpu
def test_script_floats_arrive_as_32_bit_values(host, O, pkg):
    """AviSynth hands script floats over as 32-bit values (AVS_Value.d.floating_pt; the reference reads them with avs_as_float,
    ref :762-774): JincResize(blur=0.98) computes with (double)(float)0.98 = 0.98000001907..., not with 0.98.  On float planes
    the two differ in a large share of the output samples, so the plugin path must match the oracle fed the float32-rounded
    arguments -- and an ABI caller that passes the double 0.98 gets the double's result."""
    fmt_name, sw, sh, tw, th = "RGBPS", 96, 64, 150, 100
    fmt = O.FORMATS[fmt_name]
    frames = [O.lcg_frame(fmt, sw, sh, seed=77)]
    named = dict(blur=0.98, src_left=0.1, src_top=0.3, src_width=90.7, src_height=60.1, tap=4)
    as_f32 = {k: (float(np.float32(v)) if isinstance(v, float) else v) for k, v in named.items()}
    assert as_f32["blur"] != named["blur"]
    want32 = O.OracleFilter(fmt, sw, sh, tw, th, **oracle_kwargs(as_f32)).get_frame(frames[0], threads=4)
    want64 = O.OracleFilter(fmt, sw, sh, tw, th, **oracle_kwargs(named)).get_frame(frames[0], threads=4)
    differing = sum(int(np.count_nonzero(a[:th, :tw].view(np.uint32) != b[:th, :tw].view(np.uint32))) for a, b in zip(want32, want64))
    assert differing > 0.05 * 3 * tw * th, "the case does not tell float32 arguments from doubles"
    h = Host(host)
    src = h.source(fmt, sw, sh, frames)
    clip, err = h.invoke("JincResize", src, tw, th, **named)
    assert err is None, err
    fr = host.mock_clip_get_frame(clip, 0)
    assert host.mock_clip_error(clip) is None
    got = [h.read_plane(fr, i, np.float32) for i in range(3)]
    assert_planes_equal(got, want32, fmt.plane_dims(tw, th), what="plugin path: float32-rounded script arguments")
    host.mock_frame_release(fr)
    host.mock_clip_release(clip)
    host.mock_source_release(src)
    h.close()
    # the C ABI takes doubles as they are
    f = pkg.Filter(pkg.FORMATS[fmt_name], sw, sh, tw, th, device=0, **named)
    assert_planes_equal(f.get_frame(frames[0]), want64, f.out_dims(), what="C ABI: double arguments")
    f.close()
    f = pkg.Filter(pkg.FORMATS[fmt_name], sw, sh, tw, th, device=0, **as_f32)
    assert_planes_equal(f.get_frame(frames[0]), want32, f.out_dims(), what="C ABI: float32-rounded arguments")
    f.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cpu_flags,opt,order", [(0x400 | 0x2000, None, 2), (0x400, None, 1), (0, None, 0), (0x400 | 0x2000, 0, 0),
                                                 (0x400 | 0x2000, 1, 1), (0x400 | 0x2000 | 0x10000, 3, 3)],
                         ids=["default_on_avx2_host", "default_on_sse41_host", "default_on_plain_host", "opt0", "opt1", "opt3"])
def test_simd_order_auto_gives_the_pixels_the_reference_ladder_would(host, O, monkeypatch, cpu_flags, opt, order):
    """JINCRESIZE_SIMD_ORDER=auto: the shell maps `opt` and the host's CPU flags the way the reference's
    ladder does (ref :897-899; its DEFAULT, opt=-1, is a SIMD path on any current CPU) onto jinc_filter_set_simd_order, so the
    frame is the reference binary's for that call; without the variable every call is the opt=0 result."""
    fmt = O.FORMATS["Y8"]
    sw, sh, tw, th = 640, 360, 1280, 720     # C1: opt 1 / 2 / 3 differ from opt 0 in 3 / 7 / 7 pixels
    frame = O.lcg_frame(fmt, sw, sh)
    of = O.OracleFilter(fmt, sw, sh, tw, th)
    want0 = of.get_frame(frame, threads=8)
    want = of.get_frame_simd(order, frame, threads=8) if order else want0
    named = {} if opt is None else dict(opt=opt)
    for env_value, expect in (("auto", want), (None, want0)):
        if env_value:
            monkeypatch.setenv("JINCRESIZE_SIMD_ORDER", env_value)
        else:
            monkeypatch.delenv("JINCRESIZE_SIMD_ORDER", raising=False)
        h = Host(host, cpu_flags=cpu_flags)
        src = h.source(fmt, sw, sh, [frame])
        clip, err = h.invoke("JincResize", src, tw, th, **named)
        assert err is None, err
        fr = host.mock_clip_get_frame(clip, 0)
        assert host.mock_clip_error(clip) is None
        assert_planes_equal([h.read_plane(fr, 0, np.uint8)], expect, fmt.plane_dims(tw, th), what=f"SIMD_ORDER={env_value} opt={opt}")
        host.mock_frame_release(fr)
        host.mock_clip_release(clip)
        host.mock_source_release(src)
        h.close()
    if order:
        assert int((want[0][:th, :tw] != want0[0][:th, :tw]).sum()) == {1: 3, 2: 7, 3: 7}[order]
