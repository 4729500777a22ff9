"""The VapourSynth front-end (plugin/jincresize_vs.cpp) driven by a miniature API-4 host (tests/mock_vs/):
registration surface (namespace, five functions with the reference's argument names), argument checking against the
registered signatures, the reference's create-time error texts through the shared C ABI, the alias functions, frame
requests through the two-step protocol, _ChromaLocation handling, inherited frame properties, reference counting.
The API header is self-written (the image has no VapourSynth SDK): these tests prove the shell's logic, not binary
compatibility with a real core."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import assert_planes_equal, oracle_kwargs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_vs")
LIBDIR = os.path.join(ROOT, "avisynth-jincresize_amd", "lib")


@pytest.fixture(scope="session")
def vs(pkg):
    """Builds front-end + mock host into one shared library (g++, seconds) and loads it."""
    out_dir = os.path.join(MOCK, "build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libmock_vs_plugin.so")
    srcs = [os.path.join(ROOT, "plugin", "jincresize_vs.cpp"), os.path.join(MOCK, "mock_vs_host.cpp")]
    compat = os.path.join(ROOT, "plugin", "compat")
    deps = srcs + [os.path.join(compat, "VapourSynth4.h"), os.path.join(ROOT, "include", "jincresize_hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        cmd = ["g++", "-std=c++17", "-shared", "-fPIC", "-O1", "-Wall", "-Wextra", "-Wno-unused-parameter", "-fvisibility=hidden",
               "-I" + compat, "-I" + os.path.join(ROOT, "include"), *srcs, "-L" + LIBDIR, "-ljincresize_hip", "-Wl,-rpath," + LIBDIR, "-o", so]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    L = C.CDLL(so)
    L.mockvs_new.restype = C.c_void_p
    L.mockvs_new.argtypes = [C.c_int]
    L.mockvs_free.argtypes = [C.c_void_p]
    for name in ("mockvs_plugin_namespace", "mockvs_plugin_identifier", "mockvs_last_error"):
        getattr(L, name).restype = C.c_char_p
        getattr(L, name).argtypes = [C.c_void_p]
    L.mockvs_plugin_api_version.argtypes = [C.c_void_p]
    L.mockvs_function_count.argtypes = [C.c_void_p]
    for name in ("mockvs_function_name", "mockvs_function_args", "mockvs_function_return"):
        getattr(L, name).restype = C.c_char_p
        getattr(L, name).argtypes = [C.c_void_p, C.c_int]
    L.mockvs_live_frames.restype = C.c_long
    L.mockvs_live_frames.argtypes = [C.c_void_p]
    L.mockvs_live_nodes.restype = C.c_long
    L.mockvs_live_nodes.argtypes = [C.c_void_p]
    L.mockvs_source_new.restype = C.c_void_p
    L.mockvs_source_new.argtypes = [C.c_void_p] + [C.c_int] * 11
    L.mockvs_source_frame.restype = C.c_void_p
    L.mockvs_source_frame.argtypes = [C.c_void_p, C.c_int]
    L.mockvs_source_get_frame_calls.argtypes = [C.c_void_p]
    L.mockvs_node_release.argtypes = [C.c_void_p]
    L.mockvs_frame_plane.restype = C.c_void_p
    L.mockvs_frame_plane.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mockvs_frame_prop_int.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong)]
    L.mockvs_frame_release.argtypes = [C.c_void_p]
    L.mockvs_invoke.restype = C.c_void_p
    L.mockvs_invoke.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_char_p,
                                C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_char_p)]
    L.mockvs_node_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
    L.mockvs_get_frame.restype = C.c_void_p
    L.mockvs_get_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    return L


class Core:
    """One mock core with the plugin loaded."""

    def __init__(self, L, stride_align=64):
        self.L = L
        self.h = L.mockvs_new(stride_align)

    def functions(self):
        return {self.L.mockvs_function_name(self.h, i).decode(): (self.L.mockvs_function_args(self.h, i).decode(),
                                                                    self.L.mockvs_function_return(self.h, i).decode())
                for i in range(self.L.mockvs_function_count(self.h))}

    def source(self, fmt, w, h, frames, chroma_location=-1):
        family = 2 if fmt.rgb else (1 if fmt.planes == 1 else 3)
        node = self.L.mockvs_source_new(self.h, w, h, family, int(fmt.bits == 32), fmt.bits, fmt.sample_bytes, fmt.sub_w, fmt.sub_h, fmt.planes,
                                        len(frames), chroma_location)
        for n, planes in enumerate(frames):
            fr = self.L.mockvs_source_frame(node, n)
            for i, p in enumerate(planes):
                self._plane(fr, i)[:, :] = np.ascontiguousarray(p[:self._dims(fr, i)[2]]).view(np.uint8).reshape(self._dims(fr, i)[2], -1)[:, :self._dims(fr, i)[1]]
        return node

    def _dims(self, frame, index):
        stride, row, hh = C.c_int(), C.c_int(), C.c_int()
        ptr = self.L.mockvs_frame_plane(frame, index, C.byref(stride), C.byref(row), C.byref(hh))
        return ptr, row.value, hh.value, stride.value

    def _plane(self, frame, index):
        ptr, row, hh, stride = self._dims(frame, index)
        view = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_ubyte)), shape=(hh, stride))
        return view[:, :row]

    def read_plane(self, frame, index, dtype):
        return np.ascontiguousarray(self._plane(frame, index)).view(dtype).copy()

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
        node = self.L.mockvs_invoke(self.h, name.encode(), clip, tw, th, n, names, bytes(kinds), ivals, fvals, svals)
        return (node, None) if node else (None, self.L.mockvs_last_error(self.h).decode())

    def get_frame(self, node, n):
        fr = self.L.mockvs_get_frame(self.h, node, n)
        return fr, (None if fr else self.L.mockvs_last_error(self.h).decode())

    def prop(self, frame, key):
        v = C.c_longlong()
        return v.value if self.L.mockvs_frame_prop_int(frame, key.encode(), C.byref(v)) else None

    def live(self):
        return self.L.mockvs_live_nodes(self.h), self.L.mockvs_live_frames(self.h)

    def close(self):
        self.L.mockvs_free(self.h)


def test_registration_surface(vs):
    """Namespace jinc, API 4, JincResize + the four aliases with the reference's argument names (ref :1044-1108)."""
    c = Core(vs)
    assert vs.mockvs_plugin_namespace(c.h) == b"jinc" and vs.mockvs_plugin_api_version(c.h) >> 16 == 4
    f = c.functions()
    assert f["JincResize"] == ("clip:vnode;target_width:int;target_height:int;src_left:float:opt;src_top:float:opt;src_width:float:opt;"
                               "src_height:float:opt;quant_x:int:opt;quant_y:int:opt;tap:int:opt;blur:float:opt;cplace:data:opt;threads:int:opt;"
                               "opt:int:opt;initial_capacity:int:opt;initial_factor:float:opt;", "clip:vnode;")
    alias = ("clip:vnode;target_width:int;target_height:int;src_left:float:opt;src_top:float:opt;src_width:float:opt;src_height:float:opt;"
             "quant_x:int:opt;quant_y:int:opt;cplace:data:opt;threads:int:opt;", "clip:vnode;")
    assert {k: v for k, v in f.items() if k != "JincResize"} == {n: alias for n in ("Jinc36Resize", "Jinc64Resize", "Jinc144Resize", "Jinc256Resize")}
    c.close()


ERRORS = [
    (dict(tap=0), "JincResize: tap must be between 1..16."),
    (dict(quant_y=257), "JincResize: quant_y must be between 1..256."),
    (dict(cplace="left"), "JincResize: cplace must be MPEG2, MPEG1 or topleft."),
    (dict(opt=4), "JincResize: opt higher than 3 is not allowed."),
    (dict(threads=2), "JincResize: threads must be either 0 or 1."),
    (dict(initial_factor=0.5), "JincResize: initial_factor must be eqaul to or greater than 1.0."),
]


@pytest.mark.parametrize("named,message", ERRORS, ids=["_".join(f"{k}{v}" for k, v in n.items()) for n, _ in ERRORS])
def test_create_errors_carry_the_reference_text(vs, O, named, message):
    c = Core(vs)
    fmt = O.FORMATS["YUV420P8"]
    src = c.source(fmt, 64, 48, [O.lcg_frame(fmt, 64, 48)])
    node, err = c.invoke("JincResize", src, 128, 96, **named)
    assert node is None and err == message
    vs.mockvs_node_release(src)
    assert c.live() == (0, 0)   # the failed filter released its clip reference (ref :682-687)
    c.close()


def test_the_host_rejects_arguments_the_signature_does_not_have(vs, O):
    """tap / blur / opt belong to JincResize only (ref :1061-1108: the aliases' parameter lists end with threads)."""
    c = Core(vs)
    fmt = O.FORMATS["Y8"]
    src = c.source(fmt, 64, 48, [O.lcg_frame(fmt, 64, 48)])
    node, err = c.invoke("Jinc36Resize", src, 128, 96, tap=4)
    assert node is None and "tap" in err
    node, err = c.invoke("JincResize", src, 128, 96, taps=4)
    assert node is None and "taps" in err
    vs.mockvs_node_release(src)
    assert c.live() == (0, 0)
    c.close()


def test_filter_object(vs, O, pkg):
    """Output size, fmUnordered (one frame call at a time, the instance is not re-entrant); a frame request either works (GPU
    box) or fails loudly through the frame context -- there is no CPU path."""
    c = Core(vs)
    fmt = O.FORMATS["YUV420P8"]
    src = c.source(fmt, 64, 48, [O.lcg_frame(fmt, 64, 48)])
    node, err = c.invoke("JincResize", src, 160, 120, tap=4)
    assert err is None
    w, h, n, mode = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    vs.mockvs_node_info(node, C.byref(w), C.byref(h), C.byref(n), C.byref(mode))
    assert (w.value, h.value, n.value, mode.value) == (160, 120, 1, 2)
    fr, err = c.get_frame(node, 0)
    if pkg.device_count() == 0:
        assert fr is None and "HIP device" in err
    else:
        assert err is None
        vs.mockvs_frame_release(fr)
    fr, err = c.get_frame(node, 5)
    assert fr is None and err   # the source has no such frame
    vs.mockvs_node_release(node)
    vs.mockvs_node_release(src)
    assert c.live() == (0, 0)
    c.close()


GPU_CASES = [
    ("Y8", 96, 64, 192, 128, "JincResize", {}, None),
    ("YUV420P8", 128, 96, 256, 192, "JincResize", dict(tap=4, cplace="topleft"), 2),
    ("YUV420P16", 128, 96, 200, 150, "JincResize", dict(src_left=1.5, src_top=-0.5, src_width=120.0, src_height=90.0, quant_x=64, blur=0.98), 2),
    ("YUV422P10", 128, 96, 256, 192, "Jinc36Resize", dict(cplace="MPEG1"), 2),
    ("RGBPS", 96, 64, 192, 128, "Jinc64Resize", {}, None),
    ("YUV444P8", 96, 64, 48, 32, "Jinc144Resize", dict(quant_y=32), None),
    ("RGBP8", 64, 48, 128, 96, "Jinc256Resize", {}, None),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=lambda c: f"{c[5]}_{c[0]}")
def test_frames_through_the_front_end_match_the_oracle(vs, O, case):
    """jinc.<function>(...) -> C ABI -> GPU -> frame in the host's buffers, against the oracle (doubles as given: VapourSynth does
    not round script floats to 32 bits); the aliases arrive with tap 3 / 4 / 6 / 8; _ChromaLocation is written for sub-sampled
    formats only (ref :617-625) and the other frame properties are inherited (ref :613)."""
    fmt_name, sw, sh, tw, th, fn, named, want_loc = case
    fmt = O.FORMATS[fmt_name]
    frames = [O.lcg_frame(fmt, sw, sh, seed=22345 + n) for n in range(2)]
    c = Core(vs)
    src = c.source(fmt, sw, sh, frames)
    node, err = c.invoke(fn, src, tw, th, **named)
    assert err is None, err
    kw = dict(named)
    kw.update({"Jinc36Resize": dict(tap=3), "Jinc64Resize": dict(tap=4), "Jinc144Resize": dict(tap=6), "Jinc256Resize": dict(tap=8)}.get(fn, {}))
    of = O.OracleFilter(fmt, sw, sh, tw, th, **oracle_kwargs(kw))
    dtype = {1: np.uint8, 2: np.uint16, 4: np.float32}[fmt.sample_bytes]
    for n in (1, 0):
        fr, err = c.get_frame(node, n)
        assert err is None, err
        got = [c.read_plane(fr, i, dtype) for i in range(fmt.planes)]
        assert_planes_equal(got, of.get_frame(frames[n], threads=4), fmt.plane_dims(tw, th), what=f"{fn} {fmt_name} frame {n}")
        assert c.prop(fr, "_ChromaLocation") == want_loc
        assert c.prop(fr, "_MockFrameNumber") == n
        vs.mockvs_frame_release(fr)
    vs.mockvs_node_release(node)
    vs.mockvs_node_release(src)
    assert c.live() == (0, 0)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("by_siting", [False, True], ids=["as_reference", "by_siting"])
def test_chroma_location_of_frame_zero_decides_when_cplace_is_not_given(vs, O, monkeypatch, by_siting):
    """ref :727-742 through the VapourSynth property of the same name; the property written is 2 whatever the siting, as the
    reference binary does (ref :617-625), unless JINCRESIZE_CHROMALOC=siting."""
    if by_siting:
        monkeypatch.setenv("JINCRESIZE_CHROMALOC", "siting")
    else:
        monkeypatch.delenv("JINCRESIZE_CHROMALOC", raising=False)
    fmt = O.FORMATS["YUV420P8"]
    frames = [O.lcg_frame(fmt, 128, 96)]
    for loc, cplace in ((0, "mpeg2"), (1, "mpeg1"), (2, "topleft")):
        c = Core(vs)
        src = c.source(fmt, 128, 96, frames, chroma_location=loc)
        node, err = c.invoke("JincResize", src, 256, 192)
        assert err is None
        fr, err = c.get_frame(node, 0)
        assert err is None
        got = [c.read_plane(fr, i, np.uint8) for i in range(3)]
        want = O.OracleFilter(fmt, 128, 96, 256, 192, cplace=cplace).get_frame(frames[0], threads=4)
        assert_planes_equal(got, want, fmt.plane_dims(256, 192), what=f"_ChromaLocation {loc}")
        assert c.prop(fr, "_ChromaLocation") == (loc if by_siting else 2)
        vs.mockvs_frame_release(fr)
        vs.mockvs_node_release(node)
        vs.mockvs_node_release(src)
        assert c.live() == (0, 0)
        c.close()
    c = Core(vs)
    src = c.source(fmt, 128, 96, frames, chroma_location=4)
    node, err = c.invoke("JincResize", src, 256, 192)
    assert node is None and err == "JincResize: invalid _ChromaLocation"
    vs.mockvs_node_release(src)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case,kernel", [
    (("Y8", 1280, 720, 1754, 986, "JincResize", {}), "ewa_framelane"),             # no phase structure: groups of 16 on the frame-lane kernels
    (("YUV420P8", 960, 540, 1920, 1080, "Jinc36Resize", {}), "ewa_periodic"),       # 2x: the periodic kernels' batch forms
], ids=["A137", "2x_420"])
def test_lookahead_in_the_vapoursynth_shell(vs, O, pkg, case, kernel, monkeypatch):
    """JINCRESIZE_LOOKAHEAD=32 in the VapourSynth shell -- arInitial asks for n .. n + 31, arAllFramesReady
    (fmParallelRequests: one call at a time, any order) feeds the same window ring as the AviSynth shell.  Frames pulled in
    order, then out of order and again: each is that frame's single-call result (and the oracle's), launches are 16-frame
    batches, nothing leaks."""
    monkeypatch.setenv("JINCRESIZE_LOOKAHEAD", "32")
    monkeypatch.delenv("JINCRESIZE_PIN_FRAMES", raising=False)   # pageable frames, the default
    monkeypatch.delenv("JINCRESIZE_GROUP", raising=False)
    fmt_name, sw, sh, tw, th, fn, named = case
    fmt = O.FORMATS[fmt_name]
    nframes = 64
    frames = [O.lcg_frame(fmt, sw, sh, seed=41000 + n) for n in range(nframes)]
    kw = dict(named)
    kw.update({"Jinc36Resize": dict(tap=3)}.get(fn, {}))
    single = pkg.Filter(pkg.FORMATS[fmt_name], sw, sh, tw, th, device=0, **kw)
    want = [single.get_frame(fr) for fr in frames]
    single.close()
    of = O.OracleFilter(fmt, sw, sh, tw, th, **oracle_kwargs(kw))
    c = Core(vs)
    src = c.source(fmt, sw, sh, frames)
    node, err = c.invoke(fn, src, tw, th, **named)
    assert err is None, err
    stale = pkg.last_call()
    seen = set()
    order = list(range(nframes)) + [40, 38, 39, 63, 5, 6, 4, 7]     # in order, then jumps and out-of-order neighbours
    for n in order:
        fr, err = c.get_frame(node, n)
        assert err is None, err
        seen.add(pkg.last_call())
        got = [c.read_plane(fr, i, np.uint8) for i in range(fmt.planes)]
        try:
            assert_planes_equal(got, want[n], fmt.plane_dims(tw, th), what=f"{fn} frame {n} grouped vs single")
        except AssertionError as e:   # which of the two is it, and where: the oracle decides
            ref = of.get_frame(frames[n], threads=8)
            w, h = fmt.plane_dims(tw, th)[0]
            bad = np.argwhere(got[0][:h, :w] != want[n][0][:h, :w])
            sides = (int((got[0][:h, :w] != ref[0][:h, :w]).sum()), int((want[n][0][:h, :w] != ref[0][:h, :w]).sum()))
            raise AssertionError(f"{e}; plane 0 differs in rows {bad[:, 0].min()}..{bad[:, 0].max()}, columns {bad[:, 1].min()}..{bad[:, 1].max()}; "
                                 f"samples off the oracle: grouped {sides[0]}, single {sides[1]}; launches seen so far {sorted(seen)}") from None
        if n in (0, 41):
            assert_planes_equal(got, of.get_frame(frames[n], threads=8), fmt.plane_dims(tw, th), what=f"{fn} frame {n} vs oracle")
        assert c.prop(fr, "_MockFrameNumber") == n
        vs.mockvs_frame_release(fr)
    served = {(name, k) for name, k in seen if name} - {stale}
    # full groups of 16 on the batch kernel while the clip lasts; the jumps near its end launch short groups, which a plan
    # without phase structure runs on the gather kernel (fewer than 16 frames)
    assert any(name.startswith(kernel) and k == 16 for name, k in served), served
    assert all(name.startswith(kernel) or k < 16 for name, k in served), served
    vs.mockvs_node_release(node)
    vs.mockvs_node_release(src)
    assert c.live() == (0, 0)
    c.close()
