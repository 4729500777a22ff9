"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It wraps oracle/libjinc_oracle.so (plain-C restatement of the reference opt=0 path, see
jinc_oracle.c) and restates, in Python, the parameter derivation that Create_JincResize performs
before it calls the table generator ("ref:" = /root/reference/src/JincResize.cpp).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import zlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libjinc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc/g++)."""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


class _Meta(C.Structure):
    _fields_ = [("start_x", C.c_int), ("start_y", C.c_int), ("coeff_meta", C.c_int)]


class _Table(C.Structure):
    _fields_ = [
        ("factor", C.POINTER(C.c_float)),
        ("meta", C.POINTER(_Meta)),
        ("filter_size", C.c_int),
        ("coeff_stride", C.c_int),
        ("dst_width", C.c_int),
        ("dst_height", C.c_int),
        ("factor_count", C.c_int64),
        ("cached_phases", C.c_int64),
    ]


class _TableParams(C.Structure):
    _fields_ = [
        ("quantize_x", C.c_int),
        ("quantize_y", C.c_int),
        ("samples", C.c_int),
        ("src_width", C.c_int),
        ("src_height", C.c_int),
        ("dst_width", C.c_int),
        ("dst_height", C.c_int),
        ("radius", C.c_double),
        ("crop_left", C.c_double),
        ("crop_top", C.c_double),
        ("crop_width", C.c_double),
        ("crop_height", C.c_double),
    ]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_jinc_zero.restype = C.c_double
        L.oracle_jinc_zero.argtypes = [C.c_int]
        L.oracle_jinc_sqr.restype = C.c_double
        L.oracle_jinc_sqr.argtypes = [C.c_double]
        L.oracle_lut_init.restype = None
        L.oracle_lut_init.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.oracle_table_generate.restype = C.c_int
        L.oracle_table_generate.argtypes = [C.c_void_p, C.POINTER(_TableParams), C.POINTER(_Table)]
        L.oracle_table_free.restype = None
        L.oracle_table_free.argtypes = [C.POINTER(_Table)]
        L.oracle_resize_plane.restype = None
        L.oracle_resize_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(_Table),
                                          C.c_int, C.c_float, C.c_int]
        L.oracle_resize_plane_simd.restype = None
        L.oracle_resize_plane_simd.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(_Table),
                                               C.c_int, C.c_float, C.c_int]
        L.oracle_avx2_available.restype = C.c_int
        L.oracle_resize_plane_avx2.restype = None
        L.oracle_resize_plane_avx2.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(_Table),
                                               C.c_int, C.c_float, C.c_int]
        L.oracle_avx512_available.restype = C.c_int
        L.oracle_resize_plane_avx512.restype = None
        L.oracle_resize_plane_avx512.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(_Table),
                                                 C.c_int, C.c_float, C.c_int]
        L.oracle_lcg_fill.restype = C.c_uint32
        L.oracle_lcg_fill.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.oracle_fnv1a64.restype = C.c_uint64
        L.oracle_fnv1a64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        _lib = L
    return _lib


SAMPLES = 1024  # ref :795


def jinc_zero(tap: int) -> float:
    return lib().oracle_jinc_zero(tap)


def make_lut(tap: int, blur: float) -> np.ndarray:
    """ref :794-797: radius = jinc_zeros[tap-1]; Lut::InitLut(1024, radius, blur)."""
    lut = np.zeros(SAMPLES, dtype=np.float64)
    lib().oracle_lut_init(lut.ctypes.data, SAMPLES, jinc_zero(tap), float(blur))
    return lut


def fnv1a64(buf) -> int:
    a = np.ascontiguousarray(buf)
    return int(lib().oracle_fnv1a64(a.ctypes.data, a.nbytes, 0))


class Table:
    """Owner of one oracle coefficient table (ref: EWAPixelCoeff, JincResize.h:18-25)."""

    def __init__(self, lut: np.ndarray, *, quant_x, quant_y, src_w, src_h, dst_w, dst_h, radius,
                 crop_left, crop_top, crop_width, crop_height):
        self._t = _Table()
        self.params = _TableParams(quant_x, quant_y, SAMPLES, src_w, src_h, dst_w, dst_h, radius,
                                   crop_left, crop_top, crop_width, crop_height)
        self.src_w, self.src_h = src_w, src_h
        rc = lib().oracle_table_generate(lut.ctypes.data, C.byref(self.params), C.byref(self._t))
        if rc != 0:
            raise MemoryError("oracle_table_generate failed")

    def __del__(self):
        try:
            if self._t.meta:
                lib().oracle_table_free(C.byref(self._t))
        except Exception:
            pass

    filter_size = property(lambda s: s._t.filter_size)
    coeff_stride = property(lambda s: s._t.coeff_stride)
    dst_w = property(lambda s: s._t.dst_width)
    dst_h = property(lambda s: s._t.dst_height)
    factor_count = property(lambda s: s._t.factor_count)
    cached_phases = property(lambda s: s._t.cached_phases)

    @property
    def num_sets(self) -> int:
        return self.factor_count // (self.filter_size * self.coeff_stride)

    def meta(self) -> np.ndarray:
        """(dst_h, dst_w, 3) int32 view: start_x, start_y, coeff_meta."""
        n = self.dst_w * self.dst_h * 3
        a = np.ctypeslib.as_array(C.cast(self._t.meta, C.POINTER(C.c_int)), shape=(n,))
        return a.reshape(self.dst_h, self.dst_w, 3)

    def factor(self) -> np.ndarray:
        if self.factor_count == 0:
            return np.zeros(0, np.float32)
        return np.ctypeslib.as_array(self._t.factor, shape=(self.factor_count,))

    def coeff_set(self, x: int, y: int) -> np.ndarray:
        """fs x fs coefficient block used by output pixel (x, y)."""
        fs, cs = self.filter_size, self.coeff_stride
        off = int(self.meta()[y, x, 2])
        return self.factor()[off: off + fs * cs].reshape(fs, cs)[:, :fs]

    def resize(self, src: np.ndarray, dst: np.ndarray, peak: float, threads: int = 1) -> None:
        sb = src.dtype.itemsize
        assert dst.dtype == src.dtype and src.flags.c_contiguous is not None
        assert src.shape[0] >= self.src_h and src.shape[1] >= self.src_w
        assert dst.shape == (self.dst_h, dst.shape[1]) and dst.shape[1] >= self.dst_w
        lib().oracle_resize_plane(src.ctypes.data, src.strides[0], dst.ctypes.data, dst.strides[0],
                                  C.byref(self._t), sb, float(peak), int(threads))


    def resize_simd(self, order: int, src: np.ndarray, dst: np.ndarray, min_val: float, threads: int = 1, avx2: bool = False,
                    avx512: bool = False) -> None:
        """Summation order of the reference's opt = 1 / 2 / 3 paths (simd_order.c); avx2=True: the own AVX2 code (order 2),
        avx512=True: the own AVX-512 code (order 3; only where lib().oracle_avx512_available())."""
        sb = src.dtype.itemsize
        if avx512:
            assert order == 3 and lib().oracle_avx512_available()
            lib().oracle_resize_plane_avx512(src.ctypes.data, src.strides[0], src.nbytes, dst.ctypes.data, dst.strides[0],
                                             C.byref(self._t), sb, float(min_val), int(threads))
        elif avx2:
            assert order == 2
            lib().oracle_resize_plane_avx2(src.ctypes.data, src.strides[0], src.nbytes, dst.ctypes.data, dst.strides[0],
                                           C.byref(self._t), sb, float(min_val), int(threads))
        else:
            lib().oracle_resize_plane_simd(int(order), src.ctypes.data, src.strides[0], dst.ctypes.data, dst.strides[0],
                                           C.byref(self._t), sb, float(min_val), int(threads))


# ----------------------------------------------------------------------------------------------
# Clip formats and synthetic frames
# ----------------------------------------------------------------------------------------------
@dataclass
class Format:
    """The AVS_VideoInfo facts the path reads (ref :793, :798, :826, :833-834, :903)."""
    name: str
    bits: int            # bits per component: 8..16, 32 (float)
    planes: int          # 1, 3 or 4
    sub_w: int = 0       # log2 chroma subsampling
    sub_h: int = 0
    rgb: bool = False

    @property
    def sample_bytes(self) -> int:
        return 1 if self.bits == 8 else (2 if self.bits <= 16 else 4)

    @property
    def dtype(self):
        return {1: np.uint8, 2: np.uint16, 4: np.float32}[self.sample_bytes]

    def plane_dims(self, w: int, h: int) -> List[Tuple[int, int]]:
        out = [(w, h)]
        if self.planes >= 3:
            out += [(w >> self.sub_w, h >> self.sub_h)] * 2
        if self.planes == 4:
            out.append((w, h))
        return out


def _formats():
    out = {}
    for bits, tag in ((8, "8"), (10, "10"), (12, "12"), (14, "14"), (16, "16"), (32, "S")):
        yname = "Y32" if bits == 32 else f"Y{tag}"
        out[yname] = Format(yname, bits, 1)
        for fam, sw, sh in (("420", 1, 1), ("422", 1, 0), ("444", 0, 0), ("411", 2, 0)):
            out[f"YUV{fam}P{tag}"] = Format(f"YUV{fam}P{tag}", bits, 3, sw, sh)
            out[f"YUVA{fam}P{tag}"] = Format(f"YUVA{fam}P{tag}", bits, 4, sw, sh)
        out[f"RGBP{tag}"] = Format(f"RGBP{tag}", bits, 3, rgb=True)
        out[f"RGBAP{tag}"] = Format(f"RGBAP{tag}", bits, 4, rgb=True)
    out["YV12"], out["YV16"], out["YV24"], out["YV411"] = (out["YUV420P8"], out["YUV422P8"], out["YUV444P8"],
                                                            out["YUV411P8"])
    return out


FORMATS = _formats()


def alloc_plane(w: int, h: int, dtype, align: int = 64) -> np.ndarray:
    """Zeroed plane whose pitch is row_size rounded up to `align` bytes (AviSynth+ frame layout)."""
    isz = np.dtype(dtype).itemsize
    pitch = (w * isz + align - 1) // align * align
    buf = np.zeros((h, pitch // isz), dtype=dtype)
    return buf


def lcg_frame(fmt: Format, w: int, h: int, seed: int = 12345) -> List[np.ndarray]:
    """SURVEY.md Appendix A item 4 synthetic frame: padded planes in processing order."""
    s = seed & 0xFFFFFFFF
    planes = []
    for (pw, ph) in fmt.plane_dims(w, h):
        p = alloc_plane(pw, ph, fmt.dtype)
        s = lib().oracle_lcg_fill(p.ctypes.data, p.strides[0], pw, ph, fmt.sample_bytes, fmt.bits, s)
        planes.append(p)
    return planes


def packed_bytes(planes: Sequence[np.ndarray], dims: Sequence[Tuple[int, int]]) -> bytes:
    """Output dump of Appendix A: planes in processing order, rows truncated to row_size."""
    return b"".join(np.ascontiguousarray(p[:ph, :pw]).tobytes() for p, (pw, ph) in zip(planes, dims))


def crc32_planes(planes, dims) -> str:
    c = 0
    for p, (pw, ph) in zip(planes, dims):
        c = zlib.crc32(np.ascontiguousarray(p[:ph, :pw]).tobytes(), c)
    return f"{c & 0xFFFFFFFF:08x}"


# ----------------------------------------------------------------------------------------------
# Filter-level oracle = Create_JincResize's derivation (ref :762-866) + resize_plane_c (ref :536-601)
# ----------------------------------------------------------------------------------------------
@dataclass
class OracleFilter:
    fmt: Format
    src_w: int
    src_h: int
    target_w: int
    target_h: int
    tap: int = 3
    blur: float = 1.0
    quant_x: int = 256
    quant_y: int = 256
    crop_left: float = 0.0
    crop_top: float = 0.0
    crop_width: Optional[float] = None
    crop_height: Optional[float] = None
    cplace: str = "mpeg2"
    tables: List[Table] = field(default_factory=list, init=False)

    def __post_init__(self):
        fmt = self.fmt
        # ref :762-770: defaults and "<= 0 means relative" crop
        cw = float(self.src_w) if self.crop_width is None else float(self.crop_width)
        if cw <= 0.0:
            cw = self.src_w - self.crop_left + cw
        ch = float(self.src_h) if self.crop_height is None else float(self.crop_height)
        if ch <= 0.0:
            ch = self.src_h - self.crop_top + ch
        blur = self.blur if self.blur else 1.0  # ref :772-774
        self.peak = float((1 << fmt.bits) - 1) if fmt.bits <= 16 else 0.0  # ref :793 (unused for float)
        radius = jinc_zero(self.tap)  # ref :794
        self.lut = make_lut(self.tap, blur)
        common = dict(quant_x=self.quant_x, quant_y=self.quant_y, radius=radius)
        self.tables.append(Table(self.lut, src_w=self.src_w, src_h=self.src_h, dst_w=self.target_w,
                                 dst_h=self.target_h, crop_left=self.crop_left, crop_top=self.crop_top,
                                 crop_width=cw, crop_height=ch, **common))
        self.subsampled = fmt.planes > 1 and not fmt.rgb and (fmt.sub_w or fmt.sub_h)  # ref :824-832
        if self.subsampled:
            div_w = float(1 << fmt.sub_w)  # ref :833-836
            div_h = float(1 << fmt.sub_h)
            if self.cplace in ("mpeg2", "topleft"):  # ref :838-839
                cl = (0.5 * (1.0 - float(self.src_w) / self.target_w) + self.crop_left) / div_w
            else:
                cl = self.crop_left / div_w
            if self.cplace == "topleft":  # ref :840-841
                ct = (0.5 * (1.0 - float(self.src_h) / self.target_h) + self.crop_top) / div_h
            else:
                ct = self.crop_top / div_h
            self.tables.append(Table(self.lut, src_w=self.src_w >> fmt.sub_w, src_h=self.src_h >> fmt.sub_h,
                                     dst_w=self.target_w >> fmt.sub_w, dst_h=self.target_h >> fmt.sub_h,
                                     crop_left=cl, crop_top=ct, crop_width=cw / div_w, crop_height=ch / div_h,
                                     **common))  # ref :844-862

    def table_for_plane(self, i: int) -> Table:
        """ref :552-558: U,V of subsampled formats use table 1; Y and A table 0."""
        if self.subsampled and i in (1, 2):
            return self.tables[1]
        return self.tables[0]

    def out_dims(self) -> List[Tuple[int, int]]:
        return self.fmt.plane_dims(self.target_w, self.target_h)

    def get_frame(self, src_planes: Sequence[np.ndarray], threads: int = 1) -> List[np.ndarray]:
        out = []
        for i, (pw, ph) in enumerate(self.out_dims()):
            d = alloc_plane(pw, ph, self.fmt.dtype)
            self.table_for_plane(i).resize(src_planes[i], d, self.peak, threads)
            out.append(d)
        return out

    def get_frame_simd(self, order: int, src_planes: Sequence[np.ndarray], threads: int = 1, avx2: bool = False,
                       avx512: bool = False) -> List[np.ndarray]:
        """The frame in the summation order of the reference's opt = `order` (1 SSE4.1, 2 AVX2, 3 AVX-512) path."""
        out = []
        for i, (pw, ph) in enumerate(self.out_dims()):
            d = alloc_plane(pw, ph, self.fmt.dtype)
            min_val = -0.5 if (i and not self.fmt.rgb) else 0.0   # resize_plane_sse41.cpp:20
            self.table_for_plane(i).resize_simd(order, src_planes[i], d, min_val, threads, avx2, avx512)
            out.append(d)
        return out
