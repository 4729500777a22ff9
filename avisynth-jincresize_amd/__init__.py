"""Python host mirror of the AviSynth+ JincResize plugin surface, over libjincresize_hip.so.

The product is the C-ABI library (include/jincresize_hip.h: host C++ + gfx950 HIP kernels); this
module only binds it with ctypes and mirrors the script-level interface of the reference plugin
(/root/reference/src/JincResize.cpp:1042-1111): ``JincResize(clip, target_width, target_height,
src_left, src_top, src_width, src_height, quant_x, quant_y, tap, blur, cplace, threads, opt,
initial_capacity, initial_factor)`` and the ``Jinc36Resize/Jinc64Resize/Jinc144Resize/
Jinc256Resize`` aliases.  Errors surface as ``JincError`` carrying the reference's message text.

There is no CPU fallback here: if the library is missing the import fails, and frame calls fail
loudly when no HIP device is present.

The directory name contains a hyphen, so load it with ``importlib`` (see ``__graft_entry__.py``
``load_package()``), e.g. as module ``avisynth_jincresize_amd``.
"""
from __future__ import annotations

import ctypes as C
import enum
import os
import subprocess
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JINC_LIB") or os.path.join(_HERE, "lib", "libjincresize_hip.so")  # JINC_LIB: A/B runs against another build
SIMD_ORDER_ISA_PATH = os.path.join(_HERE, "lib", "kernel_simdorder-gfx950.s")  # the one unit with (explicit) fused multiply-adds
ISA_PATHS = [os.path.join(_HERE, "lib", f"{k}-gfx950.s") for k in ("kernel_gather", "kernel_framelane", "kernel_framelane_sub", "kernel_framelane_pair", "kernel_periodic", "kernel_rowpair", "kernel_strip", "kernel_colpair", "kernel_direct", *[f"kernel_direct_walk_{t}_sx{x}" for t in ("u8", "u16", "f32") for x in (1, 2, 3, 4)], "kernel_colstrip", "kernel_quasi_fs7", "kernel_quasi_fs9", "kernel_quasi_exact_fs7",
                       "kernel_quasi_exact_fs9", "kernel_quasi_lane_fs7", "kernel_quasi_lane_fs9")]
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "jincresize_hip.h")
TEST_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "jincresize_hip_test.h")  # introspection, knobs, hooks


def build(jobs: int = 6) -> str:
    """Compile the HIP/C++ library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", _HERE, f"-j{jobs}"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libjincresize_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


class JincError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


# ---- ctypes mirror of include/jincresize_hip.h ---------------------------------------------------
class VideoInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "bits_per_component", "component_size", "num_components",
                                       "is_planar", "is_rgb", "sub_w", "sub_h")]


class Args(C.Structure):
    _fields_ = [
        ("target_width", C.c_int), ("target_height", C.c_int),
        ("src_left", C.c_double), ("src_top", C.c_double), ("src_width", C.c_double), ("src_height", C.c_double),
        ("quant_x", C.c_int), ("quant_y", C.c_int), ("tap", C.c_int), ("blur", C.c_double),
        ("cplace", C.c_char_p), ("threads", C.c_int), ("opt", C.c_int), ("initial_capacity", C.c_int),
        ("initial_factor", C.c_double), ("defined", C.c_uint),
        ("frame0_chroma_location", C.c_int),
        ("cpu_has_sse41", C.c_int), ("cpu_has_avx2", C.c_int), ("cpu_has_avx512f", C.c_int),
    ]


class PlanInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("src_width", "src_height", "dst_width", "dst_height", "filter_size", "num_sets",
                                       "periodic", "period_x", "period_y", "step_x", "step_y",
                                       "interior_x0", "interior_x1", "interior_y0", "interior_y1")] + \
               [("plan_bytes", C.c_int64)] + \
               [(n, C.c_int) for n in ("quasi", "quasi_period_x", "quasi_period_y", "quasi_step_x", "quasi_step_y")]


ARG_BITS = {"src_left": 1 << 0, "src_top": 1 << 1, "src_width": 1 << 2, "src_height": 1 << 3, "quant_x": 1 << 4,
            "quant_y": 1 << 5, "tap": 1 << 6, "blur": 1 << 7, "cplace": 1 << 8, "threads": 1 << 9, "opt": 1 << 10,
            "initial_capacity": 1 << 11, "initial_factor": 1 << 12}

EXPORTS = ["jinc_device_count", "jinc_pick_device", "jinc_last_error", "jinc_filter_create", "jinc_filter_free", "jinc_filter_output_info",
           "jinc_filter_chroma_location", "jinc_filter_set_chroma_location_mode", "jinc_filter_get_frame", "jinc_filter_process_device", "jinc_filter_sync",
           "jinc_alias_args", "jinc_filter_num_tables", "jinc_filter_plan_info", "jinc_filter_plan_pixel",
           "jinc_filter_plan_dump", "jinc_filter_plan_runs", "jinc_filter_plan_set", "jinc_filter_lut", "jinc_filter_set_kernel_mode", "jinc_filter_set_border_strips", "jinc_filter_interior_kernel", "jinc_filter_last_kernel",
           "jinc_filter_set_profiling", "jinc_filter_kernel_times", "jinc_filter_set_border_overlap", "jinc_debug_convert", "jinc_debug_buffer_range_check", "jinc_debug_set_direct_shape", "jinc_debug_last_direct_shape", "jinc_filter_set_simd_order", "jinc_debug_transport_counts", "jinc_debug_staged_frames", "jinc_debug_copy_rows", "jinc_debug_usable_cpus", "jinc_filter_periodic_support", "jinc_filter_periodic_taps", "jinc_filter_set_pipeline",
           "jinc_filter_set_pipeline_group", "jinc_filter_pipeline_group", "jinc_filter_flush", "jinc_filter_adopt_host_range", "jinc_filter_release_host_range", "jinc_batch_set_affinity", "jinc_batch_device_cpus", "jinc_debug_numa_cpus", "jinc_debug_batch_set_registrars", "jinc_debug_batch_refused", "jinc_debug_host_registrations", "jinc_debug_last_call", "jinc_filter_direct_premise", "jinc_debug_valu_pair_probe", "jinc_debug_clock_sampler_start", "jinc_debug_clock_sampler_stop",
           "jinc_filter_submit", "jinc_filter_wait", "jinc_shard_device", "jinc_batch_create", "jinc_batch_devices",
           "jinc_batch_device_of_frame", "jinc_batch_process", "jinc_batch_free", "jinc_batch_last_error",
           "jinc_filter_last_instance", "jinc_filter_last_border", "jinc_debug_last_instance", "jinc_debug_set_knob", "jinc_debug_clear_knob", "jinc_debug_get_knob", "jinc_debug_knob_name", "jinc_debug_chord_pattern"]

_lib = None
_P4 = C.c_void_p * 4
_I4 = C.c_int * 4
_S4 = C.c_size_t * 4


def lib():
    """Loads the C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc) first")
        L = C.CDLL(LIB_PATH)
        L.jinc_device_count.restype = C.c_int
        L.jinc_last_error.restype = C.c_char_p
        L.jinc_filter_create.restype = C.c_int
        L.jinc_filter_create.argtypes = [C.POINTER(VideoInfo), C.POINTER(Args), C.c_int, C.POINTER(C.c_void_p),
                                         C.c_char_p, C.c_size_t]
        L.jinc_filter_free.restype = None
        L.jinc_filter_free.argtypes = [C.c_void_p]
        L.jinc_filter_output_info.argtypes = [C.c_void_p, C.POINTER(VideoInfo)]
        L.jinc_filter_chroma_location.argtypes = [C.c_void_p]
        L.jinc_filter_set_chroma_location_mode.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_get_frame.argtypes = [C.c_void_p, _P4, _I4, _P4, _I4]
        L.jinc_filter_process_device.argtypes = [C.c_void_p, _P4, _I4, _S4, _P4, _I4, _S4, C.c_int, C.c_void_p]
        L.jinc_filter_sync.argtypes = [C.c_void_p]
        L.jinc_filter_set_pipeline.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.jinc_filter_set_pipeline_group.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.jinc_filter_pipeline_group.argtypes = [C.c_void_p]
        L.jinc_filter_flush.argtypes = [C.c_void_p]
        L.jinc_filter_direct_premise.argtypes = [C.c_void_p]
        L.jinc_filter_adopt_host_range.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.jinc_filter_release_host_range.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.jinc_batch_set_affinity.argtypes = [C.c_void_p, C.c_int]
        L.jinc_batch_device_cpus.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.jinc_debug_batch_set_registrars.argtypes = [C.c_void_p, C.c_int]
        L.jinc_debug_host_registrations.restype = C.c_longlong
        L.jinc_debug_batch_refused.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.jinc_debug_numa_cpus.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int]
        L.jinc_debug_clock_sampler_start.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_void_p)]
        L.jinc_debug_clock_sampler_stop.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.jinc_debug_valu_pair_probe.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.jinc_debug_last_call.argtypes = [C.POINTER(C.c_int)]
        L.jinc_debug_last_call.restype = C.c_char_p
        L.jinc_filter_submit.argtypes = [C.c_void_p, _P4, _I4, _P4, _I4, C.POINTER(C.c_longlong)]
        L.jinc_filter_wait.argtypes = [C.c_void_p, C.c_longlong]
        L.jinc_alias_args.argtypes = [C.c_int, C.POINTER(Args), C.POINTER(Args)]
        L.jinc_filter_num_tables.argtypes = [C.c_void_p]
        L.jinc_filter_plan_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(PlanInfo)]
        L.jinc_filter_plan_pixel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                             C.POINTER(C.c_int), C.c_void_p]
        L.jinc_filter_plan_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.jinc_filter_plan_runs.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int]
        L.jinc_filter_plan_set.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.jinc_filter_lut.argtypes = [C.c_void_p, C.c_void_p]
        L.jinc_filter_set_kernel_mode.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_set_border_overlap.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_set_border_strips.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_interior_kernel.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_interior_kernel.restype = C.c_char_p
        L.jinc_filter_last_kernel.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_last_kernel.restype = C.c_char_p
        L.jinc_debug_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]
        L.jinc_filter_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                               C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.jinc_debug_buffer_range_check.argtypes = [C.c_int]
        L.jinc_debug_set_direct_shape.argtypes = [C.c_int]
        L.jinc_filter_set_simd_order.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_periodic_support.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_periodic_taps.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.jinc_filter_periodic_taps.restype = C.c_double
        L.jinc_debug_transport_counts.argtypes = [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int]
        L.jinc_debug_staged_frames.restype = C.c_longlong
        L.jinc_debug_copy_rows.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.jinc_debug_staged_frames.argtypes = []
        L.jinc_shard_device.argtypes = [C.c_int, C.c_int]
        L.jinc_batch_create.argtypes = [C.POINTER(VideoInfo), C.POINTER(Args), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                        C.c_char_p, C.c_size_t]
        L.jinc_batch_devices.argtypes = [C.c_void_p]
        L.jinc_batch_device_of_frame.argtypes = [C.c_void_p, C.c_int]
        L.jinc_batch_process.argtypes = [C.c_void_p, C.c_int, C.c_void_p, _I4, C.c_void_p, _I4]
        L.jinc_batch_free.restype = None
        L.jinc_batch_free.argtypes = [C.c_void_p]
        L.jinc_batch_last_error.restype = C.c_char_p
        L.jinc_filter_last_instance.argtypes = [C.c_void_p, C.c_int]
        L.jinc_filter_last_instance.restype = C.c_char_p
        L.jinc_filter_last_border.argtypes = [C.c_void_p, C.c_int]
        L.jinc_debug_last_instance.restype = C.c_char_p
        L.jinc_debug_set_knob.argtypes = [C.c_int, C.c_double]
        L.jinc_debug_clear_knob.argtypes = [C.c_int]
        L.jinc_debug_get_knob.argtypes = [C.c_int, C.POINTER(C.c_double)]
        L.jinc_debug_chord_pattern.argtypes = [C.c_int, C.c_uint64]
        L.jinc_debug_chord_pattern.restype = C.c_int
        L.jinc_debug_knob_name.argtypes = [C.c_int]
        L.jinc_debug_knob_name.restype = C.c_char_p
        _lib = L
    return _lib


def device_count() -> int:
    return int(lib().jinc_device_count())


class ClockSampler:
    """Shader clock while other kernels run (jinc_debug_clock_sampler_*): `with ClockSampler(0) as c: ...; c.ghz` = (min, median, max)."""

    def __init__(self, device: int = 0, max_seconds: float = 60.0):
        self._h = C.c_void_p()
        rc = lib().jinc_debug_clock_sampler_start(int(device), float(max_seconds), C.byref(self._h))
        if rc != 0:
            raise JincError(rc, lib().jinc_last_error().decode())
        self.ghz = None

    def stop(self):
        if self._h.value:
            a, b, c = C.c_double(), C.c_double(), C.c_double()
            rc = lib().jinc_debug_clock_sampler_stop(self._h, C.byref(a), C.byref(b), C.byref(c))
            self._h = C.c_void_p()
            if rc != 0:
                raise JincError(rc, lib().jinc_last_error().decode())
            self.ghz = (a.value, b.value, c.value)
        return self.ghz

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()
        return False


def transport_counts(reset: bool = False) -> Tuple[int, int, int]:
    """(frames whose results the shader wrote into pinned host planes, frames that left by DMA copies, host ranges pinned by the
    process-wide registry) over all filter instances of this process since the last reset."""
    a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
    lib().jinc_debug_transport_counts(C.byref(a), C.byref(b), C.byref(c), int(reset))
    return a.value, b.value, c.value


def copy_rows(dst: np.ndarray, src: np.ndarray, row_bytes: int, rows: int, helpers: bool = True) -> None:
    """host_copy.cpp's plane copy between two 2-D arrays (test header; no device needed)."""
    rc = lib().jinc_debug_copy_rows(dst.ctypes.data, dst.strides[0], src.ctypes.data, src.strides[0], int(row_bytes), int(rows), int(helpers))
    if rc != 0:
        raise JincError(rc, lib().jinc_last_error().decode())


def usable_cpus() -> int:
    """CPUs the process may keep busy as the library counts them (test header)."""
    return int(lib().jinc_debug_usable_cpus())


def staged_frames() -> int:
    """Frames whose planes went through the library's own pinned buffers (register_host_buffers = 0) since the last reset of
    transport_counts."""
    return int(lib().jinc_debug_staged_frames())


def host_registrations() -> int:
    """Host ranges the library holds registered with hipHostRegister right now (registry and batch registrars; test header)."""
    return int(lib().jinc_debug_host_registrations())


def valu_pair_probe(device: int = 0, waves_per_simd: int = 8) -> Tuple[float, float]:
    """(Tops, shader clock GHz) that plain v_mul_f32 + v_add_f32 sustain on this part at the given occupancy (measurement hook)."""
    t, g = C.c_double(), C.c_double()
    rc = lib().jinc_debug_valu_pair_probe(int(device), int(waves_per_simd), C.byref(t), C.byref(g))
    if rc != 0:
        raise JincError(rc, lib().jinc_last_error().decode())
    return t.value, g.value


def last_call() -> Tuple[str, int]:
    """(interior kernel of table 0, frames) of the most recent kernel call of any filter instance in this process (test hook)."""
    n = C.c_int()
    name = lib().jinc_debug_last_call(C.byref(n))
    return (name or b"").decode(), n.value


class KernelMode(enum.IntEnum):
    """enum jinc_kernel_mode of include/jincresize_hip_test.h."""
    AUTO = 0
    GATHER = 1
    PERIODIC = 2
    ROWS = 3
    WINDOW_HALF_TILES = 4
    PACKED_RG4 = 5
    PACKED_RG8 = 6
    QUASI = 7
    QUASI_WATERFALL = 8
    DIRECT = 9
    QUASI_LANE = 10
    FRAMELANE = 11
    FRAMELANE_PAIR = 12
    QUAD = 13
    RUNS = 14
    FULL_WINDOW = 15
    FRAMELANE_SUB = 16


def knob_ids() -> dict:
    """{lower-case knob name: id} as the loaded library lists them (enum jinc_knob of the test header)."""
    out, k = {}, 0
    while True:
        name = lib().jinc_debug_knob_name(k)
        if not name:
            return out
        out[name.decode()] = k
        k += 1


def set_knob(name: str, value: float) -> None:
    """Process-wide A/B / tuning knob (test header): `name` as in knob_ids(), e.g. "quad_rg"."""
    ids = knob_ids()
    if name.lower() not in ids:
        raise KeyError(f"no such knob: {name}")
    rc = lib().jinc_debug_set_knob(ids[name.lower()], float(value))
    if rc != 0:
        raise JincError(rc, lib().jinc_last_error().decode())


def clear_knob(name: Optional[str] = None) -> None:
    """One knob (or, without a name, every knob) back to unset."""
    rc = lib().jinc_debug_clear_knob(-1 if name is None else knob_ids()[name.lower()])
    if rc != 0:
        raise JincError(rc, lib().jinc_last_error().decode())


def get_knob(name: str) -> Optional[float]:
    v = C.c_double()
    return v.value if lib().jinc_debug_get_knob(knob_ids()[name.lower()], C.byref(v)) == 1 else None


class knobs:
    """`with knobs(quad_rg=8): ...` sets knobs for a block and restores what was there before."""

    def __init__(self, **kv):
        self._kv = kv
        self._old = {}

    def __enter__(self):
        for k, v in self._kv.items():
            self._old[k] = get_knob(k)
            set_knob(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self._old.items():
            clear_knob(k) if v is None else set_knob(k, v)
        return False


# JINC_<NAME> environment variables the profiles/ scripts and recheck_rules.py pass: translated into knobs by bench.py (the
# library itself never reads the environment for them).
def apply_env_knobs(environ=None) -> dict:
    """Sets the knobs the environment names; JINC_* variables that name no knob are reported on stderr and returned under
    "unknown_variables" (a script that still sets a variable whose knob is gone measures A against A without a word).
    JINC_LIB, JINC_BENCH_* and the profiles/ scripts' own variables are not knobs by design."""
    environ = os.environ if environ is None else environ
    applied = {}
    known = {"JINC_" + n.upper() for n in knob_ids()}
    # (not knobs by design: the loader's JINC_LIB, bench.py's JINC_BENCH_*, the variables of the profiles/ scripts and of the tests)
    own = ("JINC_LIB", "JINC_FRAMES_PER_LAUNCH")
    own_prefixes = ("JINC_BENCH_", "JINC_PROFILE_", "JINC_LINE_", "JINC_TEST_")
    unknown = sorted(k for k in environ if k.startswith("JINC_") and k not in known and k not in own and not k.startswith(own_prefixes))
    if unknown:
        import sys
        print("avisynth_jincresize_amd: environment variables that name no knob (ignored): " + ", ".join(unknown), file=sys.stderr, flush=True)
        applied["unknown_variables"] = unknown
    for name in knob_ids():
        e = environ.get("JINC_" + name.upper())
        if e is None or e == "":
            continue
        if name == "pipeline_skip":
            e = {"h2d": "1", "kernels": "2"}.get(e, e)
        set_knob(name, float(e))
        applied[name] = float(e)
    return applied


def last_instance() -> str:
    """Full instantiation of the interior kernel (table 0) of the most recent kernel call of any filter instance in this process."""
    return (lib().jinc_debug_last_instance() or b"").decode()


def set_direct_shape(shape: int) -> None:
    """Process-wide A/B knob of ewa_direct_kernel's interior form (test hook): 0 per-chain fetches, 2 row walk, 3 row walk
    with 8 columns per lane (where it applies), -1 automatic."""
    rc = lib().jinc_debug_set_direct_shape(int(shape))
    if rc != 0:
        raise JincError(rc, lib().jinc_last_error().decode())


def last_direct_shape() -> int:
    """Interior form of the most recent ewa_direct_kernel interior launch in this process (test hook)."""
    return int(lib().jinc_debug_last_direct_shape())


def debug_convert(sums: np.ndarray, dtype, peak: float, device: int = 0) -> np.ndarray:
    """The kernels' sum -> sample conversion applied to `sums` on the device (test hook)."""
    sums = np.ascontiguousarray(sums, dtype=np.float32)
    out = np.zeros(sums.shape, dtype=dtype)
    rc = lib().jinc_debug_convert(sums.ctypes.data, out.ctypes.data, sums.size, out.dtype.itemsize, float(peak), device)
    if rc != 0:
        raise JincError(rc, lib().jinc_last_error().decode())
    return out


# ---- clip / format model (what an AviSynth+ host would provide) -----------------------------------
@dataclass(frozen=True)
class Format:
    name: str
    bits: int
    planes: int
    sub_w: int = 0
    sub_h: int = 0
    rgb: bool = False

    @property
    def sample_bytes(self) -> int:
        return 1 if self.bits == 8 else (2 if self.bits <= 16 else 4)

    @property
    def dtype(self):
        return {1: np.uint8, 2: np.uint16, 4: np.float32}[self.sample_bytes]

    def plane_dims(self, w: int, h: int) -> List[Tuple[int, int]]:
        dims = [(w, h)]
        if self.planes >= 3:
            dims += [(w >> self.sub_w, h >> self.sub_h)] * 2
        if self.planes == 4:
            dims.append((w, h))
        return dims

    def video_info(self, w: int, h: int) -> VideoInfo:
        return VideoInfo(w, h, self.bits, self.sample_bytes, self.planes, 1, int(self.rgb), self.sub_w, self.sub_h)


def _fmts() -> Dict[str, Format]:
    out = {}
    for bits, tag in ((8, "8"), (10, "10"), (12, "12"), (14, "14"), (16, "16"), (32, "S")):
        out[f"Y{tag if bits != 32 else '32'}"] = Format(f"Y{tag if bits != 32 else '32'}", bits, 1)
        for fam, sw, sh in (("420", 1, 1), ("422", 1, 0), ("444", 0, 0), ("411", 2, 0)):
            out[f"YUV{fam}P{tag}"] = Format(f"YUV{fam}P{tag}", bits, 3, sw, sh)
            out[f"YUVA{fam}P{tag}"] = Format(f"YUVA{fam}P{tag}", bits, 4, sw, sh)
        out[f"RGBP{tag}"] = Format(f"RGBP{tag}", bits, 3, rgb=True)
        out[f"RGBAP{tag}"] = Format(f"RGBAP{tag}", bits, 4, rgb=True)
    out["YV12"], out["YV16"], out["YV24"], out["YV411"] = out["YUV420P8"], out["YUV422P8"], out["YUV444P8"], out["YUV411P8"]
    return out


FORMATS = _fmts()


def alloc_plane(w: int, h: int, dtype, align: int = 64) -> np.ndarray:
    """Host plane with an AviSynth+-style pitch (row size rounded up to 64 bytes)."""
    isz = np.dtype(dtype).itemsize
    pitch = (w * isz + align - 1) // align * align
    return np.zeros((h, pitch // isz), dtype=dtype)


@dataclass
class Clip:
    """Minimal stand-in for an AviSynth clip: format, size, and a frame source returning planes in the
    reference's processing order (Y,U,V,A or G,B,R,A; ref JincResize.cpp:539-541)."""
    fmt: Format
    width: int
    height: int
    num_frames: int
    source: Callable[[int], Sequence[np.ndarray]]
    props: Dict[str, int] = field(default_factory=dict)  # frame properties of frame 0 (e.g. _ChromaLocation)
    planar: bool = True

    def get_frame(self, n: int) -> Sequence[np.ndarray]:
        return self.source(n)


def _build_args(fmt, width, height, target_width, target_height, frame0_chroma_location, planar, cpu_flags, alias_taps, kw):
    """jinc_video_info + jinc_args of a JincResize() call (keyword names = the script arguments)."""
    vi = fmt.video_info(width, height)
    vi.is_planar = int(planar)
    a = Args()
    a.target_width, a.target_height = int(target_width), int(target_height)
    a.frame0_chroma_location = int(frame0_chroma_location)
    a.cpu_has_sse41, a.cpu_has_avx2, a.cpu_has_avx512f = (int(bool(x)) for x in cpu_flags)
    keep = []
    for k, v in kw.items():
        if k not in ARG_BITS:
            raise TypeError(f"JincResize: unknown argument {k!r}")
        if v is None:
            continue
        a.defined |= ARG_BITS[k]
        if k == "cplace":
            b = str(v).encode()
            keep.append(b)
            a.cplace = b
        else:
            setattr(a, k, v)
    if alias_taps is not None:
        b = Args()
        rc = lib().jinc_alias_args(int(alias_taps), C.byref(a), C.byref(b))
        if rc != 0:
            raise JincError(rc, lib().jinc_last_error().decode())
        a = b
    return vi, a, keep


# register_host_buffers of set_pipeline / Batch (include/jincresize_hip.h): pageable planes copied by the CPU through pinned buffers of
# the library's own (the default) / registered once and cached (any other non-zero value) / pageable planes handed to the HIP runtime
# as they are (which maps them into the device itself behind every copy)
PIN_NONE, PIN_POOL, PIN_RUNTIME = 0, 2, 3


def numa_cpus(sysfs_root: str, bdf: str) -> List[int]:
    """batch.cpp's NUMA lookup against a sysfs tree of the caller's (test header)."""
    buf = (C.c_int * 4096)()
    n = int(lib().jinc_debug_numa_cpus(sysfs_root.encode(), bdf.encode(), buf, 4096))
    return [int(buf[i]) for i in range(max(0, min(n, 4096)))]


class Batch:
    """Frames of one clip sharded over the node's HIP devices (jinc_batch_*: frame n -> device n mod G, a plan replica
    and `streams` frames in flight per device, no exchange between devices)."""

    def __init__(self, fmt: Format, width: int, height: int, target_width: int, target_height: int, *, ndevices: int = 0,
                 streams: int = 2, register_host_buffers: bool = False, **kw):
        self.fmt = fmt
        vi, a, self._keep = _build_args(fmt, width, height, target_width, target_height, -1, True, (True, True, True), None, kw)
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().jinc_batch_create(C.byref(vi), C.byref(a), int(ndevices), int(streams), int(register_host_buffers),
                                     C.byref(self._h), err, len(err))
        if rc != 0:
            raise JincError(rc, err.value.decode())
        self.dst_w, self.dst_h = int(target_width), int(target_height)

    @property
    def devices(self) -> int:
        return int(lib().jinc_batch_devices(self._h))

    def set_affinity(self, on: bool) -> None:
        """Workers / registrars of device d on the CPUs of d's NUMA node (default) or wherever the scheduler puts them."""
        lib().jinc_batch_set_affinity(self._h, int(bool(on)))

    def set_registrars(self, n: int) -> None:
        """Test header: n registrar threads per process() call instead of one per device."""
        if lib().jinc_debug_batch_set_registrars(self._h, int(n)) != 0:
            raise JincError(-1, lib().jinc_batch_last_error().decode())

    def refused(self) -> Tuple[int, str]:
        """Test header: (host ranges hipHostRegister refused so far, what the first one was told)."""
        buf = C.create_string_buffer(256)
        return int(lib().jinc_debug_batch_refused(self._h, buf, len(buf))), buf.value.decode()

    def device_cpus(self, device_index: int) -> List[int]:
        buf = (C.c_int * 4096)()
        n = int(lib().jinc_batch_device_cpus(self._h, int(device_index), buf, 4096))
        return [int(buf[i]) for i in range(max(0, min(n, 4096)))]

    def device_of_frame(self, n: int) -> int:
        return int(lib().jinc_batch_device_of_frame(self._h, int(n)))

    def out_dims(self) -> List[Tuple[int, int]]:
        return self.fmt.plane_dims(self.dst_w, self.dst_h)

    def process(self, frames: Sequence[Sequence[np.ndarray]], outs: Optional[Sequence[Sequence[np.ndarray]]] = None):
        """frames[n] = the planes of frame n (all with the same pitches); returns outs[n] = its output planes."""
        n, np_ = len(frames), self.fmt.planes
        if outs is None:
            outs = [[alloc_plane(w, h, self.fmt.dtype) for (w, h) in self.out_dims()] for _ in range(n)]
        sp, dp = (C.c_void_p * (4 * n))(), (C.c_void_p * (4 * n))()
        spitch, dpitch = _I4(), _I4()
        for k in range(n):
            for i in range(np_):
                sp[4 * k + i], dp[4 * k + i] = frames[k][i].ctypes.data, outs[k][i].ctypes.data
                if frames[k][i].strides[0] != frames[0][i].strides[0] or outs[k][i].strides[0] != outs[0][i].strides[0]:
                    raise ValueError("all frames of a batch must share their pitches")
        for i in range(np_):
            if n:
                spitch[i], dpitch[i] = frames[0][i].strides[0], outs[0][i].strides[0]
        rc = lib().jinc_batch_process(self._h, n, sp, spitch, dp, dpitch)
        if rc != 0:
            raise JincError(rc, lib().jinc_batch_last_error().decode())
        return outs

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().jinc_batch_free(self._h)
            self._h = C.c_void_p()

    __del__ = close


def shard_device(frame: int, ndevices: int) -> int:
    """Device of frame `frame` in a shard over `ndevices` devices (jinc_shard_device; needs no GPU)."""
    return int(lib().jinc_shard_device(int(frame), int(ndevices)))


class Filter:
    """One filter instance (= one `JincResize` object of the reference)."""

    def __init__(self, fmt: Format, width: int, height: int, target_width: int, target_height: int, *,
                 device: int = 0, frame0_chroma_location: int = -1, planar: bool = True,
                 cpu_flags: Tuple[bool, bool, bool] = (True, True, True), alias_taps: Optional[int] = None, **kw):
        self.fmt = fmt
        vi, a, self._keep = _build_args(fmt, width, height, target_width, target_height, frame0_chroma_location, planar,
                                        cpu_flags, alias_taps, kw)
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = lib().jinc_filter_create(C.byref(vi), C.byref(a), int(device), C.byref(self._h), err, len(err))
        if rc != 0:
            raise JincError(rc, err.value.decode())
        out = VideoInfo()
        lib().jinc_filter_output_info(self._h, C.byref(out))
        self.src_w, self.src_h = width, height
        self.dst_w, self.dst_h = out.width, out.height
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().jinc_filter_free(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def _check(self, rc: int):
        if rc != 0:
            raise JincError(rc, lib().jinc_last_error().decode())

    @property
    def chroma_location(self) -> int:
        """What GetFrame writes to _ChromaLocation: 2 for every sub-sampled format (the reference binary's behaviour, ref
        :617-625 with d->cplace never assigned), -1 = not written; 0 / 1 / 2 by siting after set_chroma_location_mode(1)."""
        return int(lib().jinc_filter_chroma_location(self._h))

    def set_chroma_location_mode(self, mode: int) -> None:
        """0: as the reference binary (default); 1: by the siting in use (private switch, a deliberate deviation)."""
        self._check(lib().jinc_filter_set_chroma_location_mode(self._h, int(mode)))

    def out_dims(self) -> List[Tuple[int, int]]:
        return self.fmt.plane_dims(self.dst_w, self.dst_h)

    def set_kernel_mode(self, mode: int) -> None:
        self._check(lib().jinc_filter_set_kernel_mode(self._h, mode))

    def set_simd_order(self, order: int) -> None:
        """0: opt=0 results (default); 1 / 2 / 3: summation order of the reference's SSE4.1 / AVX2 / AVX-512 path."""
        self._check(lib().jinc_filter_set_simd_order(self._h, int(order)))

    # -- GetFrame on host planes (H2D, kernels, D2H) --
    def get_frame(self, src_planes: Sequence[np.ndarray]) -> List[np.ndarray]:
        n = self.fmt.planes
        outs = [alloc_plane(w, h, self.fmt.dtype) for (w, h) in self.out_dims()]
        sp, spitch, dp, dpitch = _P4(), _I4(), _P4(), _I4()
        for i in range(n):
            s = src_planes[i]
            if s.dtype != self.fmt.dtype:
                raise TypeError("plane dtype does not match the clip format")
            sp[i], spitch[i] = s.ctypes.data, s.strides[0]
            dp[i], dpitch[i] = outs[i].ctypes.data, outs[i].strides[0]
        self._check(lib().jinc_filter_get_frame(self._h, sp, spitch, dp, dpitch))
        return outs

    # -- look-ahead pipeline: several frames in flight per instance --
    def set_pipeline(self, depth: int, register_host_buffers: int = 0, group: int = 0) -> None:
        """Up to `depth` frames in flight; `group` of them coalesced into one launch (0: automatic = depth / 2).
        register_host_buffers: PIN_NONE (0 / False: the CPU copies the planes through the library's pinned buffers), PIN_POOL (non-zero /
        True: pinned once and cached by address; the caller keeps the buffers allocated) or PIN_RUNTIME (3: handed to the runtime)."""
        self._check(lib().jinc_filter_set_pipeline_group(self._h, int(depth), int(group), int(register_host_buffers)))

    @property
    def pipeline_group(self) -> int:
        return int(lib().jinc_filter_pipeline_group(self._h))

    def adopt_host_range(self, base: int, nbytes: int) -> None:
        """[base, base + nbytes) is pinned by the caller (hipHostMalloc / hipHostRegister) and stays so until close() or
        release_host_range()."""
        self._check(lib().jinc_filter_adopt_host_range(self._h, C.c_void_p(base), C.c_size_t(nbytes)))

    def release_host_range(self, base: int, nbytes: int) -> None:
        """The caller is about to unpin / free the range: frames in flight are waited for, adopted ranges touching it are forgotten."""
        self._check(lib().jinc_filter_release_host_range(self._h, C.c_void_p(base), C.c_size_t(nbytes)))

    @property
    def direct_premise(self) -> int:
        return int(lib().jinc_filter_direct_premise(self._h))

    def flush(self) -> None:
        """Launch the frames submitted so far (a client that knows no more are coming)."""
        self._check(lib().jinc_filter_flush(self._h))

    def submit(self, src_planes: Sequence[np.ndarray], dst_planes: Sequence[np.ndarray]) -> int:
        """Enqueue one frame; dst_planes (from alloc_plane) are filled when wait(ticket) returns."""
        sp, spitch, dp, dpitch = _P4(), _I4(), _P4(), _I4()
        for i in range(self.fmt.planes):
            sp[i], spitch[i] = src_planes[i].ctypes.data, src_planes[i].strides[0]
            dp[i], dpitch[i] = dst_planes[i].ctypes.data, dst_planes[i].strides[0]
        t = C.c_longlong()
        self._check(lib().jinc_filter_submit(self._h, sp, spitch, dp, dpitch, C.byref(t)))
        return t.value

    def wait(self, ticket: int) -> None:
        self._check(lib().jinc_filter_wait(self._h, C.c_longlong(ticket)))

    # -- device-resident batch (pointers are raw device addresses) --
    def process_device(self, src_ptrs, src_pitches, src_strides, dst_ptrs, dst_pitches, dst_strides, nframes: int,
                       stream: int = 0) -> None:
        sp, spitch, ss, dp, dpitch, ds = _P4(), _I4(), _S4(), _P4(), _I4(), _S4()
        for i in range(self.fmt.planes):
            sp[i], spitch[i], ss[i] = src_ptrs[i], src_pitches[i], src_strides[i]
            dp[i], dpitch[i], ds[i] = dst_ptrs[i], dst_pitches[i], dst_strides[i]
        self._check(lib().jinc_filter_process_device(self._h, sp, spitch, ss, dp, dpitch, ds, int(nframes),
                                                     C.c_void_p(stream)))

    def periodic_support(self, table: int = 0) -> int:
        """Taps per axis the periodic interior kernels execute (trimmed support on integer planes), 0: no periodic interior."""
        return int(lib().jinc_filter_periodic_support(self._h, int(table)))

    def periodic_taps(self, table: int = 0, rows_kernel=False) -> float:
        """Taps per output sample the periodic interior kernels execute (0: no periodic interior); rows_kernel: False / True = window
        and quad forms / rows kernel, 2 = the direct kernel's interior."""
        return float(lib().jinc_filter_periodic_taps(self._h, int(table), int(rows_kernel)))

    def interior_kernel(self, table: int = 0) -> str:
        return lib().jinc_filter_interior_kernel(self._h, int(table)).decode()

    def last_kernel(self, table: int = 0) -> str:
        """Interior kernel of the most recent frame call (depends on the batch size)."""
        return lib().jinc_filter_last_kernel(self._h, int(table)).decode()

    def last_border(self, table: int = 0) -> int:
        """Border kernels of the most recent frame call as bits (test header: jinc_filter_last_border)."""
        return int(lib().jinc_filter_last_border(self._h, int(table)))

    def last_instance(self, table: int = 0) -> str:
        """... with its template arguments, as rocprofv3 names it (the periodic family; the plain name otherwise)."""
        return lib().jinc_filter_last_instance(self._h, int(table)).decode()

    def set_border_strips(self, mode) -> None:
        """Border frame of exactly periodic plans: -1 by call size (default), True/1 strip kernels, 2 rows only, False/0 gather kernel."""
        self._check(lib().jinc_filter_set_border_strips(self._h, int(mode)))

    def set_border_overlap(self, enable) -> None:
        """True / False, or None for the automatic choice."""
        self._check(lib().jinc_filter_set_border_overlap(self._h, -1 if enable is None else int(bool(enable))))

    def set_profiling(self, enable: bool) -> None:
        self._check(lib().jinc_filter_set_profiling(self._h, int(enable)))

    def kernel_times(self):
        """(periodic_ms, periodic_launches, gather_ms, gather_launches) since the last call."""
        pm, gm, pn, gn = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        self._check(lib().jinc_filter_kernel_times(self._h, C.byref(pm), C.byref(pn), C.byref(gm), C.byref(gn)))
        return pm.value, pn.value, gm.value, gn.value

    def sync(self) -> None:
        self._check(lib().jinc_filter_sync(self._h))

    # -- plan introspection --
    @property
    def num_tables(self) -> int:
        return int(lib().jinc_filter_num_tables(self._h))

    def plan_info(self, table: int = 0) -> PlanInfo:
        info = PlanInfo()
        self._check(lib().jinc_filter_plan_info(self._h, table, C.byref(info)))
        return info

    def plan_dump(self, table: int = 0):
        info = self.plan_info(table)
        sx = np.zeros(info.dst_width, np.int32)
        sy = np.zeros(info.dst_height, np.int32)
        ids = np.zeros((info.dst_height, info.dst_width), np.int32)
        self._check(lib().jinc_filter_plan_dump(self._h, table, sx.ctypes.data, sy.ctypes.data, ids.ctypes.data))
        return sx, sy, ids

    def plan_runs(self, table: int = 0):
        """Rectangles of a drifting plan (what the runs form of the direct kernel walks): (array [n, 8] of set, x0, y0, sx0, sy0,
        ni, nj, first_item; number of items).  Empty for plans without runs.  Works without a device."""
        n, items = C.c_int(0), C.c_int(0)
        self._check(lib().jinc_filter_plan_runs(self._h, table, C.byref(n), C.byref(items), None, 0))
        runs = np.zeros((n.value, 8), np.int32)
        if n.value:
            self._check(lib().jinc_filter_plan_runs(self._h, table, C.byref(n), C.byref(items), runs.ctypes.data, n.value))
        return runs, items.value

    def plan_sets(self, table: int = 0) -> np.ndarray:
        info = self.plan_info(table)
        fs = info.filter_size
        out = np.zeros((info.num_sets, fs, fs), np.float32)
        for s in range(info.num_sets):
            self._check(lib().jinc_filter_plan_set(self._h, table, s, out[s].ctypes.data))
        return out

    def lut(self) -> np.ndarray:
        a = np.zeros(1024, np.float64)
        self._check(lib().jinc_filter_lut(self._h, a.ctypes.data))
        return a


# ---- script-level functions (ref JincResize.cpp:1044-1108) ---------------------------------------
def _make(clip: Clip, target_width: int, target_height: int, device: int, alias_taps: Optional[int], **kw) -> Clip:
    cl = clip.props.get("_ChromaLocation", -1)
    flt = Filter(clip.fmt, clip.width, clip.height, target_width, target_height, device=device,
                 frame0_chroma_location=cl if isinstance(cl, int) else -1, planar=clip.planar,
                 alias_taps=alias_taps, **kw)
    props = dict(clip.props)
    if flt.chroma_location >= 0:
        props["_ChromaLocation"] = flt.chroma_location  # ref :617-625

    out = Clip(clip.fmt, flt.dst_w, flt.dst_h, clip.num_frames, lambda n: flt.get_frame(clip.get_frame(n)), props)
    out.filter = flt
    return out


def JincResize(clip: Clip, target_width: int, target_height: int, src_left=None, src_top=None, src_width=None,
               src_height=None, quant_x=None, quant_y=None, tap=None, blur=None, cplace=None, threads=None, opt=None,
               initial_capacity=None, initial_factor=None, *, device: int = 0) -> Clip:
    return _make(clip, target_width, target_height, device, None, src_left=src_left, src_top=src_top,
                 src_width=src_width, src_height=src_height, quant_x=quant_x, quant_y=quant_y, tap=tap, blur=blur,
                 cplace=cplace, threads=threads, opt=opt, initial_capacity=initial_capacity,
                 initial_factor=initial_factor)


def _alias(taps: int):
    def fn(clip: Clip, target_width: int, target_height: int, src_left=None, src_top=None, src_width=None,
           src_height=None, quant_x=None, quant_y=None, cplace=None, threads=None, *, device: int = 0) -> Clip:
        return _make(clip, target_width, target_height, device, taps, src_left=src_left, src_top=src_top,
                     src_width=src_width, src_height=src_height, quant_x=quant_x, quant_y=quant_y, cplace=cplace,
                     threads=threads)
    return fn


Jinc36Resize = _alias(3)
Jinc64Resize = _alias(4)
Jinc144Resize = _alias(6)
Jinc256Resize = _alias(8)
