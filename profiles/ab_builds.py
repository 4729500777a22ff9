#!/usr/bin/env python3
"""Same-box A/B of the headline kernel ACROSS BUILDS of the library: every build runs the same C2
batch through jinc_filter_process_device in a process of its own, builds alternating, several rounds; only entry points
that every build has are bound (create / process_device / kernel timing / free).

usage: python profiles/ab_builds.py [--frames 64 1024] [--rounds 5] name=path/to/libjincresize_hip.so ...
       (internal: --child <lib> <frames> runs one measurement and prints one JSON line)"""
import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib_path, frames, steps=20):
    import torch  # first: its HIP runtime serves the process

    class VideoInfo(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("width", "height", "bits", "size", "ncomp", "planar", "rgb", "sub_w", "sub_h")]

    class Args(C.Structure):
        _fields_ = [("tw", C.c_int), ("th", C.c_int), ("sl", C.c_double), ("st", C.c_double), ("sw", C.c_double), ("sh", C.c_double),
                    ("qx", C.c_int), ("qy", C.c_int), ("tap", C.c_int), ("blur", C.c_double), ("cplace", C.c_char_p), ("threads", C.c_int),
                    ("opt", C.c_int), ("icap", C.c_int), ("ifac", C.c_double), ("defined", C.c_uint), ("f0", C.c_int),
                    ("sse41", C.c_int), ("avx2", C.c_int), ("avx512", C.c_int)]

    L = C.CDLL(lib_path)
    P4, I4, S4 = C.c_void_p * 4, C.c_int * 4, C.c_size_t * 4
    L.jinc_filter_create.argtypes = [C.POINTER(VideoInfo), C.POINTER(Args), C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    L.jinc_filter_process_device.argtypes = [C.c_void_p, P4, I4, S4, P4, I4, S4, C.c_int, C.c_void_p]
    L.jinc_filter_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.jinc_filter_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.jinc_filter_free.argtypes = [C.c_void_p]
    sw, sh, dw, dh = 1920, 1080, 3840, 2160
    vi = VideoInfo(sw, sh, 8, 1, 1, 1, 0, 0, 0)
    a = Args()
    a.tw, a.th, a.tap, a.defined, a.f0 = dw, dh, 3, 1 << 6, -1
    h = C.c_void_p()
    err = C.create_string_buffer(256)
    assert L.jinc_filter_create(C.byref(vi), C.byref(a), 0, C.byref(h), err, 256) == 0, err.value
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(12345)
    src = torch.empty((frames, sh, sw), dtype=torch.uint8, device="cuda")
    for f0 in range(0, frames, 64):
        n = min(64, frames - f0)
        src[f0:f0 + n] = torch.randint(0, 256, (n, sh, sw), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    dst = torch.zeros((frames, dh, dw), dtype=torch.uint8, device="cuda")
    sp, dp = P4(src.data_ptr()), P4(dst.data_ptr())
    spitch, dpitch, ss, ds = I4(sw), I4(dw), S4(sw * sh), S4(dw * dh)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        assert L.jinc_filter_process_device(h, sp, spitch, ss, dp, dpitch, ds, frames, C.c_void_p(stream)) == 0

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:   # clocks
        step()
        torch.cuda.synchronize()
    L.jinc_filter_set_profiling(h, 1)
    pm, gm, pn, gn = C.c_double(), C.c_double(), C.c_int(), C.c_int()
    L.jinc_filter_kernel_times(h, C.byref(pm), C.byref(pn), C.byref(gm), C.byref(gn))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    L.jinc_filter_kernel_times(h, C.byref(pm), C.byref(pn), C.byref(gm), C.byref(gn))
    crc = int(dst[frames - 1].to(torch.int64).sum().item())   # same inputs -> same outputs, build to build
    print(json.dumps({"frames": frames, "Gpix_s": frames * steps * dw * dh / wall / 1e9, "interior_ms_per_launch": pm.value / max(1, pn.value),
                      "launches": pn.value, "border_ms_per_step": gm.value / steps, "checksum_last_frame": crc}))
    L.jinc_filter_free(h)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, nargs="+", default=[64, 1024])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("builds", nargs="+")
    a = ap.parse_args()
    builds = [b.split("=", 1) for b in a.builds]
    for frames in a.frames:
        res = {n: [] for n, _ in builds}
        for _ in range(a.rounds):
            for n, path in builds:   # alternating: one process per build and measurement
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(path), str(frames)],
                                     capture_output=True, text=True, timeout=300)
                line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
                if out.returncode != 0 or not line:
                    print(f"# {n}: failed: {out.stderr[-500:]}", flush=True)
                    continue
                res[n].append(json.loads(line[-1]))
        for n, _ in builds:
            r = res[n]
            if not r:
                continue
            g = [x["Gpix_s"] for x in r]
            k = [x["interior_ms_per_launch"] for x in r]
            print(json.dumps({"build": n, "frames_per_launch": frames, "rounds": len(r), "Gpix_s_median": round(statistics.median(g), 1),
                              "Gpix_s_all": [round(x, 1) for x in g], "interior_ms_per_launch_median": round(statistics.median(k), 4),
                              "launches_per_step": r[0]["launches"] // 20, "border_ms_per_step": round(statistics.median(x["border_ms_per_step"] for x in r), 4),
                              "checksum_last_frame": r[0]["checksum_last_frame"]}), flush=True)


if __name__ == "__main__":
    main()
