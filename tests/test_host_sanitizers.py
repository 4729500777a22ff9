"""AddressSanitizer + UndefinedBehaviorSanitizer over the host side of the library (argument handling, LUT,
plan builder, plan introspection) on the CPU: the host sources are rebuilt with -fsanitize=address,undefined,
linked with the regular kernel objects and driven through the C ABI with device = -1 (no GPU involved; GPU
sanitizers are not available on this pool)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "avisynth-jincresize_amd")
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def test_host_code_is_clean_under_asan_ubsan(pkg, tmp_path):
    if not os.path.exists(CXX):
        pytest.skip("clang++ of the ROCm toolchain not found")
    kernel_objs = sorted(p for p in glob.glob(os.path.join(PKG, "build", "kernel_*.o")) if "amdgcn" not in p)
    if len(kernel_objs) < 6:
        pkg.build()
        kernel_objs = sorted(p for p in glob.glob(os.path.join(PKG, "build", "kernel_*.o")) if "amdgcn" not in p)
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    objs = []
    for name in ("filter", "filter_args", "device_plan", "dispatch", "pipeline", "host_copy", "batch", "plan", "jinc_lut", "quasi_dispatch",
                 "framelane_dispatch", "knobs"):
        o = str(tmp_path / f"{name}.o")
        subprocess.run([CXX, "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", *san, "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-c", os.path.join(PKG, "csrc", f"{name}.cpp"), "-o", o], check=True)
        objs.append(o)
    main_o = str(tmp_path / "main.o")
    subprocess.run([CXX, "-O1", "-g", "-std=c++17", *san, "-I", os.path.join(ROOT, "include"), "-c",
                    os.path.join(ROOT, "tests", "host_sanitizer", "main.cpp"), "-o", main_o], check=True)
    exe = str(tmp_path / "asan_test")
    subprocess.run([CXX, *san, main_o, *objs, *kernel_objs, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
                    "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:protect_shadow_gap=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-4000:]
    assert "periodic 1 quasi 1" in out and "smaller than the filter footprint" in out
    assert "plane copies on the helper threads: 0 wrong" in out
