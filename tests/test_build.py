"""Build products: the C-ABI library loads on a CPU-only box, exports every symbol the header
declares, contains no fused multiply-add in its device code, and the product never touches the
oracle.  CPU only -- no compute calls."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    header = open(pkg.HEADER_PATH).read() + open(pkg.TEST_HEADER_PATH).read()   # every header under include/
    declared = re.findall(r"JINC_API\s+[\w\s\*]+?\b(jinc_\w+)\s*\(", header)
    assert len(declared) >= 15
    boundary = re.findall(r"JINC_API\s+[\w\s\*]+?\b(jinc_\w+)\s*\(", open(pkg.HEADER_PATH).read())
    assert not [n for n in boundary if n.startswith(("jinc_debug", "jinc_filter_plan", "jinc_filter_set_kernel_mode"))]
    assert sorted(set(declared)) == sorted(pkg.EXPORTS)
    nm = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\sT\s+(jinc_\w+)", nm))
    assert set(declared) <= exported
    L = pkg.lib()
    for s in declared:
        assert hasattr(L, s)


def test_library_loads_without_gpu_and_reports_devices(pkg):
    n = pkg.device_count()  # hipGetDeviceCount: 0 on the CPU box, no crash
    assert n >= 0
    picks = [pkg.lib().jinc_pick_device() for _ in range(5)]
    assert picks == ([-1] * 5 if n == 0 else [k % n for k in range(picks[0], picks[0] + 5)])


def test_no_fused_multiply_add_in_device_code(pkg):
    """The opt=0 result is defined by separate v_mul_f32 / v_add_f32."""
    if not all(os.path.exists(p) for p in pkg.ISA_PATHS):
        pkg.build()
    isa = "\n".join(open(p).read() for p in pkg.ISA_PATHS)
    kernels = re.findall(r"^(_ZN4jinc\S*kernel\S*):", isa, flags=re.M)
    for name in ("ewa_gather_kernel", "ewa_periodic_kernel", "ewa_periodic_rows_kernel", "ewa_periodic_pk_kernel", "ewa_periodic_quad_kernel", "ewa_periodic_quad2_kernel", "ewa_periodic_quad8_kernel", "ewa_periodic_quad2x8_kernel",
                 "ewa_quasi_kernel", "ewa_framelane_kernel", "ewa_framelane_win_kernel", "ewa_framelane_win1k_kernel", "ewa_framelane_sub_kernel", "ewa_framelane_pair_kernel", "ewa_direct_kernel", "ewa_colstrip_kernel"):
        assert any(name in k for k in kernels), name
    fused = re.findall(r"^\s+(v_fma_f32|v_fmac_f32|v_mad_f32|v_mac_f32|v_pk_fma_f32|v_fma_mix\w*|v_mfma\w*)\b", isa, flags=re.M)
    assert fused == [], f"fused ops in device code: {sorted(set(fused))}"
    assert len(re.findall(r"^\s+v_mul_f32", isa, flags=re.M)) > 100
    assert len(re.findall(r"^\s+v_add_f32", isa, flags=re.M)) > 100
    # the packed forms (frame-pair kernel, packed periodic variant) are un-fused too: v_pk_mul_f32 + v_pk_add_f32, never v_pk_fma_f32
    assert len(re.findall(r"^\s+v_pk_mul_f32", isa, flags=re.M)) > 100 and len(re.findall(r"^\s+v_pk_add_f32", isa, flags=re.M)) > 100
    # ... except in the SIMD-order compatibility kernels, which emulate the reference's FMA paths on purpose
    so = open(pkg.SIMD_ORDER_ISA_PATH).read()
    assert re.findall(r"^\s+v_fma_f32", so, flags=re.M), "kernel_simdorder.hip must use explicit FMAs for orders 2 and 3"
    assert set(re.findall(r"^(_ZN4jinc\S*kernel\S*):", so, flags=re.M)) and all(
        "ewa_simd_order_kernel" in k for k in re.findall(r"^(_ZN4jinc\S*kernel\S*):", so, flags=re.M))
    # fp32 denormals must be preserved (float_denorm_mode_32 = 3 in every kernel descriptor)
    modes = re.findall(r"\.amdhsa_float_denorm_mode_32\s+(\d+)", isa)
    assert modes and set(modes) == {"3"}


def _vgprs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def test_dpp_reads_keep_their_distance_from_valu_writes(pkg):
    """ewa_framelane_sub_kernel multiplies by a coefficient another lane holds (v_mul_f32_dpp ... row_newbcast / quad_perm, written
    as inline assembly, which the compiler's hazard recogniser does not look into).  The hardware needs two wait states between a
    VALU write of a register and a DPP read of it: in the listing no VALU instruction within the two instructions (or s_nop
    states) in front of a DPP multiply may write that multiply's DPP operand.  (The operand comes straight from a vector load.)"""
    path = [p for p in pkg.ISA_PATHS if p.endswith("kernel_framelane_sub-gfx950.s")][0]
    if not os.path.exists(path):
        pkg.build()
    seen, hist = 0, []  # hist: (is VALU, registers written, wait states it provides)
    for line in open(path):
        t = line.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        if t.endswith(":"):
            hist = []
            continue
        parts = t.replace(",", " ").split()
        op = parts[0]
        if op == "v_mul_f32_dpp":
            seen += 1
            src, states = _vgprs(parts[2]), 0
            for valu, written, n in reversed(hist):
                if states >= 2:
                    break
                assert not (valu and written & src), f"VALU write of {sorted(written & src)} right in front of: {t}"
                states += n
        if op == "s_nop":
            hist.append((False, set(), int(parts[1], 0) + 1))
        else:
            valu = op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane"))
            hist.append((valu, _vgprs(parts[1]) if valu and len(parts) > 1 else set(), 1))
        hist = hist[-6:]
    assert seen > 1000  # every tap of every instantiation


def test_product_does_not_reference_the_oracle():
    pkg_dir = os.path.join(ROOT, "avisynth-jincresize_amd")
    product_dirs = [pkg_dir, os.path.join(ROOT, "plugin"), os.path.join(ROOT, "include")]
    for base, _, files in (entry for d in product_dirs for entry in os.walk(d)):
        if os.path.basename(base) in ("build", "lib", "__pycache__"):
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".h", ".hip", ".inc", ".hpp", "Makefile")):
                text = open(os.path.join(base, fn), errors="ignore").read()
                assert "oracle" not in text.lower() or fn == "__init__.py" and "oracle" not in text.lower(), \
                    f"{fn} mentions the oracle"
    ldd = subprocess.run(["ldd", os.path.join(pkg_dir, "lib", "libjincresize_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


def test_reference_is_not_needed_at_runtime():
    """Nothing under tests/ (gpu or not), bench.py or __graft_entry__.py may read /root/reference."""
    offenders = []
    for fn in ["bench.py", "__graft_entry__.py"] + [os.path.join("tests", f) for f in os.listdir(os.path.join(ROOT, "tests")) if f.endswith(".py")]:
        p = os.path.join(ROOT, fn)
        if not os.path.exists(p) or fn.endswith("test_build.py"):
            continue
        text = open(p).read()
        if re.search(r"open\([^)]*root/reference|listdir\([^)]*root/reference", text):
            offenders.append(fn)
    assert offenders == []
