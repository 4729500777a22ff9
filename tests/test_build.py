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
    for name in ("ewa_gather_kernel", "ewa_periodic_kernel", "ewa_periodic_rows_kernel", "ewa_periodic_pk_kernel", "ewa_periodic_quad_kernel", "ewa_periodic_quad2_kernel", "ewa_periodic_quad8_kernel", "ewa_periodic_quad2x8_kernel", "ewa_periodic_rowpair_kernel",
                 "ewa_quasi_kernel", "ewa_framelane_kernel", "ewa_framelane_win_kernel", "ewa_framelane_win1k_kernel", "ewa_framelane_sub_kernel", "ewa_framelane_pair_kernel", "ewa_direct_kernel", "ewa_colstrip_kernel", "ewa_strip_kernel", "ewa_colpair_kernel"):
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
    VALU write of a register and a DPP read of it: in the listing no VALU instruction within the two wait states in front of a
    DPP multiply may write that multiply's DPP operand -- along EVERY way control reaches the multiply (the first
    form of this test forgot its history at each label; now a label's predecessors are the instruction in front of it, unless
    that one never falls through, and every branch that names it -- a loop's back edge included).  The listing is the one the
    build saved from the very compile that produced the library (Makefile: -save-temps, lib/%.s from build/%.o)."""
    path = [p for p in pkg.ISA_PATHS if p.endswith("kernel_framelane_sub-gfx950.s")][0]
    if not os.path.exists(path):
        pkg.build()
    assert os.path.getmtime(path) <= os.path.getmtime(pkg.LIB_PATH) + 1.0, "the ISA listing is newer than the library: rebuild"
    insts, labels = [], {}   # insts: (op, parts, label-or-None in front); labels: name -> index of the instruction it precedes
    pending = []
    for line in open(path):
        t = line.split(";")[0].strip()
        if not t or t.startswith("."):
            if t.startswith(".") and t.endswith(":"):   # local labels (.LBB0_3:) begin with a dot
                pending.append(t[:-1])
            continue
        if t.endswith(":"):
            pending.append(t[:-1])
            continue
        parts = t.replace(",", " ").split()
        for name in pending:
            labels[name] = len(insts)
        insts.append((parts[0], parts, bool(pending)))
        pending = []
    branches_to = {}   # instruction index -> indices of the branches that jump to it
    for i, (op, parts, _) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")) and len(parts) > 1 and parts[1] in labels:
            branches_to.setdefault(labels[parts[1]], []).append(i)

    def falls_through(op):
        return not op.startswith(("s_branch", "s_endpgm", "s_setpc", "s_swappc"))

    def wait_states(op, parts):
        return int(parts[1], 0) + 1 if op == "s_nop" else 1

    def check(i, src, budget, text, depth=0):
        """Walks backwards from instruction i (exclusive) over every predecessor until `budget` wait states are covered."""
        if budget <= 0 or depth > 8:
            return
        preds = []
        if i > 0 and falls_through(insts[i - 1][0]):
            preds.append(i - 1)
        if insts[i][2]:
            preds += branches_to.get(i, [])
        for j in preds:
            op, parts, _ = insts[j]
            valu = op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane"))
            written = _vgprs(parts[1]) if valu and len(parts) > 1 else set()
            assert not (written & src), f"VALU write of {sorted(written & src)} ({' '.join(parts)}) within two wait states of: {text}"
            check(j, src, budget - wait_states(op, parts), text, depth + 1)

    seen = 0
    for i, (op, parts, _) in enumerate(insts):
        if op == "v_mul_f32_dpp":
            seen += 1
            check(i, _vgprs(parts[2]), 2, " ".join(parts))
    assert seen > 1000  # every tap of every instantiation
    assert branches_to, "no branch target was recognised: the label syntax of the listing has changed"


def test_no_environment_variable_steers_the_product(pkg):
    """the A/B knobs live behind include/jincresize_hip_test.h (jinc_debug_set_knob); the library neither
    spells a JINC_* variable nor imports getenv, and the host sources do not call it.  (The plugin shells keep their documented
    user switches, JINCRESIZE_*: INTEGRATION.md.)"""
    strings = subprocess.run(["strings", "-a", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines()
    assert [x for x in strings if x.startswith("JINC_")] == []
    undefined = subprocess.run(["nm", "-D", "--undefined-only", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    csrc = os.path.join(ROOT, "avisynth-jincresize_amd", "csrc")
    for fn in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, fn), errors="ignore").read(), fn
    names = pkg.knob_ids()
    assert len(names) >= 30 and all(n == n.lower() for n in names)
    assert pkg.get_knob("quad_rg") is None
    with pkg.knobs(quad_rg=8, fl_colw=0.25):
        assert pkg.get_knob("quad_rg") == 8.0 and pkg.get_knob("fl_colw") == 0.25
    assert pkg.get_knob("quad_rg") is None and pkg.get_knob("fl_colw") is None
    env = {"JINC_QUAD_RG": "4", "JINC_PIPELINE_SKIP": "kernels", "JINC_UNRELATED": "1"}
    assert pkg.apply_env_knobs(env) == {"quad_rg": 4.0, "pipeline_skip": 2.0, "unknown_variables": ["JINC_UNRELATED"]}  # (reported, not silently ignored: ADVICE r5)
    pkg.clear_knob()
    assert pkg.get_knob("quad_rg") is None


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


def test_library_is_not_unloadable(pkg):
    """The library parks helper threads (csrc/host_copy.cpp) and keeps process-wide registries: dlclose() by a host that unloads its
    plugins must not unmap it (-z nodelete, avisynth-jincresize_amd/Makefile)."""
    import subprocess
    out = subprocess.run(["readelf", "-d", pkg.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "NODELETE" in out
