"""profiles/recheck_rules.py re-measures the dispatch rules on a GPU box; here (CPU) only that its list is not stale: every
configuration it names exists in bench.py, every JINC_<KNOB> variable it sets names a knob of the test header's enum jinc_knob
(bench.py translates the variable into jinc_debug_set_knob; the library reads no environment) that the library's sources consult,
every kernel mode is one jinc_filter_set_kernel_mode accepts."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_check_names_things_that_exist():
    tool = _load(os.path.join(ROOT, "profiles", "recheck_rules.py"), "recheck_rules")
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_for_rules")
    csrc = os.path.join(ROOT, "avisynth-jincresize_amd", "csrc")
    sources = "\n".join(open(os.path.join(csrc, f), errors="ignore").read() for f in os.listdir(csrc) if f.endswith((".cpp", ".hip", ".h", ".inc")))
    header = open(os.path.join(ROOT, "include", "jincresize_hip_test.h")).read()
    assert len(tool.CHECKS) >= 15
    rules_named = 0
    for rule, cfg, frames, choice, other, other_name in tool.CHECKS:
        assert cfg in bench.CONFIGS, (rule, cfg)
        assert frames >= 1 and other_name
        for variant in (choice, other):
            for knob in variant.get("env", {}):
                assert knob.startswith("JINC_") and f"JINC_KNOB_{knob[5:]}" in header, f"{rule}: {knob} is not a knob of the test header"
                assert f"JINC_KNOB_{knob[5:]}" in sources, f"{rule}: no source consults {knob}"
            args = variant.get("args", [])
            if "--kernel-mode" in args:
                assert 0 <= int(args[args.index("--kernel-mode") + 1]) <= 16
        m = re.match(r"(k[A-Z]\w+)", rule)
        if m:
            assert m.group(1) in sources, f"{rule}: no such constant in the sources"
            rules_named += 1
    assert rules_named >= 6
