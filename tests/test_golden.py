"""Committed golden vectors (tests/golden/vectors.npz, made by tests/golden/make_golden.py).
CPU: the oracle still reproduces them.  GPU: the HIP path reproduces them through the C ABI.

What these vectors are and are not: the ORACLE wrote them (make_golden.py runs oracle/, nothing of the reference: it cannot be
built in this image), so the GPU half of this file is the oracle comparison of tests/test_gpu_parity.py under another name -- a
regression net that pins today's oracle output against tomorrow's edits of the oracle itself, not a pin of the oracle to the
reference.  The only reference-derived figures in the repository are the crc32 / sha prefixes and table statistics of
tests/golden/kat.json (recorded by the survey from a build of the reference; see DESIGN.md section 2, "parity unpinned")."""
import json
import os

import numpy as np
import pytest

from conftest import assert_planes_equal

HERE = os.path.join(os.path.dirname(__file__), "golden")
INDEX = json.load(open(os.path.join(HERE, "vectors.json")))
VEC = np.load(os.path.join(HERE, "vectors.npz"))
TO_SCRIPT = {"crop_left": "src_left", "crop_top": "src_top", "crop_width": "src_width", "crop_height": "src_height"}


def _planes(name, kind, n, alloc, fmt_dtype):
    out = []
    for i in range(n):
        a = VEC[f"{name}.{kind}{i}"]
        p = alloc(a.shape[1], a.shape[0], fmt_dtype)
        p[:, :a.shape[1]] = a
        out.append(p)
    return out


@pytest.mark.parametrize("v", INDEX, ids=lambda v: v["name"])
def test_oracle_reproduces_golden(O, v):
    F = O.FORMATS[v["format"]]
    src = _planes(v["name"], "src", v["planes"], O.alloc_plane, F.dtype)
    flt = O.OracleFilter(F, *v["src"], *v["dst"], **v["args"])
    out = flt.get_frame(src)
    want = [VEC[f"{v['name']}.dst{i}"] for i in range(v["planes"])]
    assert_planes_equal(out, want, flt.out_dims(), v["name"])
    assert O.crc32_planes(out, flt.out_dims()) == v["crc32"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1], ids=["auto", "gather"])
@pytest.mark.parametrize("v", INDEX, ids=lambda v: v["name"])
def test_hip_path_reproduces_golden(gpu_pkg, v, mode):
    F = gpu_pkg.FORMATS[v["format"]]
    src = _planes(v["name"], "src", v["planes"], gpu_pkg.alloc_plane, F.dtype)
    kw = {TO_SCRIPT.get(k, k): x for k, x in v["args"].items()}
    f = gpu_pkg.Filter(F, *v["src"], *v["dst"], device=0, **kw)
    f.set_kernel_mode(mode)
    out = f.get_frame(src)
    want = [VEC[f"{v['name']}.dst{i}"] for i in range(v["planes"])]
    assert_planes_equal(out, want, f.out_dims(), v["name"])
    f.close()
