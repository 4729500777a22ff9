"""pytest configuration: `gpu` marker, package/oracle loaders, shared helpers."""
import os
import sys

import numpy as np
import pytest

# PyTorch-ROCm bundles its own HIP runtime; when a process uses both torch and libjincresize_hip.so the
# runtime that is loaded first serves both, and only torch's own copy is known to work for torch.  The
# device-batch test passes torch tensors to the C ABI, so torch goes first (plumbing, not the product).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes binding of libjincresize_hip.so). Builds it if the .so is missing."""
    p = entry.load_package()
    if not os.path.exists(p.LIB_PATH):  # (the ISA listings are only needed by tests/test_build.py)
        p.build()
    p.lib()
    return p


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    o = entry.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    if pkg.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible (the HIP path must not be skipped silently)")
    return pkg


@pytest.fixture(autouse=True)
def _knobs_do_not_outlive_a_test():
    """The library's A/B knobs are process-wide (test header): whatever a test set is unset again when it ends, pass or fail."""
    yield
    p = sys.modules.get(entry.PKG_NAME)
    if p is not None and getattr(p, "_lib", None) is not None:
        p.clear_knob()


def crop_planes(planes, dims):
    return [np.ascontiguousarray(p[:h, :w]) for p, (w, h) in zip(planes, dims)]


def assert_planes_equal(got, want, dims, what=""):
    """Bit-exact comparison of the visible part of each plane (padding excluded)."""
    for i, (w, h) in enumerate(dims):
        a = np.ascontiguousarray(got[i][:h, :w])
        b = np.ascontiguousarray(want[i][:h, :w])
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            y, x = bad[0]
            raise AssertionError(f"{what}: plane {i} differs at {len(bad)} samples; first (x={x}, y={y}): "
                                 f"got {got[i][y, x]!r}, want {want[i][y, x]!r}")


def oracle_kwargs(kw):
    """Maps JincResize script-argument names to OracleFilter field names."""
    m = {"src_left": "crop_left", "src_top": "crop_top", "src_width": "crop_width", "src_height": "crop_height"}
    return {m.get(k, k): v for k, v in kw.items()}
