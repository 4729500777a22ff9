"""jinc_filter_adopt_host_range: a host that pins its own frame memory (here: one torch pinned-memory pool that holds every
source and destination plane) tells the instance once; frames inside it travel like frames the instance pinned itself."""
import numpy as np
import pytest

from conftest import assert_planes_equal

pytestmark = pytest.mark.gpu


def test_planes_in_a_caller_pinned_pool(gpu_pkg, O):
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th = "Y8", 320, 180, 438, 246
    n = 40
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    spitch, dpitch = 320, 448
    pool = torch.empty(n * (spitch * sh + dpitch * th) + 4096, dtype=torch.uint8, pin_memory=True)
    host = pool.numpy()
    srcs, dsts, frames = [], [], []
    off = 0
    for k in range(n):
        s = host[off:off + spitch * sh].reshape(sh, spitch); off += spitch * sh
        d = host[off:off + dpitch * th].reshape(th, dpitch); off += dpitch * th
        fr = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=77 + k)
        s[:, :sw] = fr[0][:, :sw]
        d[:] = 0xEE
        srcs.append([s]); dsts.append([d]); frames.append(fr)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_pipeline(32, False)                       # the instance itself pins nothing
    f.adopt_host_range(pool.data_ptr(), pool.numel())
    f.adopt_host_range(pool.data_ptr() + 4096, 1 << 20)   # a range inside a known one: accepted, nothing to do
    with pytest.raises(gpu_pkg.JincError):
        f.adopt_host_range(np.zeros(1 << 16, np.uint8).ctypes.data, 1 << 16)   # pageable memory is not pinned
    tickets = [f.submit(srcs[k], dsts[k]) for k in range(n)]
    for k in reversed(range(n)):
        f.wait(tickets[k])
    assert f.last_kernel(0).startswith("ewa_framelane") or n % 16   # groups of 16 frames of a plan without phase structure
    for k in range(n):
        assert_planes_equal(dsts[k], of.get_frame(frames[k], threads=4), f.out_dims(), what=f"frame {k}")
        assert (dsts[k][0][:, tw:] == 0xEE).all(), "bytes between the rows of a destination plane were written"
    f.close()
