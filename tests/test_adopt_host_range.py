"""jinc_filter_adopt_host_range: a host that pins its own frame memory (here: one torch pinned-memory pool that holds every
source and destination plane) tells the instance once; frames inside it travel like frames the instance pinned itself."""
import numpy as np
import pytest

from conftest import assert_planes_equal, fresh_mapping

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


def test_pieces_of_one_pool_adopted_one_by_one_become_one_range(gpu_pkg, O):
    """a registrar that pins a frame pool piece by piece hands the pieces over one by one; a plane that begins in
    one piece and ends in the next must still travel by the shader, so adopted ranges that continue each other (in host and
    in device addresses) are merged."""
    torch = pytest.importorskip("torch")
    fmt, sw, sh, tw, th = "Y8", 320, 180, 438, 246
    n = 24
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    spitch, dpitch = 320, 448
    per_frame = spitch * sh + dpitch * th
    pool = torch.empty(n * per_frame + 8192, dtype=torch.uint8, pin_memory=True)
    host = pool.numpy()
    off = 100      # deliberately not aligned to anything: planes straddle the piece boundaries below
    srcs, dsts, frames = [], [], []
    for k in range(n):
        s = host[off:off + spitch * sh].reshape(sh, spitch); off += spitch * sh
        d = host[off:off + dpitch * th].reshape(th, dpitch); off += dpitch * th
        fr = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=177 + k)
        s[:, :sw] = fr[0][:, :sw]
        d[:] = 0xEE
        srcs.append([s]); dsts.append([d]); frames.append(fr)
    f = gpu_pkg.Filter(gpu_pkg.FORMATS[fmt], sw, sh, tw, th, device=0)
    f.set_pipeline(16, False)
    piece = 3 * 4096 * 7    # pieces of 21 pages, in an order that needs merging on both sides
    total = pool.numel()
    starts = list(range(0, total, piece))
    for a in starts[1::2] + starts[0::2]:
        f.adopt_host_range(pool.data_ptr() + a, min(piece, total - a))
    gpu_pkg.transport_counts(reset=True)
    tickets = [f.submit(srcs[k], dsts[k]) for k in range(n)]
    for t in tickets:
        f.wait(t)
    by_shader, by_dma, _ = gpu_pkg.transport_counts()
    assert (by_shader, by_dma) == (n, 0)
    for k in range(n):
        assert_planes_equal(dsts[k], of.get_frame(frames[k], threads=4), f.out_dims(), what=f"frame {k}")
    f.close()


def test_batch_registrar_on_one_contiguous_unaligned_pool_keeps_the_shader_transport(gpu_pkg, O, pooling_host):
    """jinc_batch_process pins 16 frames at a time, clipped to what is pinned already; with one contiguous,
    unaligned frame pool every chunk boundary falls inside a plane.  Every frame must still leave by the shader."""
    fmt, sw, sh, tw, th = "Y8", 320, 180, 438, 246
    n = 48
    F = gpu_pkg.FORMATS[fmt]
    of = O.OracleFilter(O.FORMATS[fmt], sw, sh, tw, th)
    spitch, dpitch = 320, 448
    per_frame = spitch * sh + dpitch * th
    pool = fresh_mapping(n * per_frame + 8192)            # pageable, a mapping of its own: the registrar pins it
    off = 52
    srcs, dsts, frames = [], [], []
    for k in range(n):
        s = pool[off:off + spitch * sh].reshape(sh, spitch); off += spitch * sh
        d = pool[off:off + dpitch * th].reshape(th, dpitch); off += dpitch * th
        fr = O.lcg_frame(O.FORMATS[fmt], sw, sh, seed=277 + k)
        s[:, :sw] = fr[0][:, :sw]
        srcs.append([s]); dsts.append([d]); frames.append(fr)
    b = gpu_pkg.Batch(F, sw, sh, tw, th, ndevices=1, streams=32, register_host_buffers=True)
    gpu_pkg.transport_counts(reset=True)
    b.process(srcs, dsts)
    by_shader, by_dma, _ = gpu_pkg.transport_counts()
    assert (by_shader, by_dma) == (n, 0)
    for k in range(n):
        assert_planes_equal(dsts[k], of.get_frame(frames[k], threads=4), F.plane_dims(tw, th), what=f"frame {k}")
    b.close()


def test_waits_on_tickets_that_were_never_issued_are_errors(gpu_pkg, O):
    """a ticket the instance never handed out is not 'complete'."""
    f = gpu_pkg.Filter(gpu_pkg.FORMATS["Y8"], 64, 48, 128, 96, device=0)
    f.set_pipeline(4, False)
    src = O.lcg_frame(O.FORMATS["Y8"], 64, 48)
    dst = [gpu_pkg.alloc_plane(128, 96, np.uint8)]
    t = f.submit(src, dst)
    f.wait(t)
    f.wait(t)            # a completed ticket may be waited on again
    for bad in (t + 1, t + 100, -1):
        with pytest.raises(gpu_pkg.JincError):
            f.wait(bad)
    f.close()
