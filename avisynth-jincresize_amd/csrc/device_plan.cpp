// device_plan.cpp -- device-resident plans of a filter instance: upload of the compact plan, and the launch planning
// for every kernel family (which kernel computes which part of the output plane).  See DESIGN.md section 4.
#include <algorithm>
#include <cstring>
#include <mutex>

#include "filter_internal.h"
#include "knobs.h"

namespace jinc {
namespace host {

namespace {

// Tables leave the host through a pinned bounce buffer of the library's own (hipHostMalloc), never straight from the vectors
// they were built in: handed a pageable pointer, the runtime maps that memory for the device on the fly, and in a process whose
// allocator recycles pages (glibc trims its heap) such on-the-fly mappings of recycled heap pages have ended in GPU memory
// access faults INSIDE these uploads (round 6: jinc_filter_create of test_framelane_pair, the library holding no registration
// of its own at the time -- profiles/round6/README.md).  Create-time traffic: a CPU copy of a few hundred KB does not matter.
struct Bounce {
    std::mutex mutex;
    void* host = nullptr;
    static constexpr size_t kBytes = size_t(4) << 20;
};
Bounce& bounce() {
    static Bounce& b = *new Bounce;  // (outlives static destruction, like the pin registry)
    return b;
}

// `stream`: the instance's own stream where the caller has one (the copy is queued there and waited for: the device memory it
// fills is then only ever touched from that stream -- with the copies on the null stream instead, create / frame / free cycles
// without a device-wide synchronisation in between left freed plan memory unreturned: 18 MiB per cycle of a tap-12 plan,
// profiles/round6/leak_ab.log); nullptr: a synchronous copy.
void upload(void* dev, const void* host, size_t bytes, const char* what, hipStream_t stream = nullptr) {
    if (!bytes) return;
    auto copy = [&](void* d, const void* h, size_t n) {
        if (stream) {
            hip_check(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, stream), what);
            hip_check(hipStreamSynchronize(stream), what);
        } else {
            hip_check(hipMemcpy(d, h, n, hipMemcpyHostToDevice), what);
        }
    };
    if (!knobs::flag(JINC_KNOB_UPLOAD_BOUNCE, true)) {  // A/B: straight from the caller's (pageable) memory
        copy(dev, host, bytes);
        return;
    }
    Bounce& b = bounce();
    std::lock_guard<std::mutex> lock(b.mutex);
    if (!b.host) hip_check(hipHostMalloc(&b.host, Bounce::kBytes, hipHostMallocPortable), "hipHostMalloc(upload buffer)");
    for (size_t at = 0; at < bytes; at += Bounce::kBytes) {
        const size_t n = std::min(Bounce::kBytes, bytes - at);
        std::memcpy(b.host, static_cast<const char*>(host) + at, n);
        copy(static_cast<char*>(dev) + at, b.host, n);  // waited for: the buffer is free again
    }
}

void download(void* host, const void* dev, size_t bytes, const char* what) {
    if (!bytes) return;
    Bounce& b = bounce();
    std::lock_guard<std::mutex> lock(b.mutex);
    if (!b.host) hip_check(hipHostMalloc(&b.host, Bounce::kBytes, hipHostMallocPortable), "hipHostMalloc(upload buffer)");
    for (size_t at = 0; at < bytes; at += Bounce::kBytes) {
        const size_t n = std::min(Bounce::kBytes, bytes - at);
        hip_check(hipMemcpy(b.host, static_cast<const char*>(dev) + at, n, hipMemcpyDeviceToHost), what);
        std::memcpy(static_cast<char*>(host) + at, b.host, n);
    }
}

}  // namespace

// (for the test hooks of filter.cpp: no copy of this library hands caller memory to the runtime, the debug entries included)
void bounce_upload(void* dev, const void* host, size_t bytes, const char* what) { upload(dev, host, bytes, what); }
void bounce_download(void* host, const void* dev, size_t bytes, const char* what) { download(host, dev, bytes, what); }

namespace {


// Smallest stride P <= 8 such that at least 90 % of the interior coordinates keep their class when
// stepping by P (1 if there is none).  Exact for periodic plans; for drifting ratios (1.5x, 3x) it is
// the nominal period, and the gather kernel's waterfall absorbs the deviations.
int dominant_period(const std::vector<int32_t>& cls) {
    const int n = static_cast<int>(cls.size());
    for (int P = 1; P <= 8; ++P) {
        long long same = 0, total = 0;
        for (int i = 0; i + P < n; ++i) {
            if (cls[i] < 0 || cls[i + P] < 0) continue;
            ++total;
            same += cls[i] == cls[i + P];
        }
        if (total > 0 && same * 10 >= total * 9) return P;
    }
    return 1;
}

void upload_table(const jinc::PlanePlan& p, DeviceTable& t, hipStream_t stream) {
    struct Piece {
        const void* host;
        size_t bytes;
        size_t offset;
    };
    std::vector<Piece> pieces;
    size_t off = 0;
    auto add = [&](const void* host, size_t bytes) {
        off = align_up(off, 256);
        pieces.push_back({host, bytes, off});
        off += bytes;
        return pieces.size() - 1;
    };
    const size_t i_cs = add(p.col_start.data(), p.col_start.size() * 4);
    const size_t i_rs = add(p.row_start.data(), p.row_start.size() * 4);
    const size_t i_cc = add(p.col_class.data(), p.col_class.size() * 4);
    const size_t i_rc = add(p.row_class.data(), p.row_class.size() * 4);
    const size_t i_is = add(p.interior_set.data(), p.interior_set.size() * 4);
    const size_t i_bc = add(p.bcol_set.data(), p.bcol_set.size() * 4);
    const size_t i_br = add(p.brow_set.data(), p.brow_set.size() * 4);
    // device layout of a set: fs rows of padded_fs floats (row stride a multiple of 16 bytes, zero padded)
    const int fsp = (p.fs + 3) & ~3;
    std::vector<float> padded(static_cast<size_t>(p.num_sets) * p.fs * fsp, 0.f);
    for (int s = 0; s < p.num_sets; ++s)
        for (int ly = 0; ly < p.fs; ++ly)
            std::memcpy(&padded[(static_cast<size_t>(s) * p.fs + ly) * fsp], p.set_ptr(s) + static_cast<size_t>(ly) * p.fs,
                        sizeof(float) * p.fs);
    // slack around the coefficient array: the direct kernel fetches whole coefficient blocks (<= 16 floats), its row walk
    // coefficient rows of the neighbouring sets (kDirectCoeffSlackRows, never used)
    const size_t slack = static_cast<size_t>(jinc::kDirectCoeffSlackRows) * fsp * sizeof(float) + 256;
    off = align_up(off, 256) + slack;
    const size_t i_co = add(padded.data(), padded.size() * 4);
    t.bytes = align_up(off, 256) + slack;
    hip_check(hipMalloc(&t.blob, t.bytes), "hipMalloc(plan)");
    char* base = static_cast<char*>(t.blob);
    for (const Piece& pc : pieces)
        upload(base + pc.offset, pc.host, pc.bytes, "plan upload", stream);

    auto ptr_i = [&](size_t i) { return reinterpret_cast<const int32_t*>(base + pieces[i].offset); };
    t.plan.col_start = ptr_i(i_cs);
    t.plan.row_start = ptr_i(i_rs);
    t.plan.col_class = ptr_i(i_cc);
    t.plan.row_class = ptr_i(i_rc);
    t.plan.interior_set = ptr_i(i_is);
    t.plan.bcol_set = ptr_i(i_bc);
    t.plan.brow_set = ptr_i(i_br);
    t.plan.coeffs = reinterpret_cast<const float*>(base + pieces[i_co].offset);
    t.plan.src_w = p.g.src_w;
    t.plan.src_h = p.g.src_h;
    t.plan.dst_w = p.g.dst_w;
    t.plan.dst_h = p.g.dst_h;
    t.plan.fs = p.fs;
    t.plan.n_col_classes = p.n_col_classes;
    t.plan.gather_period_x = dominant_period(p.col_class);
    t.plan.gather_period_y = dominant_period(p.row_class);
}

jinc::RectList border_frame(const jinc::PlanePlan& p, int x_end, int y_end);

// ewa_periodic_quad_kernel's coefficient pairs for 2x up-scales whose two phases per axis share their window origin
// (filter sizes 7 and 9): quad[ly][q][8 or 10 pairs][p] = (set(p = 0, q), set(p = 1, q))[ly][lx]; a kernel row of one q is 16
// dwords (fs 7: one s_load_dwordx16, the eighth pair is padding) or 20 (fs 9), the two q of a kernel row are adjacent.
// `sets`: the four phase sets, FS x FS floats each, phase q * 2 + p (the plan's, or their trimmed copies).
// (FS rows of NX taps each; NX = 0: FS)
void attach_quad(DeviceTable& t, jinc::PeriodicArgs& pa, int FS, const std::vector<const float*>& sets, int NX = 0) {
    if (NX == 0) NX = FS;
    if (!t.use_periodic || (FS != 6 && FS != 7 && FS != 8 && FS != 9) || pa.px != 2 || pa.py != 2 || pa.start_x[0] != pa.start_x[1] ||
        pa.start_y[0] != pa.start_y[1])
        return;
    const int PR = NX == 9 ? 10 : 8;  // pairs per (kernel row, q), padded: 16 / 20 dwords
    std::vector<float> q(static_cast<size_t>(2) * FS * PR * 2, 0.f);
    for (int qy = 0; qy < 2; ++qy)
        for (int ly = 0; ly < FS; ++ly)
            for (int lx = 0; lx < NX; ++lx)
                for (int px = 0; px < 2; ++px)
                    q[((static_cast<size_t>(ly) * 2 + qy) * PR + lx) * 2 + px] = sets[static_cast<size_t>(qy * 2 + px)][ly * NX + lx];
    void* dev = nullptr;
    hip_check(hipMalloc(&dev, q.size() * sizeof(float)), "hipMalloc(quad coefficients)");
    t.lane_blobs.push_back(dev);  // freed with the table
    upload(dev, q.data(), q.size() * sizeof(float), "quad coefficient upload");
    pa.quad = static_cast<const float*>(dev);
}

// ewa_periodic_rowpair_kernel's coefficient pairs (kernels.h PeriodicArgs::rowpair): `sets` = the plan's phase sets (phase q * px +
// p), ny rows of n floats each.  `trimmed`: per (q, kernel row) the taps BOTH phases p leave out on either side -- zero coefficients
// in front of and behind the row's span, exact to skip for finite samples -- are recorded for the kernel; otherwise none (the
// reference's full window, every tap executed).  Only for two phases per period in x that share their window origin.
void attach_rowpair(DeviceTable& t, jinc::PeriodicArgs& pa, int n, int ny, const std::vector<const float*>& sets, bool trimmed) {
    pa.rowpair = nullptr;
    pa.rowpair_n = pa.rowpair_ny = pa.rowpair_stride = 0;
    if (!jinc::rowpair_supported(n) || ny > n || ny < 1 || pa.px != 2 || pa.py > jinc::PeriodicArgs::kRowPairMaxPhases || pa.start_x[0] != pa.start_x[1] ||
        pa.start_x[0] != pa.min_sx)
        return;
    const int stride = (2 * n + 3) & ~3;  // floats per kernel row: n pairs, padded to 16 bytes
    const int widest = std::min(5, (n - 2) / 2);
    std::vector<float> c(static_cast<size_t>(pa.py) * ny * stride, 0.f);
    for (int q = 0; q < pa.py; ++q) {
        uint64_t bits = 0;
        for (int ly = 0; ly < ny; ++ly) {
            int tr = widest;
            for (int px = 0; px < 2; ++px) {
                const float* r = sets[static_cast<size_t>(q * 2 + px)] + ly * n;
                for (int lx = 0; lx < n; ++lx) c[(static_cast<size_t>(q) * ny + ly) * stride + 2 * lx + px] = r[lx];
                int lead = 0, trail = 0;
                while (lead < n && r[lead] == 0.f) ++lead;
                while (trail < n - lead && r[n - 1 - trail] == 0.f) ++trail;
                tr = std::min(tr, std::min(lead, trail));
            }
            bits |= static_cast<uint64_t>(trimmed ? tr : 0) << (3 * ly);
        }
        if (trimmed) {  // kernel rows that are zero throughout for both p, in front of and behind the rest (border rows: shifted windows)
            auto zero_row = [&](int ly) {
                for (int px = 0; px < 2; ++px)
                    for (int lx = 0; lx < n; ++lx)
                        if (sets[static_cast<size_t>(q * 2 + px)][ly * n + lx] != 0.f) return false;
                return true;
            };
            int first = 0, last = ny;  // [first, last)
            while (first < ny - 1 && zero_row(first)) ++first;
            while (last > first + 1 && zero_row(last - 1)) --last;
            bits |= static_cast<uint64_t>(first) << 54 | static_cast<uint64_t>(last) << 59;
        }
        pa.rowpair_trim[q] = bits;
    }
    void* dev = nullptr;
    hip_check(hipMalloc(&dev, c.size() * sizeof(float)), "hipMalloc(row-pair coefficients)");
    t.lane_blobs.push_back(dev);  // freed with the table
    upload(dev, c.data(), c.size() * sizeof(float), "row-pair coefficient upload");
    pa.rowpair = static_cast<const float*>(dev);
    pa.rowpair_n = n;
    pa.rowpair_ny = ny;
    pa.rowpair_stride = stride;
}

// Trimmed support of the periodic interior (integer planes).  The reference's window is filter_size x filter_size taps, but
// the EWA disc does not fill it: taps beyond the radius carry the coefficient 0.0f (LUT index >= samples, ref :277-281), for
// the 2x up-scale with tap 3 the whole first kernel row and column of all four phase sets.  A tap whose coefficient is zero
// contributes float(sample) * 0 = +-0 to a chain that starts at +0 and therefore is never -0: r + (+-0) == r bit for bit, so
// leaving the tap out is exact -- for FINITE samples.  8 ... 16-bit samples always are; a float sample may be an infinity
// or a NaN, whose product with 0 is a NaN the reference propagates: float planes take the trimmed support frame by frame,
// where a scan of the call's source planes found nothing but finite samples (dispatch.cpp, kernel_scan.hip).  Here: the bounding box of
// the non-zero coefficients over the interior's phase sets, squared up; the kernels of the periodic family then run with
// filter size trim_fs on copies of the sets cut to the box, window origins moved by the box's corner.
void trim_periodic(const jinc::PlanePlan& p, DeviceTable& t, bool integer_samples) {
    t.trim_fs = 0;
    t.trim_nx = 0;
    t.trim_needs_finite = !integer_samples;  // float planes: only frames without infinities / NaNs (dispatch.cpp, kernel_scan.hip)
    if (!t.use_periodic) return;
    if (!knobs::flag(JINC_KNOB_TRIM, true)) return;  // A/B knob: TRIM = 0 keeps the full window
    const jinc::PeriodicArgs& pa = t.periodic;
    const int fs = p.fs, nphase = pa.px * pa.py;
    int r0 = fs, r1 = -1, c0 = fs, c1 = -1;
    for (int ph = 0; ph < nphase; ++ph) {
        const float* s = p.set_ptr(pa.set[ph]);
        for (int ly = 0; ly < fs; ++ly)
            for (int lx = 0; lx < fs; ++lx)
                if (s[ly * fs + lx] != 0.f) r0 = std::min(r0, ly), r1 = std::max(r1, ly), c0 = std::min(c0, lx), c1 = std::max(c1, lx);
    }
    if (r1 < 0) return;  // nothing but zeros: leave it to the full window
    // (fs - 1) rows x fs columns (fs 7 / 9, four phases with one window origin): chroma planes sited as MPEG-2 at 2x with tap 3 / 4 -- the
    // disc spans fs - 1 source rows, and fs columns because the siting shifts it by an eighth of a sample.  Only the two-periods-per-lane
    // quad forms take it (fs taps per kernel row: PeriodicArgs::quad_taps; ewa_periodic_quad2_kernel / ewa_periodic_quad2x8_kernel);
    // every other kernel of the family keeps the full window.
    if ((fs == 7 || fs == 9) && nphase == 4 && r1 - r0 + 1 == fs - 1 && c1 - c0 + 1 == fs && pa.px == 2 && pa.py == 2 && pa.start_x[0] == pa.start_x[1] &&
        pa.start_y[0] == pa.start_y[1]) {
        const int nr = fs - 1;
        std::vector<float> dense(static_cast<size_t>(4) * nr * fs, 0.f);
        for (int ph = 0; ph < 4; ++ph)
            for (int ly = 0; ly < nr; ++ly)
                for (int lx = 0; lx < fs; ++lx) dense[(static_cast<size_t>(ph) * nr + ly) * fs + lx] = p.set_ptr(pa.set[ph])[(r0 + ly) * fs + lx];
        jinc::PeriodicArgs tr = pa;
        tr.coeffs = nullptr;  // (no kernel but the quad forms reads this variant)
        tr.quad = nullptr;
        for (int q = 0; q < pa.py; ++q) tr.start_y[q] = pa.start_y[q] + r0;
        tr.min_sy = pa.min_sy + r0;
        tr.quad_taps = fs;
        {   // per (kernel row, q): the zero coefficients in front of / behind the row's span for BOTH phases p (kernels.h quad_span7)
            const bool off = !knobs::flag(JINC_KNOB_QUAD_INNER, true);  // A/B knob
            uint64_t spans = 0;
            for (int q = 0; q < 2 && !off; ++q)
                for (int ly = 0; ly < nr; ++ly) {
                    int lead = 3, trail = 3;
                    for (int px = 0; px < 2; ++px) {
                        const float* r = &dense[(static_cast<size_t>(q * 2 + px) * nr + ly) * fs];
                        int a = 0, b = 0;
                        while (a < fs && r[a] == 0.f) ++a;
                        while (b < fs - a && r[fs - 1 - b] == 0.f) ++b;
                        lead = std::min(lead, a), trail = std::min(trail, b);
                    }
                    spans |= static_cast<uint64_t>(lead) << (4 * (2 * ly + q)) | static_cast<uint64_t>(trail) << (4 * (2 * ly + q) + 2);
                }
            tr.quad_span7 = spans;
        }
        t.periodic_trim = tr;
        std::vector<const float*> sets;
        for (int ph = 0; ph < 4; ++ph) sets.push_back(dense.data() + static_cast<size_t>(ph) * nr * fs);
        attach_rowpair(t, t.periodic_trim, fs, nr, sets, true);
        attach_quad(t, t.periodic_trim, nr, sets, fs);
        if (t.periodic_trim.quad) {
            t.trim_fs = nr;
            t.trim_nx = fs;
        }
        return;
    }
    // n taps per kernel row x ny kernel rows.  The window and quad forms are square (n = ny = the larger side of the box); the
    // rows kernel (n >= 10 or n <= 5) walks its kernel rows in a rolled loop and takes the row count at run time
    // (PeriodicArgs::rows_ny), so there the box keeps its own height: chroma planes sited as MPEG-2 at 2x with tap 8 need all 17
    // columns but only 16 rows.
    int n = std::max(3, c1 - c0 + 1), ny = std::max(3, r1 - r0 + 1);
    const bool rows_family = n >= 10 || n <= 5;
    if (!(rows_family && ny < n)) n = ny = std::max(n, ny);
    if (n >= fs && ny >= fs) return;
    n = std::min(n, fs), ny = std::min(ny, fs);
    r0 = std::min(r0, fs - ny);
    c0 = std::min(c0, fs - n);
    const int row = (n + 3) & ~3;  // floats per coefficient row on the device, as upload_table
    std::vector<float> cut(static_cast<size_t>(nphase) * ny * row, 0.f), dense(static_cast<size_t>(nphase) * ny * n, 0.f);
    for (int ph = 0; ph < nphase; ++ph) {
        const float* s = p.set_ptr(pa.set[ph]);
        for (int ly = 0; ly < ny; ++ly)
            for (int lx = 0; lx < n; ++lx) {
                const float c = s[(r0 + ly) * fs + (c0 + lx)];
                cut[(static_cast<size_t>(ph) * ny + ly) * row + lx] = c;
                dense[(static_cast<size_t>(ph) * ny + ly) * n + lx] = c;
            }
    }
    // per phase and kernel row: the zero coefficients in front of and behind the row's span (the disc's chord), the smaller of
    // the two counts -- the rows kernel leaves that many taps out on either side (kernel_periodic.hip rows_kernel_row)
    std::vector<int32_t> row_trim(static_cast<size_t>(nphase) * 32, 0);
    if (n <= 32)
        for (int ph = 0; ph < nphase; ++ph)
            for (int ly = 0; ly < ny; ++ly) {
                const float* r = &dense[(static_cast<size_t>(ph) * ny + ly) * n];
                int lead = 0, trail = 0;
                while (lead < n && r[lead] == 0.f) ++lead;
                while (trail < n - lead && r[n - 1 - trail] == 0.f) ++trail;
                row_trim[static_cast<size_t>(ph) * 32 + ly] = std::min(5, std::min(std::min(lead, trail), (n - 1) / 2));
            }
    {   // taps per sample the rows kernel executes with these spans (reports): it offers 0 .. 5 (n >= 12) or 0 .. 2 (n >= 6) taps
        // per side, none below
        const int widest = n >= 12 ? 5 : n >= 6 ? 2 : 0;
        double taps = 0;
        for (int ph = 0; ph < nphase; ++ph)
            for (int ly = 0; ly < ny; ++ly) taps += n - 2 * std::min(widest, n <= 32 ? row_trim[static_cast<size_t>(ph) * 32 + ly] : 0);
        t.trim_rows_taps = taps / nphase;
    }
    const size_t cut_bytes = align_up(cut.size() * sizeof(float), 256);
    void* dev = nullptr;
    hip_check(hipMalloc(&dev, cut_bytes + row_trim.size() * sizeof(int32_t)), "hipMalloc(trimmed coefficient sets)");
    t.lane_blobs.push_back(dev);  // freed with the table
    upload(dev, cut.data(), cut.size() * sizeof(float), "trimmed coefficient upload");
    upload(static_cast<char*>(dev) + cut_bytes, row_trim.data(), row_trim.size() * sizeof(int32_t), "row trim upload");
    jinc::PeriodicArgs tr = pa;
    tr.coeffs = static_cast<const float*>(dev);
    tr.row_trim = reinterpret_cast<const int32_t*>(static_cast<char*>(dev) + cut_bytes);
    tr.quad = nullptr;
    for (int ph = 0; ph < nphase; ++ph) tr.set[ph] = ph;
    for (int q = 0; q < pa.px; ++q) tr.start_x[q] = pa.start_x[q] + c0;
    for (int q = 0; q < pa.py; ++q) tr.start_y[q] = pa.start_y[q] + r0;
    tr.min_sx = pa.min_sx + c0;
    tr.min_sy = pa.min_sy + r0;
    tr.rows_ny = ny < n ? ny : 0;
    t.periodic_trim = tr;
    t.trim_fs = n;
    t.trim_nx = n;
    {   // the packed phase-pair form of the rows kernel on this support (12 .. 17 taps per kernel row)
        std::vector<const float*> sets;
        for (int ph = 0; ph < nphase; ++ph) sets.push_back(dense.data() + static_cast<size_t>(ph) * ny * n);
        if (pa.px == 2) attach_rowpair(t, t.periodic_trim, n, ny, sets, true);
    }
    if (nphase == 4 && ny == n) {
        std::vector<const float*> sets;
        for (int ph = 0; ph < 4; ++ph) sets.push_back(dense.data() + static_cast<size_t>(ph) * n * n);
        attach_quad(t, t.periodic_trim, n, sets);
        if (n == 6 && t.periodic_trim.quad) {  // kernel rows whose first and last tap are zero for both p of a q: the chord in the box's edge rows
            const bool off = !knobs::flag(JINC_KNOB_QUAD_INNER, true);  // A/B knob
            uint32_t inner = 0;
            for (int q = 0; q < 2 && !off; ++q)
                for (int ly = 0; ly < n; ++ly) {
                    bool zero = true;
                    for (int px = 0; px < 2; ++px) {
                        const float* r = sets[static_cast<size_t>(q * 2 + px)] + ly * n;
                        zero = zero && r[0] == 0.f && r[n - 1] == 0.f;
                    }
                    if (zero) inner |= 1u << (2 * ly + q);
                }
            t.periodic_trim.quad_inner = inner;
        }
        if (n == 8 && t.periodic_trim.quad) {  // 8 x 8 support: taps every (kernel row, q) leaves out per side for both p
            const bool off = !knobs::flag(JINC_KNOB_QUAD_INNER, true);
            uint32_t tr8 = 0;
            for (int q = 0; q < 2 && !off; ++q)
                for (int ly = 0; ly < n; ++ly) {
                    int trim = 3;
                    for (int px = 0; px < 2; ++px) {
                        const float* r = sets[static_cast<size_t>(q * 2 + px)] + ly * n;
                        int lead = 0, trail = 0;
                        while (lead < n && r[lead] == 0.f) ++lead;
                        while (trail < n && r[n - 1 - trail] == 0.f) ++trail;
                        trim = std::min(trim, std::min(lead, trail));
                    }
                    tr8 |= static_cast<uint32_t>(trim) << (2 * (2 * ly + q));
                }
            t.periodic_trim.quad_trim8 = tr8;
        }
    }
}

// Decides how the output plane is split between the periodic kernel and the gather kernel.
void plan_launches(const jinc::PlanePlan& p, DeviceTable& t) {
    const int W = p.g.dst_w, H = p.g.dst_h;
    t.whole = jinc::RectList{};
    t.whole.n = 1;
    t.whole.w[0] = W;
    t.whole.h[0] = H;
    // Without an exactly periodic interior the lanes of an item rarely share a coefficient set (drifting classes,
    // or no structure at all): the gather kernel then skips its uniform passes and fetches coefficients per lane
    // (1.5x tap 8: 16 -> 26 Gpix/s, 1.37x: 41 -> 65, 5/6: 47 -> 57).
    t.whole.private_sets = !p.periodic;
    t.use_periodic = false;
    if (!p.periodic || !jinc::periodic_supported(p.fs, p.px, p.py, p.sx, p.sy)) return;

    jinc::PeriodicArgs pa;
    pa.coeffs = t.plan.coeffs;
    pa.px = p.px;
    pa.py = p.py;
    pa.ix0 = p.ix0;
    pa.iy0 = p.iy0;
    pa.ni = (p.ix1 - p.ix0) / p.px;
    pa.nj = (p.iy1 - p.iy0) / p.py;
    if (pa.ni < 1 || pa.nj < 1) return;
    int min_sx = INT32_MAX, max_sx = INT32_MIN, min_sy = INT32_MAX, max_sy = INT32_MIN;
    for (int q = 0; q < p.px; ++q) {
        pa.start_x[q] = p.col_start[p.ix0 + q];
        min_sx = std::min(min_sx, pa.start_x[q]);
        max_sx = std::max(max_sx, pa.start_x[q]);
    }
    for (int q = 0; q < p.py; ++q) {
        pa.start_y[q] = p.row_start[p.iy0 + q];
        min_sy = std::min(min_sy, pa.start_y[q]);
        max_sy = std::max(max_sy, pa.start_y[q]);
    }
    // The kernel's LDS tile has room for a phase spread of one source sample per axis.
    if (max_sx - min_sx > 1 || max_sy - min_sy > 1) return;
    pa.min_sx = min_sx;
    pa.min_sy = min_sy;
    for (int q = 0; q < p.py; ++q)
        for (int r = 0; r < p.px; ++r)
            pa.set[q * p.px + r] = p.interior_set[static_cast<size_t>(p.row_class[p.iy0 + q]) * p.n_col_classes +
                                                  p.col_class[p.ix0 + r]];
    pa.src_w = p.g.src_w;
    pa.src_h = p.g.src_h;
    pa.dst_h = p.g.dst_h;
    t.periodic = pa;
    t.use_periodic = true;
    if (pa.px == 2) {  // the rows kernel's packed phase-pair form on the reference's full window (every tap executed: no trims)
        std::vector<const float*> sets;
        for (int ph = 0; ph < pa.px * pa.py; ++ph) sets.push_back(p.set_ptr(pa.set[ph]));
        attach_rowpair(t, t.periodic, p.fs, p.fs, sets, false);
    }

    t.border_rects = border_frame(p, p.ix0 + p.px * pa.ni, p.iy0 + p.py * pa.nj);
}

// The up-to-four rectangles around the interior block [ix0, x_end) x [iy0, y_end) of the output plane.
jinc::RectList border_frame(const jinc::PlanePlan& p, int x_end, int y_end) {
    const int W = p.g.dst_w, H = p.g.dst_h;
    jinc::RectList r;
    auto add = [&](int x0, int y0, int w, int h) {
        if (w <= 0 || h <= 0) return;
        r.x0[r.n] = x0;
        r.y0[r.n] = y0;
        r.w[r.n] = w;
        r.h[r.n] = h;
        ++r.n;
    };
    add(0, 0, W, p.iy0);                     // top rows
    add(0, y_end, W, H - y_end);             // bottom rows
    add(0, p.iy0, p.ix0, y_end - p.iy0);     // left columns
    add(x_end, p.iy0, W - x_end, y_end - p.iy0);  // right columns
    return r;
}

// Quasi-periodic interior (see kernels.h): used when the plan is not exactly periodic but its window
// origins are affine per residue, or when forced for A/B runs.
void plan_quasi(const jinc::PlanePlan& p, DeviceTable& t) {
    t.use_quasi = false;
    int px, py, sx, sy;
    // An exactly periodic plan keeps ITS period here too (the affine-origin period may be shorter): the border frame is
    // laid out once per table for the extent ix0 + px * ni, whichever interior kernel then runs.
    if (p.periodic) {
        px = p.px, py = p.py, sx = p.sx, sy = p.sy;
    } else if (p.quasi) {
        px = p.qpx, py = p.qpy, sx = p.qsx, sy = p.qsy;
    } else {
        return;
    }
    if (!jinc::quasi_supported(p.fs, px, py, sx, sy, p.n_col_classes, p.n_row_classes)) return;
    jinc::QuasiArgs qa;
    qa.coeffs = t.plan.coeffs;
    qa.col_class = t.plan.col_class;
    qa.row_class = t.plan.row_class;
    qa.interior_set = t.plan.interior_set;
    qa.n_col_classes = p.n_col_classes;
    qa.n_row_classes = p.n_row_classes;
    qa.px = px, qa.py = py, qa.sx = sx, qa.sy = sy;
    qa.exact = p.periodic ? 1 : 2;  // 1: one set per phase; 2: drifting classes, per-lane coefficient registers
    qa.ix0 = p.ix0, qa.iy0 = p.iy0;
    qa.ni = (p.ix1 - p.ix0) / px;
    qa.nj = (p.iy1 - p.iy0) / py;
    if (qa.ni < 1 || qa.nj < 1) return;
    int min_sx = INT32_MAX, max_sx = INT32_MIN, min_sy = INT32_MAX, max_sy = INT32_MIN;
    for (int k = 0; k < px; ++k) {
        qa.start_x[k] = p.col_start[p.ix0 + k];
        min_sx = std::min(min_sx, qa.start_x[k]);
        max_sx = std::max(max_sx, qa.start_x[k]);
    }
    for (int k = 0; k < py; ++k) {
        qa.start_y[k] = p.row_start[p.iy0 + k];
        min_sy = std::min(min_sy, qa.start_y[k]);
        max_sy = std::max(max_sy, qa.start_y[k]);
    }
    qa.min_sx = min_sx, qa.min_sy = min_sy;
    qa.src_w = p.g.src_w, qa.src_h = p.g.src_h, qa.dst_h = p.g.dst_h;
    if (p.periodic)
        for (int q = 0; q < py; ++q)
            for (int r = 0; r < px; ++r)
                qa.phase_set[q * px + r] = p.interior_set[static_cast<size_t>(p.row_class[p.iy0 + q]) * p.n_col_classes +
                                                          p.col_class[p.ix0 + r]];
    if (!jinc::quasi_configure(qa, p.fs, max_sx - min_sx, max_sy - min_sy)) return;
    t.quasi = qa;
    t.use_quasi = true;
    if (!t.use_periodic) {
        t.border_rects = border_frame(p, p.ix0 + px * qa.ni, p.iy0 + py * qa.nj);
        t.border_rects.private_sets = !p.periodic;
        t.border_rects.unit_stride = !p.periodic;
    }
}

// Drifting plans (window origins affine per residue, phase classes changing now and then along an axis: 1.5x, 3x, 8/3 x 9/4 ...)
// with filter sizes the direct kernel's row walk covers: along each axis the periods of a phase fall into a few runs of constant
// class (1280 x 720 -> 1920 x 1080: 9..14 runs per column phase, 4..7 per row phase; 120 distinct interior sets), and a
// (column run) x (row run) rectangle of one phase pair has ONE coefficient set and exactly periodic windows -- the direct
// kernel's premise, per rectangle.  kernels.h DirectRun; the rectangles of a plane are one launch.
void plan_runs(const jinc::PlanePlan& p, DeviceTable& t) {
    t.use_runs = false;
    if (p.periodic || !p.quasi) return;
    const int px = p.qpx, py = p.qpy, sx = p.qsx, sy = p.qsy;
    if (!jinc::direct_runs_supported(p.fs, px, py, sx, sy)) return;
    const int ni = (p.ix1 - p.ix0) / px, nj = (p.iy1 - p.iy0) / py;
    if (ni < 1 || nj < 1) return;
    static_assert(sizeof(jinc::PlanRun) == sizeof(jinc::DirectRun) && offsetof(jinc::PlanRun, first_item) == offsetof(jinc::DirectRun, first_wave),
                  "the host's rectangle is what the kernel reads");
    std::vector<jinc::PlanRun> runs;
    std::vector<int32_t> item_run;
    if (!jinc::build_plan_runs(p, runs, item_run)) return;  // (plan.cpp: the list itself is host code, tested without a device)
    const size_t run_bytes = runs.size() * sizeof(jinc::DirectRun), item_bytes = item_run.size() * sizeof(int32_t);
    char* dev = nullptr;
    hip_check(hipMalloc(reinterpret_cast<void**>(&dev), run_bytes + item_bytes), "hipMalloc(direct runs)");
    t.lane_blobs.push_back(dev);  // freed with the table
    upload(dev, runs.data(), run_bytes, "direct runs upload");
    upload(dev + run_bytes, item_run.data(), item_bytes, "direct runs upload");
    jinc::DirectArgs da;
    da.coeffs = t.plan.coeffs;
    da.fs = p.fs;
    da.coeff_row = (p.fs + 3) & ~3;
    da.px = px, da.py = py, da.sx = sx, da.sy = sy;
    da.ix0 = p.ix0, da.iy0 = p.iy0, da.ni = ni, da.nj = nj;
    da.dst_h = p.g.dst_h;
    da.plan = t.plan;
    da.runs = reinterpret_cast<const jinc::DirectRun*>(dev);
    da.item_run = reinterpret_cast<const int32_t*>(dev + run_bytes);
    da.n_items = static_cast<int>(item_run.size());
    t.runs = da;
    t.use_runs = true;
    if (!t.use_quasi) {  // (plan_quasi lays out the same border frame for the filter sizes it covers)
        t.border_rects = border_frame(p, p.ix0 + px * ni, p.iy0 + py * nj);
        t.border_rects.private_sets = true;
        t.border_rects.unit_stride = true;
    }
}

// Exactly periodic plans: kernel_direct.hip can take the interior (it is the choice for down-scales and taps > 8,
// which the register/LDS kernels do not cover) and, for every interior kernel, the border rows and columns.
void plan_direct(const jinc::PlanePlan& p, DeviceTable& t) {
    t.use_direct = false;
    if (!p.periodic || !jinc::direct_supported(p.fs, p.px, p.py, p.sx, p.sy)) return;
    jinc::DirectArgs da;
    da.coeffs = t.plan.coeffs;
    da.fs = p.fs;
    da.coeff_row = (p.fs + 3) & ~3;
    da.px = p.px, da.py = p.py, da.sx = p.sx, da.sy = p.sy;
    da.ix0 = p.ix0, da.iy0 = p.iy0;
    da.ni = (p.ix1 - p.ix0) / p.px;
    da.nj = (p.iy1 - p.iy0) / p.py;
    if (da.ni < 1 || da.nj < 1) return;
    for (int k = 0; k < p.px; ++k) da.start_x[k] = p.col_start[p.ix0 + k];
    for (int k = 0; k < p.py; ++k) da.start_y[k] = p.row_start[p.iy0 + k];
    da.dst_h = p.g.dst_h;
    da.plan = t.plan;
    const int x_end = p.ix0 + p.px * da.ni, y_end = p.iy0 + p.py * da.nj;
    const int W = p.g.dst_w, H = p.g.dst_h;

    t.row_strips = da;
    t.row_strips.line0[0] = 0, t.row_strips.line_n[0] = p.iy0;
    t.row_strips.line0[1] = y_end, t.row_strips.line_n[1] = H - y_end;
    jinc::RectList c;
    auto add = [&](int x0, int y0, int w, int h) {
        if (w <= 0 || h <= 0) return;
        c.x0[c.n] = x0, c.y0[c.n] = y0, c.w[c.n] = w, c.h[c.n] = h;
        ++c.n;
    };
    add(0, 0, p.ix0, H);
    add(x_end, 0, W - x_end, H);
    t.column_rects = c;
    c = jinc::RectList{};
    add(0, 0, p.ix0, p.iy0);
    add(x_end, 0, W - x_end, p.iy0);
    add(0, y_end, p.ix0, H - y_end);
    add(x_end, y_end, W - x_end, H - y_end);
    c.private_sets = true;  // corner pixels own a coefficient set each
    c.unit_stride = true;
    t.corner_rects = c;

    jinc::ColStripArgs ca;
    ca.coeffs = t.plan.coeffs;
    ca.fs = p.fs, ca.coeff_row = da.coeff_row;
    ca.py = p.py, ca.sy = p.sy, ca.iy0 = p.iy0, ca.nj = da.nj;
    int min_sy = INT32_MAX, max_sy = INT32_MIN;
    for (int k = 0; k < p.py; ++k) {
        ca.start_y[k] = da.start_y[k];
        min_sy = std::min(min_sy, da.start_y[k]);
        max_sy = std::max(max_sy, da.start_y[k]);
    }
    ca.min_sy = min_sy, ca.spread_y = max_sy - min_sy;
    ca.x0[0] = 0, ca.nx[0] = p.ix0;
    ca.x0[1] = x_end, ca.nx[1] = W - x_end;
    for (int s = 0; s < 2; ++s) {
        if (ca.nx[s] <= 0) continue;
        int lo = INT32_MAX, hi = INT32_MIN;  // window origins are non-decreasing in x, but do not rely on it
        for (int x = ca.x0[s]; x < ca.x0[s] + ca.nx[s]; ++x) {
            lo = std::min(lo, p.col_start[x]);
            hi = std::max(hi, p.col_start[x] + p.fs);
        }
        ca.src_c0[s] = lo, ca.src_w[s] = hi - lo;
    }
    ca.plan = t.plan;
    t.use_colstrip = jinc::colstrip_configure(ca);
    t.col_strips = ca;

    // The strip kernels take ONE coefficient set per (border row, column phase) / (border column, row phase).  That
    // holds when the border pixels' coefficients repeat with the interior's period (integer ratios, exact down-scales)
    // -- but a plan can have a periodic interior and still private border sets: for 3/2 on a small frame the
    // interior classes have not drifted yet, while the reference computes every border pixel's coefficients from its
    // float-accumulated position, so no two are equal (found by the widened random sweep).  Check, do not assume.
    bool uniform = true;
    for (int y = 0; y < H && uniform; ++y) {
        if (y >= p.iy0 && y < y_end) continue;
        for (int r = 0; r < p.px && uniform; ++r) {
            const int s0 = p.set_of(p.ix0 + r, y);
            for (int i = 1; i < da.ni; ++i)
                if (p.set_of(p.ix0 + p.px * i + r, y) != s0) {
                    uniform = false;
                    break;
                }
        }
    }
    for (int x = 0; x < W && uniform; ++x) {
        if (x >= p.ix0 && x < x_end) continue;
        for (int q = 0; q < p.py && uniform; ++q) {
            const int s0 = p.set_of(x, p.iy0 + q);
            for (int j = 1; j < da.nj; ++j)
                if (p.set_of(x, p.iy0 + p.py * j + q) != s0) {
                    uniform = false;
                    break;
                }
        }
    }
    t.strips_ok = uniform;

    // kernel_strip.hip (round 5): the strips of filter sizes up to 9 at source step 1 with one register window per lane.  A group
    // = consecutive border lines that share their window origin across the strip (all of them, for the plans seen: the reference
    // shifts every border window back to the image's first / last fs lines, ref :395-418); at most four groups per orientation.
    t.use_strip_rows = t.use_strip_cols = false;
    if (uniform) {
        auto build = [&](int axis, jinc::StripArgs& sa) -> bool {
            const bool rows = axis == 0;
            const int P = rows ? p.px : p.py, S = rows ? p.sx : p.sy;
            sa = jinc::StripArgs{};
            sa.coeffs = t.plan.coeffs;
            sa.fs = p.fs, sa.axis = axis, sa.P = P, sa.S = S;
            sa.i0 = rows ? p.ix0 : p.iy0;
            sa.ni = rows ? da.ni : da.nj;
            int lo = INT32_MAX, hi = INT32_MIN;
            for (int k = 0; k < P; ++k) {
                sa.start[k] = rows ? da.start_x[k] : da.start_y[k];
                lo = std::min(lo, sa.start[k]), hi = std::max(hi, sa.start[k]);
            }
            sa.min_start = lo, sa.spread = hi - lo;
            if (P > 16 || !jinc::strip_supported(p.fs, P, S, sa.spread)) return false;
            sa.src_w = p.g.src_w, sa.src_h = p.g.src_h, sa.dst_h = p.g.dst_h;
            const std::vector<int32_t>& origin_of = rows ? p.row_start : p.col_start;
            const int ends[2][2] = {{0, rows ? p.iy0 : p.ix0}, {rows ? y_end : x_end, rows ? H : W}};
            std::vector<int32_t> sets;
            for (const auto& e : ends)
                for (int l = e[0]; l < e[1];) {
                    int m = l;
                    while (m < e[1] && origin_of[static_cast<size_t>(m)] == origin_of[static_cast<size_t>(l)]) ++m;
                    if (sa.ngroups == 4) return false;
                    const int g = sa.ngroups++;
                    sa.line0[g] = l, sa.nlines[g] = m - l, sa.origin[g] = origin_of[static_cast<size_t>(l)];
                    sa.set_base[g] = static_cast<int>(sets.size()) / P;
                    for (int line = l; line < m; ++line)
                        for (int k = 0; k < P; ++k) sets.push_back(rows ? p.set_of(p.ix0 + k, line) : p.set_of(line, p.iy0 + k));
                    l = m;
                }
            if (sa.ngroups == 0) return false;
            void* dev = nullptr;
            hip_check(hipMalloc(&dev, sets.size() * sizeof(int32_t)), "hipMalloc(strip sets)");
            t.lane_blobs.push_back(dev);  // freed with the table
            upload(dev, sets.data(), sets.size() * sizeof(int32_t), "strip set upload");
            sa.sets = static_cast<const int32_t*>(dev);
            return true;
        };
        t.use_strip_rows = build(0, t.strip_rows);
        t.use_strip_cols = build(1, t.strip_cols);
    }

    for (int q = 0; q < p.py; ++q)
        for (int r = 0; r < p.px; ++r)
            da.set[q * p.px + r] = p.interior_set[static_cast<size_t>(p.row_class[p.iy0 + q]) * p.n_col_classes +
                                                  p.col_class[p.ix0 + r]];
    t.direct = da;
    t.use_direct = true;
    if (!t.use_periodic && !t.use_quasi) t.border_rects = border_frame(p, x_end, y_end);  // fallback border (gather)
    if (!t.strips_ok) t.border_rects.private_sets = t.border_rects.unit_stride = true;  // coefficients per lane
}

// Border rows of 2x up-scales with 10 .. 17 taps per kernel row on ewa_periodic_rowpair_kernel (round 5): the rows of one end of the
// plane, interior columns, are one launch of the interior's packed kernel (its one-period-row form) with the rows as its "row phases" -- every
// border row has one coefficient set per column phase (strips_ok) and all rows of an end share their window origin (the reference
// shifts every border window back to the image's first / last fs lines, ref :395-418), which is all that kernel asks of a phase.
// ewa_direct_kernel's row strips, which these launches replace, ran at about a sixth of the interior's rate per tap (C3: 0.10 ms per
// plane and launch for 1.5 % of the samples).
// Integer planes leave out the zero taps of each (row, kernel row) chord as the interior does, and the kernel rows that are zero
// throughout (the windows are shifted against the disc); float planes execute every tap.
void plan_rowpair_rows(const jinc::PlanePlan& p, DeviceTable& t, bool integer_samples) {
    t.rowpair_rows.clear();
    if (!t.use_periodic || !t.use_direct || !t.strips_ok || !jinc::rowpair_strip_supported(p.fs)) return;
    const jinc::PeriodicArgs& pa = t.periodic;
    if (pa.px != 2 || pa.start_x[0] != pa.start_x[1] || pa.start_x[0] != pa.min_sx) return;
    const int H = p.g.dst_h, y_end = pa.iy0 + pa.py * pa.nj;
    const int ends[2][2] = {{0, pa.iy0}, {y_end, H}};
    std::vector<jinc::PeriodicArgs> launches;
    for (const auto& e : ends) {
        const int m = e[1] - e[0];
        if (m <= 0) continue;
        if (m > jinc::PeriodicArgs::kRowPairMaxPhases) return;
        const int origin = p.row_start[static_cast<size_t>(e[0])];
        for (int y = e[0]; y < e[1]; ++y)
            if (p.row_start[static_cast<size_t>(y)] != origin) return;
        if (origin < 0 || origin + p.fs > p.g.src_h) return;
        jinc::PeriodicArgs ra = pa;
        ra.quad = nullptr, ra.row_trim = nullptr, ra.coeffs = nullptr, ra.rows_ny = 0;
        ra.edge = jinc::PeriodicArgs::EdgeColumns{};
        ra.py = m, ra.iy0 = e[0], ra.nj = 1;
        ra.rowpair_strip_phases = (m + (m + 7) / 8 - 1) / ((m + 7) / 8);  // groups of at most eight rows (a row per wave), evenly: 17 -> 6 + 6 + 5
        for (int q = 0; q < 8; ++q) ra.start_y[q] = origin;
        ra.min_sy = origin;
        std::vector<const float*> sets;
        for (int q = 0; q < m; ++q)
            for (int px = 0; px < 2; ++px) sets.push_back(p.set_ptr(p.set_of(pa.ix0 + px, e[0] + q)));
        attach_rowpair(t, ra, p.fs, p.fs, sets, integer_samples);
        if (!ra.rowpair) return;
        launches.push_back(ra);
    }
    t.rowpair_rows = launches;
}

// Border columns on ewa_colpair_kernel (round 5; kernels.h ColPairArgs): exactly periodic plans at source step 1 whose border columns
// repeat their sets with the interior's period (strips_ok) and share one window origin per side.
void plan_colpair(const jinc::PlanePlan& p, DeviceTable& t) {
    t.use_colpair = false;
    t.colpair = jinc::ColPairArgs{};
    if (!t.use_direct || !t.strips_ok || p.sy != 1 || p.py > 4) return;
    const jinc::DirectArgs& da = t.direct;
    jinc::ColPairArgs ca;
    ca.fs = p.fs, ca.py = p.py, ca.iy0 = da.iy0, ca.nj = da.nj;
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int q = 0; q < p.py; ++q) {
        ca.start_y[q] = da.start_y[q];
        lo = std::min(lo, da.start_y[q]), hi = std::max(hi, da.start_y[q]);
    }
    ca.min_sy = lo, ca.spread = hi - lo;
    if (!jinc::colpair_supported(p.fs, p.py, p.sy, ca.spread)) return;
    ca.src_w = p.g.src_w, ca.src_h = p.g.src_h, ca.dst_h = p.g.dst_h;
    const int fs = p.fs, x_end = da.ix0 + da.px * da.ni, W = p.g.dst_w;
    const int side_x0[2] = {0, x_end}, side_n[2] = {da.ix0, W - x_end};
    constexpr int kMaxGroups = jinc::ColPairArgs::kMaxGroups;
    const size_t block = static_cast<size_t>(fs) * 4 * fs;  // floats per (side, group, q)
    std::vector<float> blob(static_cast<size_t>(2) * kMaxGroups * p.py * block, 0.f);
    bool any = false;
    for (int s = 0; s < 2; ++s) {
        const int n = side_n[s];
        if (n <= 0) continue;
        if (n > 4 * kMaxGroups) return;
        const int origin = p.col_start[static_cast<size_t>(side_x0[s])];
        for (int k = 1; k < n; ++k)
            if (p.col_start[static_cast<size_t>(side_x0[s] + k)] != origin) return;
        if (origin < 0 || origin + fs > p.g.src_w) return;
        for (int k = 0; k < n; ++k)
            for (int q = 0; q < p.py; ++q) {
                const float* set = p.set_ptr(p.set_of(side_x0[s] + k, da.iy0 + q));
                float* dst = &blob[(static_cast<size_t>(s * kMaxGroups + k / 4) * p.py + q) * block];
                for (int ly = 0; ly < fs; ++ly)
                    for (int lx = 0; lx < fs; ++lx) dst[(static_cast<size_t>(ly) * fs + lx) * 4 + k % 4] = set[ly * fs + lx];
            }
        ca.n[s] = n, ca.x0[s] = side_x0[s], ca.origin[s] = origin;
        any = true;
    }
    if (!any) return;
    void* dev = nullptr;
    hip_check(hipMalloc(&dev, blob.size() * sizeof(float)), "hipMalloc(column pair coefficients)");
    t.lane_blobs.push_back(dev);  // freed with the table
    upload(dev, blob.data(), blob.size() * sizeof(float), "column pair coefficient upload");
    ca.coeffs = static_cast<const float*>(dev);
    t.colpair = ca;
    t.use_colpair = true;
}

// Border columns inside the interior kernel (ewa_periodic_quad2_kernel / ewa_periodic_quad2x8_kernel, integer planes): the columns left and right of the
// interior, interior rows only (the corners stay with the corner kernel), are computed by the first and the last tile column of the
// interior launch from the source tile it has staged anyway.  As kernels of their own these 7 + 5 columns of C2 cost 0.28 ms of a
// 9.8 ms step -- 4.4 M scattered 64-byte lines per launch (profiles/round5/strip_ab.log) -- for 0.05 ms worth of arithmetic.
// What has to hold (checked here, not assumed): the strips repeat their sets with the interior's period (strips_ok), every column of a
// side has ONE window origin whose fs columns lie in the edge tile (or one column in front of it), and the sets' kernel rows outside
// the interior's trimmed rows are zero (the columns share the interior's row phases, so they are for the plans seen).
void plan_edge_columns(const jinc::PlanePlan& p, DeviceTable& t, bool integer_samples) {
    t.use_edge_cols = false;
    t.edge_cols = jinc::PeriodicArgs::EdgeColumns{};
    // the two-periods-per-lane quad forms: 6 kernel rows of filter size 7 (ewa_periodic_quad2_kernel), 8 of filter size 9 (..quad2x8..)
    if (!integer_samples || !t.use_periodic || !t.use_direct || !t.strips_ok || !t.periodic_trim.quad) return;
    if (!((t.trim_fs == 6 && p.fs == 7) || (t.trim_fs == 8 && p.fs == 9))) return;
    const jinc::PeriodicArgs& pa = t.periodic_trim;
    if (pa.px != 2 || pa.py != 2 || pa.start_y[0] != pa.start_y[1] || pa.start_x[0] != pa.start_x[1]) return;
    const int fs = p.fs, nr = t.trim_fs, ncp = (fs + 3) & ~3, r0 = pa.min_sy - t.periodic.min_sy;
    if (r0 < 0 || r0 + nr > fs) return;
    const int x_end = pa.ix0 + 2 * pa.ni, W = p.g.dst_w;
    constexpr int kTileCols = 128, kMax = jinc::PeriodicArgs::EdgeColumns::kMaxPerSide;
    const int kLdsCols = kTileCols + nr;  // Quad2Cfg / Quad2x8Cfg (kernel_periodic.hip)
    const size_t block = static_cast<size_t>(nr) * ncp;
    jinc::PeriodicArgs::EdgeColumns e;
    std::vector<float> blob(static_cast<size_t>(2) * kMax * 2 * block, 0.f);
    const int side_x0[2] = {0, x_end}, side_n[2] = {pa.ix0, W - x_end};
    bool any = false;
    for (int s = 0; s < 2; ++s) {
        const int n = side_n[s];
        if (n <= 0) continue;
        if (n > kMax) return;
        const int origin = p.col_start[static_cast<size_t>(side_x0[s])];
        for (int k = 1; k < n; ++k)
            if (p.col_start[static_cast<size_t>(side_x0[s] + k)] != origin) return;
        const int tile_x = s == 0 ? 0 : (pa.ni - 1) / kTileCols;
        const int lds_col = origin - pa.min_sx - kTileCols * tile_x;
        if (lds_col < -1 || lds_col + fs > kLdsCols || origin < 0 || origin + fs > p.g.src_w) return;
        for (int k = 0; k < n; ++k)
            for (int q = 0; q < 2; ++q) {
                const float* set = p.set_ptr(p.set_of(side_x0[s] + k, pa.iy0 + q));
                for (int ly = 0; ly < fs; ++ly)
                    for (int lx = 0; lx < fs; ++lx) {
                        const float c = set[ly * fs + lx];
                        if (ly < r0 || ly >= r0 + nr) {
                            if (c != 0.f) return;  // a tap outside the staged rows
                        } else {
                            blob[(static_cast<size_t>((s * kMax + k) * 2 + q)) * block + static_cast<size_t>(ly - r0) * ncp + lx] = c;
                        }
                    }
            }
        e.n[s] = n, e.x0[s] = side_x0[s], e.lds_col[s] = lds_col, e.tile_x[s] = tile_x;
        any = true;
    }
    if (!any) return;
    void* dev = nullptr;
    hip_check(hipMalloc(&dev, blob.size() * sizeof(float)), "hipMalloc(edge column coefficients)");
    t.lane_blobs.push_back(dev);  // freed with the table
    upload(dev, blob.data(), blob.size() * sizeof(float), "edge column coefficient upload");
    e.coeffs = static_cast<const float*>(dev);
    t.edge_cols = e;
    t.use_edge_cols = true;
}

// The direct kernel's interior on the trimmed support (integer planes; see trim_periodic for why leaving out taps whose
// coefficient is 0.0f is exact): the bounding box of the phase sets' non-zero coefficients, squared up.  The kernel takes
// the filter size at run time, so this is a copy of the arguments with a smaller fs, window origins moved by the box's
// corner and the sets cut to the box -- laid out like the plan's coefficient array, slack rows around it included (the row
// walk fetches coefficient rows of the "neighbouring sets", never used; upload_table).
void trim_direct(const jinc::PlanePlan& p, DeviceTable& t, bool integer_samples) {
    t.direct_trim_fs = 0;
    if (!t.use_direct || !integer_samples) return;
    if (!knobs::flag(JINC_KNOB_TRIM, true)) return;
    const jinc::DirectArgs& da = t.direct;
    const int fs = p.fs, nphase = da.px * da.py;
    int r0 = fs, r1 = -1, c0 = fs, c1 = -1;
    for (int ph = 0; ph < nphase; ++ph) {
        const float* s = p.set_ptr(da.set[ph]);
        for (int ly = 0; ly < fs; ++ly)
            for (int lx = 0; lx < fs; ++lx)
                if (s[ly * fs + lx] != 0.f) r0 = std::min(r0, ly), r1 = std::max(r1, ly), c0 = std::min(c0, lx), c1 = std::max(c1, lx);
    }
    if (r1 < 0) return;
    const int n = std::max(1, std::max(r1 - r0 + 1, c1 - c0 + 1));
    if (n >= fs || !jinc::direct_supported(n, da.px, da.py, da.sx, da.sy)) return;
    r0 = std::min(r0, fs - n);
    c0 = std::min(c0, fs - n);
    const int row = (n + 3) & ~3;
    const size_t slack_floats = static_cast<size_t>(jinc::kDirectCoeffSlackRows) * row + 64;
    std::vector<float> cut(2 * slack_floats + static_cast<size_t>(nphase) * n * row, 0.f);
    for (int ph = 0; ph < nphase; ++ph) {
        const float* s = p.set_ptr(da.set[ph]);
        for (int ly = 0; ly < n; ++ly)
            for (int lx = 0; lx < n; ++lx) cut[slack_floats + (static_cast<size_t>(ph) * n + ly) * row + lx] = s[(r0 + ly) * fs + (c0 + lx)];
    }
    void* dev = nullptr;
    hip_check(hipMalloc(&dev, cut.size() * sizeof(float)), "hipMalloc(trimmed coefficient sets of the direct kernel)");
    t.lane_blobs.push_back(dev);
    upload(dev, cut.data(), cut.size() * sizeof(float), "trimmed coefficient upload");
    jinc::DirectArgs tr = da;
    tr.coeffs = static_cast<const float*>(dev) + slack_floats;
    tr.fs = n;
    tr.coeff_row = row;
    for (int ph = 0; ph < nphase; ++ph) tr.set[ph] = ph;
    for (int k = 0; k < da.px; ++k) tr.start_x[k] = da.start_x[k] + c0;
    for (int k = 0; k < da.py; ++k) tr.start_y[k] = da.start_y[k] + r0;
    t.direct_trim = tr;
    t.direct_trim_fs = n;
}

}  // namespace

// kernel_direct.hip passes the row offset of its segment fetches as the buffer instructions' scalar offset and relies
// on the hardware range check covering it (measured on gfx950; LLVM's intrinsic documentation says otherwise).  Checked
// once per device on the device itself; 1 = covered, 0 = not (the direct kernel is then not used), < 0 = HIP error.
int buffer_range_check_covers_soffset(int device) {
    static std::atomic<int> cache[64];  // 0: unknown, 1: not covered, 2: covered
    if (device < 0 || device >= 64) return 0;
    const int c = cache[device].load();
    if (c != 0) return c - 1;
    constexpr uint32_t N = 4096;
    std::vector<uint32_t> h(2 * N / 4), r(128, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < h.size(); ++i) h[i] = i;
    uint32_t *d = nullptr, *o = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&d, 2 * N) != hipSuccess) return -1;
    if (hipMalloc(&o, 128 * 4) != hipSuccess) {
        (void)hipFree(d);
        return -1;
    }
    bool ok = true;
    try {
        upload(d, h.data(), 2 * N, "probe upload");
        ok = jinc::launch_soffset_probe(d, N, o, nullptr) == 0;
        if (ok) download(r.data(), o, 128 * 4, "probe download");
    } catch (const std::exception&) {
        ok = false;
    }
    (void)hipFree(d);
    (void)hipFree(o);
    if (!ok) return -1;
    bool covered = true;
    for (uint32_t l = 0; l < 64; ++l) {
        covered = covered && r[l] == (l < 32 ? (N - 128) / 4 + l : 0u);  // in range up to the descriptor's end, zero past it
        covered = covered && r[64 + l] == 0u;                            // scalar offset alone past the end
    }
    cache[device].store(covered ? 2 : 1);
    return covered ? 1 : 0;
}

// kernel_direct.hip fetches whole segments as naturally aligned dwords through a buffer resource that ends with the
// aligned dword holding the plane's last sample, so it cannot touch memory outside the plane's own dwords.  It needs
// 4-byte multiples for pitch and frame stride (the plane base may be anywhere) and 32-bit offsets.
bool direct_fetch_is_safe(size_t frame_stride, int nframes, uint64_t plane_bytes, int pitch, int fs) {
    if (plane_bytes + static_cast<uint64_t>(pitch) * (fs + 16) + 64 >= (1ull << 32)) return false;
    if (pitch % 4 != 0) return false;
    return nframes <= 1 || frame_stride % 4 == 0;
}
// Readable bytes from the aligned-down plane base: up to the end of the aligned dword that holds the last sample.
uint32_t direct_src_bytes(const void* base, uint64_t plane_bytes) {
    const uint64_t mis = reinterpret_cast<uintptr_t>(base) & 3u;
    return static_cast<uint32_t>((mis + plane_bytes + 3) & ~3ull);
}

namespace {
// Rectangles whose pixels own private coefficient sets (border frame of drifting plans, corners of periodic plans):
// a lane-major copy of exactly those coefficients, in the gather kernel's item order, turns its per-lane coefficient
// fetches (64 cache lines per wave and fetch) into contiguous ones.  See RectList::lane_coeffs.
void attach_lane_coeffs(const jinc::PlanePlan& p, DeviceTable& t, jinc::RectList& rects, hipStream_t stream) {
    if (!rects.private_sets || rects.n <= 0) return;
    const int fs = p.fs, fsp = (p.fs + 3) & ~3;
    const size_t item_floats = static_cast<size_t>(fs) * fsp * 64;
    long long total = 0;
    for (int r = 0; r < rects.n; ++r) {
        int axis, P;
        jinc::gather_rect_layout(t.plan, rects.w[r], rects.h[r], rects.unit_stride, axis, P);
        rects.lane_item_base[r] = total;
        total += jinc::gather_item_count(rects.w[r], rects.h[r], axis, P);
    }
    if (total <= 0 || static_cast<unsigned long long>(total) * item_floats * sizeof(float) > (512ull << 20)) return;
    std::vector<float> buf(static_cast<size_t>(total) * item_floats, 0.f);
    for (int r = 0; r < rects.n; ++r) {
        int axis, P;
        jinc::gather_rect_layout(t.plan, rects.w[r], rects.h[r], rects.unit_stride, axis, P);
        const int along = axis == 0 ? rects.w[r] : rects.h[r], across = axis == 0 ? rects.h[r] : rects.w[r];
        const int blocks = (along + 64 * P - 1) / (64 * P);
        for (int line = 0; line < across; ++line)
            for (int res = 0; res < P; ++res)
                for (int ba = 0; ba < blocks; ++ba) {
                    float* item = buf.data() + static_cast<size_t>(rects.lane_item_base[r] + static_cast<long long>(line * P + res) * blocks + ba) * item_floats;
                    for (int l = 0; l < 64; ++l) {
                        const int coord = ba * 64 * P + P * l + res;
                        if (coord >= along) break;
                        const int x = rects.x0[r] + (axis == 0 ? coord : line), y = rects.y0[r] + (axis == 0 ? line : coord);
                        const float* src = p.set_ptr(p.set_of(x, y));
                        for (int ly = 0; ly < fs; ++ly)
                            for (int lx = 0; lx < fs; ++lx)
                                item[((static_cast<size_t>(ly) * (fsp / 4) + lx / 4) * 64 + l) * 4 + lx % 4] = src[ly * fs + lx];
                    }
                }
    }
    void* dev = nullptr;
    hip_check(hipMalloc(&dev, buf.size() * sizeof(float)), "hipMalloc(lane-major coefficients)");
    t.lane_blobs.push_back(dev);
    upload(dev, buf.data(), buf.size() * sizeof(float), "lane-major coefficient upload", stream);
    rects.lane_coeffs = static_cast<const float*>(dev);
}

}  // namespace

void init_device(jinc_filter& f, int device) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) throw HipError("JincResize: no HIP device available.");
    if (device >= count) throw HipError("JincResize: HIP device index out of range.");
    hip_check(hipSetDevice(device), "hipSetDevice");
    f.device = device;
    hip_check(hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking), "hipStreamCreate");
    {   // The side stream carries the small border kernels: at the highest priority its workgroups are dispatched as
        // soon as slots free up instead of queueing behind the interior kernel, which can hold every wave slot.
        int least = 0, greatest = 0;
        hip_check(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
        hip_check(hipStreamCreateWithPriority(&f.aux_stream, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
    }
    for (int k = 0; k < jinc_filter::kForkEvents; ++k) {
        hip_check(hipEventCreateWithFlags(&f.ev_fork[k], hipEventDisableTiming), "hipEventCreate");
        hip_check(hipEventCreateWithFlags(&f.ev_join[k], hipEventDisableTiming), "hipEventCreate");
    }
    {   // a probe that could not run (-1) is a broken device, not "the premise does not hold": the latter silently moves every
        // exactly periodic down-scale and tap > 8 plan to the gather kernel at a fraction of the rate
        const int probe = buffer_range_check_covers_soffset(device);
        if (probe < 0) throw HipError("JincResize: the buffer range-check probe could not run on the HIP device.");
        f.direct_premise = probe == 1;
    }
    f.tables.resize(f.plans.size());
    for (size_t i = 0; i < f.plans.size(); ++i) {
        upload_table(f.plans[i], f.tables[i], f.stream);
        plan_launches(f.plans[i], f.tables[i]);
        if (f.tables[i].use_periodic && f.tables[i].periodic.px * f.tables[i].periodic.py == 4) {
            std::vector<const float*> sets;
            for (int ph = 0; ph < 4; ++ph) sets.push_back(f.plans[i].set_ptr(f.tables[i].periodic.set[ph]));
            attach_quad(f.tables[i], f.tables[i].periodic, f.plans[i].fs, sets);
        }
        trim_periodic(f.plans[i], f.tables[i], f.vi_in.component_size < 4);
        plan_quasi(f.plans[i], f.tables[i]);
        plan_direct(f.plans[i], f.tables[i]);
        trim_direct(f.plans[i], f.tables[i], f.vi_in.component_size < 4);
        plan_edge_columns(f.plans[i], f.tables[i], f.vi_in.component_size < 4);
        plan_rowpair_rows(f.plans[i], f.tables[i], f.vi_in.component_size < 4);
        plan_colpair(f.plans[i], f.tables[i]);
        plan_runs(f.plans[i], f.tables[i]);
        {   // every interior variant of a table must cover the same extent: the border frame is laid out once
            const DeviceTable& t = f.tables[i];
            int ex = -1, ey = -1;
            auto same = [&](int x_end, int y_end) {
                if (ex < 0) ex = x_end, ey = y_end;
                if (ex != x_end || ey != y_end) throw std::runtime_error("JincResize: interior kernels disagree about the interior extent.");
            };
            if (t.use_periodic) same(t.periodic.ix0 + t.periodic.px * t.periodic.ni, t.periodic.iy0 + t.periodic.py * t.periodic.nj);
            if (t.use_quasi) same(t.quasi.ix0 + t.quasi.px * t.quasi.ni, t.quasi.iy0 + t.quasi.py * t.quasi.nj);
            if (t.use_direct) same(t.direct.ix0 + t.direct.px * t.direct.ni, t.direct.iy0 + t.direct.py * t.direct.nj);
            if (t.use_runs) same(t.runs.ix0 + t.runs.px * t.runs.ni, t.runs.iy0 + t.runs.py * t.runs.nj);
        }
        attach_lane_coeffs(f.plans[i], f.tables[i], f.tables[i].border_rects, f.stream);
        attach_lane_coeffs(f.plans[i], f.tables[i], f.tables[i].corner_rects, f.stream);
        if ((f.tables[i].use_runs || (f.tables[i].use_quasi && !f.plans[i].periodic)) && f.tables[i].border_rects.n > 0) {
            f.tables[i].use_fl_border =
                jinc::framelane_configure(f.plans[i], f.tables[i].border_rects, f.vi_in.component_size, 64, f.tables[i].fl_border);
            f.tables[i].fl_border.plan = f.tables[i].plan;
        }
        // border columns (and corners) of exactly periodic plans in batches; window sizes of the frame-lane kernel's sliding-window
        // form only (C2 841 -> 853 Gpix/s, C1 740 -> 767; 4K -> 1080p with fs 13 on the row-segment form 205 -> 196: round4/fl_cols_ab.log)
        if (f.tables[i].use_direct && f.tables[i].column_rects.n > 0 && f.plans[i].fs <= 9) {
            f.tables[i].use_fl_cols =
                jinc::framelane_configure(f.plans[i], f.tables[i].column_rects, f.vi_in.component_size, 64, f.tables[i].fl_cols);
            f.tables[i].fl_cols.plan = f.tables[i].plan;
        }
        f.tables[i].use_framelane =
            jinc::framelane_configure(f.plans[i], f.tables[i].whole, f.vi_in.component_size, 64, f.tables[i].fl_whole);
        f.tables[i].fl_whole.plan = f.tables[i].plan;
        f.tables[i].use_framelane_pair = f.tables[i].use_framelane &&
            jinc::framelane_pair_configure(f.plans[i], f.tables[i].whole, f.vi_in.component_size, 128, f.tables[i].fl_pair);
        f.tables[i].fl_pair.plan = f.tables[i].plan;
    }
}

}  // namespace host
}  // namespace jinc
