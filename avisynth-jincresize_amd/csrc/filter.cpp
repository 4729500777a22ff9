// filter.cpp -- host side of libjincresize_hip.so: the C ABI declared in include/jincresize_hip.h.
//
// Mirrors the reference plugin's filter life cycle for the accelerated path
// ("ref:" = /root/reference/src/JincResize.cpp):
//   jinc_filter_create      <- Create_JincResize      (ref :654-984)  same defaults, same checks in
//                                                      the same order, same error strings
//   jinc_filter_get_frame   <- process_frame call in JincResize_GetFrame (ref :615)
//   jinc_filter_free        <- free_JincResize         (ref :632-647)
//   jinc_alias_args         <- resizer()/resizer_jincresize<taps> (ref :1007-1040)
// There is no CPU fallback: without a HIP device every frame call fails loudly.
#include "filter_internal.h"
#include "knobs.h"

using namespace jinc::host;

namespace {
thread_local std::string g_last_error;
}  // namespace

namespace jinc {
namespace host {
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
}  // namespace host
}  // namespace jinc

namespace {

template <typename Fn>
int guarded(Fn&& fn) {
    try {
        fn();
        g_last_error.clear();
        return JINC_OK;
    } catch (const ArgError& e) {
        return fail(JINC_ERR_INVALID_ARG, e.what());
    } catch (const HipError& e) {
        return fail(JINC_ERR_HIP, e.what());
    } catch (const std::bad_alloc&) {
        return fail(JINC_ERR_NOMEM, "JincResize: out of memory.");
    } catch (const std::exception& e) {
        return fail(JINC_ERR_UNSUPPORTED, e.what());
    }
}

const jinc::PlanePlan* table_or_null(const jinc_filter* f, int table) {
    if (!f || table < 0 || table >= static_cast<int>(f->plans.size())) return nullptr;
    return &f->plans[table];
}

}  // namespace

extern "C" {

int jinc_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int jinc_pick_device(void) {
    static std::atomic<unsigned> counter{0};
    const int n = jinc_device_count();
    if (n <= 0) return -1;
    return static_cast<int>(counter.fetch_add(1) % static_cast<unsigned>(n));
}

const char* jinc_last_error(void) { return g_last_error.c_str(); }

int jinc_filter_create(const jinc_video_info* vi, const jinc_args* args, int device, jinc_filter** out, char* err,
                       size_t err_len) {
    if (out) *out = nullptr;
    if (err && err_len) err[0] = '\0';
    if (!vi || !args || !out) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    std::unique_ptr<jinc_filter> f(new (std::nothrow) jinc_filter());
    if (!f) return fail(JINC_ERR_NOMEM, "JincResize: out of memory.");
    int rc = guarded([&] {
        configure(*f, *vi, *args);
        if (device >= 0) init_device(*f, device);
    });
    if (rc == JINC_ERR_HIP && device >= 0 && jinc_device_count() == 0) rc = JINC_ERR_NO_DEVICE;
    if (rc != JINC_OK) {
        if (err && err_len) {
            std::strncpy(err, g_last_error.c_str(), err_len - 1);
            err[err_len - 1] = '\0';
        }
        return rc;
    }
    *out = f.release();
    return JINC_OK;
}

void jinc_filter_free(jinc_filter* f) { delete f; }

int jinc_filter_output_info(const jinc_filter* f, jinc_video_info* out_vi) {
    if (!f || !out_vi) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    *out_vi = f->vi_out;
    return JINC_OK;
}

int jinc_filter_chroma_location(const jinc_filter* f) {
    if (!f) return -1;
    return f->chroma_location_mode == JINC_CHROMA_LOCATION_BY_SITING ? f->chroma_location_by_siting : f->chroma_location;
}

int jinc_filter_set_chroma_location_mode(jinc_filter* f, int mode) {
    if (!f || (mode != JINC_CHROMA_LOCATION_AS_REFERENCE && mode != JINC_CHROMA_LOCATION_BY_SITING))
        return fail(JINC_ERR_INVALID_ARG, "JincResize: chroma location mode must be 0 or 1.");
    f->chroma_location_mode = mode;
    return JINC_OK;
}

int jinc_filter_get_frame(jinc_filter* f, const void* const src[4], const int src_pitch[4], void* const dst[4],
                          const int dst_pitch[4]) {
    if (!f || !src || !dst || !src_pitch || !dst_pitch) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        drain_pipeline(*f);  // a synchronous frame must not overtake frames still in the pipeline
        wait_frame(*f, submit_frame(*f, src, src_pitch, dst, dst_pitch));
    });
}

int jinc_filter_set_pipeline_group(jinc_filter* f, int depth, int group, int register_host_buffers) {
    if (!f || depth < 1 || depth > kMaxPipelineDepth) return fail(JINC_ERR_INVALID_ARG, "JincResize: pipeline depth must be 1..256.");
    if (group < 0 || group > depth) return fail(JINC_ERR_INVALID_ARG, "JincResize: frames per launch must be 0 (automatic) .. depth.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        configure_pipeline(*f, depth, group, register_host_buffers);  // 0 staged, 3 the runtime's own mapping, anything else cached registrations
    });
}

int jinc_filter_set_pipeline(jinc_filter* f, int depth, int register_host_buffers) {
    return jinc_filter_set_pipeline_group(f, depth, 0, register_host_buffers);
}

int jinc_filter_pipeline_group(const jinc_filter* f) { return f ? f->group_frames : 0; }

int jinc_filter_submit(jinc_filter* f, const void* const src[4], const int src_pitch[4], void* const dst[4],
                       const int dst_pitch[4], long long* ticket) {
    if (!f || !src || !dst || !src_pitch || !dst_pitch || !ticket) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        *ticket = submit_frame(*f, src, src_pitch, dst, dst_pitch);
    });
}

int jinc_filter_adopt_host_range(jinc_filter* f, void* base, size_t bytes) {
    if (!f || !base || !bytes) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        adopt_host_range(*f, base, bytes);
    });
}

int jinc_filter_release_host_range(jinc_filter* f, void* base, size_t bytes) {
    if (!f || !base || !bytes) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        release_host_range(*f, base, bytes);
    });
}

long long jinc_debug_staged_frames(void) { return staged_frames(); }

int jinc_debug_usable_cpus(void) { return copy_lanes_cpus(); }

int jinc_debug_copy_rows(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes, int rows, int may_use_helpers) {
    if (!dst || !src || rows < 0 || dst_pitch < row_bytes || src_pitch < row_bytes) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad copy.");
    copy_plane_rows(static_cast<char*>(dst), dst_pitch, static_cast<const char*>(src), src_pitch, row_bytes, rows, may_use_helpers != 0);
    return JINC_OK;
}

int jinc_debug_transport_counts(long long* by_shader, long long* by_dma, long long* pinned_ranges, int reset) {
    transport_counts(by_shader, by_dma, pinned_ranges, reset != 0);
    return JINC_OK;
}

int jinc_filter_flush(jinc_filter* f) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        if (f->open_group >= 0) launch_open_group(*f);
    });
}

int jinc_filter_wait(jinc_filter* f, long long ticket) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        wait_frame(*f, ticket);
    });
}

int jinc_filter_process_device(jinc_filter* f, const void* const src[4], const int src_pitch[4],
                               const size_t src_frame_stride[4], void* const dst[4], const int dst_pitch[4],
                               const size_t dst_frame_stride[4], int nframes, void* hip_stream) {
    if (!f || !src || !dst || !src_pitch || !dst_pitch) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    if (nframes < 1 || nframes > 65535) return fail(JINC_ERR_INVALID_ARG, "JincResize: nframes must be in 1..65535.");
    if (nframes > 1 && (!src_frame_stride || !dst_frame_stride))
        return fail(JINC_ERR_INVALID_ARG, "JincResize: frame strides are required for nframes > 1.");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        hipStream_t s = static_cast<hipStream_t>(hip_stream);  // NULL = the HIP null stream, ordered with the caller's default-stream work
        enqueue(*f, src, src_pitch, src_frame_stride, dst, dst_pitch, dst_frame_stride, nframes, s);
    });
}

int jinc_filter_sync(jinc_filter* f) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        drain_pipeline(*f);
        hip_check(hipStreamSynchronize(f->stream), "stream sync");
    });
}

int jinc_alias_args(int taps, const jinc_args* in, jinc_args* out) {
    if (!in || !out) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (taps != 3 && taps != 4 && taps != 6 && taps != 8)
        return fail(JINC_ERR_INVALID_ARG, "JincResize: alias tap count must be 3, 4, 6 or 8.");
    jinc_args a{};
    a.target_width = in->target_width;
    a.target_height = in->target_height;
    const unsigned forwarded = JINC_ARG_SRC_LEFT | JINC_ARG_SRC_TOP | JINC_ARG_SRC_WIDTH | JINC_ARG_SRC_HEIGHT |
                               JINC_ARG_QUANT_X | JINC_ARG_QUANT_Y | JINC_ARG_CPLACE | JINC_ARG_THREADS;
    a.defined = (in->defined & forwarded) | JINC_ARG_TAP;
    a.src_left = in->src_left;
    a.src_top = in->src_top;
    a.src_width = in->src_width;
    a.src_height = in->src_height;
    a.quant_x = in->quant_x;
    a.quant_y = in->quant_y;
    a.cplace = in->cplace;
    a.threads = in->threads;
    a.tap = taps;
    a.frame0_chroma_location = in->frame0_chroma_location;
    a.cpu_has_sse41 = in->cpu_has_sse41;
    a.cpu_has_avx2 = in->cpu_has_avx2;
    a.cpu_has_avx512f = in->cpu_has_avx512f;
    *out = a;
    return JINC_OK;
}

int jinc_filter_num_tables(const jinc_filter* f) { return f ? static_cast<int>(f->plans.size()) : 0; }

int jinc_filter_plan_info(const jinc_filter* f, int table, jinc_plan_info* out) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p || !out) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table index.");
    jinc_plan_info i{};
    i.src_width = p->g.src_w;
    i.src_height = p->g.src_h;
    i.dst_width = p->g.dst_w;
    i.dst_height = p->g.dst_h;
    i.filter_size = p->fs;
    i.num_sets = p->num_sets;
    i.periodic = p->periodic ? 1 : 0;
    i.period_x = p->px;
    i.period_y = p->py;
    i.step_x = p->sx;
    i.step_y = p->sy;
    i.interior_x0 = p->ix0;
    i.interior_x1 = p->ix1;
    i.interior_y0 = p->iy0;
    i.interior_y1 = p->iy1;
    i.plan_bytes = static_cast<int64_t>(4) * (p->col_start.size() + p->row_start.size() + p->col_class.size() +
                                              p->row_class.size() + p->interior_set.size() + p->bcol_set.size() +
                                              p->brow_set.size() + p->coeffs.size());
    i.quasi = p->quasi ? 1 : 0;
    i.quasi_period_x = p->qpx;
    i.quasi_period_y = p->qpy;
    i.quasi_step_x = p->qsx;
    i.quasi_step_y = p->qsy;
    *out = i;
    return JINC_OK;
}

int jinc_filter_plan_pixel(const jinc_filter* f, int table, int x, int y, int* start_x, int* start_y, float* coeffs) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p || x < 0 || y < 0 || x >= p->g.dst_w || y >= p->g.dst_h)
        return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table index or pixel.");
    if (start_x) *start_x = p->col_start[x];
    if (start_y) *start_y = p->row_start[y];
    if (coeffs) std::memcpy(coeffs, p->set_ptr(p->set_of(x, y)), sizeof(float) * p->fs * p->fs);
    return JINC_OK;
}

int jinc_filter_plan_dump(const jinc_filter* f, int table, int* start_x, int* start_y, int* set_id) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table index.");
    if (start_x) std::memcpy(start_x, p->col_start.data(), sizeof(int) * p->g.dst_w);
    if (start_y) std::memcpy(start_y, p->row_start.data(), sizeof(int) * p->g.dst_h);
    if (set_id)
        for (int y = 0; y < p->g.dst_h; ++y)
            for (int x = 0; x < p->g.dst_w; ++x) set_id[static_cast<size_t>(y) * p->g.dst_w + x] = p->set_of(x, y);
    return JINC_OK;
}

int jinc_filter_plan_runs(const jinc_filter* f, int table, int* n_runs, int* n_items, int32_t* runs, int capacity) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p || !n_runs || !n_items) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table index or null argument.");
    std::vector<jinc::PlanRun> list;
    std::vector<int32_t> item_run;
    if (!jinc::build_plan_runs(*p, list, item_run)) list.clear(), item_run.clear();
    *n_runs = static_cast<int>(list.size());
    *n_items = static_cast<int>(item_run.size());
    if (runs && capacity > 0)
        std::memcpy(runs, list.data(), sizeof(jinc::PlanRun) * std::min<size_t>(list.size(), static_cast<size_t>(capacity)));
    return JINC_OK;
}

int jinc_filter_plan_set(const jinc_filter* f, int table, int set, float* coeffs) {
    const jinc::PlanePlan* p = table_or_null(f, table);
    if (!p || !coeffs || set < 0 || set >= p->num_sets) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad table or set index.");
    std::memcpy(coeffs, p->set_ptr(set), sizeof(float) * p->fs * p->fs);
    return JINC_OK;
}

int jinc_filter_lut(const jinc_filter* f, double* lut1024) {
    if (!f || !lut1024) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    std::memcpy(lut1024, f->lut.v.data(), sizeof(double) * jinc::kLutSamples);
    return JINC_OK;
}

int jinc_filter_set_profiling(jinc_filter* f, int enable) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    f->profiling = enable != 0;
    return JINC_OK;
}

int jinc_filter_kernel_times(jinc_filter* f, double* periodic_ms, int* periodic_launches, double* gather_ms,
                             int* gather_launches) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    if (f->device < 0) return fail(JINC_ERR_NO_DEVICE, "JincResize: filter was created without a HIP device (device < 0).");
    return guarded([&] {
        hip_check(hipSetDevice(f->device), "hipSetDevice");
        auto collect = [](std::vector<EventPair>& v, double* ms_out, int* n_out) {
            double total = 0.0;
            for (auto& e : v) {
                hip_check(hipEventSynchronize(e.stop), "hipEventSynchronize");
                float ms = 0.f;
                hip_check(hipEventElapsedTime(&ms, e.start, e.stop), "hipEventElapsedTime");
                total += ms;
                (void)hipEventDestroy(e.start);
                (void)hipEventDestroy(e.stop);
            }
            if (ms_out) *ms_out = total;
            if (n_out) *n_out = static_cast<int>(v.size());
            v.clear();
        };
        collect(f->ev_periodic, periodic_ms, periodic_launches);
        collect(f->ev_gather, gather_ms, gather_launches);
    });
}

int jinc_debug_convert(const float* sums, void* out, int n, int sample_bytes, float peak, int device) {
    if (!sums || !out || n < 0 || (sample_bytes != 1 && sample_bytes != 2 && sample_bytes != 4))
        return fail(JINC_ERR_INVALID_ARG, "JincResize: bad argument.");
    return guarded([&] {
        hip_check(hipSetDevice(device), "hipSetDevice");
        float* d_in = nullptr;
        void* d_out = nullptr;
        hip_check(hipMalloc(&d_in, sizeof(float) * (n + 1)), "hipMalloc");
        hip_check(hipMalloc(&d_out, static_cast<size_t>(sample_bytes) * (n + 1)), "hipMalloc");
        bounce_upload(d_in, sums, sizeof(float) * n, "upload of the sums");
        hip_check(static_cast<hipError_t>(jinc::launch_debug_convert(d_in, d_out, n, sample_bytes, peak, nullptr)), "convert launch");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        bounce_download(out, d_out, static_cast<size_t>(sample_bytes) * n, "download of the samples");
        (void)hipFree(d_in);
        (void)hipFree(d_out);
    });
}

const char* jinc_filter_interior_kernel(const jinc_filter* f, int table) {
    if (!f || f->device < 0 || table < 0 || table >= static_cast<int>(f->tables.size())) return "";
    const DeviceTable& t = f->tables[table];
    const int m = f->kernel_mode;
    const bool quasi = t.use_quasi && (m == 7 || m == 8 || m == 10 || (m != 1 && !t.use_periodic));
    const bool periodic = t.use_periodic && m != 1 && m != 7 && m != 8 && m != 10;
    if (t.use_direct && m != 1 && (m == 9 || (!periodic && !quasi))) return "ewa_direct_kernel";
    if (t.use_runs && f->direct_premise && (m == 14 || (m == 0 && t.plan.fs >= 9))) return "ewa_direct_runs_kernel";
    if (quasi) return "ewa_quasi_kernel";
    if (periodic) {
        const int fs = (t.trim_fs > 0 && t.trim_nx == t.trim_fs && !f->full_window && m != 5 && m != 6) ? t.trim_fs : t.plan.fs;
        if (m == 5 || m == 6) return t.plan.fs == 7 ? "ewa_periodic_pk_kernel" : "ewa_periodic_kernel";
        if (m == 3 || fs < 6 || fs > 9) return "ewa_periodic_rows_kernel";
        return "ewa_periodic_kernel";
    }
    return "ewa_gather_kernel";
}

namespace {
// Does the periodic family run `t` on its trimmed support under the filter's kernel mode?  (csrc/dispatch.cpp Choice::trimmed
// without what depends on the call: float planes take it from Rules::kFloatTrimMinTaps taps per plane and call on.)
bool runs_trimmed(const jinc_filter* f, const DeviceTable& t) {
    return t.trim_fs > 0 && !f->full_window && f->kernel_mode != 5 && f->kernel_mode != 6;
}
}  // namespace

int jinc_filter_periodic_support(const jinc_filter* f, int table) {
    if (!f || f->device < 0 || table < 0 || table >= static_cast<int>(f->tables.size())) return 0;
    const DeviceTable& t = f->tables[table];
    if (!t.use_periodic) return 0;
    return runs_trimmed(f, t) ? t.trim_fs : t.plan.fs;
}

double jinc_filter_periodic_taps(const jinc_filter* f, int table, int rows_kernel) {
    if (!f || f->device < 0 || table < 0 || table >= static_cast<int>(f->tables.size())) return 0;
    const DeviceTable& t = f->tables[table];
    if (rows_kernel == 2) {  // the direct kernel's interior
        if (!t.use_direct) return 0;
        const int n = (t.direct_trim_fs > 0 && !f->full_window) ? t.direct_trim_fs : t.plan.fs;
        return static_cast<double>(n) * n;
    }
    if (!t.use_periodic) return 0;
    const bool trimmed = runs_trimmed(f, t);
    if (rows_kernel == 4) {  // ewa_periodic_rowpair_kernel: the spans both phases p of a (q, kernel row) share
        const jinc::PeriodicArgs& pa = (trimmed && t.trim_nx == t.trim_fs) ? t.periodic_trim : t.periodic;
        double taps = 0;
        for (int q = 0; q < pa.py; ++q) {
            const int first = static_cast<int>(pa.rowpair_trim[q] >> 54) & 31, last = static_cast<int>(pa.rowpair_trim[q] >> 59) & 31;  // kernel rows executed
            for (int ly = first; ly < (last ? last : pa.rowpair_ny); ++ly) taps += pa.rowpair_n - 2 * static_cast<int>((pa.rowpair_trim[q] >> (3 * ly)) & 7u);
        }
        return pa.rowpair ? taps / pa.py : 0.0;
    }
    if (rows_kernel == 3 && t.trim_fs == 8 && t.trim_nx == 9 && trimmed)  // 8 rows x 9 columns; with the MPEG-2 chords 60 of the 72
        return (jinc::quad_span9_fits(t.periodic_trim.quad_span7, jinc::kQuadSpan9Mpeg2) || jinc::quad_span9_fits(t.periodic_trim.quad_span7, jinc::kQuadSpan9Mpeg2Swapped))
                   ? jinc::quad_span9_taps(jinc::kQuadSpan9Mpeg2) / 2.0
                   : 72.0;
    if (rows_kernel == 3 && t.trim_fs == 6 && t.trim_nx == 7 && trimmed)  // 6 rows x 7 columns; with the MPEG-2 chords 36 of the 42
        return (jinc::quad_span7_fits(t.periodic_trim.quad_span7, jinc::PeriodicArgs::kQuadSpan7Mpeg2) ||
                jinc::quad_span7_fits(t.periodic_trim.quad_span7, jinc::PeriodicArgs::kQuadSpan7Mpeg2Swapped))
                   ? jinc::quad_span7_taps(jinc::PeriodicArgs::kQuadSpan7Mpeg2) / 2.0
                   : 42.0;
    if (rows_kernel == 3 && t.trim_fs == 6 && trimmed &&  // ewa_periodic_quad2_kernel: chord rows on four taps (half the samples each)
        (t.periodic_trim.quad_inner & jinc::PeriodicArgs::kQuadInnerTap3) == jinc::PeriodicArgs::kQuadInnerTap3)
        return 34.0;
    if (rows_kernel == 3 && t.trim_fs == 8 && trimmed && t.periodic_trim.quad &&  // quad forms on the 8 x 8 support with the tap-4 pattern
        jinc::quad8_pattern_fits(t.periodic_trim.quad_trim8, jinc::kQuad8TrimTap4Value))
        return 56.0;
    if (trimmed && t.trim_nx == t.trim_fs) return rows_kernel == 1 ? t.trim_rows_taps : static_cast<double>(t.trim_fs) * t.trim_fs;
    return static_cast<double>(t.plan.fs) * t.plan.fs;
}

int jinc_filter_set_simd_order(jinc_filter* f, int order) {
    if (!f || order < 0 || order > 3) return fail(JINC_ERR_INVALID_ARG, "JincResize: SIMD order must be 0..3.");
    f->simd_order = order;
    return JINC_OK;
}

int jinc_debug_buffer_range_check(int device) {
    const int r = buffer_range_check_covers_soffset(device);
    if (r < 0) return fail(JINC_ERR_HIP, "JincResize: the buffer range-check probe could not run.");
    return r;
}

int jinc_debug_set_direct_shape(int shape) {
    if (shape != -1 && shape != 0 && shape != 2 && shape != 3) return fail(JINC_ERR_INVALID_ARG, "JincResize: direct shape must be -1, 0, 2 or 3.");
    jinc::set_direct_shape(shape);
    return JINC_OK;
}

int jinc_debug_last_direct_shape(void) { return jinc::last_direct_shape(); }

const char* jinc_filter_last_kernel(const jinc_filter* f, int table) {
    if (!f || f->device < 0 || table < 0 || table >= static_cast<int>(f->tables.size())) return "";
    return f->tables[table].last_kernel;
}

const char* jinc_filter_last_instance(const jinc_filter* f, int table) {
    if (!f || f->device < 0 || table < 0 || table >= static_cast<int>(f->tables.size())) return "";
    return f->tables[table].last_instance.c_str();
}

int jinc_debug_set_knob(int knob, double value) {
    if (knob < 0 || knob >= JINC_KNOB_COUNT) return fail(JINC_ERR_INVALID_ARG, "JincResize: no such knob.");
    if (knob == JINC_KNOB_DIRECT_SHAPE) return jinc_debug_set_direct_shape(static_cast<int>(value));
    jinc::knobs::set(knob, value);
    return JINC_OK;
}

int jinc_debug_clear_knob(int knob) {
    if (knob >= JINC_KNOB_COUNT) return fail(JINC_ERR_INVALID_ARG, "JincResize: no such knob.");
    jinc::knobs::clear(knob);
    if (knob < 0 || knob == JINC_KNOB_DIRECT_SHAPE) jinc::set_direct_shape(-1);
    return JINC_OK;
}

int jinc_debug_get_knob(int knob, double* value) {
    if (knob < 0 || knob >= JINC_KNOB_COUNT) return fail(JINC_ERR_INVALID_ARG, "JincResize: no such knob.");
    if (!jinc::knobs::is_set(knob)) return 0;
    if (value) *value = jinc::knobs::get(knob, 0.0);
    return 1;
}

const char* jinc_debug_knob_name(int knob) { return jinc::knobs::name(knob); }
int jinc_debug_chord_pattern(int taps_per_row, uint64_t spans) {
    if (taps_per_row == 7)
        return jinc::quad_span7_fits(spans, jinc::PeriodicArgs::kQuadSpan7Mpeg2) ? 1 : jinc::quad_span7_fits(spans, jinc::PeriodicArgs::kQuadSpan7Mpeg2Swapped) ? 2 : 0;
    if (taps_per_row == 9) return jinc::quad_span9_fits(spans, jinc::kQuadSpan9Mpeg2) ? 1 : jinc::quad_span9_fits(spans, jinc::kQuadSpan9Mpeg2Swapped) ? 2 : 0;
    return -1;
}

// Shader-clock sampler beside the kernels being timed (kernel_probe.hip).
struct jinc_clock_sampler {
    int device = 0;
    hipStream_t stream = nullptr, control = nullptr;
    int* stop = nullptr;                 // DEVICE memory, raised by a memset on the control stream (a flag in pinned host
                                         // memory is not seen by a running kernel here: the samplers ran to their time limit)
    unsigned long long* out = nullptr;   // pinned host memory, 2 x kSamplers
    static constexpr int kSamplers = 8;
};

int jinc_debug_clock_sampler_start(int device, double max_seconds, jinc_clock_sampler** out) {
    if (!out || max_seconds <= 0.0 || max_seconds > 120.0) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad argument.");
    *out = nullptr;
    std::unique_ptr<jinc_clock_sampler> s(new jinc_clock_sampler());
    s->device = device;
    const int rc = guarded([&] {
        hip_check(hipSetDevice(device), "hipSetDevice");
        // default priority: a dispatch that stays active on a HIGH-priority stream throttles the wave launch of every other
        // queue for as long as it lives (measured: eight sleeping sampler waves at the highest priority cost the direct
        // kernels 10-15 %: 1080p -> 720p 256 -> 223 Gpix/s; profiles/round3/clock_sampler_priority.log)
        hip_check(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipStreamCreateWithFlags(&s->control, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipMalloc(reinterpret_cast<void**>(&s->stop), 256), "hipMalloc");
        hip_check(hipMemsetAsync(s->stop, 0, 256, s->control), "hipMemsetAsync");
        hip_check(hipStreamSynchronize(s->control), "stream sync");
        hip_check(hipHostMalloc(reinterpret_cast<void**>(&s->out), sizeof(unsigned long long) * 2 * jinc_clock_sampler::kSamplers, hipHostMallocDefault),
                  "hipHostMalloc");
        std::memset(s->out, 0, sizeof(unsigned long long) * 2 * jinc_clock_sampler::kSamplers);
        hip_check(static_cast<hipError_t>(jinc::launch_clock_sampler(s->stop, s->out, jinc_clock_sampler::kSamplers, max_seconds, s->stream)),
                  "clock sampler launch");
    });
    if (rc != JINC_OK) {
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        if (s->stop) (void)hipFree(s->stop);
        if (s->out) (void)hipHostFree(s->out);
        if (s->stream) (void)hipStreamDestroy(s->stream);
        if (s->control) (void)hipStreamDestroy(s->control);
        return rc;
    }
    *out = s.release();
    return JINC_OK;
}

int jinc_debug_clock_sampler_stop(jinc_clock_sampler* s, double* ghz_min, double* ghz_median, double* ghz_max) {
    if (!s) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    const int rc = guarded([&] {
        hip_check(hipSetDevice(s->device), "hipSetDevice");
        hip_check(hipMemsetAsync(s->stop, 1, 4, s->control), "hipMemsetAsync(stop flag)");  // any non-zero value stops the samplers
        hip_check(hipStreamSynchronize(s->control), "stream sync");
        hip_check(hipStreamSynchronize(s->stream), "stream sync");
        std::vector<double> ghz;
        for (int k = 0; k < jinc_clock_sampler::kSamplers; ++k)
            if (s->out[2 * k + 1] > 0) ghz.push_back(static_cast<double>(s->out[2 * k]) / static_cast<double>(s->out[2 * k + 1]) * 0.1);
        std::sort(ghz.begin(), ghz.end());
        if (ghz.empty()) throw HipError("JincResize: the clock samplers did not run.");
        if (ghz_min) *ghz_min = ghz.front();
        if (ghz_max) *ghz_max = ghz.back();
        if (ghz_median) *ghz_median = 0.5 * (ghz[(ghz.size() - 1) / 2] + ghz[ghz.size() / 2]);
    });
    (void)hipFree(s->stop);
    (void)hipHostFree(s->out);
    (void)hipStreamDestroy(s->stream);
    (void)hipStreamDestroy(s->control);
    delete s;
    return rc;
}

int jinc_debug_valu_pair_probe(int device, int waves_per_simd, double* tops, double* shader_clock_ghz) {
    if (waves_per_simd < 1 || waves_per_simd > 8 || !tops) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad argument.");
    return guarded([&] {
        hip_check(hipSetDevice(device), "hipSetDevice");
        hipDeviceProp_t prop;
        hip_check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
        const int blocks = prop.multiProcessorCount * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
        const int iters = 100000;                                     // ~15 ms at 8 waves per SIMD
        float* out = nullptr;
        hip_check(hipMalloc(&out, sizeof(float) * 256 * blocks), "hipMalloc");
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hip_check(hipEventCreate(&e0), "hipEventCreate");
        hip_check(hipEventCreate(&e1), "hipEventCreate");
        jinc_clock_sampler* cs = nullptr;
        double best = 1e30, ghz = 0.0;
        for (int rep = 0; rep < 4; ++rep) {  // the first launch ramps the clock; the last one is sampled
            if (rep == 3 && shader_clock_ghz) (void)jinc_debug_clock_sampler_start(device, 5.0, &cs);
            hip_check(hipEventRecord(e0, nullptr), "hipEventRecord");
            hip_check(static_cast<hipError_t>(jinc::launch_valu_pair_probe(out, blocks, iters, nullptr)), "probe launch");
            hip_check(hipEventRecord(e1, nullptr), "hipEventRecord");
            hip_check(hipEventSynchronize(e1), "hipEventSynchronize");
            float ms = 0.f;
            hip_check(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
            if (rep > 0) best = std::min(best, static_cast<double>(ms));
        }
        if (cs) (void)jinc_debug_clock_sampler_stop(cs, nullptr, &ghz, nullptr);
        *tops = 16.0 * iters * 256.0 * blocks / (best * 1e-3) / 1e12;
        if (shader_clock_ghz) *shader_clock_ghz = ghz;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(out);
    });
}

const char* jinc_debug_last_call(int* nframes) {
    if (nframes) *nframes = last_call_frames_in_process();
    return last_interior_kernel_in_process();
}

int jinc_filter_last_border(const jinc_filter* f, int table) {
    if (!f || f->device < 0 || table < 0 || table >= static_cast<int>(f->tables.size())) return 0;
    return f->tables[table].last_border;
}

const char* jinc_debug_last_instance(void) { return last_interior_instance_in_process(); }

int jinc_filter_direct_premise(const jinc_filter* f) { return (f && f->device >= 0) ? (f->direct_premise ? 1 : 0) : -1; }

int jinc_filter_set_border_strips(jinc_filter* f, int enable) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    f->border_strips = enable < 0 ? -1 : enable > 4 ? 1 : enable;  // -1: by call size; 2: rows as strips, columns on the gather kernel; 3: ewa_strip_kernel; 4: 3 + columns inside the interior kernel
    g_last_error.clear();
    return JINC_OK;
}

int jinc_filter_set_kernel_mode(jinc_filter* f, int mode) {
    if (!f || mode < 0 || mode > 16) return fail(JINC_ERR_INVALID_ARG, "JincResize: bad kernel mode.");
    f->full_window = mode == 15;  // 15 = the automatic choice, but on the reference's full window (no trimmed support)
    f->kernel_mode = mode == 15 ? 0 : mode;
    return JINC_OK;
}

int jinc_filter_set_border_overlap(jinc_filter* f, int enable) {
    if (!f) return fail(JINC_ERR_INVALID_ARG, "JincResize: null argument.");
    f->overlap_border = enable < 0 ? -1 : (enable != 0);
    return JINC_OK;
}

}  // extern "C"
