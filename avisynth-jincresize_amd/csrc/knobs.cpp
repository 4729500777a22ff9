// knobs.cpp -- storage of the A/B knob table and of the "which instantiation did the launcher pick" note (knobs.h).
#include "knobs.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace jinc {
namespace knobs {

namespace {
struct Slot {
    std::atomic<bool> set{false};
    std::atomic<double> value{0.0};
};
Slot g_slots[JINC_KNOB_COUNT];

// lower-case on purpose: nothing in the library spells an environment variable's name (tests/test_build.py)
const char* const kNames[JINC_KNOB_COUNT] = {
    "trim", "quad_inner", "runs_fl_border_frames", "plane_fork", "plane_pair", "quasi_split", "fl_sub", "float_trim_min_taps",
    "float_trim_min_fs", "quad8", "fl_cols_frames", "quad_rg", "quad2x8", "float_scan", "fl_fill_weight", "fl_variant", "fl_1k",
    "fl_lds_kb", "fl_colw", "fl_threads", "flp_lds_kb", "flp_colw", "flp_threads", "blit_workgroups", "group_shares",
    "pipeline_skip", "pipeline_dma", "d2h_priority", "quasi_lds_kb", "blit_setprio", "direct_shape", "gather_passes",
    "rows_pair", "rowpair_small", "strip_lds", "edge_cols", "rowpair_rows", "colpair", "upload_bounce", "copy_threads", "stage_bands", "stage_defer_kb",
};

thread_local char t_instance[192] = "";
thread_local const char* t_kernel = nullptr;
thread_local bool t_noted = false;
}  // namespace

namespace {
std::atomic<long long> g_live_host_registrations{0};
}
void count_host_registration(int delta) { g_live_host_registrations += delta; }
long long live_host_registrations() { return g_live_host_registrations.load(); }
std::mutex& host_registration_mutex() {
    static std::mutex& m = *new std::mutex;  // (leaked: instances may be freed while the process's statics are being destroyed)
    return m;
}

bool is_set(int id) { return id >= 0 && id < JINC_KNOB_COUNT && g_slots[id].set.load(std::memory_order_acquire); }

double get(int id, double unset_value) { return is_set(id) ? g_slots[id].value.load(std::memory_order_relaxed) : unset_value; }

void set(int id, double value) {
    if (id < 0 || id >= JINC_KNOB_COUNT) return;
    g_slots[id].value.store(value, std::memory_order_relaxed);
    g_slots[id].set.store(true, std::memory_order_release);
}

void clear(int id) {
    for (int k = 0; k < JINC_KNOB_COUNT; ++k)
        if (id < 0 || id == k) g_slots[k].set.store(false, std::memory_order_release);
}

const char* name(int id) { return id >= 0 && id < JINC_KNOB_COUNT ? kNames[id] : nullptr; }

void note_instance(const char* kernel, const char* fmt, ...) {
    char targs[128];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(targs, sizeof(targs), fmt, ap);
    va_end(ap);
    std::snprintf(t_instance, sizeof(t_instance), "%s<%s>", kernel, targs);
    t_kernel = kernel;
    t_noted = true;
}

const char* take_instance(const char** kernel) {
    if (kernel) *kernel = t_noted ? t_kernel : nullptr;
    if (!t_noted) return "";
    t_noted = false;
    return t_instance;
}

}  // namespace knobs
}  // namespace jinc
