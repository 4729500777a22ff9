// knobs.h -- the process-wide table of A/B and tuning knobs (ids: enum jinc_knob of include/jincresize_hip_test.h).
// The product's code never reads the environment for them: a knob is unset until a test, bench.py or a profiles/ script sets
// it through jinc_debug_set_knob; every site that consults one names the default it uses while the knob is unset.
#pragma once
#include <mutex>
#include "../../include/jincresize_hip_test.h"

namespace jinc {
namespace knobs {

bool is_set(int id);
double get(int id, double unset_value);
inline int geti(int id, int unset_value) { return is_set(id) ? static_cast<int>(get(id, 0.0)) : unset_value; }
// an on/off knob: `unset_value` while unset, otherwise value != 0
inline bool flag(int id, bool unset_value) { return is_set(id) ? get(id, 0.0) != 0.0 : unset_value; }

// hipHostRegister / hipHostUnregister calls of this library that succeeded, process-wide (pipeline.cpp's registry and batch.cpp's
// registrars): registered - unregistered = host ranges the library holds pinned right now (test header: jinc_debug_host_registrations).
void count_host_registration(int delta);
long long live_host_registrations();
// One hipHostRegister / hipHostUnregister call at a time, process-wide (pipeline.cpp's registry and batch.cpp's registrars take it
// around each call).  Ranges of neighbouring buffers share pages (registrations cover exact bytes), and the one test that had four
// threads register such neighbours side by side while the device worked through earlier ones is the one that ended in GPU memory
// access faults twice in six runs (profiles/round6/README.md); none of the isolated probes reproduces it, so this is a precaution.
std::mutex& host_registration_mutex();

void set(int id, double value);
void clear(int id);  // id < 0: every knob
const char* name(int id);  // lower-case name ("quad_rg"), nullptr outside 0 .. JINC_KNOB_COUNT - 1

// Full template instantiation of the kernel a launcher has just launched on the calling thread, spelled as rocprofv3 prints it
// ("ewa_periodic_quad2_kernel<unsigned char, 8, 1026u, 6>"): launchers call note_instance, dispatch.cpp takes the note after
// the interior launch (test header: jinc_filter_last_instance).  take_instance returns "" (and no kernel name) when nothing was
// noted since the last take.
// `kernel` must be a string literal (the kernel's plain name; it outlives the call), `fmt` prints the template arguments.
void note_instance(const char* kernel, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* take_instance(const char** kernel = nullptr);
template <typename T> constexpr const char* type_name();
template <> constexpr const char* type_name<unsigned char>() { return "unsigned char"; }
template <> constexpr const char* type_name<unsigned short>() { return "unsigned short"; }
template <> constexpr const char* type_name<float>() { return "float"; }

}  // namespace knobs
}  // namespace jinc
