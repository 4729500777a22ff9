// kernel_quasi_lane_fs7.hip -- ewa_quasi_kernel (drifting phases, per-lane coefficient registers) for filter size 7
// (see kernel_quasi_impl.inc).
#define JINC_QUASI_FS 7
#define JINC_QUASI_EXACT 2
#include "kernel_quasi_impl.inc"
