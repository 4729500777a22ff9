// kernel_quasi_lane_fs9.hip -- ewa_quasi_kernel (drifting phases, per-lane coefficient registers) for filter size 9
// (see kernel_quasi_impl.inc).
#define JINC_QUASI_FS 9
#define JINC_QUASI_EXACT 2
#include "kernel_quasi_impl.inc"
