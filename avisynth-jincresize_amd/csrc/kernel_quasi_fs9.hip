// kernel_quasi_fs9.hip -- ewa_quasi_kernel (drifting phases) for filter size 9 (see kernel_quasi_impl.inc).
#define JINC_QUASI_FS 9
#define JINC_QUASI_EXACT 0
#include "kernel_quasi_impl.inc"
