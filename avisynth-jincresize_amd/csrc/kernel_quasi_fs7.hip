// kernel_quasi_fs7.hip -- ewa_quasi_kernel (drifting phases) for filter size 7 (see kernel_quasi_impl.inc).
#define JINC_QUASI_FS 7
#define JINC_QUASI_EXACT 0
#include "kernel_quasi_impl.inc"
