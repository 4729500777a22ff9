// kernel_quasi_fs7.hip -- ewa_quasi_kernel instantiated for filter size 7 (see kernel_quasi_impl.inc).
#define JINC_QUASI_FS 7
#include "kernel_quasi_impl.inc"
