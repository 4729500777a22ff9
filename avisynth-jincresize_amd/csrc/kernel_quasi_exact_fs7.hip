// kernel_quasi_exact_fs7.hip -- ewa_quasi_kernel, exactly periodic variant, filter size 7 (see kernel_quasi_impl.inc).
#define JINC_QUASI_FS 7
#define JINC_QUASI_EXACT 1
#include "kernel_quasi_impl.inc"
