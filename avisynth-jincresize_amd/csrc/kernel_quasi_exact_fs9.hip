// kernel_quasi_exact_fs9.hip -- ewa_quasi_kernel, exactly periodic variant, filter size 9 (see kernel_quasi_impl.inc).
#define JINC_QUASI_FS 9
#define JINC_QUASI_EXACT 1
#include "kernel_quasi_impl.inc"
