// placeholder -- replaced by the parallel decoder
#include "pjd_device_common.h"
#include "pjd_kernels.h"
void pjd_launch_build_tables(hipStream_t, const PjdDevBatch &) {}
void pjd_launch_huff_sync(hipStream_t, const PjdDevBatch &) {}
void pjd_launch_huff_fix(hipStream_t, const PjdDevBatch &) {}
void pjd_launch_huff_carry(hipStream_t, const PjdDevBatch &) {}
void pjd_launch_huff_write(hipStream_t, const PjdDevBatch &) {}
