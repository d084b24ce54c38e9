/* Developer-build-only entry points (HOLD_DEV=1 python -m hold_amd.build -> libholdhip_dev.so; hold_amd/csrc/dev/diag.hip).
 * NOT part of the product's C ABI: include/hold_hip.h declares what libholdhip.so exports, this header what only the
 * developer library adds. */
#ifndef HOLD_HIP_DEV_H
#define HOLD_HIP_DEV_H
#include "../../../include/hold_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* diagnostic: pure v_mfma_f32_32x32x2_f32 issue loop (blocks x 256 threads, iters x 64 MFMAs per wave);
 * out needs blocks*256 floats; random_operands != 0 feeds 32 pseudo-random operand values per lane (realistic
 * switching power).  Used only to calibrate the MFMA ceiling at the sustained clock. */
int hold_diag_mfma_peak(float* out, int32_t blocks, int32_t iters, int32_t random_operands, hold_stream_t stream);
/* diagnostic: same MFMA count with A operands read from LDS (mode 2) and B streamed from wsrc (mode 3; >= 64 Ki floats);
 * out needs blocks*512 floats */
int hold_diag_mfma_lds(float* out, const float* wsrc, int32_t blocks, int32_t iters, int32_t mode, hold_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif /* HOLD_HIP_DEV_H */
