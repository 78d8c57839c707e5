#include "common.hpp"
#include "../../include/asr_hip.h"
extern "C" int asr_version(void) { return 1; }

// One wave that does nothing for `microseconds` (wall clock, 100 MHz): staggers the two half batches of asr/pipeline.py by
// less than a recurrence, so that the projections of one half fall into the recurrence of the other.  Bounded (<= 100 ms).
__global__ void stream_delay_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int asr_stream_delay(void* stream, int microseconds) {
    if (microseconds <= 0) return ASR_OK;
    if (microseconds > 100000) microseconds = 100000;
    hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ERR_LAUNCH;
}

// `blocks` workgroups that each hold `lds_bytes` of LDS and idle for `microseconds`: makes CUs temporarily unavailable to
// whatever is launched behind it on another stream (tests of the persistent GRU launches' behaviour when they cannot be
// fully resident at once; diagnostic only, nothing on the train path calls it).  Bounded (<= 200 ms).
__global__ void occupy_kernel(unsigned long long ticks) {
    extern __shared__ unsigned char hog[];
    if (threadIdx.x == 0) hog[0] = 1;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int asr_occupy_cus(void* stream, int microseconds, int lds_bytes, int blocks) {
    if (microseconds <= 0 || blocks <= 0 || lds_bytes < 0 || lds_bytes > 160 * 1024) return ASR_ERR_BAD_ARG;
    if (microseconds > 200000) microseconds = 200000;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return ASR_ERR_LAUNCH;
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(64), (size_t)lds_bytes, (hipStream_t)stream,
                       (unsigned long long)microseconds * 100ull);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ERR_LAUNCH;
}
