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
