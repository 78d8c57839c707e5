#include "common.hpp"
#include "../../include/asr_hip.h"
extern "C" int asr_version(void) { return 1; }
extern "C" int asr_act_dtype(void) { return ASR_ACT_IS_F16; }

// One wave that does nothing for `microseconds` (wall clock, 100 MHz): a timed gap on a stream (experiments).  Bounded (<= 100 ms).
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

// Stand-in for a RESIDENT COLLECTIVE (RCCL's ring kernels: a few dozen workgroups that copy and reduce between device buffers for as long as
// their peers take): `blocks` workgroups of 256 threads, each holding `lds_bytes` of LDS, stream buf -> buf + bytes / 2 (16 B per lane,
// read + add + write) round after round until `microseconds` have passed.  One-GPU measurement of what such a kernel does to a persistent
// recurrence launched beside it (tools/gru_beside_collective.py); nothing on the train path calls it.
__global__ __launch_bounds__(256) void traffic_kernel(float4* __restrict__ buf, size_t n16, unsigned long long ticks) {
    extern __shared__ unsigned char hog[];
    if (threadIdx.x == 0) hog[0] = 1;
    const size_t half = n16 / 2;
    const unsigned long long t0 = wall_clock64();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            if (i >= half) i -= half * (i / half);
            const float4 a = buf[i], b = buf[i + half];
            buf[i + half] = make_float4(a.x + b.x * 0.5f, a.y + b.y * 0.5f, a.z + b.z * 0.5f, a.w + b.w * 0.5f);
            i += (size_t)gridDim.x * blockDim.x;
        }
    }
}
extern "C" int asr_stream_traffic(void* stream, int microseconds, int lds_bytes, int blocks, void* buf, long long bytes) {
    if (microseconds <= 0 || blocks <= 0 || lds_bytes < 0 || lds_bytes > 160 * 1024 || !buf || bytes < 4096 || (((uintptr_t)buf) & 15)) return ASR_ERR_BAD_ARG;
    if (microseconds > 200000) microseconds = 200000;
    if (lds_bytes > 64 * 1024 &&
        hipFuncSetAttribute((const void*)traffic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return ASR_ERR_LAUNCH;
    hipLaunchKernelGGL(traffic_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, (float4*)buf, (size_t)(bytes / 16),
                       (unsigned long long)microseconds * 100ull);
    return hipGetLastError() == hipSuccess ? ASR_OK : ASR_ERR_LAUNCH;
}
