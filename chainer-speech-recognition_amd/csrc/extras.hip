// The function layers of the reference's asr.nn that no recipe uses but its API offers (asr/nn/nn.py:18-23 CReLU, :42-43 LogSoftmax,
// :58-63 Softmax, :77-93 AveragePooling2D / ND, :105-113 MaxPoolingND, :123-133 Unpooling2D, :220-231 GaussianNoise): thin wrappers
// over Chainer functions there, small HBM-bound kernels here.  All on the path's physical layout -- rows of C contiguous channels
// (T, B, H, C) bf16 -- so "axis 1" of the reference's (B, C, H, T) is the contiguous one: a softmax row is one wave's coalesced read.
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace extras {

constexpr int kThreads = 256;
static inline int grid_for(long long n) {
    long long g = (n + kThreads - 1) / kThreads;
    if (g > 8192) g = 8192;
    return g < 1 ? 1 : (int)g;
}

// chainer.functions.crelu(x, axis=1): concat(relu(x), relu(-x)) along the channels
__global__ void crelu_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long rows, int C) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float v = bf16_to_f32(x[i]);
        y[r * 2 * C + c] = f32_to_bf16(fmaxf(v, 0.f));
        y[r * 2 * C + C + c] = f32_to_bf16(fmaxf(-v, 0.f));
    }
}
__global__ void crelu_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, long long rows,
                                 int C) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float v = bf16_to_f32(x[i]);
        const float gp = bf16_to_f32(dy[r * 2 * C + c]), gn = bf16_to_f32(dy[r * 2 * C + C + c]);
        dx[i] = f32_to_bf16(v > 0.f ? gp : (v < 0.f ? -gn : 0.f));
    }
}

// softmax / log_softmax over the C channels of a row: one wave per row
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long rows, int C, int logform) {
    const int lane = threadIdx.x & 63;
    for (long long r = blockIdx.x * 4LL + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const uint16_t* xr = x + r * C;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, bf16_to_f32(xr[c]));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += __expf(bf16_to_f32(xr[c]) - m);
        s = wave_sum(s);
        const float ls = __logf(s), inv = 1.0f / s;
        for (int c = lane; c < C; c += 64) {
            const float d = bf16_to_f32(xr[c]) - m;
            y[r * C + c] = f32_to_bf16(logform ? d - ls : __expf(d) * inv);
        }
    }
}
// softmax: dx = y (dy - sum(dy y));  log_softmax: dx = dy - exp(y) sum(dy)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const uint16_t* __restrict__ y, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
                                                          long long rows, int C, int logform) {
    const int lane = threadIdx.x & 63;
    for (long long r = blockIdx.x * 4LL + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const uint16_t* yr = y + r * C;
        const uint16_t* gr = dy + r * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += logform ? bf16_to_f32(gr[c]) : bf16_to_f32(gr[c]) * bf16_to_f32(yr[c]);
        s = wave_sum(s);
        for (int c = lane; c < C; c += 64) {
            const float yy = bf16_to_f32(yr[c]), g = bf16_to_f32(gr[c]);
            dx[r * C + c] = f32_to_bf16(logform ? g - __expf(yy) * s : yy * (g - s));
        }
    }
}

// chainer.functions.average_pooling_2d(x, (k, 1), stride (k, 1), pad 0): cover_all is False there, every window is whole
__global__ void avgpool_h_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long R, int Hin, int Hout, int C, int k) {
    const long long n = R * Hout * C;
    const float inv = 1.0f / (float)k;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ho = (int)((i / C) % Hout);
        const long long r = i / ((long long)C * Hout);
        const uint16_t* src = x + (r * Hin + (long long)ho * k) * C + c;
        float s = 0.f;
        for (int j = 0; j < k; ++j) s += bf16_to_f32(src[(long long)j * C]);
        y[i] = f32_to_bf16(s * inv);
    }
}
__global__ void avgpool_h_bwd_kernel(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, long long R, int Hin, int Hout, int C, int k) {
    const long long n = R * Hin * C;
    const float inv = 1.0f / (float)k;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int h = (int)((i / C) % Hin);
        const long long r = i / ((long long)C * Hin);
        const int ho = h / k;
        dx[i] = ho < Hout ? f32_to_bf16(bf16_to_f32(dy[(r * Hout + ho) * C + c]) * inv) : (uint16_t)0;
    }
}

// chainer.functions.unpooling_2d(x, (k, 1), stride (k, 1), pad 0): every input row is repeated over its k output rows; Hout =
// k (Hin - 1) + 1 with cover_all (the last window is cut to one row), k Hin without
__global__ void unpool_h_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long R, int Hin, int Hout, int C, int k) {
    const long long n = R * Hout * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ho = (int)((i / C) % Hout);
        const long long r = i / ((long long)C * Hout);
        y[i] = x[(r * Hin + ho / k) * C + c];
    }
}
__global__ void unpool_h_bwd_kernel(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, long long R, int Hin, int Hout, int C, int k) {
    const long long n = R * Hin * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int h = (int)((i / C) % Hin);
        const long long r = i / ((long long)C * Hin);
        float s = 0.f;
        for (int j = 0; j < k; ++j) {
            const int ho = h * k + j;
            if (ho < Hout) s += bf16_to_f32(dy[(r * Hout + ho) * C + c]);
        }
        dx[i] = f32_to_bf16(s);
    }
}

// x + N(0, std^2): counter-based (seed, element index) Box-Muller, one normal per element (asr/nn/nn.py:220-231: the reference's
// `mean` argument is never used there either)
__device__ __forceinline__ uint32_t hash32(uint32_t v) {
    v ^= v >> 16; v *= 0x7feb352dU; v ^= v >> 15; v *= 0x846ca68bU; v ^= v >> 16;
    return v;
}
__global__ void gaussian_noise_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long n, float stdv, uint32_t seed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t a = hash32((uint32_t)i ^ seed), b = hash32((uint32_t)(i >> 32) + 0x9e3779b9U + a);
        const float u1 = ((a >> 8) + 1) * (1.0f / 16777217.0f), u2 = (b >> 8) * (1.0f / 16777216.0f);     // u1 in (0, 1]
        const float g = sqrtf(-2.0f * __logf(u1)) * __cosf(6.2831853f * u2);
        y[i] = f32_to_bf16(bf16_to_f32(x[i]) + stdv * g);
    }
}

}  // namespace extras
}  // namespace asr

using namespace asr;
using namespace asr::extras;

extern "C" int asr_crelu_fwd(void* stream, const void* x, void* y, long long rows, int C) {
    if (!x || !y || rows <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(crelu_fwd_kernel, dim3(grid_for(rows * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, rows, C);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_crelu_bwd(void* stream, const void* x, const void* dy, void* dx, long long rows, int C) {
    if (!x || !dy || !dx || rows <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(crelu_bwd_kernel, dim3(grid_for(rows * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)dy,
                       (uint16_t*)dx, rows, C);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_softmax_fwd(void* stream, const void* x, void* y, long long rows, int C, int log_form) {
    if (!x || !y || rows <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    long long g = (rows + 3) / 4;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, rows, C, log_form);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_softmax_bwd(void* stream, const void* y, const void* dy, void* dx, long long rows, int C, int log_form) {
    if (!y || !dy || !dx || rows <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    long long g = (rows + 3) / 4;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)y, (const uint16_t*)dy, (uint16_t*)dx,
                       rows, C, log_form);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_avgpool_h_fwd(void* stream, const void* x, void* y, long long R, int Hin, int C, int k) {
    if (!x || !y || R <= 0 || Hin <= 0 || C <= 0 || k <= 0 || Hin < k) return ASR_ERR_BAD_ARG;
    const int Hout = (Hin - k) / k + 1;
    hipLaunchKernelGGL(avgpool_h_fwd_kernel, dim3(grid_for(R * Hout * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, R,
                       Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_avgpool_h_bwd(void* stream, const void* dy, void* dx, long long R, int Hin, int C, int k) {
    if (!dy || !dx || R <= 0 || Hin <= 0 || C <= 0 || k <= 0 || Hin < k) return ASR_ERR_BAD_ARG;
    const int Hout = (Hin - k) / k + 1;
    hipLaunchKernelGGL(avgpool_h_bwd_kernel, dim3(grid_for(R * Hin * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)dy, (uint16_t*)dx, R,
                       Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_unpool_h_fwd(void* stream, const void* x, void* y, long long R, int Hin, int Hout, int C, int k) {
    if (!x || !y || R <= 0 || Hin <= 0 || C <= 0 || k <= 0 || Hout <= 0 || Hout > Hin * k) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(unpool_h_fwd_kernel, dim3(grid_for(R * Hout * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, R,
                       Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_unpool_h_bwd(void* stream, const void* dy, void* dx, long long R, int Hin, int Hout, int C, int k) {
    if (!dy || !dx || R <= 0 || Hin <= 0 || C <= 0 || k <= 0 || Hout <= 0 || Hout > Hin * k) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(unpool_h_bwd_kernel, dim3(grid_for(R * Hin * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)dy, (uint16_t*)dx, R,
                       Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_gaussian_noise(void* stream, const void* x, void* y, long long n, float stdv, unsigned int seed) {
    if (!x || !y || n <= 0 || !(stdv >= 0.f)) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(gaussian_noise_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n, stdv, seed);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
