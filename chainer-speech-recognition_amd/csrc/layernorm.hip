// Layer normalisation over the contiguous (H, C) block of every (t, b) row  (gfx950).
//
// Replaces nn.LayerNormalization (asr/nn/nn.py:240-265) = NormalizeLayer (asr/nn/layernorm.py:29-64) followed by
// scale/bias along the channel axis.  Reference semantics kept: statistics over axes (1, 2) = (C, H) of the logical
// (B, C, H, T) array, biased variance, NO epsilon in the forward (asr/nn/layernorm.py:42-48 ignores `eps`).
//   y[r][h][c] = (x - mean_r) / std_r * gamma[c] + beta[c]
//   dx = (g - mean(g) - xhat * mean(g * xhat)) / std,  g = dy * gamma   (closed form of asr/nn/layernorm.py:50-61)
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace ln {

template <typename T>
__device__ __forceinline__ float ldf(const T* p, long long i) {
    if (sizeof(T) == 2) return bf16_to_f32((uint16_t)p[i]);
    return (float)p[i];
}
template <typename T>
__device__ __forceinline__ void stf(T* p, long long i, float v) {
    if (sizeof(T) == 2) p[i] = (T)f32_to_bf16(v); else p[i] = (T)v;
}

template <typename XT, typename YT>
__global__ __launch_bounds__(256) void fwd_kernel(const XT* __restrict__ x, YT* __restrict__ y,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out, int D,
                                                  int C) {
    __shared__ float scratch[32];
    const long long row = blockIdx.x;
    const XT* xr = x + row * D;
    YT* yr = y + row * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) s += ldf(xr, i);
    const float mean = block_sum(s, scratch) / (float)D;
    float v = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float d = ldf(xr, i) - mean;
        v += d * d;
    }
    const float var = block_sum(v, scratch) / (float)D;
    const float rstd = 1.0f / sqrtf(var);
    if (threadIdx.x == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const int c = i % C;
        stf(yr, i, (ldf(xr, i) - mean) * rstd * gamma[c] + beta[c]);
    }
}

template <typename XT, typename GT, typename DT>
__global__ __launch_bounds__(256) void bwd_kernel(const XT* __restrict__ x, const GT* __restrict__ dy,
                                                  const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                  const float* __restrict__ rstd_in, DT* __restrict__ dx, int D, int C) {
    __shared__ float scratch[32];
    const long long row = blockIdx.x;
    const XT* xr = x + row * D;
    const GT* gr = dy + row * D;
    DT* dr = dx + row * D;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float g = ldf(gr, i) * gamma[i % C];
        const float xh = (ldf(xr, i) - mean) * rstd;
        s1 += g;
        s2 += g * xh;
    }
    const float m1 = block_sum(s1, scratch) / (float)D;
    const float m2 = block_sum(s2, scratch) / (float)D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const float g = ldf(gr, i) * gamma[i % C];
        const float xh = (ldf(xr, i) - mean) * rstd;
        stf(dr, i, (g - m1 - xh * m2) * rstd);
    }
}

// dgamma[c] += sum_{r, h} dy * xhat ; dbeta[c] += sum dy.  grid.x: blocks of 64 inner positions, grid.y: row chunks.
template <typename XT, typename GT>
__global__ __launch_bounds__(256) void param_grad_kernel(const XT* __restrict__ x, const GT* __restrict__ dy,
                                                         const float* __restrict__ mean_in,
                                                         const float* __restrict__ rstd_in, long long rows, int D, int C,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float pg[4][64], pb[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    const long long chunk = (rows + gridDim.y - 1) / gridDim.y;
    const long long r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
    float sg = 0.f, sb = 0.f;
    if (i < D)
        for (long long r = r0 + w; r < r1; r += 4) {
            const float g = ldf(dy, r * D + i);
            sg += g * (ldf(x, r * D + i) - mean_in[r]) * rstd_in[r];
            sb += g;
        }
    pg[w][threadIdx.x & 63] = sg;
    pb[w][threadIdx.x & 63] = sb;
    __syncthreads();
    if (w == 0 && i < D) {
        const int t = threadIdx.x;
        atomicAdd(dgamma + (i % C), pg[0][t] + pg[1][t] + pg[2][t] + pg[3][t]);
        atomicAdd(dbeta + (i % C), pb[0][t] + pb[1][t] + pb[2][t] + pb[3][t]);
    }
}

}  // namespace ln
}  // namespace asr

using namespace asr;
using namespace asr::ln;

extern "C" int asr_layernorm_fwd(void* stream, const void* x, int x_bf16, void* y, int y_bf16, const float* gamma,
                                 const float* beta, float* mean, float* rstd, long long rows, int D, int C) {
    if (!x || !y || !gamma || !beta || !mean || !rstd || rows <= 0 || D <= 0 || C <= 0 || D % C) return ASR_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)rows), b(256);
    if (x_bf16 && y_bf16)
        hipLaunchKernelGGL((fwd_kernel<uint16_t, uint16_t>), g, b, 0, s, (const uint16_t*)x, (uint16_t*)y, gamma, beta, mean, rstd, D, C);
    else if (x_bf16)
        hipLaunchKernelGGL((fwd_kernel<uint16_t, float>), g, b, 0, s, (const uint16_t*)x, (float*)y, gamma, beta, mean, rstd, D, C);
    else if (y_bf16)
        hipLaunchKernelGGL((fwd_kernel<float, uint16_t>), g, b, 0, s, (const float*)x, (uint16_t*)y, gamma, beta, mean, rstd, D, C);
    else
        hipLaunchKernelGGL((fwd_kernel<float, float>), g, b, 0, s, (const float*)x, (float*)y, gamma, beta, mean, rstd, D, C);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

template <typename XT, typename GT>
static int launch_bwd(hipStream_t s, const void* x, const void* dy, const float* gamma, const float* mean,
                      const float* rstd, void* dx, int dx_bf16, float* dgamma, float* dbeta, long long rows, int D,
                      int C) {
    const dim3 g((unsigned)rows), b(256);
    if (dx) {
        if (dx_bf16)
            hipLaunchKernelGGL((bwd_kernel<XT, GT, uint16_t>), g, b, 0, s, (const XT*)x, (const GT*)dy, gamma, mean, rstd, (uint16_t*)dx, D, C);
        else
            hipLaunchKernelGGL((bwd_kernel<XT, GT, float>), g, b, 0, s, (const XT*)x, (const GT*)dy, gamma, mean, rstd, (float*)dx, D, C);
        ASR_LAUNCH_CHECK();
    }
    if (dgamma && dbeta) {
        int chunks = (int)((rows + 127) / 128);
        if (chunks > 64) chunks = 64;
        hipLaunchKernelGGL((param_grad_kernel<XT, GT>), dim3(cdiv(D, 64), chunks), dim3(256), 0, s, (const XT*)x,
                           (const GT*)dy, mean, rstd, rows, D, C, dgamma, dbeta);
        ASR_LAUNCH_CHECK();
    }
    return ASR_OK;
}

extern "C" int asr_layernorm_bwd(void* stream, const void* x, int x_bf16, const void* dy, int dy_bf16,
                                 const float* gamma, const float* mean, const float* rstd, void* dx, int dx_bf16,
                                 float* dgamma, float* dbeta, long long rows, int D, int C) {
    if (!x || !dy || !gamma || !mean || !rstd || rows <= 0 || D <= 0 || C <= 0 || D % C) return ASR_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (x_bf16 && dy_bf16) return launch_bwd<uint16_t, uint16_t>(s, x, dy, gamma, mean, rstd, dx, dx_bf16, dgamma, dbeta, rows, D, C);
    if (x_bf16) return launch_bwd<uint16_t, float>(s, x, dy, gamma, mean, rstd, dx, dx_bf16, dgamma, dbeta, rows, D, C);
    if (dy_bf16) return launch_bwd<float, uint16_t>(s, x, dy, gamma, mean, rstd, dx, dx_bf16, dgamma, dbeta, rows, D, C);
    return launch_bwd<float, float>(s, x, dy, gamma, mean, rstd, dx, dx_bf16, dgamma, dbeta, rows, D, C);
}
