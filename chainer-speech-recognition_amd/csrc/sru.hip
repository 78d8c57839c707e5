// Simple Recurrent Unit scan kernels for gfx950 -- the MI355X counterpart of the reference's only hand-written CUDA
// (asr/nn/sru.py:7-193: `forward` :17-73, `backward` :75-191).
//
//   f = sigmoid(U_f + b_f)   r = sigmoid(U_r + b_r)   c_t = f (c_{t-1} - z) + z   h_t = r (g(c_t) - x_t) + x_t
//   g = tanh or identity;  U = W x with W rows [z; f; r] (asr/nn/sru.py:295-296) comes from asr_gemm_nt.
//
// The reference keeps (B, D, T) arrays with time contiguous, one thread per (b, d) column: neighbouring threads are a
// whole sequence apart.  Here x / H / C are (T, B, D) and U is (T*B, 3D), so the 64 lanes of a wave read 64 adjacent
// features of one time step (coalesced) and the serial loop walks t.  Bias gradients are reduced in registers over t
// and added with one atomic per column (the reference stores (B, 2D, T) partials and sums them afterwards, :427).
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace sru {

__device__ __forceinline__ float sigm(float x) { return tanhf(x * 0.5f) * 0.5f + 0.5f; }   // asr/nn/sru.py:11-15

__global__ __launch_bounds__(256) void fwd_kernel(const uint16_t* __restrict__ x, const float* __restrict__ U,
                                                  const float* __restrict__ bias, const float* __restrict__ c0,
                                                  const float* __restrict__ mask, uint16_t* __restrict__ H,
                                                  float* __restrict__ C, float* __restrict__ cT, int T, int B, int D,
                                                  int use_tanh) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;     // (b, d)
    if (col >= B * D) return;
    const int d = col % D;
    const float bf = bias[d], br = bias[D + d];
    const float mk = mask ? mask[col] : 1.f;
    float c = c0 ? c0[col] : 0.f;
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * B * D + col;
        const float* u = U + ((size_t)t * B + col / D) * 3 * D + d;
        const float z = u[0];
        const float f = sigm(u[D] + bf);
        const float r = sigm(u[2 * D] + br);
        const float xt = bf16_to_f32(x[row]) * mk;
        c = f * (c - z) + z;
        C[row] = c;
        const float g = use_tanh ? tanhf(c) : c;
        H[row] = f32_to_bf16(r * (g - xt) + xt);
    }
    cT[col] = c;
}

__global__ __launch_bounds__(256) void bwd_kernel(const uint16_t* __restrict__ x, const float* __restrict__ U,
                                                  const float* __restrict__ bias, const float* __restrict__ C,
                                                  const float* __restrict__ c0, const float* __restrict__ mask,
                                                  const uint16_t* __restrict__ gH, const float* __restrict__ gcT,
                                                  uint16_t* __restrict__ gU, uint16_t* __restrict__ gxh,
                                                  float* __restrict__ gbias, float* __restrict__ gc0, int T, int B, int D,
                                                  int use_tanh) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= B * D) return;
    const int d = col % D, b = col / D;
    const float bf = bias[d], br = bias[D + d];
    const float mk = mask ? mask[col] : 1.f;
    const float cinit = c0 ? c0[col] : 0.f;
    float gc = gcT ? gcT[col] : 0.f;
    float sbf = 0.f, sbr = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)t * B * D + col;
        const size_t urow = ((size_t)t * B + b) * 3 * D + d;
        const float z = U[urow];
        const float f = sigm(U[urow + D] + bf);
        const float r = sigm(U[urow + 2 * D] + br);
        const float xt = bf16_to_f32(x[row]) * mk;
        const float gh = gH ? bf16_to_f32(gH[row]) : 0.f;
        const float c = C[row];
        const float cp = t == 0 ? cinit : C[row - (size_t)B * D];
        const float g = use_tanh ? tanhf(c) : c;
        const float gbr = gh * (g - xt) * (1.f - r) * r;                 // asr/nn/sru.py:158
        const float gtanh = use_tanh ? (1.f - g * g) : 1.f;
        const float gct = gh * r * gtanh;                                // :162
        const float gbf = (gct + gc) * (cp - z) * (1.f - f) * f;         // :163
        gxh[row] = f32_to_bf16(gh * (1.f - r));                          // :166
        gU[urow] = f32_to_bf16((gct + gc) * (1.f - f));                  // :169
        gU[urow + D] = f32_to_bf16(gbf);                                 // :170
        gU[urow + 2 * D] = f32_to_bf16(gbr);                             // :171
        gc = (gct + gc) * f;                                             // :174
        sbf += gbf;
        sbr += gbr;
    }
    gc0[col] = gc;
    atomicAdd(gbias + d, sbf);
    atomicAdd(gbias + D + d, sbr);
}

// out = (a + b) * mask[(b, d)]  on (T, B, D) bf16: highway gradient + projection gradient (asr/nn/sru.py:422-425)
__global__ void combine_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b2,
                               const float* __restrict__ mask, uint16_t* __restrict__ out, long long n, int BD) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = bf16_to_f32(a[i]) + bf16_to_f32(b2[i]);
        if (mask) v *= mask[i % BD];
        out[i] = f32_to_bf16(v);
    }
}
// x * mask (the reference multiplies X in place before the projection, asr/nn/sru.py:336-337)
__global__ void mask_kernel(const uint16_t* __restrict__ x, const float* __restrict__ mask, uint16_t* __restrict__ out,
                            long long n, int BD) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = f32_to_bf16(bf16_to_f32(x[i]) * mask[i % BD]);
}

}  // namespace sru
}  // namespace asr

using namespace asr;
using namespace asr::sru;

extern "C" int asr_sru_fwd(void* stream, const void* x_bf16, const float* U, const float* bias, const float* c0,
                           const float* mask, void* H_bf16, float* C, float* cT, int T, int B, int D, int use_tanh) {
    if (!x_bf16 || !U || !bias || !H_bf16 || !C || !cT || T <= 0 || B <= 0 || D <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(fwd_kernel, dim3(cdiv((long long)B * D, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x_bf16, U, bias, c0, mask, (uint16_t*)H_bf16, C, cT, T, B, D, use_tanh);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sru_bwd(void* stream, const void* x_bf16, const float* U, const float* bias, const float* C,
                           const float* c0, const float* mask, const void* gH_bf16, const float* gcT, void* gU_bf16,
                           void* gxh_bf16, float* gbias, float* gc0, int T, int B, int D, int use_tanh) {
    if (!x_bf16 || !U || !bias || !C || !gU_bf16 || !gxh_bf16 || !gbias || !gc0 || T <= 0 || B <= 0 || D <= 0)
        return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(bwd_kernel, dim3(cdiv((long long)B * D, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x_bf16, U, bias, C, c0, mask, (const uint16_t*)gH_bf16, gcT, (uint16_t*)gU_bf16,
                       (uint16_t*)gxh_bf16, gbias, gc0, T, B, D, use_tanh);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sru_combine(void* stream, const void* a_bf16, const void* b_bf16, const float* mask, void* out_bf16,
                               long long n, int BD) {
    if (!a_bf16 || !out_bf16 || n <= 0 || BD <= 0) return ASR_ERR_BAD_ARG;
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (b_bf16)
        hipLaunchKernelGGL(combine_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a_bf16,
                           (const uint16_t*)b_bf16, mask, (uint16_t*)out_bf16, n, BD);
    else {
        if (!mask) return ASR_ERR_BAD_ARG;
        hipLaunchKernelGGL(mask_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a_bf16, mask,
                           (uint16_t*)out_bf16, n, BD);
    }
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
