// (Bi)GRU recurrence for gfx950: one launch per time step, both directions in the same launch.
//
// API surface replaced: nn.GRU / nn.NStepBiGRU, which reach the reference's `asr.nn` namespace through
// `from chainer.links import *` (asr/nn/nn.py:3).  Chainer's source is not under /root/reference and the reference never
// instantiates a GRU, so the gate convention is the cuDNN / torch.nn.GRU one:
//     r = sigmoid(gi_r + gh_r)   z = sigmoid(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)   h' = (1 - z) * n + z * h
//     gi = x W_ih^T + b_ih  (one big MFMA GEMM for all time steps, asr_gemm_nt)      gh = h W_hh^T + b_hh  (here)
//
// Layouts (rows are (t, b) pairs, time-major):
//     gi, dgi, dgh : [T*B][ndir*3H]   gate order r | z | n inside each direction
//     hseq (f32) and hseq16 (bf16 copy used as the MFMA operand of the next step): [T*B][ndir*H]
//     gates (saved for backward, f32): [T*B][ndir][4][H] = r | z | n | q   with q = gh_n (incl. bias)
//     whh  bf16 [ndir][3H][H]      whhT bf16 [ndir][H][3H]
// A workgroup owns 16 hidden units of one direction: 3 gate tiles x ceil(B/16) batch tiles of v_mfma_f32_16x16x32_bf16,
// the K range split over its 4 waves and reduced through LDS; operands come straight from L2 (k-contiguous, 16 B/lane).
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace gru {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

union Frag {
    bf16x8 v;
    uint4 u;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    const float t = (1.0f - e) / (1.0f + e);
    return copysignf(t, x);
}

constexpr int MT = 2;   // 16-row batch tiles per pass (32 utterances); larger batches loop

__global__ __launch_bounds__(256) void fwd_step_kernel(const float* __restrict__ gi, const uint16_t* __restrict__ whh,
                                                       const float* __restrict__ bhh, float* __restrict__ hseq,
                                                       uint16_t* __restrict__ hseq16, float* __restrict__ gates, int T,
                                                       int B, int H, int ndir, int s) {
    __shared__ __attribute__((aligned(16))) float4 part[4 * MT * 3 * 64];
    const int d = blockIdx.y, j0 = blockIdx.x * 16;
    const int t = d == 0 ? s : T - 1 - s;
    const int tp = d == 0 ? t - 1 : t + 1;
    const bool first = s == 0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nks = H >> 5;
    const size_t hs = (size_t)ndir * H;
    for (int b0 = 0; b0 < B; b0 += 16 * MT) {
        if (!first) {
            f32x4 acc[MT][3];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int ks = w; ks < nks; ks += 4) {
                const int k = ks * 32 + 8 * (lane >> 4);
                Frag a[MT], bb[3];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int row = b0 + m * 16 + (lane & 15);
                    a[m].u = row < B ? *reinterpret_cast<const uint4*>(hseq16 + ((size_t)tp * B + row) * hs + d * H + k)
                                     : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    bb[g].u = *reinterpret_cast<const uint4*>(whh + ((size_t)(d * 3 + g) * H + j0 + (lane & 15)) * H + k);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m].v, bb[g].v, acc[m][g], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    part[((w * MT + m) * 3 + g) * 64 + lane] = make_float4(acc[m][g][0], acc[m][g][1], acc[m][g][2], acc[m][g][3]);
        }
        __syncthreads();
        for (int idx = tid; idx < MT * 256; idx += 256) {
            const int bl = idx >> 4, j = idx & 15;
            const int b = b0 + bl;
            if (b >= B) continue;
            const int m = bl >> 4, row = bl & 15;
            const int pl = (row >> 2) * 16 + j, pr = row & 3;
            float gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float sum = bhh[(d * 3 + g) * H + j0 + j];
                if (!first) {
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        const float4 v = part[((ww * MT + m) * 3 + g) * 64 + pl];
                        sum += pr == 0 ? v.x : (pr == 1 ? v.y : (pr == 2 ? v.z : v.w));
                    }
                }
                gh[g] = sum;
            }
            const size_t rowi = (size_t)t * B + b;
            const float* gir = gi + rowi * (3 * hs) + (size_t)d * 3 * H + j0 + j;
            const float r = sigmoidf_(gir[0] + gh[0]);
            const float z = sigmoidf_(gir[H] + gh[1]);
            const float n = tanhf_(gir[2 * H] + r * gh[2]);
            const float hp = first ? 0.f : hseq[((size_t)tp * B + b) * hs + d * H + j0 + j];
            const float h = (1.0f - z) * n + z * hp;
            hseq[rowi * hs + d * H + j0 + j] = h;
            hseq16[rowi * hs + d * H + j0 + j] = f32_to_bf16(h);
            float* gs = gates + (rowi * ndir + d) * 4 * H + j0 + j;
            gs[0] = r; gs[H] = z; gs[2 * H] = n; gs[3 * H] = gh[2];
        }
        __syncthreads();
    }
}

// backward step: dh_t = dy_t + carry + dgh_{next} W_hh ; gate gradients ; carry <- dh_t * z_t
__global__ __launch_bounds__(256) void bwd_step_kernel(const uint16_t* __restrict__ dy, const float* __restrict__ gates,
                                                       const float* __restrict__ hseq,
                                                       const uint16_t* __restrict__ whhT, uint16_t* __restrict__ dgi,
                                                       uint16_t* __restrict__ dgh, float* __restrict__ carry, int T,
                                                       int B, int H, int ndir, int s) {
    __shared__ __attribute__((aligned(16))) float4 part[4 * MT * 64];
    const int d = blockIdx.y, j0 = blockIdx.x * 16;
    const int t = d == 0 ? T - 1 - s : s;          // reverse of the forward order
    const int tn = d == 0 ? t + 1 : t - 1;         // step processed just before in this sweep
    const int tp = d == 0 ? t - 1 : t + 1;         // forward-order predecessor (h_{prev})
    const bool first = s == 0;                     // nothing flows in from "next"
    const bool has_prev = d == 0 ? t > 0 : t < T - 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nks = (3 * H) >> 5;
    const size_t hs = (size_t)ndir * H, gs3 = (size_t)ndir * 3 * H;
    for (int b0 = 0; b0 < B; b0 += 16 * MT) {
        if (!first) {
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int ks = w; ks < nks; ks += 4) {
                const int k = ks * 32 + 8 * (lane >> 4);
                Frag a[MT], bb;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int row = b0 + m * 16 + (lane & 15);
                    a[m].u = row < B ? *reinterpret_cast<const uint4*>(dgh + ((size_t)tn * B + row) * gs3 + (size_t)d * 3 * H + k)
                                     : make_uint4(0, 0, 0, 0);
                }
                bb.u = *reinterpret_cast<const uint4*>(whhT + ((size_t)d * H + j0 + (lane & 15)) * (3 * H) + k);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m].v, bb.v, acc[m], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
                part[(w * MT + m) * 64 + lane] = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        }
        __syncthreads();
        for (int idx = tid; idx < MT * 256; idx += 256) {
            const int bl = idx >> 4, j = idx & 15;
            const int b = b0 + bl;
            if (b >= B) continue;
            const int m = bl >> 4, row = bl & 15;
            const int pl = (row >> 2) * 16 + j, pr = row & 3;
            const size_t rowi = (size_t)t * B + b;
            float dh = bf16_to_f32(dy[rowi * H + j0 + j]);
            float* cp = carry + ((size_t)d * B + b) * H + j0 + j;
            if (!first) {
                dh += *cp;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float4 v = part[(ww * MT + m) * 64 + pl];
                    dh += pr == 0 ? v.x : (pr == 1 ? v.y : (pr == 2 ? v.z : v.w));
                }
            }
            const float* gs = gates + (rowi * ndir + d) * 4 * H + j0 + j;
            const float r = gs[0], z = gs[H], n = gs[2 * H], q = gs[3 * H];
            const float hp = has_prev ? hseq[((size_t)tp * B + b) * hs + d * H + j0 + j] : 0.f;
            const float dn = dh * (1.0f - z);
            const float dz = dh * (hp - n);
            const float dan = dn * (1.0f - n * n);
            const float daz = dz * z * (1.0f - z);
            const float dq = dan * r;
            const float dar = dan * q * r * (1.0f - r);
            *cp = dh * z;
            uint16_t* gi_o = dgi + rowi * gs3 + (size_t)d * 3 * H + j0 + j;
            uint16_t* gh_o = dgh + rowi * gs3 + (size_t)d * 3 * H + j0 + j;
            const uint16_t ar = f32_to_bf16(dar), az = f32_to_bf16(daz);
            gi_o[0] = ar; gi_o[H] = az; gi_o[2 * H] = f32_to_bf16(dan);
            gh_o[0] = ar; gh_o[H] = az; gh_o[2 * H] = f32_to_bf16(dq);
        }
        __syncthreads();
    }
}

// y = hf + hb (or a copy for one direction): f32 state -> bf16 layer output
__global__ void merge_dirs_kernel(const float* __restrict__ hseq, uint16_t* __restrict__ y, long long rows, int H,
                                  int ndir) {
    const long long n = rows * H;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / H;
        const int j = (int)(i - r * H);
        float v = hseq[r * ndir * H + j];
        if (ndir == 2) v += hseq[r * ndir * H + H + j];
        y[i] = f32_to_bf16(v);
    }
}

}  // namespace gru
}  // namespace asr

using namespace asr;
using namespace asr::gru;

static int check_dims(int T, int B, int H, int ndir) {
    if (T <= 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return ASR_ERR_BAD_ARG;
    if (H % 32) return ASR_ERR_UNSUPPORTED;       // MFMA K step and 16-unit workgroup slices
    return ASR_OK;
}

extern "C" int asr_gru_fwd(void* stream, const float* gi, const void* whh_bf16, const float* bhh, float* hseq,
                           void* hseq_bf16, float* gates, void* y_bf16, int T, int B, int H, int ndir) {
    if (!gi || !whh_bf16 || !bhh || !hseq || !hseq_bf16 || !gates) return ASR_ERR_BAD_ARG;
    const int rc = check_dims(T, B, H, ndir);
    if (rc != ASR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(H / 16, ndir), block(256);
    for (int s = 0; s < T; ++s)
        hipLaunchKernelGGL(fwd_step_kernel, grid, block, 0, st, gi, (const uint16_t*)whh_bf16, bhh, hseq,
                           (uint16_t*)hseq_bf16, gates, T, B, H, ndir, s);
    ASR_LAUNCH_CHECK();
    if (y_bf16) {
        const long long n = (long long)T * B * H;
        long long g = (n + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(merge_dirs_kernel, dim3((unsigned)g), dim3(256), 0, st, hseq, (uint16_t*)y_bf16,
                           (long long)T * B, H, ndir);
        ASR_LAUNCH_CHECK();
    }
    return ASR_OK;
}

extern "C" int asr_gru_bwd(void* stream, const void* dy_bf16, const float* gates, const float* hseq,
                           const void* whhT_bf16, void* dgi_bf16, void* dgh_bf16, float* carry_ws, int T, int B, int H,
                           int ndir) {
    if (!dy_bf16 || !gates || !hseq || !whhT_bf16 || !dgi_bf16 || !dgh_bf16 || !carry_ws) return ASR_ERR_BAD_ARG;
    const int rc = check_dims(T, B, H, ndir);
    if (rc != ASR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(H / 16, ndir), block(256);
    for (int s = 0; s < T; ++s)
        hipLaunchKernelGGL(bwd_step_kernel, grid, block, 0, st, (const uint16_t*)dy_bf16, gates, hseq,
                           (const uint16_t*)whhT_bf16, (uint16_t*)dgi_bf16, (uint16_t*)dgh_bf16, carry_ws, T, B, H, ndir, s);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
