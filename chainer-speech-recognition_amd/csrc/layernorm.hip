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

// A row of zero variance has rstd = inf (the reference divides by std without an epsilon, asr/nn/layernorm.py:42-48: its forward
// output is NaN there, and so is ours).  Such rows occur legitimately: frames beyond an utterance's length are all-zero after a
// length-aware recurrent layer with zero biases.  The backward kernels give them no gradient and take none from them
// (xhat := 0, dx := 0) instead of spreading 0 * NaN into the parameter gradients.
__device__ __forceinline__ float usable_rstd(float r) { return r < INFINITY ? r : 0.f; }

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
    const float mean = mean_in[row], rstd = usable_rstd(rstd_in[row]);
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
            sg += g * (ldf(x, r * D + i) - mean_in[r]) * usable_rstd(rstd_in[r]);
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


// float4 variants for f32 rows with D % 4 == 0 and C % 4 == 0 (the logits path: D = C = V)
__global__ __launch_bounds__(256) void fwd_f32x4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out, int D,
                                                        int C) {
    __shared__ float scratch[32];
    const long long row = blockIdx.x;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    float4* yr = reinterpret_cast<float4*>(y + row * D);
    const int n4 = D >> 2;
    float s = 0.f;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) { const float4 v = xr[i]; s += (v.x + v.y) + (v.z + v.w); }
    const float mean = block_sum(s, scratch) / (float)D;
    float q = 0.f;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = xr[i];
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, scratch) / (float)D);
    if (threadIdx.x == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = xr[i];
        const int c = (i * 4) % C;
        const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
        yr[i] = make_float4((v.x - mean) * rstd * g.x + b.x, (v.y - mean) * rstd * g.y + b.y, (v.z - mean) * rstd * g.z + b.z,
                            (v.w - mean) * rstd * g.w + b.w);
    }
}
// The logits path again, one WAVE per row and the row in registers (D <= 256 NV floats): one read of x, the two statistics and --
// optionally -- the row's log-sum-exp of y by wave shuffles, no workgroup barrier, one write of y.  gamma / beta stay in registers
// over the rows a wave walks (C == D: a lane always owns the same columns).  lse (may be null): what a CTC-family loss on y needs
// first (asr_ctc_forward_lse), formed here while y is in registers instead of by two more passes over the 384 MB of logits.
// Replaces three passes over x behind three barriers (fwd_f32x4_kernel: 0.22 ms and 1.07 GB of traffic on (32000, 3000)).
template <int NV>
__global__ __launch_bounds__(256) void fwd_rows_f32_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out, float* __restrict__ lse_out, long long rows, int D) {
    const int n4 = D >> 2, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float4 g[NV], b[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = lane + 64 * k;
        g[k] = i < n4 ? *reinterpret_cast<const float4*>(gamma + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        b[k] = i < n4 ? *reinterpret_cast<const float4*>(beta + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_d = 1.0f / (float)D;
    for (long long row = (long long)blockIdx.x * 4 + wid; row < rows; row += (long long)gridDim.x * 4) {
        const float4* xr = reinterpret_cast<const float4*>(x + row * D);
        float4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            v[k] = i < n4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (lane + 64 * k < n4) {
                const float a0 = v[k].x - mean, a1 = v[k].y - mean, a2 = v[k].z - mean, a3 = v[k].w - mean;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d);
        if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
        float4* yr = reinterpret_cast<float4*>(y + row * D);
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = lane + 64 * k;
            if (i < n4) {
                v[k] = make_float4((v[k].x - mean) * rstd * g[k].x + b[k].x, (v[k].y - mean) * rstd * g[k].y + b[k].y,
                                   (v[k].z - mean) * rstd * g[k].z + b[k].z, (v[k].w - mean) * rstd * g[k].w + b[k].w);
                yr[i] = v[k];
                m = fmaxf(m, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
            }
        }
        if (lse_out) {
            m = wave_max(m);
            float e = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (lane + 64 * k < n4) e += (__expf(v[k].x - m) + __expf(v[k].y - m)) + (__expf(v[k].z - m) + __expf(v[k].w - m));
            e = wave_sum(e);
            if (lane == 0) lse_out[row] = m + __logf(e);
        }
    }
}

__global__ __launch_bounds__(256) void bwd_f32x4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                        const float* __restrict__ rstd_in, float* __restrict__ dx, int D,
                                                        int C) {
    __shared__ float scratch[32];
    const long long row = blockIdx.x;
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    const float4* gr = reinterpret_cast<const float4*>(dy + row * D);
    float4* dr = reinterpret_cast<float4*>(dx + row * D);
    const int n4 = D >> 2;
    const float mean = mean_in[row], rstd = usable_rstd(rstd_in[row]);
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = xr[i], gy = gr[i];
        const float4 gm = *reinterpret_cast<const float4*>(gamma + (i * 4) % C);
        const float g0 = gy.x * gm.x, g1 = gy.y * gm.y, g2 = gy.z * gm.z, g3 = gy.w * gm.w;
        s1 += (g0 + g1) + (g2 + g3);
        s2 += (g0 * (v.x - mean) + g1 * (v.y - mean) + g2 * (v.z - mean) + g3 * (v.w - mean)) * rstd;
    }
    const float m1 = block_sum(s1, scratch) / (float)D;
    const float m2 = block_sum(s2, scratch) / (float)D;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = xr[i], gy = gr[i];
        const float4 gm = *reinterpret_cast<const float4*>(gamma + (i * 4) % C);
        dr[i] = make_float4((gy.x * gm.x - m1 - (v.x - mean) * rstd * m2) * rstd, (gy.y * gm.y - m1 - (v.y - mean) * rstd * m2) * rstd,
                            (gy.z * gm.z - m1 - (v.z - mean) * rstd * m2) * rstd, (gy.w * gm.w - m1 - (v.w - mean) * rstd * m2) * rstd);
    }
}

// Backward of the logits normalisation in ONE sweep: float32 x and dy (D % 4 == 0, D <= 4096), dx in bf16 or f32, and the
// parameter-gradient column sums carried in registers over the rows of a workgroup (a thread always owns the same
// columns) -> partial[workgroup][2][D], folded into dgamma / dbeta by fold_partials_kernel.  Replaces bwd_f32x4 +
// param_grad (+ the bf16 cast of dx in the layer in front): 1.9 GB of traffic -> 0.96 GB on the (32000, 3000) logits.
// The next row's loads are issued before the current row's reduction; one barrier per row (parity-buffered scratch).
template <typename DT, int NV>
__global__ __launch_bounds__(256) void bwd_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                           const float* __restrict__ rstd_in, DT* __restrict__ dx,
                                                           float* __restrict__ partial, long long rows, int D, int C) {
    __shared__ float red[2][2][4];
    const int n4 = D >> 2, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float4 gm[NV], ag[NV], ab[NV], v[NV], gy[NV], vn[NV], gn[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + 256 * k;
        gm[k] = i < n4 ? *reinterpret_cast<const float4*>(gamma + (i * 4) % C) : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        vn[k] = gn[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    long long row = blockIdx.x;
    if (row < rows) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k;
            if (i < n4) {
                vn[k] = reinterpret_cast<const float4*>(x + row * D)[i];
                gn[k] = reinterpret_cast<const float4*>(dy + row * D)[i];
            }
        }
    }
    int par = 0;
    const float invD = 1.0f / (float)D;
    for (; row < rows; row += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { v[k] = vn[k]; gy[k] = gn[k]; }
        const float mean = mean_in[row], rstd = usable_rstd(rstd_in[row]);
        const long long nxt = row + gridDim.x;
        if (nxt < rows) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = tid + 256 * k;
                if (i < n4) {
                    vn[k] = reinterpret_cast<const float4*>(x + nxt * D)[i];
                    gn[k] = reinterpret_cast<const float4*>(dy + nxt * D)[i];
                }
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {      // lanes beyond the row hold zeros (gamma = 0, dy = 0)
            v[k] = make_float4((v[k].x - mean) * rstd, (v[k].y - mean) * rstd, (v[k].z - mean) * rstd, (v[k].w - mean) * rstd);
            const float g0 = gy[k].x * gm[k].x, g1 = gy[k].y * gm[k].y, g2 = gy[k].z * gm[k].z, g3 = gy[k].w * gm[k].w;
            s1 += (g0 + g1) + (g2 + g3);
            s2 += (g0 * v[k].x + g1 * v[k].y) + (g2 * v[k].z + g3 * v[k].w);
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[par][0][wid] = s1; red[par][1][wid] = s2; }
        __syncthreads();
        const float m1 = ((red[par][0][0] + red[par][0][1]) + (red[par][0][2] + red[par][0][3])) * invD;
        const float m2 = ((red[par][1][0] + red[par][1][1]) + (red[par][1][2] + red[par][1][3])) * invD;
        par ^= 1;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k;
            if (i < n4) {
                const float d0 = (gy[k].x * gm[k].x - m1 - v[k].x * m2) * rstd, d1 = (gy[k].y * gm[k].y - m1 - v[k].y * m2) * rstd;
                const float d2 = (gy[k].z * gm[k].z - m1 - v[k].z * m2) * rstd, d3 = (gy[k].w * gm[k].w - m1 - v[k].w * m2) * rstd;
                if (dx) {
                    if (sizeof(DT) == 2) {
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(d0) | ((unsigned)f32_to_bf16(d1) << 16);
                        o.y = (unsigned)f32_to_bf16(d2) | ((unsigned)f32_to_bf16(d3) << 16);
                        reinterpret_cast<uint2*>(dx + row * D)[i] = o;
                    } else {
                        reinterpret_cast<float4*>(dx + row * D)[i] = make_float4(d0, d1, d2, d3);
                    }
                }
                ag[k].x += gy[k].x * v[k].x; ag[k].y += gy[k].y * v[k].y; ag[k].z += gy[k].z * v[k].z; ag[k].w += gy[k].w * v[k].w;
                ab[k].x += gy[k].x; ab[k].y += gy[k].y; ab[k].z += gy[k].z; ab[k].w += gy[k].w;
            }
        }
    }
    if (partial) {
        float* pg = partial + (size_t)blockIdx.x * 2 * D;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k;
            if (i < n4) {
                reinterpret_cast<float4*>(pg)[i] = ag[k];
                reinterpret_cast<float4*>(pg + D)[i] = ab[k];
            }
        }
    }
}
// dgamma[c] += sum over workgroups and positions of partial[.][0][i], i % C == c; dbeta likewise from partial[.][1][i]
// (planes = 3: a third plane of plain column sums goes to `extra` -- the bias gradient of the layer in front, csrc/ctc_ln.hip)
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ partial, int G, int D, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int planes,
                                                            float* __restrict__ extra) {
    __shared__ float acc[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const int chunk = (G + gridDim.y - 1) / gridDim.y;
    const int g0 = blockIdx.y * chunk, g1 = min(G, g0 + chunk);
    float s = 0.f;
    if (col < planes * D)
        for (int g = g0 + w; g < g1; g += 4) s += partial[(size_t)g * planes * D + col];
    acc[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && col < planes * D) {
        const int t = threadIdx.x;
        const float total = (acc[0][t] + acc[1][t]) + (acc[2][t] + acc[3][t]);
        if (col < D) atomicAdd(dgamma + col % C, total);
        else if (col < 2 * D) atomicAdd(dbeta + (col - D) % C, total);
        else atomicAdd(extra + (col - 2 * D), total);
    }
}


// ------------------------------------------------------------------------------------------------ batch normalisation
// chainer.links.BatchNormalization reaches the reference API through `from chainer.links import *` (asr/nn/nn.py:3).
// Statistics per channel over every other axis; in the (T, B, H, C) layout that is a column reduction over R = T*B*H rows.
//   reduce : mode 0: out0[c] += sum_r x, out1[c] += sum_r x^2                      (float64 accumulators)
//            mode 1: out0[c] += sum_r gy, out1[c] += sum_r gy * (x - mean[c]) * rstd[c]
//   finish : mean = s/R, var = q/R - mean^2 (biased, the normalising one); rstd = 1/sqrt(var + eps); running averages
//            avg = decay*avg + (1-decay)*batch with the UNBIASED variance var*R/(R-1) (Chainer's update rule)
//   fwd    : y = gamma (x - mean) rstd + beta
//   bwd    : dx = gamma rstd (gy - sgy/R - xhat sgx/R); dgamma += sgx, dbeta += sgy
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gy,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        long long R, int C, int rows_per_block, double* __restrict__ out0,
                                                        double* __restrict__ out1) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;                 // a block covers few rows: float partials, float64 across blocks
        const float mu = MODE ? mean[c] : 0.f, rs = MODE ? rstd[c] : 0.f;
        for (long long r = r0; r < r1; ++r) {
            const float xv = bf16_to_f32(x[r * C + c]);
            if (MODE == 0) { a += xv; b += xv * xv; }
            else { const float g = bf16_to_f32(gy[r * C + c]); a += g; b += g * (xv - mu) * rs; }
        }
        atomicAdd(out0 + c, (double)a);
        atomicAdd(out1 + c, (double)b);
    }
}

__global__ void bn_finish_kernel(const double* __restrict__ s, const double* __restrict__ q, long long R, int C, float eps,
                                 float decay, float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ avg_mean,
                                 float* __restrict__ avg_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = s[c] / (double)R;
    double var = q[c] / (double)R - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (avg_mean) {
        const double unbiased = R > 1 ? var * (double)R / (double)(R - 1) : var;
        avg_mean[c] = decay * avg_mean[c] + (1.f - decay) * (float)mu;
        avg_var[c] = decay * avg_var[c] + (1.f - decay) * (float)unbiased;
    }
}

__global__ void bn_fwd_kernel(const uint16_t* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                              const float* __restrict__ gamma, const float* __restrict__ beta, long long n, int C,
                              uint16_t* __restrict__ y) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        y[i] = f32_to_bf16(gamma[c] * (bf16_to_f32(x[i]) - mean[c]) * rstd[c] + beta[c]);
    }
}

__global__ void bn_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gy, const float* __restrict__ mean,
                              const float* __restrict__ rstd, const float* __restrict__ gamma, const double* __restrict__ sgy,
                              const double* __restrict__ sgx, long long R, int C, uint16_t* __restrict__ dx) {
    const long long n = R * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float xhat = (bf16_to_f32(x[i]) - mean[c]) * rstd[c];
        const float a = (float)(sgy[c] / (double)R), b = (float)(sgx[c] / (double)R);
        dx[i] = f32_to_bf16(gamma[c] * rstd[c] * (bf16_to_f32(gy[i]) - a - xhat * b));
    }
}

__global__ void bn_param_grad_kernel(const double* __restrict__ sgy, const double* __restrict__ sgx, int C,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    atomicAdd(dgamma + c, (float)sgx[c]);
    atomicAdd(dbeta + c, (float)sgy[c]);
}

}  // namespace ln
}  // namespace asr

using namespace asr;
using namespace asr::ln;

// rows in registers: float32 in and out, one parameter per column (C == D), whole float4s
static bool fwd_rows_ok(const void* x, const void* y, const float* gamma, const float* beta, int D, int C) {
    return C == D && (D & 3) == 0 && D <= 4096 && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0;
}
static int launch_fwd_rows(hipStream_t s, const float* x, float* y, const float* gamma, const float* beta, float* mean, float* rstd,
                           float* lse, long long rows, int D) {
    long long blocks = (rows + 3) / 4;
    if (blocks > 256 * 3) blocks = 256 * 3;
    const int nv = cdiv(D >> 2, 64);
#define ASR_LNF(NV) hipLaunchKernelGGL(fwd_rows_f32_kernel<NV>, dim3((unsigned)blocks), dim3(256), 0, s, x, y, gamma, beta, mean, rstd, lse, rows, D)
    if (nv <= 4) ASR_LNF(4); else if (nv <= 8) ASR_LNF(8); else if (nv <= 12) ASR_LNF(12); else ASR_LNF(16);
#undef ASR_LNF
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_layernorm_fwd_lse_ok(int D, int C) { return C == D && (D & 3) == 0 && D <= 4096; }

extern "C" int asr_layernorm_fwd_lse(void* stream, const float* x, float* y, const float* gamma, const float* beta, float* mean,
                                     float* rstd, float* lse, long long rows, int D) {
    if (!x || !y || !gamma || !beta || !mean || !rstd || rows <= 0 || D <= 0) return ASR_ERR_BAD_ARG;
    if (!fwd_rows_ok(x, y, gamma, beta, D, D)) return ASR_ERR_UNSUPPORTED;
    return launch_fwd_rows((hipStream_t)stream, x, y, gamma, beta, mean, rstd, lse, rows, D);
}

extern "C" int asr_layernorm_fwd(void* stream, const void* x, int x_bf16, void* y, int y_bf16, const float* gamma,
                                 const float* beta, float* mean, float* rstd, long long rows, int D, int C) {
    if (!x || !y || !gamma || !beta || !mean || !rstd || rows <= 0 || D <= 0 || C <= 0 || D % C) return ASR_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!x_bf16 && !y_bf16 && fwd_rows_ok(x, y, gamma, beta, D, C))
        return launch_fwd_rows(s, (const float*)x, (float*)y, gamma, beta, mean, rstd, nullptr, rows, D);
    const dim3 g((unsigned)rows), b(256);
    if (!x_bf16 && !y_bf16 && (D & 3) == 0 && (C & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
        hipLaunchKernelGGL(fwd_f32x4_kernel, g, b, 0, s, (const float*)x, (float*)y, gamma, beta, mean, rstd, D, C);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
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
    if (dx && sizeof(XT) == 4 && sizeof(GT) == 4 && !dx_bf16 && (D & 3) == 0 && (C & 3) == 0 &&
        ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0) {
        hipLaunchKernelGGL(bwd_f32x4_kernel, g, b, 0, s, (const float*)x, (const float*)dy, gamma, mean, rstd, (float*)dx, D, C);
        ASR_LAUNCH_CHECK();
    } else if (dx) {
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

// workgroups of the one-sweep backward: enough rows each to amortise the partial sums, enough of them to fill the chip
static int rows_grid(long long rows) {
    long long g = (rows + 15) / 16;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" long long asr_layernorm_bwd_rows_ws_bytes(long long rows, int D) {
    if (rows <= 0 || D <= 0 || (D & 3) || D > 4096) return 0;
    return (long long)rows_grid(rows) * 2 * D * (long long)sizeof(float);
}

extern "C" int asr_layernorm_bwd_rows(void* stream, const float* x, const float* dy, const float* gamma, const float* mean,
                                      const float* rstd, void* dx, int dx_bf16, float* dgamma, float* dbeta, long long rows,
                                      int D, int C, void* ws, long long ws_bytes) {
    if (!x || !dy || !gamma || !mean || !rstd || rows <= 0 || D <= 0 || C <= 0 || D % C) return ASR_ERR_BAD_ARG;
    if ((D & 3) || (C & 3) || D > 4096 || ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15)) return ASR_ERR_UNSUPPORTED;
    const bool params = dgamma && dbeta;
    if (params && (!ws || ws_bytes < asr_layernorm_bwd_rows_ws_bytes(rows, D))) return ASR_ERR_BAD_ARG;
    if (!dx && !params) return ASR_OK;
    hipStream_t s = (hipStream_t)stream;
    const int G = rows_grid(rows);
    float* partial = params ? (float*)ws : nullptr;
    const int nv = (D / 4 + 255) / 256;
#define ASR_LNB(NV)                                                                                                       \
    do {                                                                                                                  \
        if (dx_bf16)                                                                                                      \
            hipLaunchKernelGGL((bwd_rows_f32_kernel<uint16_t, NV>), dim3(G), dim3(256), 0, s, x, dy, gamma, mean, rstd,  \
                               (uint16_t*)dx, partial, rows, D, C);                                                       \
        else                                                                                                              \
            hipLaunchKernelGGL((bwd_rows_f32_kernel<float, NV>), dim3(G), dim3(256), 0, s, x, dy, gamma, mean, rstd,     \
                               (float*)dx, partial, rows, D, C);                                                          \
    } while (0)
    switch (nv) {
        case 1: ASR_LNB(1); break;
        case 2: ASR_LNB(2); break;
        case 3: ASR_LNB(3); break;
        default: ASR_LNB(4); break;
    }
#undef ASR_LNB
    ASR_LAUNCH_CHECK();
    if (params) {
        int chunks = (G + 15) / 16;
        if (chunks > 16) chunks = 16;
        hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(2 * D, 64), chunks), dim3(256), 0, s, partial, G, D, C, dgamma, dbeta, 2, (float*)nullptr);
        ASR_LAUNCH_CHECK();
    }
    return ASR_OK;
}

// dgamma / dbeta += the per-workgroup column sums of a one-sweep backward (also used by csrc/ctc_ln.hip)
extern "C" int asr_layernorm_fold_partials(void* stream, const float* partial, int G, int D, int C, float* dgamma, float* dbeta,
                                           float* extra) {
    if (!partial || !dgamma || !dbeta || G <= 0 || D <= 0 || C <= 0 || D % C) return ASR_ERR_BAD_ARG;
    int chunks = (G + 15) / 16;
    if (chunks > 16) chunks = 16;
    const int planes = extra ? 3 : 2;
    hipLaunchKernelGGL(asr::ln::fold_partials_kernel, dim3(cdiv(planes * D, 64), chunks), dim3(256), 0, (hipStream_t)stream, partial, G, D, C,
                       dgamma, dbeta, planes, extra);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// ------------------------------------------------------------------------------------------------ weight normalisation
// asr/nn/convolution_2d.py:21-25,62-64: W = g * V / (||V|| + 1e-9), norm over everything but the output channel.
// One workgroup per output channel (rows of K = Ci*kh*kw floats).
namespace asr {
namespace wn {

__global__ __launch_bounds__(256) void fwd_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                                  float* __restrict__ W, float* __restrict__ norm_out, int K) {
    __shared__ float scratch[32];
    const int co = blockIdx.x;
    const float* v = V + (size_t)co * K;
    float s = 0.f;
    for (int i = threadIdx.x; i < K; i += blockDim.x) s += v[i] * v[i];
    const float norm = sqrtf(block_sum(s, scratch)) + 1e-9f;
    if (threadIdx.x == 0) norm_out[co] = norm;
    const float sc = g[co] / norm;
    for (int i = threadIdx.x; i < K; i += blockDim.x) W[(size_t)co * K + i] = v[i] * sc;
}
// asr/nn/convolution_2d.py:92-93: gg = sum(gW * Vn), gV = g * (gW - gg * Vn) / norm   (accumulated)
__global__ __launch_bounds__(256) void bwd_kernel(const float* __restrict__ gW, const float* __restrict__ V,
                                                  const float* __restrict__ g, const float* __restrict__ norm,
                                                  float* __restrict__ gV, float* __restrict__ gg, int K) {
    __shared__ float scratch[32];
    const int co = blockIdx.x;
    const float nr = norm[co], gc = g[co];
    const float* v = V + (size_t)co * K;
    const float* w = gW + (size_t)co * K;
    float s = 0.f;
    for (int i = threadIdx.x; i < K; i += blockDim.x) s += w[i] * v[i] / nr;
    const float ggc = block_sum(s, scratch);
    if (threadIdx.x == 0) gg[co] += ggc;
    for (int i = threadIdx.x; i < K; i += blockDim.x) gV[(size_t)co * K + i] += gc * (w[i] - ggc * v[i] / nr) / nr;
}
// per-channel mean and (population) standard deviation of x (rows, C) f32: data-dependent init (:152-167)
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, long long rows, int C,
                                                            float* __restrict__ mean, float* __restrict__ stdv) {
    __shared__ float scratch[32];
    const int c = blockIdx.x;
    float s = 0.f;
    for (long long r = threadIdx.x; r < rows; r += blockDim.x) s += x[r * C + c];
    const float m = block_sum(s, scratch) / (float)rows;
    float v = 0.f;
    for (long long r = threadIdx.x; r < rows; r += blockDim.x) { const float d = x[r * C + c] - m; v += d * d; }
    const float var = block_sum(v, scratch) / (float)rows;
    if (threadIdx.x == 0) { mean[c] = m; stdv[c] = sqrtf(var); }
}
// y = x * scale[c] + shift[c]  (f32 in, bf16 out)
__global__ void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                      const float* __restrict__ shift, uint16_t* __restrict__ y, long long n, int C) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        y[i] = f32_to_bf16(x[i] * scale[c] + shift[c]);
    }
}
// g = 1 / std, b = -mean / std
__global__ void init_kernel(const float* __restrict__ mean, const float* __restrict__ stdv, float* __restrict__ g,
                            float* __restrict__ b, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) { g[c] = 1.f / stdv[c]; b[c] = -mean[c] / stdv[c]; }
}

}  // namespace wn
}  // namespace asr

extern "C" int asr_weightnorm_fwd(void* stream, const float* V, const float* g, float* W, float* norm, int Co, int K) {
    if (!V || !g || !W || !norm || Co <= 0 || K <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(asr::wn::fwd_kernel, dim3(Co), dim3(256), 0, (hipStream_t)stream, V, g, W, norm, K);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_weightnorm_bwd(void* stream, const float* gW, const float* V, const float* g, const float* norm,
                                  float* gV, float* gg, int Co, int K) {
    if (!gW || !V || !g || !norm || !gV || !gg || Co <= 0 || K <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(asr::wn::bwd_kernel, dim3(Co), dim3(256), 0, (hipStream_t)stream, gW, V, g, norm, gV, gg, K);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_channel_stats(void* stream, const float* x, long long rows, int C, float* mean, float* stdv) {
    if (!x || !mean || !stdv || rows <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(asr::wn::channel_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, rows, C, mean, stdv);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_channel_affine(void* stream, const float* x, const float* scale, const float* shift, void* y_bf16,
                                  long long n, int C) {
    if (!x || !scale || !shift || !y_bf16 || n <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    long long gsz = (n + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(asr::wn::channel_affine_kernel, dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                       (uint16_t*)y_bf16, n, C);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_weightnorm_init(void* stream, const float* mean, const float* stdv, float* g, float* b, int C) {
    if (!mean || !stdv || !g || !b || C <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(asr::wn::init_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, stdv, g, b, C);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_batchnorm_stats(void* stream, const void* x_bf16, long long R, int C, float eps, float decay, double* ws2C,
                                   float* mean, float* rstd, float* avg_mean, float* avg_var) {
    if (!x_bf16 || !ws2C || !mean || !rstd || R <= 0 || C <= 0 || (avg_mean == nullptr) != (avg_var == nullptr)) return ASR_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws2C, 0, sizeof(double) * 2 * C, s) != hipSuccess) return ASR_ERR_LAUNCH;
    const int rpb = 64;
    hipLaunchKernelGGL(asr::ln::bn_reduce_kernel<0>, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(256), 0, s, (const uint16_t*)x_bf16,
                       (const uint16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, R, C, rpb, ws2C, ws2C + C);
    hipLaunchKernelGGL(asr::ln::bn_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, ws2C, ws2C + C, R, C, eps, decay, mean, rstd,
                       avg_mean, avg_var);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

namespace asr { namespace ln {
__global__ void rsqrt_eps_kernel(const float* __restrict__ var, float eps, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)(1.0 / sqrt((double)var[i] + (double)eps));
}
} }

// inference-mode batch normalisation: rstd from the running variance (chainer.links.BatchNormalization with
// chainer.config.train == False normalises with avg_var + eps)
extern "C" int asr_rsqrt_eps(void* stream, const float* var, float eps, float* out, int n) {
    if (!var || !out || n <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(asr::ln::rsqrt_eps_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, var, eps, out, n);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_batchnorm_fwd(void* stream, const void* x_bf16, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, long long R, int C, void* y_bf16) {
    if (!x_bf16 || !mean || !rstd || !gamma || !beta || !y_bf16 || R <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    const long long n = R * C;
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(asr::ln::bn_fwd_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16, mean, rstd,
                       gamma, beta, n, C, (uint16_t*)y_bf16);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_batchnorm_bwd(void* stream, const void* x_bf16, const void* gy_bf16, const float* mean, const float* rstd,
                                 const float* gamma, long long R, int C, double* ws2C, void* dx_bf16, float* dgamma_acc,
                                 float* dbeta_acc) {
    if (!x_bf16 || !gy_bf16 || !mean || !rstd || !gamma || !ws2C || R <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws2C, 0, sizeof(double) * 2 * C, s) != hipSuccess) return ASR_ERR_LAUNCH;
    const int rpb = 64;
    hipLaunchKernelGGL(asr::ln::bn_reduce_kernel<1>, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(256), 0, s, (const uint16_t*)x_bf16,
                       (const uint16_t*)gy_bf16, mean, rstd, R, C, rpb, ws2C, ws2C + C);
    if (dx_bf16) {
        const long long n = R * C;
        long long g = (n + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(asr::ln::bn_bwd_kernel, dim3((unsigned)g), dim3(256), 0, s, (const uint16_t*)x_bf16, (const uint16_t*)gy_bf16, mean,
                           rstd, gamma, ws2C, ws2C + C, R, C, (uint16_t*)dx_bf16);
    }
    if (dgamma_acc && dbeta_acc)
        hipLaunchKernelGGL(asr::ln::bn_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, s, ws2C, ws2C + C, C, dgamma_acc, dbeta_acc);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
