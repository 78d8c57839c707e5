// HBM-bound layout / activation / pooling kernels of the conv front-end (gfx950).
// Internal activation layout: (T, B, H, C) bf16 -- time-major, channel-last -- so that
//   * maxout pairs (asr/nn/nn.py:45-50) are adjacent elements,
//   * max-pooling over height (asr/nn/nn.py:95-103) and layer-norm over (C, H) (asr/nn/layernorm.py:42-45) stay
//     inside one contiguous H*C block per (t, b),
//   * the (B, C*H, T) reshape feeding the recurrent stack (run/ctc/sru/model.py:114) and the per-time-step split
//     (asr/model/cnn.py:41-44) are free views.
// The reference's logical (B, C, H, T) order is recovered by a permuted view on the Python side.
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace ew {

constexpr int kThreads = 256;
static inline int grid_for(long long n) {
    long long g = (n + kThreads - 1) / kThreads;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------ cast / transpose
__global__ void cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = f32_to_bf16(src[i]);
}
// 8 elements per thread: two float4 in, one 16-B store out (n % 8 == 0, 16-B aligned)
__global__ void cast_bf16_vec8_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const float4 a = src[2 * i], b = src[2 * i + 1];
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
        o.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
        o.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
        o.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
        dst[i] = o;
    }
}
// dst[c][r] = src[r][c]   (small weight matrices; 32x32 LDS tile)
__global__ void transpose_cast_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[(size_t)c * rows + r] = f32_to_bf16(tile[tx][i]);
    }
}
// All bf16 weight copies of a model in ONE launch (they are rebuilt after every optimiser step: 27 launches of ~6 us
// each in front of the GEMMs otherwise).  jobs: njobs x 6 long long {src, dst, rows, cols, transpose, first_tile},
// first_tile ascending; a tile is 64 x 64 elements of src.
__global__ void __launch_bounds__(256) cast_many_kernel(const long long* __restrict__ jobs, int njobs) {
    __shared__ float tile[64][65];
    __shared__ long long first_tile[256];
    // which job owns this tile: the first_tile column goes to LDS in one load per thread and is searched there (a linear walk over
    // the table in global memory cost the workgroups of the last jobs ~40 dependent L2 round trips before their first byte of work)
    int j = 0;
    if (njobs <= 256) {
        if ((int)threadIdx.x < njobs) first_tile[threadIdx.x] = jobs[threadIdx.x * 6 + 5];
        __syncthreads();
        int lo = 0, hi = njobs - 1;                 // last job whose first_tile <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (first_tile[mid] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        j = lo;
    } else {
        while (j + 1 < njobs && (long long)blockIdx.x >= jobs[(j + 1) * 6 + 5]) ++j;
    }
    const float* __restrict__ src = (const float*)jobs[j * 6 + 0];
    uint16_t* __restrict__ dst = (uint16_t*)jobs[j * 6 + 1];
    const int rows = (int)jobs[j * 6 + 2], cols = (int)jobs[j * 6 + 3];
    const bool transpose = jobs[j * 6 + 4] != 0;
    const int local = (int)((long long)blockIdx.x - jobs[j * 6 + 5]);
    const int tiles_c = (cols + 63) / 64;
    const int r0 = (local / tiles_c) * 64, c0 = (local % tiles_c) * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16 threads, 4 columns each, 4 row passes
    const bool vec = (cols & 3) == 0;
    if (!transpose) {
        for (int i = ty; i < 64; i += 16) {
            const int r = r0 + i, c = c0 + tx * 4;
            if (r >= rows || c >= cols) continue;
            if (vec) {
                const float4 v = *(const float4*)(src + (size_t)r * cols + c);
                uint2 o;
                o.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
                o.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
                *(uint2*)(dst + (size_t)r * cols + c) = o;
            } else {
                for (int k = 0; k < 4 && c + k < cols; ++k) dst[(size_t)r * cols + c + k] = f32_to_bf16(src[(size_t)r * cols + c + k]);
            }
        }
        return;
    }
    for (int i = ty; i < 64; i += 16) {
        const int r = r0 + i, c = c0 + tx * 4;
        if (vec && r < rows && c < cols) {
            const float4 v = *(const float4*)(src + (size_t)r * cols + c);
            tile[i][tx * 4 + 0] = v.x; tile[i][tx * 4 + 1] = v.y; tile[i][tx * 4 + 2] = v.z; tile[i][tx * 4 + 3] = v.w;
        } else {
            for (int k = 0; k < 4; ++k) tile[i][tx * 4 + k] = (r < rows && c + k < cols) ? src[(size_t)r * cols + c + k] : 0.f;
        }
    }
    __syncthreads();
    const bool vecr = (rows & 3) == 0;
    for (int i = ty; i < 64; i += 16) {       // dst row = source column c0 + i; 4 consecutive source rows per thread
        const int c = c0 + i, r = r0 + tx * 4;
        if (c >= cols || r >= rows) continue;
        if (vecr) {
            uint2 o;
            o.x = (unsigned)f32_to_bf16(tile[tx * 4 + 0][i]) | ((unsigned)f32_to_bf16(tile[tx * 4 + 1][i]) << 16);
            o.y = (unsigned)f32_to_bf16(tile[tx * 4 + 2][i]) | ((unsigned)f32_to_bf16(tile[tx * 4 + 3][i]) << 16);
            *(uint2*)(dst + (size_t)c * rows + r) = o;
        } else {
            for (int k = 0; k < 4 && r + k < rows; ++k) dst[(size_t)c * rows + r + k] = f32_to_bf16(tile[tx * 4 + k][i]);
        }
    }
}
__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dst[i] = bf16_to_f32(src[i]);
}

// generic 4-d permuting copy: dst is dense (d0,d1,d2,d3); src element strides given per dst dim
template <typename SrcT, typename DstT>
__global__ void permute4_kernel(const SrcT* __restrict__ src, DstT* __restrict__ dst, int d0, int d1, int d2, int d3,
                                long long s0, long long s1, long long s2, long long s3) {
    const long long n = (long long)d0 * d1 * d2 * d3;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int i3 = r % d3; r /= d3;
        const int i2 = r % d2; r /= d2;
        const int i1 = r % d1; r /= d1;
        const int i0 = (int)r;
        const SrcT v = src[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
        float f;
        if (sizeof(SrcT) == 2) f = bf16_to_f32((uint16_t)v); else f = (float)v;
        if (sizeof(DstT) == 2) dst[i] = (DstT)f32_to_bf16(f); else dst[i] = (DstT)f;
    }
}

// dense (T, B, H, Cpad) bf16 from any strided (T, B, H, C) source, zero channels behind C
template <typename SrcT>
__global__ void pack_input_pad_kernel(const SrcT* __restrict__ x, long long sT, long long sB, long long sH, long long sC, int T,
                                      int B, int H, int C, int Cpad, uint16_t* __restrict__ out) {
    const long long n = (long long)T * B * H * Cpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int c = (int)(r % Cpad); r /= Cpad;
        const int h = (int)(r % H); r /= H;
        const int b = (int)(r % B);
        const int t = (int)(r / B);
        uint16_t v = 0;
        if (c < C) {
            const SrcT s = x[t * sT + b * sB + h * sH + c * sC];
            if (sizeof(SrcT) == 2) v = (uint16_t)s; else v = f32_to_bf16((float)s);
        }
        out[i] = v;
    }
}

// The reference's loader hands (B, C, H, T) float32 with time innermost (asr/data/loaders/base.py): the gather above then reads 4
// useful bytes per 128-B line (34 us for the model's 36 MB).  Time-innermost sources go through an LDS tile instead: a workgroup
// reads 64 frames of 8 heights x C channels along time (256-B runs) and writes (t, b, h, 8 channels) rows, 128 contiguous bytes
// per frame.
constexpr int PIP = 72;       // tile pitch per frame in elements: 8 heights x 8 channels + 8 (16-B aligned rows)
__global__ __launch_bounds__(256) void pack_input_pad_time_kernel(const float* __restrict__ x, long long sB, long long sH, long long sC,
                                                                 int T, int B, int H, int C, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[64 * PIP];
    const int t0 = blockIdx.x * 64, b = blockIdx.y, h0 = blockIdx.z * 8;
    const int tl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * PIP / 2; i += 256) reinterpret_cast<uint32_t*>(tile)[i] = 0u;      // (channels >= C stay zero)
    __syncthreads();
    for (int rr = rg; rr < C * 8; rr += 4) {
        const int c = rr >> 3, hh = rr & 7;
        if (h0 + hh < H && t0 + tl < T)
            tile[tl * PIP + hh * 8 + c] = f32_to_bf16(x[(long long)(t0 + tl) + b * sB + (h0 + hh) * sH + c * sC]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int p = threadIdx.x + 256 * k, t = p >> 3, hh = p & 7;
        if (t0 + t < T && h0 + hh < H)
            *reinterpret_cast<uint4*>(out + (((long long)(t0 + t) * B + b) * H + h0 + hh) * 8) = *reinterpret_cast<const uint4*>(tile + t * PIP + hh * 8);
    }
}

// ------------------------------------------------------------------------------------------------ im2col / col2im
// col[(t, b, ho)][(kh, kw, ci)] = x[t + kw - pt, b, ho + kh - ph, ci]   (zero outside), row pitch Kp >= KH*KW*Cin,
// t in [0, Tout).  pt = KW-1, Tout = T is the causal convolution; Tout = T + KW-1 the reference's padded output.
template <typename SrcT>
__global__ void im2col_kernel(const SrcT* __restrict__ x, long long sT, long long sB, long long sH, long long sC, int T,
                              int B, int Hin, int Cin, int KH, int KW, int ph, int pt, int Tout, int Hout, int Kp,
                              uint16_t* __restrict__ col) {
    const long long rows = (long long)Tout * B * Hout;
    const long long n = rows * Kp;
    const int Kreal = KH * KW * Cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        long long row = i / Kp;
        uint16_t out = 0;
        if (k < Kreal) {
            const int ci = k % Cin;
            const int kw = (k / Cin) % KW;
            const int kh = k / (Cin * KW);
            const int ho = (int)(row % Hout); row /= Hout;
            const int b = (int)(row % B);
            const int t = (int)(row / B);
            const int ti = t + kw - pt, hi = ho + kh - ph;
            if (ti >= 0 && ti < T && hi >= 0 && hi < Hin) {
                const SrcT v = x[ti * sT + b * sB + hi * sH + ci * sC];
                if (sizeof(SrcT) == 2) out = (uint16_t)v; else out = f32_to_bf16((float)v);
            }
        }
        col[i] = out;
    }
}
// the same for any input layout / channel count, 8 columns per thread and one 16-B store (Kp % 8 == 0): the per-element form
// above spends its time in 2-byte stores (0.24 ms for the first layer's 117 MB column matrix)
template <typename SrcT>
__global__ void im2col_chunk8_kernel(const SrcT* __restrict__ x, long long sT, long long sB, long long sH, long long sC, int T,
                                     int B, int Hin, int Cin, int KH, int KW, int ph, int pt, int Tout, int Hout, int Kp,
                                     uint4* __restrict__ col) {
    const int chunks = Kp >> 3;
    const long long n = (long long)Tout * B * Hout * chunks;
    const int Kreal = KH * KW * Cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % chunks);
        long long row = i / chunks;
        const int ho = (int)(row % Hout); row /= Hout;
        const int b = (int)(row % B);
        const int t = (int)(row / B);
        int k = c8 * 8;
        int ci = k % Cin, kw = (k / Cin) % KW, kh = k / (Cin * KW);
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint16_t v = 0;
            if (k + e < Kreal) {
                const int ti = t + kw - pt, hi = ho + kh - ph;
                if (ti >= 0 && ti < T && hi >= 0 && hi < Hin) {
                    const SrcT s = x[ti * sT + b * sB + hi * sH + ci * sC];
                    if (sizeof(SrcT) == 2) v = (uint16_t)s; else v = f32_to_bf16((float)s);
                }
            }
            o[e] = v;
            if (++ci == Cin) { ci = 0; if (++kw == KW) { kw = 0; ++kh; } }
        }
        uint4 pk;
        pk.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); pk.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        pk.z = (uint32_t)o[4] | ((uint32_t)o[5] << 16); pk.w = (uint32_t)o[6] | ((uint32_t)o[7] << 16);
        col[i] = pk;
    }
}
// dx[t, b, h, ci] = sum_{kh,kw} dcol[(t - kw + pt, b, h - kh + ph)][(kh, kw, ci)]   (gather form, no atomics)
__global__ void col2im_kernel(const uint16_t* __restrict__ dcol, int T, int B, int Hin, int Cin, int KH, int KW, int ph,
                              int pt, int Tout, int Hout, int Kp, uint16_t* __restrict__ dx) {
    const long long n = (long long)T * B * Hin * Cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int ci = (int)(r % Cin); r /= Cin;
        const int h = (int)(r % Hin); r /= Hin;
        const int b = (int)(r % B);
        const int t = (int)(r / B);
        float acc = 0.f;
        for (int kh = 0; kh < KH; ++kh) {
            const int ho = h - kh + ph;
            if (ho < 0 || ho >= Hout) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int to = t - kw + pt;
                if (to < 0 || to >= Tout) continue;
                acc += bf16_to_f32(dcol[(((long long)to * B + b) * Hout + ho) * Kp + (kh * KW + kw) * Cin + ci]);
            }
        }
        dx[i] = f32_to_bf16(acc);
    }
}

// ------------------------------------------------------------------------------------------------ maxout(2)
// y[r][c] = max(x[r][2c], x[r][2c+1]); rows of width 2*C (x) / C (y).  n = rows * C outputs.
__global__ void maxout2_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long n) {
    const uint32_t* x2 = reinterpret_cast<const uint32_t*>(x);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t p = x2[i];
        const float a = bf16_to_f32((uint16_t)(p & 0xffff)), b = bf16_to_f32((uint16_t)(p >> 16));
        y[i] = (b > a) ? (uint16_t)(p >> 16) : (uint16_t)(p & 0xffff);     // ties -> first element (argmax)
    }
}
__global__ void maxout2_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                   uint16_t* __restrict__ dx, long long n) {
    const uint32_t* x2 = reinterpret_cast<const uint32_t*>(x);
    uint32_t* dx2 = reinterpret_cast<uint32_t*>(dx);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t p = x2[i];
        const float a = bf16_to_f32((uint16_t)(p & 0xffff)), b = bf16_to_f32((uint16_t)(p >> 16));
        const uint32_t g = dy[i];
        dx2[i] = (b > a) ? (g << 16) : g;
    }
}

// ------------------------------------------------------------------------------------------------ max-pool over H
// x (R, Hin, C) -> y (R, Hout, C); window k, stride k, cover_all (window may overhang the end)
__global__ void maxpool_h_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long R, int Hin,
                                     int Hout, int C, int k) {
    const long long n = R * Hout * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ho = (int)((i / C) % Hout);
        const long long r = i / ((long long)C * Hout);
        const uint16_t* src = x + (r * Hin) * C + c;
        float m = -INFINITY;
        uint16_t mv = 0xff80;   // -inf in bf16
        for (int j = 0; j < k; ++j) {
            const int h = ho * k + j;
            if (h >= Hin) break;
            const uint16_t v = src[(long long)h * C];
            const float f = bf16_to_f32(v);
            if (f > m) { m = f; mv = v; }
        }
        y[i] = mv;
    }
}
__global__ void maxpool_h_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                     uint16_t* __restrict__ dx, long long R, int Hin, int Hout, int C, int k) {
    const long long n = R * Hout * C;     // one thread per window: writes its k inputs
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ho = (int)((i / C) % Hout);
        const long long r = i / ((long long)C * Hout);
        const uint16_t* src = x + (r * Hin) * C + c;
        uint16_t* dst = dx + (r * Hin) * C + c;
        float m = -INFINITY;
        int am = 0;
        for (int j = 0; j < k; ++j) {
            const int h = ho * k + j;
            if (h >= Hin) break;
            const float f = bf16_to_f32(src[(long long)h * C]);
            if (f > m) { m = f; am = j; }
        }
        const uint16_t g = dy[i];
        for (int j = 0; j < k; ++j) {
            const int h = ho * k + j;
            if (h >= Hin) break;
            dst[(long long)h * C] = (j == am) ? g : (uint16_t)0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ add / column sum
__global__ void add_bf16_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ y,
                                long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = f32_to_bf16(bf16_to_f32(a[i]) + bf16_to_f32(b[i]));
}
// out[c] += sum_r x[r][c]; grid.x over column blocks of 64, grid.y over row chunks; one atomic per (block, column)
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, long long rows, int cols, int ld, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    const long long chunk = (rows + gridDim.y - 1) / gridDim.y;
    const long long r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
    float s = 0.f;
    if (c < cols)
        for (long long r = r0 + w; r < r1; r += 4) {
            const T v = x[r * ld + c];
            s += sizeof(T) == 2 ? bf16_to_f32((uint16_t)v) : (float)v;
        }
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < cols) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}


// ------------------------------------------------------------------------------------------------ activations
// kind: 0 relu, 1 clipped_relu(alpha = z), 2 leaky_relu(alpha = slope), 3 elu(alpha), 4 sigmoid, 5 tanh,
//       6 hard_sigmoid, 7 softplus(alpha = beta)      (the activation wrappers of asr/nn/nn.py:11-73)
__device__ __forceinline__ float act_f(int kind, float a, float x) {
    switch (kind) {
        case 0: return fmaxf(x, 0.f);
        case 1: return fminf(fmaxf(x, 0.f), a);
        case 2: return x >= 0.f ? x : a * x;
        case 3: return x >= 0.f ? x : a * (__expf(x) - 1.f);
        case 4: return 1.f / (1.f + __expf(-x));
        case 5: return tanhf(x);
        case 6: return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f);
        default: { const float bx = a * x; return (bx > 20.f ? bx : log1pf(__expf(bx))) / a; }
    }
}
__device__ __forceinline__ float act_df(int kind, float a, float x) {
    switch (kind) {
        case 0: return x > 0.f ? 1.f : 0.f;
        case 1: return (x > 0.f && x < a) ? 1.f : 0.f;
        case 2: return x >= 0.f ? 1.f : a;
        case 3: return x >= 0.f ? 1.f : a * __expf(x);
        case 4: { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f - s); }
        case 5: { const float t = tanhf(x); return 1.f - t * t; }
        case 6: return (x > -2.5f && x < 2.5f) ? 0.2f : 0.f;
        default: return 1.f / (1.f + __expf(-a * x));
    }
}
__global__ void act_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long n, int kind, float a) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = f32_to_bf16(act_f(kind, a, bf16_to_f32(x[i])));
}
__global__ void act_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
                               long long n, int kind, float a) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dx[i] = f32_to_bf16(bf16_to_f32(dy[i]) * act_df(kind, a, bf16_to_f32(x[i])));
}
// GLU (asr/nn/nn.py:279-280): rows of 2C channels [A | B] -> C channels, H = A * sigmoid(B)
__global__ void glu_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long rows, int C) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float a = bf16_to_f32(x[r * 2 * C + c]), b = bf16_to_f32(x[r * 2 * C + C + c]);
        y[i] = f32_to_bf16(a / (1.f + __expf(-b)));
    }
}
__global__ void glu_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
                               long long rows, int C) {
    const long long n = rows * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        const float a = bf16_to_f32(x[r * 2 * C + c]), b = bf16_to_f32(x[r * 2 * C + C + c]);
        const float s = 1.f / (1.f + __expf(-b)), g = bf16_to_f32(dy[i]);
        dx[r * 2 * C + c] = f32_to_bf16(g * s);
        dx[r * 2 * C + C + c] = f32_to_bf16(g * a * s * (1.f - s));
    }
}
// dropout with a counter-based hash RNG (seed, element index) so that backward regenerates the mask
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__global__ void dropout_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long n, float ratio,
                               uint32_t seed) {
    const float scale = 1.f / (1.f - ratio);
    const uint32_t thr = (uint32_t)(ratio * 4294967296.0);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t r = hash32((uint32_t)i * 0x9e3779b9U + hash32(seed + (uint32_t)(i >> 32)));
        y[i] = r >= thr ? f32_to_bf16(bf16_to_f32(x[i]) * scale) : (uint16_t)0;
    }
}


// ------------------------------------------------------------------------------------------------ conv weight (un)packing
// pack:   dst[co][k] (or dst[k][co] when transposed) = W[co][ci][kh][kw], k = (kh*KW + kw)*Ci + ci, zero for k >= K
__global__ void conv_weight_pack_kernel(const float* __restrict__ W, uint16_t* __restrict__ dst, int Co, int Ci, int KH,
                                        int KW, int Kp, int transpose) {
    const long long n = (long long)Co * Kp;
    const int K = KH * KW * Ci;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(i / Kp), k = (int)(i - (long long)co * Kp);
        float v = 0.f;
        if (k < K) {
            const int ci = k % Ci, kw = (k / Ci) % KW, kh = k / (Ci * KW);
            v = W[(((long long)co * Ci + ci) * KH + kh) * KW + kw];
        }
        if (transpose) dst[(long long)k * Co + co] = f32_to_bf16(v); else dst[i] = f32_to_bf16(v);
    }
}
// backward-data operand of the implicit-GEMM convolution: dst[ci][(kh*KW + kw)*Co + co] = W[co][ci][kh][kw]
__global__ void conv_weight_pack_bwd_kernel(const float* __restrict__ W, uint16_t* __restrict__ dst, int Co, int Ci, int KH,
                                            int KW) {
    const long long n = (long long)Co * Ci * KH * KW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int co = (int)(r % Co); r /= Co;
        const int kw = (int)(r % KW); r /= KW;
        const int kh = (int)(r % KH);
        const int ci = (int)(r / KH);
        dst[i] = f32_to_bf16(W[(((long long)co * Ci + ci) * KH + kh) * KW + kw]);
    }
}
// unpack: gW[co][ci][kh][kw] += scratch[co][(kh*KW + kw)*Cs + ci]   (Cs >= Ci: channel pitch of the scratch rows)
__global__ void conv_weight_grad_unpack_kernel(const float* __restrict__ scratch, float* __restrict__ gW, int Co, int Ci,
                                               int KH, int KW, int Kp, int Cs, int copies) {
    const long long n = (long long)Co * Ci * KH * KW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        long long r = i;
        const int kw = (int)(r % KW); r /= KW;
        const int kh = (int)(r % KH); r /= KH;
        const int ci = (int)(r % Ci);
        const int co = (int)(r / Ci);
        const long long o = (long long)co * Kp + (kh * KW + kw) * Cs + ci;
        float v = scratch[o];
        for (int c = 1; c < copies; ++c) v += scratch[o + (long long)c * Co * Kp];           // (per-XCD copies: asr_conv_tn_acc_copies)
        gW[i] += v;                // (one writer per element: a float atomic here ran at the memory side, 45 us for 123 k elements)
    }
}


// 16-byte variants for the internal (T, B, H, C) bf16 layout with C % 8 == 0: one thread moves 8 channels
__global__ void im2col_vec8_kernel(const uint16_t* __restrict__ x, int T, int B, int Hin, int Cin, int KH, int KW, int ph,
                                   int pt, int Tout, int Hout, int Kp, uint16_t* __restrict__ col) {
    const int c8n = Cin >> 3;
    const int per_row = KH * KW * c8n;
    const long long n = (long long)Tout * B * Hout * per_row;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        long long r = i / c8n;
        const int kw = (int)(r % KW); r /= KW;
        const int kh = (int)(r % KH); r /= KH;
        const int ho = (int)(r % Hout); r /= Hout;
        const int b = (int)(r % B);
        const int t = (int)(r / B);
        const int ti = t + kw - pt, hi = ho + kh - ph;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ti >= 0 && ti < T && hi >= 0 && hi < Hin)
            v = *reinterpret_cast<const uint4*>(x + ((((long long)ti * B + b) * Hin + hi) * Cin + c8 * 8));
        *reinterpret_cast<uint4*>(col + (((long long)t * B + b) * Hout + ho) * Kp + (kh * KW + kw) * Cin + c8 * 8) = v;
    }
}
__global__ void col2im_vec8_kernel(const uint16_t* __restrict__ dcol, int T, int B, int Hin, int Cin, int KH, int KW,
                                   int ph, int pt, int Tout, int Hout, int Kp, uint16_t* __restrict__ dx) {
    const int c8n = Cin >> 3;
    const long long n = (long long)T * B * Hin * c8n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        long long r = i / c8n;
        const int h = (int)(r % Hin); r /= Hin;
        const int b = (int)(r % B);
        const int t = (int)(r / B);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < KH; ++kh) {
            const int ho = h - kh + ph;
            if (ho < 0 || ho >= Hout) continue;
            for (int kw = 0; kw < KW; ++kw) {
                const int to = t - kw + pt;
                if (to < 0 || to >= Tout) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(dcol + (((long long)to * B + b) * Hout + ho) * Kp + (kh * KW + kw) * Cin + c8 * 8);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += bf16_to_f32((uint16_t)(w[e] & 0xffff));
                    acc[2 * e + 1] += bf16_to_f32((uint16_t)(w[e] >> 16));
                }
            }
        }
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(acc[0]) | ((uint32_t)f32_to_bf16(acc[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(acc[2]) | ((uint32_t)f32_to_bf16(acc[3]) << 16);
        o.z = (uint32_t)f32_to_bf16(acc[4]) | ((uint32_t)f32_to_bf16(acc[5]) << 16);
        o.w = (uint32_t)f32_to_bf16(acc[6]) | ((uint32_t)f32_to_bf16(acc[7]) << 16);
        *reinterpret_cast<uint4*>(dx + i * 8) = o;
    }
}

// maxout / max-pool, 8 outputs per thread
__device__ __forceinline__ uint32_t max_bf16x2(uint32_t a, uint32_t b) {      // element-wise max, first wins ties
    const float a0 = bf16_to_f32((uint16_t)(a & 0xffff)), a1 = bf16_to_f32((uint16_t)(a >> 16));
    const float b0 = bf16_to_f32((uint16_t)(b & 0xffff)), b1 = bf16_to_f32((uint16_t)(b >> 16));
    return (b0 > a0 ? (b & 0xffffu) : (a & 0xffffu)) | (b1 > a1 ? (b & 0xffff0000u) : (a & 0xffff0000u));
}
__global__ void maxout2_fwd_vec_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long n8) {
    const uint4* x4 = reinterpret_cast<const uint4*>(x);
    uint4* y4 = reinterpret_cast<uint4*>(y);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const uint4 a = x4[2 * i], b = x4[2 * i + 1];
        const uint32_t in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t r = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t p = in[2 * e + h];
                const float lo = bf16_to_f32((uint16_t)(p & 0xffff)), hi = bf16_to_f32((uint16_t)(p >> 16));
                r |= (hi > lo ? (p >> 16) : (p & 0xffffu)) << (16 * h);
            }
            out[e] = r;
        }
        y4[i] = make_uint4(out[0], out[1], out[2], out[3]);
    }
}
__global__ void maxout2_bwd_vec_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                       uint16_t* __restrict__ dx, long long n8) {
    const uint4* x4 = reinterpret_cast<const uint4*>(x);
    const uint4* g4 = reinterpret_cast<const uint4*>(dy);
    uint4* d4 = reinterpret_cast<uint4*>(dx);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const uint4 a = x4[2 * i], b = x4[2 * i + 1], g = g4[i];
        const uint32_t in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const uint32_t gg[4] = {g.x, g.y, g.z, g.w};
        uint32_t out[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t p = in[e];
            const float lo = bf16_to_f32((uint16_t)(p & 0xffff)), hi = bf16_to_f32((uint16_t)(p >> 16));
            const uint32_t gv = (gg[e >> 1] >> (16 * (e & 1))) & 0xffffu;
            out[e] = hi > lo ? (gv << 16) : gv;
        }
        d4[2 * i] = make_uint4(out[0], out[1], out[2], out[3]);
        d4[2 * i + 1] = make_uint4(out[4], out[5], out[6], out[7]);
    }
}
// nn.Maxout(2) followed by nn.MaxPooling2D((k, 1)) (the conv blocks of the recipes: asr/nn/nn.py:45-50 + :95-103) in one
// pass, 8 output channels per thread: y[r][ho][c] = max_{h in window} max(x[r][h][2c], x[r][h][2c+1]).  The unfused pair
// writes and re-reads the maxout output (and its gradient): 674 -> 364 MB forward, 1.2 -> 0.68 GB backward on the first
// block.  Ties as in the two kernels it replaces: first channel of the pair, first row of the window.
__global__ void maxout2_pool_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long R, int Hin,
                                        int Hout, int C, int k) {
    const int c8n = C >> 3;
    const long long n = R * Hout * c8n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        const int ho = (int)((i / c8n) % Hout);
        const long long r = i / ((long long)c8n * Hout);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        for (int j = 0; j < k; ++j) {
            const int h = ho * k + j;
            if (h >= Hin) break;
            const uint4* src = reinterpret_cast<const uint4*>(x + ((r * Hin + h) * 2 * C + c8 * 16));
            const uint4 a = src[0], b = src[1];
            const uint32_t in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float lo = bf16_to_f32((uint16_t)(in[e] & 0xffff)), hi = bf16_to_f32((uint16_t)(in[e] >> 16));
                const float v = hi > lo ? hi : lo;
                if (v > m[e]) m[e] = v;
            }
        }
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(m[0]) | ((uint32_t)f32_to_bf16(m[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(m[2]) | ((uint32_t)f32_to_bf16(m[3]) << 16);
        o.z = (uint32_t)f32_to_bf16(m[4]) | ((uint32_t)f32_to_bf16(m[5]) << 16);
        o.w = (uint32_t)f32_to_bf16(m[6]) | ((uint32_t)f32_to_bf16(m[7]) << 16);
        *reinterpret_cast<uint4*>(y + ((r * Hout + ho) * C + c8 * 8)) = o;
    }
}
// db (optional, 2 C floats, accumulated): column sums of dx = the bias gradient of the convolution in front.  dx is a scatter of dy
// (every element goes to one of the 2 k inputs of its window), so the sums come from dy and the winners' channel bits while both are
// in registers -- the separate column-sum pass read all of dx again (311 MB for the first block of the BASELINE model).  Needs
// 256 % (C / 8) == 0: then a thread meets the same eight channel pairs in every round of its grid-stride loop.
__global__ void maxout2_pool_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                        uint16_t* __restrict__ dx, float* __restrict__ db, long long R, int Hin, int Hout, int C, int k) {
    __shared__ float red[16 * (256 + 32)];
    const int c8n = C >> 3;
    const long long n = R * Hout * c8n;
    float bsum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bsum[e] = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        const int ho = (int)((i / c8n) % Hout);
        const long long r = i / ((long long)c8n * Hout);
        float m[8];
        int win[8];                 // 2 * (row of the window) + (second channel of the pair won)
#pragma unroll
        for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; win[e] = 0; }
        for (int j = 0; j < k; ++j) {
            const int h = ho * k + j;
            if (h >= Hin) break;
            const uint4* src = reinterpret_cast<const uint4*>(x + ((r * Hin + h) * 2 * C + c8 * 16));
            const uint4 a = src[0], b = src[1];
            const uint32_t in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float lo = bf16_to_f32((uint16_t)(in[e] & 0xffff)), hi = bf16_to_f32((uint16_t)(in[e] >> 16));
                const bool second = hi > lo;
                const float v = second ? hi : lo;
                if (v > m[e]) { m[e] = v; win[e] = 2 * j + (second ? 1 : 0); }
            }
        }
        const uint4 g = *reinterpret_cast<const uint4*>(dy + ((r * Hout + ho) * C + c8 * 8));
        const uint32_t gg[4] = {g.x, g.y, g.z, g.w};
        if (db) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gv = bf16_to_f32((uint16_t)((gg[e >> 1] >> (16 * (e & 1))) & 0xffffu));
                bsum[2 * e] += (win[e] & 1) ? 0.f : gv;
                bsum[2 * e + 1] += (win[e] & 1) ? gv : 0.f;
            }
        }
        for (int j = 0; j < k; ++j) {
            const int h = ho * k + j;
            if (h >= Hin) break;
            uint32_t out[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t gv = (gg[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                out[e] = (win[e] >> 1) == j ? ((win[e] & 1) ? (gv << 16) : gv) : 0u;
            }
            uint4* dst = reinterpret_cast<uint4*>(dx + ((r * Hin + h) * 2 * C + c8 * 16));
            dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
            dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
        }
    }
    if (db) {       // threads tid, tid + c8n, ... hold the same channels: one atomic per channel and workgroup
        // image [e][thread], pitch 256 + c8n floats: a wave writes 64 consecutive floats per e; the folding threads (e, c8) of a wave read
        // banks e c8n + c8 (mod 32): at most two per bank.  (The [thread][16] image was a 16-way bank conflict on every access: 0.88 of
        // this kernel's LDS cycles, profiles/r03_pmc_sq.csv; pitch 256 still put the eight e of a wave on the same banks: 0.60.)
        const int pitch = 256 + (c8n < 32 ? c8n : 32);      // (c8n >= 32: a wave's folding threads span at most two e)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[e * pitch + threadIdx.x] = bsum[e];
        __syncthreads();
        // 16 c8n = 2 C sums to fold: more than one round of the 256 threads once C > 128 (a single `if (t < 16 c8n)`
        // round silently dropped the sums of e >= 256 / c8n, i.e. half of the bias gradient at C = 256)
        for (int t = threadIdx.x; t < c8n * 16; t += blockDim.x) {
            const int e = t / c8n, c8 = t - e * c8n;
            float s = 0.f;
            for (int u = c8; u < 256; u += c8n) s += red[e * pitch + u];
            atomicAdd(db + c8 * 16 + e, s);
        }
    }
}
__global__ void maxpool_h_fwd_vec_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, long long R, int Hin,
                                         int Hout, int C, int k) {
    const int c8n = C >> 3;
    const long long n = R * Hout * c8n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        const int ho = (int)((i / c8n) % Hout);
        const long long r = i / ((long long)c8n * Hout);
        const uint16_t* src = x + (r * Hin) * C + c8 * 8;
        uint4 m = *reinterpret_cast<const uint4*>(src + (long long)(ho * k) * C);
        for (int j = 1; j < k; ++j) {
            const int h = ho * k + j;
            if (h >= Hin) break;
            const uint4 v = *reinterpret_cast<const uint4*>(src + (long long)h * C);
            m.x = max_bf16x2(m.x, v.x); m.y = max_bf16x2(m.y, v.y); m.z = max_bf16x2(m.z, v.z); m.w = max_bf16x2(m.w, v.w);
        }
        *reinterpret_cast<uint4*>(y + ((r * Hout + ho) * C + c8 * 8)) = m;
    }
}

// column sums with 16-byte loads: a workgroup walks a chunk of rows; thread = (row lane, 8 columns)
__global__ __launch_bounds__(256) void colsum_vec_kernel(const uint16_t* __restrict__ x, long long rows, int cols, int ld,
                                                         float* __restrict__ out) {
    extern __shared__ float red[];                   // [row lanes][cols]
    const int tpr = cols >> 3;                       // threads per row
    const int rl = threadIdx.x / tpr, c8 = threadIdx.x - rl * tpr, nrl = blockDim.x / tpr;
    const long long chunk = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto add = [&](const uint4& v) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[2 * e] += bf16_to_f32((uint16_t)(w[e] & 0xffff));
            acc[2 * e + 1] += bf16_to_f32((uint16_t)(w[e] >> 16));
        }
    };
    if (rl < nrl) {
        // one workgroup per CU (the atomics) means four waves per CU: eight rows in flight per thread, or the loop is one
        // dependent load after the other (43 us for the model's 41 MB, 1.3 TB/s)
        constexpr int RU = 8;
        long long r = r0 + rl;
        for (; r + (long long)(RU - 1) * nrl < r1; r += (long long)RU * nrl) {
            uint4 v[RU];
#pragma unroll
            for (int q = 0; q < RU; ++q) v[q] = *reinterpret_cast<const uint4*>(x + (r + (long long)q * nrl) * ld + c8 * 8);
#pragma unroll
            for (int q = 0; q < RU; ++q) add(v[q]);
        }
        for (; r < r1; r += nrl) add(*reinterpret_cast<const uint4*>(x + r * ld + c8 * 8));
    }
    if (rl < nrl)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl * cols + c8 * 8 + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float s2 = 0.f;
        for (int q = 0; q < nrl; ++q) s2 += red[q * cols + c];
        atomicAdd(out + c, s2);
    }
}

// the same for rows wider than 256 x 8 columns (the 3000-wide logit gradient): a thread owns up to NC chunks of 8 columns
// (chunk = thread + 256 k), a workgroup a contiguous run of rows; no cross-thread reduction, one atomic per column and workgroup
template <int NC>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const uint16_t* __restrict__ x, long long rows, int cols, int ld,
                                                          float* __restrict__ out) {
    const int n8 = cols >> 3;
    const long long chunk = (rows + gridDim.x - 1) / gridDim.x;
    const long long r0 = blockIdx.x * chunk, r1 = min(rows, r0 + chunk);
    float acc[NC][8];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    constexpr int RU = 4;                            // rows in flight per thread (one 16-byte load each and column chunk)
    for (long long r = r0; r < r1; r += RU) {
        uint4 v[RU][NC];
#pragma unroll
        for (int q = 0; q < RU; ++q)
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int c8 = threadIdx.x + 256 * k;
                v[q][k] = (c8 < n8 && r + q < r1) ? *reinterpret_cast<const uint4*>(x + (r + q) * ld + c8 * 8) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
        for (int q = 0; q < RU; ++q)
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const uint32_t w[4] = {v[q][k].x, v[q][k].y, v[q][k].z, v[q][k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[k][2 * e] += bf16_to_f32((uint16_t)(w[e] & 0xffff));
                    acc[k][2 * e + 1] += bf16_to_f32((uint16_t)(w[e] >> 16));
                }
            }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c8 = threadIdx.x + 256 * k;
        if (c8 < n8)
#pragma unroll
            for (int e = 0; e < 8; ++e) atomicAdd(out + c8 * 8 + e, acc[k][e]);
    }
}

}  // namespace ew
}  // namespace asr

using namespace asr;
using namespace asr::ew;

extern "C" int asr_cast_bf16(void* stream, const float* src, void* dst, int rows, int cols, int transpose) {
    if (!src || !dst || rows <= 0 || cols <= 0) return ASR_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!transpose) {
        const long long n = (long long)rows * cols;
        if ((n & 7) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0)
            hipLaunchKernelGGL(cast_bf16_vec8_kernel, dim3(grid_for(n >> 3)), dim3(kThreads), 0, s, (const float4*)src, (uint4*)dst, n >> 3);
        else
            hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, src, (uint16_t*)dst, n);
    } else {
        hipLaunchKernelGGL(transpose_cast_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, s, src,
                           (uint16_t*)dst, rows, cols);
    }
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_cast_bf16_many(void* stream, const long long* jobs_dev, int njobs, long long total_tiles) {
    if (!jobs_dev || njobs <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffffLL) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(cast_many_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_bf16_to_f32(void* stream, const void* src, float* dst, long long n) {
    if (!src || !dst || n <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)src, dst, n);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_permute4(void* stream, const void* src, int src_bf16, void* dst, int dst_bf16, int d0, int d1, int d2,
                            int d3, long long s0, long long s1, long long s2, long long s3) {
    if (!src || !dst || d0 <= 0 || d1 <= 0 || d2 <= 0 || d3 <= 0) return ASR_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)d0 * d1 * d2 * d3;
    const dim3 g(grid_for(n)), b(kThreads);
    if (src_bf16 && dst_bf16)
        hipLaunchKernelGGL((permute4_kernel<uint16_t, uint16_t>), g, b, 0, s, (const uint16_t*)src, (uint16_t*)dst, d0, d1, d2, d3, s0, s1, s2, s3);
    else if (src_bf16)
        hipLaunchKernelGGL((permute4_kernel<uint16_t, float>), g, b, 0, s, (const uint16_t*)src, (float*)dst, d0, d1, d2, d3, s0, s1, s2, s3);
    else if (dst_bf16)
        hipLaunchKernelGGL((permute4_kernel<float, uint16_t>), g, b, 0, s, (const float*)src, (uint16_t*)dst, d0, d1, d2, d3, s0, s1, s2, s3);
    else
        hipLaunchKernelGGL((permute4_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, d0, d1, d2, d3, s0, s1, s2, s3);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_im2col(void* stream, const void* x, int x_bf16, long long sT, long long sB, long long sH, long long sC,
                          int T, int B, int Hin, int Cin, int KH, int KW, int pad_h, int pad_t, int Tout, int Kp,
                          void* col) {
    if (!x || !col || T <= 0 || B <= 0 || Hin <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || pad_h < 0 || pad_t < 0) return ASR_ERR_BAD_ARG;
    const int Hout = Hin + 2 * pad_h - KH + 1;
    if (Hout <= 0 || Kp < KH * KW * Cin || Tout <= 0 || Tout > T + 2 * pad_t - KW + 1) return ASR_ERR_BAD_ARG;
    const long long n = (long long)Tout * B * Hout * Kp;
    hipStream_t s = (hipStream_t)stream;
    if (x_bf16 && (Cin & 7) == 0 && (Kp & 7) == 0 && sC == 1 && sH == Cin && sB == (long long)Hin * Cin &&
        sT == (long long)B * Hin * Cin && ((((uintptr_t)x) | ((uintptr_t)col)) & 15) == 0 && Kp == KH * KW * Cin) {
        const long long nv = (long long)Tout * B * Hout * KH * KW * (Cin >> 3);
        hipLaunchKernelGGL(im2col_vec8_kernel, dim3(grid_for(nv)), dim3(kThreads), 0, s, (const uint16_t*)x, T, B, Hin, Cin, KH,
                           KW, pad_h, pad_t, Tout, Hout, Kp, (uint16_t*)col);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    if ((Kp & 7) == 0 && (((uintptr_t)col) & 15) == 0) {
        const long long nc = n >> 3;
        if (x_bf16)
            hipLaunchKernelGGL(im2col_chunk8_kernel<uint16_t>, dim3(grid_for(nc)), dim3(kThreads), 0, s, (const uint16_t*)x, sT, sB, sH,
                               sC, T, B, Hin, Cin, KH, KW, pad_h, pad_t, Tout, Hout, Kp, (uint4*)col);
        else
            hipLaunchKernelGGL(im2col_chunk8_kernel<float>, dim3(grid_for(nc)), dim3(kThreads), 0, s, (const float*)x, sT, sB, sH, sC,
                               T, B, Hin, Cin, KH, KW, pad_h, pad_t, Tout, Hout, Kp, (uint4*)col);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    if (x_bf16)
        hipLaunchKernelGGL(im2col_kernel<uint16_t>, dim3(grid_for(n)), dim3(kThreads), 0, s, (const uint16_t*)x, sT, sB, sH,
                           sC, T, B, Hin, Cin, KH, KW, pad_h, pad_t, Tout, Hout, Kp, (uint16_t*)col);
    else
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(n)), dim3(kThreads), 0, s, (const float*)x, sT, sB, sH, sC,
                           T, B, Hin, Cin, KH, KW, pad_h, pad_t, Tout, Hout, Kp, (uint16_t*)col);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_col2im(void* stream, const void* dcol, int T, int B, int Hin, int Cin, int KH, int KW, int pad_h,
                          int pad_t, int Tout, int Kp, void* dx) {
    if (!dcol || !dx || T <= 0 || B <= 0 || Hin <= 0 || Cin <= 0 || pad_h < 0 || pad_t < 0) return ASR_ERR_BAD_ARG;
    const int Hout = Hin + 2 * pad_h - KH + 1;
    if (Hout <= 0 || Kp < KH * KW * Cin || Tout <= 0 || Tout > T + 2 * pad_t - KW + 1) return ASR_ERR_BAD_ARG;
    const long long n = (long long)T * B * Hin * Cin;
    if ((Cin & 7) == 0 && (Kp & 7) == 0 && ((((uintptr_t)dcol) | ((uintptr_t)dx)) & 15) == 0) {
        hipLaunchKernelGGL(col2im_vec8_kernel, dim3(grid_for(n >> 3)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const uint16_t*)dcol, T, B, Hin, Cin, KH, KW, pad_h, pad_t, Tout, Hout, Kp, (uint16_t*)dx);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)dcol, T,
                       B, Hin, Cin, KH, KW, pad_h, pad_t, Tout, Hout, Kp, (uint16_t*)dx);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_maxout2_fwd(void* stream, const void* x, void* y, long long n_out) {
    if (!x || !y || n_out <= 0) return ASR_ERR_BAD_ARG;
    if ((n_out & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
        hipLaunchKernelGGL(maxout2_fwd_vec_kernel, dim3(grid_for(n_out >> 3)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const uint16_t*)x, (uint16_t*)y, n_out >> 3);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    hipLaunchKernelGGL(maxout2_fwd_kernel, dim3(grid_for(n_out)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, n_out);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_maxout2_bwd(void* stream, const void* x, const void* dy, void* dx, long long n_out) {
    if (!x || !dy || !dx || n_out <= 0) return ASR_ERR_BAD_ARG;
    if ((n_out & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15) == 0) {
        hipLaunchKernelGGL(maxout2_bwd_vec_kernel, dim3(grid_for(n_out >> 3)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, n_out >> 3);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    hipLaunchKernelGGL(maxout2_bwd_kernel, dim3(grid_for(n_out)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, n_out);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_maxout2_pool_fwd(void* stream, const void* x, void* y, long long R, int Hin, int C, int k) {
    if (!x || !y || R <= 0 || Hin <= 0 || C <= 0 || k <= 0) return ASR_ERR_BAD_ARG;
    if ((C & 7) || ((((uintptr_t)x) | ((uintptr_t)y)) & 15)) return ASR_ERR_UNSUPPORTED;
    const int Hout = (Hin + k - 1) / k;
    hipLaunchKernelGGL(maxout2_pool_fwd_kernel, dim3(grid_for(R * Hout * (C >> 3))), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, R, Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_maxout2_pool_bwd_db(void* stream, const void* x, const void* dy, void* dx, float* db, long long R, int Hin, int C, int k) {
    if (!x || !dy || !dx || R <= 0 || Hin <= 0 || C <= 0 || k <= 0) return ASR_ERR_BAD_ARG;
    if ((C & 7) || ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx)) & 15)) return ASR_ERR_UNSUPPORTED;
    if (db && (kThreads != 256 || (256 % (C >> 3)) != 0)) return ASR_ERR_UNSUPPORTED;
    const int Hout = (Hin + k - 1) / k;
    int grid = grid_for(R * Hout * (C >> 3));
    // one float atomic per channel and workgroup, all of them on the same 2 C addresses at the end of the launch: that tail, not the
    // streaming, sets the size of the grid.  T=1000, B=32, first block (H=38, 311 MB in and out) / second block (H=11) of the BASELINE model,
    // us with the bias sums (without: 127 / 32): 4096 workgroups 248 / 213, 2048 190 / 121, 1024 166 / 78, 768 154 / 66, 512 157 / 58,
    // 384 190 / 58, 256 245 / 65 (round 3 measured 768 as starving the streaming part, with the 16-way conflicted reduction image)
    if (db && grid > 512) grid = 512;
    hipLaunchKernelGGL(maxout2_pool_bwd_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, db, R, Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_maxout2_pool_bwd(void* stream, const void* x, const void* dy, void* dx, long long R, int Hin, int C, int k) {
    return asr_maxout2_pool_bwd_db(stream, x, dy, dx, nullptr, R, Hin, C, k);
}
extern "C" int asr_maxout2_pool_bwd_db_ok(int C) { return kThreads == 256 && (C & 7) == 0 && (256 % (C >> 3)) == 0; }

extern "C" int asr_maxpool_h_fwd(void* stream, const void* x, void* y, long long R, int Hin, int C, int k) {
    if (!x || !y || R <= 0 || Hin <= 0 || C <= 0 || k <= 0) return ASR_ERR_BAD_ARG;
    const int Hout = Hin <= k ? 1 : cdiv(Hin - k, k) + 1;     // cover_all = True, stride = k
    const long long n = R * Hout * C;
    if ((C & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
        hipLaunchKernelGGL(maxpool_h_fwd_vec_kernel, dim3(grid_for(n >> 3)), dim3(kThreads), 0, (hipStream_t)stream,
                           (const uint16_t*)x, (uint16_t*)y, R, Hin, Hout, C, k);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    hipLaunchKernelGGL(maxpool_h_fwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, R, Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_maxpool_h_bwd(void* stream, const void* x, const void* dy, void* dx, long long R, int Hin, int C,
                                 int k) {
    if (!x || !dy || !dx || R <= 0 || Hin <= 0 || C <= 0 || k <= 0) return ASR_ERR_BAD_ARG;
    const int Hout = Hin <= k ? 1 : cdiv(Hin - k, k) + 1;
    const long long n = R * Hout * C;
    hipLaunchKernelGGL(maxpool_h_bwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, R, Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_add_bf16(void* stream, const void* a, const void* b, void* y, long long n) {
    if (!a || !b || !y || n <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)a,
                       (const uint16_t*)b, (uint16_t*)y, n);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_colsum_acc(void* stream, const void* x, int x_bf16, long long rows, int cols, int ld, float* out) {
    if (!x || !out || rows <= 0 || cols <= 0 || ld < cols) return ASR_ERR_BAD_ARG;
    if (x_bf16 && (cols & 7) == 0 && cols <= 2048 && (ld & 7) == 0 && (((uintptr_t)x) & 15) == 0) {
        const int tpr = cols >> 3;
        const int nrl = 256 / tpr;
        if (nrl >= 1) {
            // (rows / 512 left most of the chip idle on the model's (32000, 640) gradients: 63 workgroups, 118 us for 41 MB)
            // and one float atomic per column and workgroup serialises per address (~12 ns each): one workgroup per CU
            int blocks = (int)((rows + 31) / 32);
            if (blocks > 256) blocks = 256;
            if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(colsum_vec_kernel, dim3(blocks), dim3(256), sizeof(float) * nrl * cols, (hipStream_t)stream,
                               (const uint16_t*)x, rows, cols, ld, out);
            ASR_LAUNCH_CHECK();
            return ASR_OK;
        }
    }
    if (x_bf16 && (cols & 7) == 0 && cols <= 4 * 256 * 8 && (ld & 7) == 0 && (((uintptr_t)x) & 15) == 0) {
        int blocks = (int)((rows + 31) / 32);
        if (blocks > 512) blocks = 512;
        const int nc = cdiv(cols >> 3, 256);
        if (nc <= 2) hipLaunchKernelGGL(colsum_wide_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, rows, cols, ld, out);
        else hipLaunchKernelGGL(colsum_wide_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, rows, cols, ld, out);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    int chunks = (int)((rows + 255) / 256);
    if (chunks > 128) chunks = 128;
    const dim3 g(cdiv(cols, 64), chunks);
    if (x_bf16)
        hipLaunchKernelGGL(colsum_kernel<uint16_t>, g, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, rows, cols, ld, out);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, g, dim3(256), 0, (hipStream_t)stream, (const float*)x, rows, cols, ld, out);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_activation_fwd(void* stream, const void* x, void* y, long long n, int kind, float alpha) {
    if (!x || !y || n <= 0 || kind < 0 || kind > 7) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (uint16_t*)y, n, kind, alpha);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_activation_bwd(void* stream, const void* x, const void* dy, void* dx, long long n, int kind,
                                  float alpha) {
    if (!x || !dy || !dx || n <= 0 || kind < 0 || kind > 7) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (const uint16_t*)dy, (uint16_t*)dx, n, kind, alpha);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_glu_fwd(void* stream, const void* x, void* y, long long rows, int C) {
    if (!x || !y || rows <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for(rows * C)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, rows, C);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_glu_bwd(void* stream, const void* x, const void* dy, void* dx, long long rows, int C) {
    if (!x || !dy || !dx || rows <= 0 || C <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for(rows * C)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, rows, C);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_dropout(void* stream, const void* x, void* y, long long n, float ratio, unsigned int seed) {
    if (!x || !y || n <= 0 || ratio < 0.f || ratio >= 1.f) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (uint16_t*)y, n, ratio, seed);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_conv_weight_pack(void* stream, const float* W, void* dst, int Co, int Ci, int KH, int KW, int Kp,
                                    int transpose) {
    if (!W || !dst || Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0 || Kp < KH * KW * Ci) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(grid_for((long long)Co * Kp)), dim3(kThreads), 0, (hipStream_t)stream,
                       W, (uint16_t*)dst, Co, Ci, KH, KW, Kp, transpose);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_conv_weight_pack_bwd(void* stream, const float* W, void* dst, int Co, int Ci, int KH, int KW) {
    if (!W || !dst || Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(conv_weight_pack_bwd_kernel, dim3(grid_for((long long)Co * Ci * KH * KW)), dim3(kThreads), 0, (hipStream_t)stream,
                       W, (uint16_t*)dst, Co, Ci, KH, KW);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_conv_weight_grad_unpack_copies(void* stream, const float* scratch, int copies, float* gW, int Co, int Ci, int KH,
                                                  int KW, int Kp, int Cs) {
    if (Cs <= 0) Cs = Ci;
    if (!scratch || !gW || Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0 || Cs < Ci || Kp < KH * KW * Cs || copies < 1 || copies > 8)
        return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(conv_weight_grad_unpack_kernel, dim3(grid_for((long long)Co * Ci * KH * KW)), dim3(kThreads), 0,
                       (hipStream_t)stream, scratch, gW, Co, Ci, KH, KW, Kp, Cs, copies);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_conv_weight_grad_unpack(void* stream, const float* scratch, float* gW, int Co, int Ci, int KH, int KW,
                                           int Kp, int Cs) {
    return asr_conv_weight_grad_unpack_copies(stream, scratch, 1, gW, Co, Ci, KH, KW, Kp, Cs);
}

extern "C" int asr_pack_input_pad(void* stream, const void* x, int x_bf16, long long sT, long long sB, long long sH, long long sC,
                                  int T, int B, int H, int C, int Cpad, void* out_bf16) {
    if (!x || !out_bf16 || T <= 0 || B <= 0 || H <= 0 || C <= 0 || Cpad < C) return ASR_ERR_BAD_ARG;
    const long long n = (long long)T * B * H * Cpad;
    if (!x_bf16 && sT == 1 && Cpad == 8 && C <= 8 && B <= 65535 && (H + 7) / 8 <= 65535 && (((uintptr_t)out_bf16) & 15) == 0) {
        hipLaunchKernelGGL(pack_input_pad_time_kernel, dim3((T + 63) / 64, B, (H + 7) / 8), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           sB, sH, sC, T, B, H, C, (uint16_t*)out_bf16);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    if (x_bf16)
        hipLaunchKernelGGL(pack_input_pad_kernel<uint16_t>, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x,
                           sT, sB, sH, sC, T, B, H, C, Cpad, (uint16_t*)out_bf16);
    else
        hipLaunchKernelGGL(pack_input_pad_kernel<float>, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, (const float*)x, sT,
                           sB, sH, sC, T, B, H, C, Cpad, (uint16_t*)out_bf16);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
