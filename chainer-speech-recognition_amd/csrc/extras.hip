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
// chainer.functions.upsampling_2d (asr/nn/nn.py:135-146): the inverse of a max pooling given the pooling's argmax positions
// (`indexes` of a MaxPooling2D function object: the window-local position of every maximum; ksize (k, 1), stride = ksize here, so the
// position is the row inside the window).  maxpool_h_indexes forms them (first maximum, as max_pooling_2d routes its gradient);
// upsample: y[r][h k + j][c] = x[r][h][c] where j = idx[r][h][c], zero elsewhere; backward: dx[r][h][c] = dy[r][h k + idx][c].
__global__ void maxpool_h_indexes_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ idx, long long R, int Hin, int Hout, int C, int k) {
    const long long n = R * Hout * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ho = (int)((i / C) % Hout);
        const long long r = i / ((long long)C * Hout);
        float m = -INFINITY;
        int best = 0;
        for (int j = 0; j < k; ++j) {
            const int h = ho * k + j;
            if (h >= Hin) break;
            const float v = bf16_to_f32(x[(r * Hin + h) * C + c]);
            if (v > m) { m = v; best = j; }
        }
        idx[i] = (uint8_t)best;
    }
}
__global__ void upsample_h_fwd_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ idx, uint16_t* __restrict__ y, long long R, int Hin,
                                      int Hout, int C, int k) {
    const long long n = R * Hout * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ho = (int)((i / C) % Hout);
        const long long r = i / ((long long)C * Hout);
        const long long src = (r * Hin + ho / k) * C + c;
        y[i] = (int)idx[src] == ho % k ? x[src] : (uint16_t)0;
    }
}
__global__ void upsample_h_bwd_kernel(const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx, uint16_t* __restrict__ dx, long long R,
                                      int Hin, int Hout, int C, int k) {
    const long long n = R * Hin * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int h = (int)((i / C) % Hin);
        const long long r = i / ((long long)C * Hin);
        const int ho = h * k + (int)idx[i];
        dx[i] = ho < Hout ? dy[(r * Hout + ho) * C + c] : (uint16_t)0;
    }
}

// chainer.functions.spatial_pyramid_pooling_2d (asr/nn/nn.py:115-121), max pooling: level l cuts the (H, T) plane of every (utterance,
// channel) into 2^l x 2^l bins -- window (ceil(H / 2^l), ceil(T / 2^l)), stride = window, padded by ((2^l kh - H + 1) / 2, likewise in
// time) with -inf, as Chainer's MaxPooling2D(ksize, pad, cover_all) does -- and keeps the maximum of every bin.  Output row b holds,
// level after level, [c][by][bx] (the reference's reshape to (B, C 4^l, 1, 1) and concat along axis 1): here (B, sum_l 4^l, C) with the
// channels innermost, which the wrapper permutes.  x is the physical (T, B, H, C) tensor.  One thread per (b, bin, c): channels run
// across lanes, so every load of the window sweep is a coalesced row segment.  pos (optional) receives the flat (h T + t) position
// of the first maximum in (h, t) row-major order -- Chainer's argmax over the flattened window -- for the backward scatter.
struct SppBin { int level, by, bx, kh, kw, ph, pt; };
__device__ __forceinline__ SppBin spp_bin(int bin, int H, int T) {
    SppBin s;
    int l = 0, base = 0;
    while (bin >= base + (1 << (2 * l))) { base += 1 << (2 * l); ++l; }
    const int nb = 1 << l, o = bin - base;
    s.level = l; s.by = o / nb; s.bx = o - s.by * nb;
    s.kh = (H + nb - 1) / nb; s.kw = (T + nb - 1) / nb;
    s.ph = (nb * s.kh - H + 1) / 2; s.pt = (nb * s.kw - T + 1) / 2;
    return s;
}
__global__ void spp_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int* __restrict__ pos, int T, int B, int H, int C, int bins) {
    const long long n = (long long)B * bins * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int bin = (int)((i / C) % bins);
        const int b = (int)(i / ((long long)C * bins));
        const SppBin s = spp_bin(bin, H, T);
        const int h0 = max(0, s.by * s.kh - s.ph), h1 = min(H, s.by * s.kh - s.ph + s.kh);
        const int t0 = max(0, s.bx * s.kw - s.pt), t1 = min(T, s.bx * s.kw - s.pt + s.kw);
        float m = -INFINITY;
        int best = -1;
        for (int h = h0; h < h1; ++h)
            for (int t = t0; t < t1; ++t) {
                const float v = bf16_to_f32(x[(((long long)t * B + b) * H + h) * C + c]);
                if (v > m) { m = v; best = h * T + t; }
            }
        y[i] = f32_to_bf16(m);
        if (pos) pos[i] = best;
    }
}
// dx32 (T, B, H, C) float32, zeroed by the caller: several levels route their gradient to the same element
__global__ void spp_bwd_kernel(const uint16_t* __restrict__ dy, const int* __restrict__ pos, float* __restrict__ dx32, int T, int B, int H, int C, int bins) {
    const long long n = (long long)B * bins * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int p = pos[i];
        if (p < 0) continue;
        const int c = (int)(i % C);
        const int b = (int)(i / ((long long)C * bins));
        const int h = p / T, t = p - h * T;
        atomicAdd(dx32 + (((long long)t * B + b) * H + h) * C + c, bf16_to_f32(dy[i]));
    }
}

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

extern "C" int asr_maxpool_h_indexes(void* stream, const void* x, void* idx_u8, long long R, int Hin, int C, int k) {
    if (!x || !idx_u8 || R <= 0 || Hin <= 0 || C <= 0 || k <= 0 || k > 255) return ASR_ERR_BAD_ARG;
    const int Hout = Hin <= k ? 1 : cdiv(Hin - k, k) + 1;       // cover_all, stride = k
    hipLaunchKernelGGL(maxpool_h_indexes_kernel, dim3(grid_for(R * Hout * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint8_t*)idx_u8,
                       R, Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_upsample_h_fwd(void* stream, const void* x, const void* idx_u8, void* y, long long R, int Hin, int Hout, int C, int k) {
    if (!x || !idx_u8 || !y || R <= 0 || Hin <= 0 || C <= 0 || k <= 0 || Hout <= 0 || Hout > Hin * k) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(upsample_h_fwd_kernel, dim3(grid_for(R * Hout * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (const uint8_t*)idx_u8, (uint16_t*)y, R, Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_upsample_h_bwd(void* stream, const void* dy, const void* idx_u8, void* dx, long long R, int Hin, int Hout, int C, int k) {
    if (!dy || !idx_u8 || !dx || R <= 0 || Hin <= 0 || C <= 0 || k <= 0 || Hout <= 0 || Hout > Hin * k) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(upsample_h_bwd_kernel, dim3(grid_for(R * Hin * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)dy,
                       (const uint8_t*)idx_u8, (uint16_t*)dx, R, Hin, Hout, C, k);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_spp_bins(int pyramid_height) {
    if (pyramid_height <= 0 || pyramid_height > 8) return -1;
    int n = 0;
    for (int l = 0; l < pyramid_height; ++l) n += 1 << (2 * l);
    return n;
}
extern "C" int asr_spp_fwd(void* stream, const void* x, void* y, int* pos, int T, int B, int H, int C, int pyramid_height) {
    const int bins = asr_spp_bins(pyramid_height);
    if (!x || !y || T <= 0 || B <= 0 || H <= 0 || C <= 0 || bins <= 0 || (long long)H * T > 0x7fffffffLL) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(spp_fwd_kernel, dim3(grid_for((long long)B * bins * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y,
                       pos, T, B, H, C, bins);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
extern "C" int asr_spp_bwd(void* stream, const void* dy, const int* pos, float* dx32, int T, int B, int H, int C, int pyramid_height) {
    const int bins = asr_spp_bins(pyramid_height);
    if (!dy || !pos || !dx32 || T <= 0 || B <= 0 || H <= 0 || C <= 0 || bins <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(spp_bwd_kernel, dim3(grid_for((long long)B * bins * C)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)dy, pos, dx32,
                       T, B, H, C, bins);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
