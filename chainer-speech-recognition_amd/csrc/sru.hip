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

// ------------------------------------------------------------------------------------------------ chunked scans
// The kernels above are the reference's shape: one thread per (b, d) column walks all T steps -- B D / 64 waves (256 at B = 32,
// D = 512: one per CU), every step a dependent chain of global loads, ~2 % of the HBM roofline.  But the cell recurrence is LINEAR
// in c:   c_t = f_t c_{t-1} + (1 - f_t) z_t,   and so is the backward one:   gc_{t-1} = f_t (gc_t + gct_t)   with f, z, gct
// functions of the saved arrays only.  So time is cut into NC chunks:
//   summary pass  every (chunk, column) thread reduces its chunk to (P, S) with  state_out = P state_in + S  (reads f, z: 8 B of
//                 the 20 B per element forward; f, r, C, gH: 14 of 28 backward),
//   apply pass    every (chunk, column) thread folds the summaries of the chunks in front of it into its entry state (<= NC - 1
//                 fused multiply-adds) and runs the reference's loop over its chunk, writing the outputs.
// NC x more waves in flight (>= 8 per CU), two adjacent features per lane (8-B U / C accesses, packed bf16 x / H / gU), loops
// unrolled four steps so that a lane has ~100 B of loads in flight; sigmoid / tanh through v_exp_f32 + v_rcp_f32.  Bias gradients:
// registers -> LDS over the batch rows of a workgroup -> one atomic per (workgroup, column).
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    return copysignf(1.0f - 2.0f * e * __builtin_amdgcn_rcpf(1.0f + e), x);
}
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 bf2(const uint16_t* p) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
    return make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
}

constexpr int SX = 64, SY = 4;        // workgroup: 64 feature pairs x 4 batch rows

struct Col {
    int b, d;
    bool ok;
};
__device__ __forceinline__ Col my_column(int B, int D) {
    Col c;
    c.d = 2 * (blockIdx.x * SX + threadIdx.x);
    c.b = blockIdx.y * SY + threadIdx.y;
    c.ok = c.d < D && c.b < B;
    return c;
}

// forward summaries: (P, S) of chunk k = blockIdx.z (the last chunk needs none)
__global__ __launch_bounds__(SX * SY) void fwd_summary_kernel(const float* __restrict__ U, const float* __restrict__ bias, float4* __restrict__ ws,
                                                              int T, int B, int D, int Tc) {
    const Col c = my_column(B, D);
    if (!c.ok) return;
    const int k = blockIdx.z, t0 = k * Tc, t1 = min(T, t0 + Tc);
    const float2 bf = ld2(bias + c.d);
    const size_t us = (size_t)B * 3 * D;
    const float* u = U + ((size_t)t0 * B + c.b) * 3 * D + c.d;
    float2 P = make_float2(1.f, 1.f), S = make_float2(0.f, 0.f);
#pragma unroll 4
    for (int t = t0; t < t1; ++t, u += us) {
        const float2 z = ld2(u), fp = ld2(u + D);
        const float fx = fsig(fp.x + bf.x), fy = fsig(fp.y + bf.y);
        P.x *= fx; P.y *= fy;
        S.x = fx * (S.x - z.x) + z.x;
        S.y = fy * (S.y - z.y) + z.y;
    }
    ws[((size_t)k * B + c.b) * (D / 2) + (c.d >> 1)] = make_float4(P.x, P.y, S.x, S.y);
}

template <bool TANH>
__global__ __launch_bounds__(SX * SY) void fwd_apply_kernel(const uint16_t* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
                                                            const float* __restrict__ c0, const float* __restrict__ mask, const float4* __restrict__ ws,
                                                            uint16_t* __restrict__ H, float* __restrict__ C, float* __restrict__ cT, int T, int B, int D,
                                                            int Tc, int NC) {
    const Col cl = my_column(B, D);
    if (!cl.ok) return;
    const int k = blockIdx.z, t0 = k * Tc, t1 = min(T, t0 + Tc);
    const size_t col = (size_t)cl.b * D + cl.d;
    const float2 bf = ld2(bias + cl.d), br = ld2(bias + D + cl.d);
    const float2 mk = mask ? ld2(mask + col) : make_float2(1.f, 1.f);
    float2 c = c0 ? ld2(c0 + col) : make_float2(0.f, 0.f);
    for (int kk = 0; kk < k; ++kk) {            // entry state of this chunk
        const float4 ps = ws[((size_t)kk * B + cl.b) * (D / 2) + (cl.d >> 1)];
        c.x = ps.x * c.x + ps.z;
        c.y = ps.y * c.y + ps.w;
    }
    const size_t us = (size_t)B * 3 * D, xs = (size_t)B * D;
    const float* u = U + ((size_t)t0 * B + cl.b) * 3 * D + cl.d;
    size_t row = (size_t)t0 * xs + col;
#pragma unroll 4
    for (int t = t0; t < t1; ++t, u += us, row += xs) {
        const float2 z = ld2(u), fp = ld2(u + D), rp = ld2(u + 2 * D);
        const float2 xr = bf2(x + row);
        const float fx = fsig(fp.x + bf.x), fy = fsig(fp.y + bf.y);
        const float rx = fsig(rp.x + br.x), ry = fsig(rp.y + br.y);
        const float xx = xr.x * mk.x, xy = xr.y * mk.y;
        c.x = fx * (c.x - z.x) + z.x;
        c.y = fy * (c.y - z.y) + z.y;
        *reinterpret_cast<float2*>(C + row) = c;
        const float gx = TANH ? ftanh(c.x) : c.x, gy = TANH ? ftanh(c.y) : c.y;
        *reinterpret_cast<uint32_t*>(H + row) = pack_bf16x2(rx * (gx - xx) + xx, ry * (gy - xy) + xy);
    }
    if (k == NC - 1) *reinterpret_cast<float2*>(cT + col) = c;
}

// backward summaries: gc entering chunk k - 1 (from the right) = P gc_in + S, chunk k = blockIdx.z + 1 (chunk 0 needs none)
template <bool TANH>
__global__ __launch_bounds__(SX * SY) void bwd_summary_kernel(const float* __restrict__ U, const float* __restrict__ bias, const float* __restrict__ C,
                                                              const uint16_t* __restrict__ gH, float4* __restrict__ ws, int T, int B, int D, int Tc) {
    const Col cl = my_column(B, D);
    if (!cl.ok) return;
    const int k = blockIdx.z + 1, t0 = k * Tc, t1 = min(T, t0 + Tc);
    const size_t col = (size_t)cl.b * D + cl.d;
    const float2 bf = ld2(bias + cl.d), br = ld2(bias + D + cl.d);
    const size_t us = (size_t)B * 3 * D, xs = (size_t)B * D;
    const float* u = U + ((size_t)(t1 - 1) * B + cl.b) * 3 * D + cl.d;
    size_t row = (size_t)(t1 - 1) * xs + col;
    float2 P = make_float2(1.f, 1.f), S = make_float2(0.f, 0.f);
    // a missing gH (set_materialize_grads(False): only cT received a gradient): the load stays unconditional (any valid row; no branch in
    // the loop) and the VALUE is selected afterwards -- scaling by 0 would turn the Inf / NaN bit patterns those bytes may hold into NaN
    const uint16_t* ghp = gH ? gH : reinterpret_cast<const uint16_t*>(C);
    const bool has_gh = gH != nullptr;
#pragma unroll 4
    for (int t = t1 - 1; t >= t0; --t, u -= us, row -= xs) {
        const float2 fp = ld2(u + D), rp = ld2(u + 2 * D), c = ld2(C + row);
        float2 gh = bf2(ghp + row);
        gh = has_gh ? gh : make_float2(0.f, 0.f);           // v_cndmask on a kernel-uniform predicate
        const float fx = fsig(fp.x + bf.x), fy = fsig(fp.y + bf.y);
        const float rx = fsig(rp.x + br.x), ry = fsig(rp.y + br.y);
        const float gx = TANH ? ftanh(c.x) : c.x, gy = TANH ? ftanh(c.y) : c.y;
        const float gctx = gh.x * rx * (TANH ? (1.f - gx * gx) : 1.f), gcty = gh.y * ry * (TANH ? (1.f - gy * gy) : 1.f);
        S.x = (gctx + S.x) * fx;
        S.y = (gcty + S.y) * fy;
        P.x *= fx; P.y *= fy;
    }
    ws[((size_t)k * B + cl.b) * (D / 2) + (cl.d >> 1)] = make_float4(P.x, P.y, S.x, S.y);
}

template <bool TANH>
__global__ __launch_bounds__(SX * SY) void bwd_apply_kernel(const uint16_t* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias,
                                                            const float* __restrict__ C, const float* __restrict__ c0, const float* __restrict__ mask,
                                                            const uint16_t* __restrict__ gH, const float* __restrict__ gcT, const float4* __restrict__ ws,
                                                            uint16_t* __restrict__ gU, uint16_t* __restrict__ gxh, float* __restrict__ gbias,
                                                            float* __restrict__ gc0, int T, int B, int D, int Tc, int NC) {
    __shared__ float red[SY][2][2 * SX];
    const Col cl = my_column(B, D);
    const int k = blockIdx.z, t0 = k * Tc, t1 = min(T, t0 + Tc);
    float sbfx = 0.f, sbfy = 0.f, sbrx = 0.f, sbry = 0.f;
    if (cl.ok) {
        const size_t col = (size_t)cl.b * D + cl.d;
        const float2 bf = ld2(bias + cl.d), br = ld2(bias + D + cl.d);
        const float2 mk = mask ? ld2(mask + col) : make_float2(1.f, 1.f);
        const float2 cinit = c0 ? ld2(c0 + col) : make_float2(0.f, 0.f);
        float2 gc = gcT ? ld2(gcT + col) : make_float2(0.f, 0.f);
        for (int kk = NC - 1; kk > k; --kk) {       // gc entering this chunk from the right
            const float4 ps = ws[((size_t)kk * B + cl.b) * (D / 2) + (cl.d >> 1)];
            gc.x = ps.x * gc.x + ps.z;
            gc.y = ps.y * gc.y + ps.w;
        }
        const size_t us = (size_t)B * 3 * D, xs = (size_t)B * D;
        const float* u = U + ((size_t)(t1 - 1) * B + cl.b) * 3 * D + cl.d;
        uint16_t* gu = gU + ((size_t)(t1 - 1) * B + cl.b) * 3 * D + cl.d;
        size_t row = (size_t)(t1 - 1) * xs + col;
        float2 c = ld2(C + row);
        // No branch inside the loop: gfx9 counts loads and stores in ONE in-order counter, and behind a branch the compiler waits for
        // vmcnt(0) -- every step then waited for the loads it had just issued (the unrolled steps' loads no longer overlapped).  The
        // t == 0 state and a missing gH are selected AFTER an always-valid load (clamped address / the x row).
        const uint16_t* ghp = gH ? gH : x;
        const bool has_gh = gH != nullptr;
        // software pipeline: the operands of step t - 1 are asked for before step t is worked on (the last step asks for its own row
        // again: always a valid address), so that a lane has two steps of loads in flight across the transcendental chain of a step
        struct In { float2 z, fp, rp, cp; uint32_t xw, gw; };
        auto ask = [&](const float* uu, size_t rr, int tt) {
            In in;
            in.z = ld2(uu); in.fp = ld2(uu + D); in.rp = ld2(uu + 2 * D);
            in.xw = *reinterpret_cast<const uint32_t*>(x + rr);
            in.gw = *reinterpret_cast<const uint32_t*>(ghp + rr);
            in.cp = ld2(C + (tt == 0 ? rr : rr - xs));
            return in;
        };
        In cur = ask(u, row, t1 - 1);
#pragma unroll 2
        for (int t = t1 - 1; t >= t0; --t, u -= us, gu -= us, row -= xs) {
            const bool more = t > t0;
            const In nxt = ask(more ? u - us : u, more ? row - xs : row, more ? t - 1 : t);
            const float2 z = cur.z, fp = cur.fp, rp = cur.rp;
            const float2 xr = make_float2(__uint_as_float(cur.xw << 16), __uint_as_float(cur.xw & 0xffff0000u));
            float2 gh = make_float2(__uint_as_float(cur.gw << 16), __uint_as_float(cur.gw & 0xffff0000u));
            gh = has_gh ? gh : make_float2(0.f, 0.f);       // selected, never scaled: x may hold Inf / NaN patterns (0 * Inf = NaN)
            float2 cp = cur.cp;
            if (t == 0) cp = cinit;
            const float fx = fsig(fp.x + bf.x), fy = fsig(fp.y + bf.y);
            const float rx = fsig(rp.x + br.x), ry = fsig(rp.y + br.y);
            const float xx = xr.x * mk.x, xy = xr.y * mk.y;
            const float gx = TANH ? ftanh(c.x) : c.x, gy = TANH ? ftanh(c.y) : c.y;
            const float gbrx = gh.x * (gx - xx) * (1.f - rx) * rx, gbry = gh.y * (gy - xy) * (1.f - ry) * ry;      // asr/nn/sru.py:158
            const float gctx = gh.x * rx * (TANH ? (1.f - gx * gx) : 1.f), gcty = gh.y * ry * (TANH ? (1.f - gy * gy) : 1.f);   // :162
            const float tx = gctx + gc.x, ty = gcty + gc.y;
            const float gbfx = tx * (cp.x - z.x) * (1.f - fx) * fx, gbfy = ty * (cp.y - z.y) * (1.f - fy) * fy;    // :163
            *reinterpret_cast<uint32_t*>(gxh + row) = pack_bf16x2(gh.x * (1.f - rx), gh.y * (1.f - ry));           // :166
            *reinterpret_cast<uint32_t*>(gu) = pack_bf16x2(tx * (1.f - fx), ty * (1.f - fy));                      // :169
            *reinterpret_cast<uint32_t*>(gu + D) = pack_bf16x2(gbfx, gbfy);                                         // :170
            *reinterpret_cast<uint32_t*>(gu + 2 * D) = pack_bf16x2(gbrx, gbry);                                     // :171
            gc.x = tx * fx; gc.y = ty * fy;                                                                         // :174
            sbfx += gbfx; sbfy += gbfy; sbrx += gbrx; sbry += gbry;
            c = cp;
            cur = nxt;
        }
        if (k == 0) *reinterpret_cast<float2*>(gc0 + col) = gc;
    }
    red[threadIdx.y][0][2 * threadIdx.x] = sbfx; red[threadIdx.y][0][2 * threadIdx.x + 1] = sbfy;
    red[threadIdx.y][1][2 * threadIdx.x] = sbrx; red[threadIdx.y][1][2 * threadIdx.x + 1] = sbry;
    __syncthreads();
    if (threadIdx.y < 2) {          // rows 0 / 1 of the thread block fold the b_f / b_r sums of the SY batch rows
        const int which = threadIdx.y, d = 2 * (blockIdx.x * SX + threadIdx.x);
        if (d < D) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int y = 0; y < SY; ++y) { a0 += red[y][which][2 * threadIdx.x]; a1 += red[y][which][2 * threadIdx.x + 1]; }
            atomicAdd(gbias + which * D + d, a0);
            atomicAdd(gbias + which * D + d + 1, a1);
        }
    }
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

// number of time chunks of the chunked scans (0: the one-thread-per-column kernels serve): enough waves for >= 8 per CU, chunks
// of at least 16 steps; odd D or misaligned operands fall back
static int sru_chunks(int T, int B, int D) {
    if ((D & 1) || T < 32) return 0;
    const long long waves = ((long long)B * (D / 2) + 63) / 64;
    long long nc = (2048 + waves - 1) / waves;
    if (nc > 32) nc = 32;
    if (nc > T / 16) nc = T / 16;
    if (nc < 2) return 0;
    const int Tc = (int)((T + nc - 1) / nc);
    return (T + Tc - 1) / Tc;           // no empty chunk at the end
}

extern "C" size_t asr_sru_ws_bytes(int T, int B, int D) {
    const int nc = sru_chunks(T, B, D);
    return nc ? (size_t)nc * B * (D / 2) * sizeof(float4) : 0;
}

static bool aligned8(const void* p) { return (((uintptr_t)p) & 7) == 0; }

extern "C" int asr_sru_fwd(void* stream, const void* x_bf16, const float* U, const float* bias, const float* c0,
                           const float* mask, void* H_bf16, float* C, float* cT, int T, int B, int D, int use_tanh, void* ws,
                           size_t ws_bytes) {
    if (ASR_ACT_IS_F16) return ASR_ERR_UNSUPPORTED;      // (bfloat16 bits handled directly: common.hpp)
    if (!x_bf16 || !U || !bias || !H_bf16 || !C || !cT || T <= 0 || B <= 0 || D <= 0) return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nc = sru_chunks(T, B, D);
    if (nc && ws && ws_bytes >= asr_sru_ws_bytes(T, B, D) && aligned8(U) && aligned8(C) && aligned8(bias) && aligned8(cT) && (!c0 || aligned8(c0)) &&
        (!mask || aligned8(mask)) && (((uintptr_t)ws) & 15) == 0 && (((uintptr_t)x_bf16 | (uintptr_t)H_bf16) & 3) == 0) {
        const int Tc = (T + nc - 1) / nc;
        const dim3 block(SX, SY), grid((D / 2 + SX - 1) / SX, (B + SY - 1) / SY, nc);
        hipLaunchKernelGGL(fwd_summary_kernel, dim3(grid.x, grid.y, nc - 1), block, 0, st, U, bias, (float4*)ws, T, B, D, Tc);
        if (use_tanh)
            hipLaunchKernelGGL(fwd_apply_kernel<true>, grid, block, 0, st, (const uint16_t*)x_bf16, U, bias, c0, mask, (const float4*)ws,
                               (uint16_t*)H_bf16, C, cT, T, B, D, Tc, nc);
        else
            hipLaunchKernelGGL(fwd_apply_kernel<false>, grid, block, 0, st, (const uint16_t*)x_bf16, U, bias, c0, mask, (const float4*)ws,
                               (uint16_t*)H_bf16, C, cT, T, B, D, Tc, nc);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    hipLaunchKernelGGL(fwd_kernel, dim3(cdiv((long long)B * D, 256)), dim3(256), 0, st,
                       (const uint16_t*)x_bf16, U, bias, c0, mask, (uint16_t*)H_bf16, C, cT, T, B, D, use_tanh);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sru_bwd(void* stream, const void* x_bf16, const float* U, const float* bias, const float* C,
                           const float* c0, const float* mask, const void* gH_bf16, const float* gcT, void* gU_bf16,
                           void* gxh_bf16, float* gbias, float* gc0, int T, int B, int D, int use_tanh, void* ws, size_t ws_bytes) {
    if (ASR_ACT_IS_F16) return ASR_ERR_UNSUPPORTED;      // (bfloat16 bits handled directly: common.hpp)
    if (!x_bf16 || !U || !bias || !C || !gU_bf16 || !gxh_bf16 || !gbias || !gc0 || T <= 0 || B <= 0 || D <= 0)
        return ASR_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nc = sru_chunks(T, B, D);
    if (nc && ws && ws_bytes >= asr_sru_ws_bytes(T, B, D) && aligned8(U) && aligned8(C) && aligned8(bias) && aligned8(gc0) && (!c0 || aligned8(c0)) &&
        (!mask || aligned8(mask)) && (!gcT || aligned8(gcT)) && (((uintptr_t)ws) & 15) == 0 &&
        (((uintptr_t)x_bf16 | (uintptr_t)gU_bf16 | (uintptr_t)gxh_bf16 | (uintptr_t)gH_bf16) & 3) == 0) {
        const int Tc = (T + nc - 1) / nc;
        const dim3 block(SX, SY), grid((D / 2 + SX - 1) / SX, (B + SY - 1) / SY, nc);
        if (use_tanh) {
            hipLaunchKernelGGL(bwd_summary_kernel<true>, dim3(grid.x, grid.y, nc - 1), block, 0, st, U, bias, C, (const uint16_t*)gH_bf16, (float4*)ws, T, B, D, Tc);
            hipLaunchKernelGGL(bwd_apply_kernel<true>, grid, block, 0, st, (const uint16_t*)x_bf16, U, bias, C, c0, mask, (const uint16_t*)gH_bf16, gcT,
                               (const float4*)ws, (uint16_t*)gU_bf16, (uint16_t*)gxh_bf16, gbias, gc0, T, B, D, Tc, nc);
        } else {
            hipLaunchKernelGGL(bwd_summary_kernel<false>, dim3(grid.x, grid.y, nc - 1), block, 0, st, U, bias, C, (const uint16_t*)gH_bf16, (float4*)ws, T, B, D, Tc);
            hipLaunchKernelGGL(bwd_apply_kernel<false>, grid, block, 0, st, (const uint16_t*)x_bf16, U, bias, C, c0, mask, (const uint16_t*)gH_bf16, gcT,
                               (const float4*)ws, (uint16_t*)gU_bf16, (uint16_t*)gxh_bf16, gbias, gc0, T, B, D, Tc, nc);
        }
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    hipLaunchKernelGGL(bwd_kernel, dim3(cdiv((long long)B * D, 256)), dim3(256), 0, st,
                       (const uint16_t*)x_bf16, U, bias, C, c0, mask, (const uint16_t*)gH_bf16, gcT, (uint16_t*)gU_bf16,
                       (uint16_t*)gxh_bf16, gbias, gc0, T, B, D, use_tanh);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sru_combine(void* stream, const void* a_bf16, const void* b_bf16, const float* mask, void* out_bf16,
                               long long n, int BD) {
    if (ASR_ACT_IS_F16) return ASR_ERR_UNSUPPORTED;      // (bfloat16 bits handled directly: common.hpp)
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
