// CTC and Gram-CTC loss + gradient for gfx950.
//
// Replaces (reference file:line):
//   Chainer F.connectionist_temporal_classification   call sites run/ctc/cnn/train.py:162,191
//   asr/loss/gram_ctc.py:219-297 (GramCTC.forward/backward), :142-178 (alpha/beta), :180-217 (label prob)
//
// The reference multiplies dense (B, N, N) log connection matrices every time step; the lattice only has
// the diagonals k in {0,1,2} (CTC) / {0,1,2,3,5,6,7} (Gram-CTC), so each node reads <= 7 neighbours from LDS.
//
// Four kernels, all on the caller's stream:
//   prep     (B workgroups)      path labels + per-node edge bitmask
//   rows     (T*B workgroups)    log-sum-exp of every logit row, gather log p on the path        [HBM: read T*B*V]
//   lattice  (2*B workgroups)    alpha (blockIdx.y=0) and beta (blockIdx.y=1) recursions, state in LDS;
//                                f64 accumulation, f32 exp/log only on differences <= 0
//   grad     (T*B workgroups)    occupancy scatter into an LDS row, grad = (softmax - occ) * scale [HBM: read+write T*B*V]
#include "common.hpp"
#include "ctc_ws.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace ctc {

// diagonal offset of edge class j: CTC {0,1,2}; Gram-CTC {0,1,2,3,5,6,7}
template <int NK>
__device__ __forceinline__ constexpr int koff(int j) { return NK == 3 ? j : (j < 4 ? j : j + 1); }

// ------------------------------------------------------------------------------------------------ prep
template <bool GRAM>
__global__ void prep_kernel(const int* __restrict__ uni, const int* __restrict__ big, const int* __restrict__ l_len,
                            int Lmax, int Sp, int V, int blank, int* __restrict__ path_label,
                            int* __restrict__ path_mask, int* __restrict__ path_len) {
    const int b = blockIdx.x;
    int len = l_len ? l_len[b] : Lmax;
    len = min(max(len, 0), Lmax);
    const int S = (GRAM ? 3 : 2) * len + 1;
    if (threadIdx.x == 0) path_len[b] = S;
    const int* u = uni + (size_t)b * Lmax;
    const int* g = GRAM ? big + (size_t)b * Lmax : nullptr;
    for (int s = threadIdx.x; s < Sp; s += blockDim.x) {
        int label = -1, mask = 0;
        if (s < S) {
            if (!GRAM) {
                const int i = (s - 1) >> 1;
                const bool odd = s & 1;
                label = odd ? u[i] : blank;
                if (label < 0 || label >= V) label = -1;
                if (label >= 0) {
                    mask = 1;
                    if (s >= 1) mask |= 2;
                    if (odd && i >= 1 && u[i] != u[i - 1]) mask |= 4;
                }
            } else {
                const int kind = s % 3, i = s / 3;
                auto alive_at = [&](int q) -> bool {   // node q of this path is usable
                    if (q < 0 || q >= S) return false;
                    const int kq = q % 3, iq = q / 3;
                    const int l = kq == 0 ? blank : (kq == 1 ? u[iq] : g[iq]);
                    return l >= 0 && l < V;
                };
                label = kind == 0 ? blank : (kind == 1 ? u[i] : g[i]);
                if (label < 0 || label >= V) label = -1;
                if (label >= 0) {
                    mask = 1;                                                        // k = 0
                    if (kind != 2 && alive_at(s - 1)) mask |= 1 << 1;                 // k = 1
                    if (kind != 2 && alive_at(s - 2)) mask |= 1 << 2;                 // k = 2
                    if (kind == 1 && i >= 1 && u[i] != u[i - 1] && alive_at(s - 3)) mask |= 1 << 3;   // k = 3
                    if (kind == 2 && alive_at(s - 5)) mask |= 1 << 4;                 // k = 5
                    if (kind == 2 && i >= 2 && g[i] != g[i - 2] && alive_at(s - 6)) mask |= 1 << 5;   // k = 6
                    if (kind == 2 && alive_at(s - 7)) mask |= 1 << 6;                 // k = 7
                }
            }
        }
        path_label[(size_t)b * Sp + s] = label;
        path_mask[(size_t)b * Sp + s] = mask;
    }
}

// ------------------------------------------------------------------------------------------------ rows
// One workgroup per (t, b) row of logits: lse = log sum exp, then lp[b][t][s] = x[label_s] - lse.
__global__ __launch_bounds__(256) void rows_kernel(const float* __restrict__ xs, const int* __restrict__ x_len,
                                                   const int* __restrict__ path_label, int T, int B, int V, int Sp,
                                                   float* __restrict__ lse_out, float* __restrict__ lp, const float* __restrict__ lse_in) {
    __shared__ float scratch[32];
    if (lse_in) {       // the producer of the logits formed the row's log-sum-exp while it had the row in registers: gather only,
                        // one wave per row, four rows per workgroup
        const int row = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (row >= T * B) return;
        const int t = row / B, b = row - t * B;
        const int xl = x_len ? min(x_len[b], T) : T;
        if (t >= xl) return;
        const float* x = xs + (size_t)row * V;
        const float lse = lse_in[row];
        if (lane == 0) lse_out[row] = lse;
        const int* pl = path_label + (size_t)b * Sp;
        float* out = lp + ((size_t)b * T + t) * Sp;
        for (int s = lane; s < Sp; s += 64) {
            const int l = pl[s];
            out[s] = l >= 0 ? x[l] - lse : -INFINITY;
        }
        return;
    }
    const int row = blockIdx.x;            // row = t * B + b
    const int t = row / B, b = row - t * B;
    const int xl = x_len ? min(x_len[b], T) : T;
    if (t >= xl) return;
    const float* x = xs + (size_t)row * V;
    float m = -INFINITY;
    const bool vec = ((V & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (int i = threadIdx.x; i < (V >> 2); i += blockDim.x) {
            const float4 v = x4[i];
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
        }
    } else {
        for (int i = threadIdx.x; i < V; i += blockDim.x) m = fmaxf(m, x[i]);
    }
    m = block_max(m, scratch);
    float sum = 0.f;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (int i = threadIdx.x; i < (V >> 2); i += blockDim.x) {
            const float4 v = x4[i];
            sum += __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
        }
    } else {
        for (int i = threadIdx.x; i < V; i += blockDim.x) sum += __expf(x[i] - m);
    }
    sum = block_sum(sum, scratch);
    const float lse = m + __logf(sum);
    if (threadIdx.x == 0) lse_out[row] = lse;
    const int* pl = path_label + (size_t)b * Sp;
    float* out = lp + ((size_t)b * T + t) * Sp;
    for (int s = threadIdx.x; s < Sp; s += blockDim.x) {
        const int l = pl[s];
        out[s] = l >= 0 ? x[l] - lse : -INFINITY;
    }
}

// ------------------------------------------------------------------------------------------------ lattice
// log(sum_j exp(v_j)) over the inputs the mask names.  The offset m only has to be NEAR the maximum (the identity holds for any m),
// so it is found in float32 -- one v_max3_f32 per three inputs instead of a chain of canonicalising v_max_f64 / v_cndmask pairs --
// and floored at -1e30: dead inputs (-inf) then give exp(-inf) = 0 and an all-dead node log(0) = -inf without a branch.  The
// differences v_j - m are formed in float64 (exact), the transcendentals in float32 on arguments <= ~1e-3: absolute error ~1e-7
// as before.  The step of the lattice is one dependent chain (LDS read -> ... -> LDS write, ~0.3 us); this form has 16 instructions
// on it instead of 35 (v_log_f32 x ln 2 instead of the denormal-safe logf sequence: the sum lies in [1, NK]).
template <int NK>
__device__ __forceinline__ double lse_masked(const double* v, int mask) {
    float f[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) f[j] = (mask & (1 << j)) ? (float)v[j] : -INFINITY;
    float m32 = -1e30f;
#pragma unroll
    for (int j = 0; j < NK; ++j) m32 = __builtin_fmaxf(m32, f[j]);
    const double m = (double)m32;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const float e = __expf((float)(v[j] - m));
        acc += (mask & (1 << j)) ? e : 0.f;
    }
    return m + (double)(__builtin_amdgcn_logf(acc) * 0.69314718f);       // acc == 0 (no live input): log2 -> -inf
}

// __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: in the one-node-per-thread loops it made every time step wait for
// the alpha / beta store it had just issued AND for the lp prefetch of four steps ahead -- a memory round trip per step on a chain of
// 1000 steps.  The exchange between the steps is LDS only: wait for the LDS operations, then the barrier; the prefetched values are
// waited for where they are used (the compiler counts vmcnt), the stores never.
#define ASR_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int NK>
__global__ __launch_bounds__(1024) void lattice_kernel(const float* __restrict__ lp, const int* __restrict__ x_len,
                                                       const int* __restrict__ path_label,
                                                       const int* __restrict__ path_mask,
                                                       const int* __restrict__ path_len, int T, int B, int Sp,
                                                       double* __restrict__ alpha, double* __restrict__ beta,
                                                       double* __restrict__ total, float* __restrict__ loss) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* buf0 = reinterpret_cast<double*>(smem);          // Sp + 8 doubles each, 8 guard slots in front
    double* buf1 = buf0 + (Sp + 16);
    int* mask_s = reinterpret_cast<int*>(buf1 + (Sp + 16));  // Sp + 8 ints (8 guard slots behind)
    const int b = blockIdx.x;
    const bool backward = blockIdx.y == 1;
    const int xl = x_len ? min(x_len[b], T) : T;
    const int S = path_len[b];
    const float* lpb = lp + (size_t)b * T * Sp;
    double* outb = (backward ? beta : alpha) + (size_t)b * T * Sp;
    const int* pm = path_mask + (size_t)b * Sp;
    const int* pl = path_label + (size_t)b * Sp;

    // guards: reading s - k (forward) or s + k (backward) outside [0, Sp) sees -inf / mask 0
    for (int i = threadIdx.x; i < Sp + 16; i += blockDim.x) {
        buf0[i] = -INFINITY;
        buf1[i] = -INFINITY;
    }
    for (int i = threadIdx.x; i < Sp + 8; i += blockDim.x) mask_s[i] = i < Sp ? pm[i] : 0;
    __syncthreads();
    double* prev = buf0 + 8;   // index s in [-8, Sp + 8)
    double* cur = buf1 + 8;

    if (xl <= 0) {
        if (!backward && threadIdx.x == 0) { total[b] = -INFINITY; loss[b] = 1e10f; }
        return;
    }

    const bool one_node = (int)blockDim.x >= Sp;          // one node per thread: lp is prefetched PF steps ahead in registers
    constexpr int PF = 4;
    if (!backward && one_node) {
        if (threadIdx.x == 0) prev[0] = 0.0;
        __syncthreads();
        const int sidx = threadIdx.x;
        const int mk = mask_s[sidx];
        // Whole blocks of PF steps run without a branch and every step issues exactly one store and one load (the prefetch index is
        // clamped, not guarded): gfx9 counts loads and stores in ONE in-order counter, and only in straight-line code can the
        // compiler wait for "all but the 2 (PF - 1) youngest" -- behind a branch it waits for vmcnt(0), i.e. every step paid the
        // round trip of the alpha store and the prefetch it had just issued.
        float lq[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) lq[j] = lpb[(size_t)min(j, xl - 1) * Sp + sidx];
        int t0 = 0;
        for (; t0 + PF <= xl; t0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int t = t0 + j;
                double v[NK];
#pragma unroll
                for (int q = 0; q < NK; ++q) v[q] = prev[sidx - koff<NK>(q)];
                const double a = lse_masked<NK>(v, mk) + (double)lq[j];
                cur[sidx] = a;
                outb[(size_t)t * Sp + sidx] = a;
                lq[j] = lpb[(size_t)min(t + PF, xl - 1) * Sp + sidx];
                ASR_LDS_BARRIER();
                double* tmp = prev; prev = cur; cur = tmp;
            }
        }
#pragma unroll
        for (int j = 0; j < PF - 1; ++j) {         // the last xl % PF steps: their log-probabilities are in lq[0 .. ] already
            const int t = t0 + j;
            if (t < xl) {                          // uniform over the workgroup
                double v[NK];
#pragma unroll
                for (int q = 0; q < NK; ++q) v[q] = prev[sidx - koff<NK>(q)];
                const double a = lse_masked<NK>(v, mk) + (double)lq[j];
                cur[sidx] = a;
                outb[(size_t)t * Sp + sidx] = a;
                ASR_LDS_BARRIER();
                double* tmp = prev; prev = cur; cur = tmp;
            }
        }
        if (threadIdx.x == 0) {
            double v[3] = {-INFINITY, -INFINITY, -INFINITY};
            v[0] = prev[S - 1];
            if (S >= 2) v[1] = prev[S - 2];
            if (NK != 3 && S >= 3) v[2] = prev[S - 3];
            const double tot = lse_masked<3>(v, 7);
            total[b] = tot;
            loss[b] = tot == -INFINITY ? 1e10f : (float)(-tot);
        }
    } else if (backward && one_node) {
        const int sidx = threadIdx.x;
        bool fin;
        if (NK == 3) fin = (sidx == S - 1) || (sidx == S - 2 && S >= 2);
        else fin = (sidx == S - 1) || (S >= 3 && (sidx == S - 2 || sidx == S - 3));
        fin = fin && pl[sidx] >= 0;
        const double bt0 = fin ? 0.0 : -INFINITY;
        outb[(size_t)(xl - 1) * Sp + sidx] = bt0;
        prev[sidx] = bt0 + (double)lpb[(size_t)(xl - 1) * Sp + sidx];
        int mkd = 0;                                 // edge s -> s + k belongs to the destination's mask
#pragma unroll
        for (int q = 0; q < NK; ++q) mkd |= ((mask_s[sidx + koff<NK>(q)] >> q) & 1) << q;
        __syncthreads();
        float lq[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) lq[j] = lpb[(size_t)max(xl - 2 - j, 0) * Sp + sidx];
        int t0 = xl - 2;
        for (; t0 - (PF - 1) >= 0; t0 -= PF) {      // whole blocks of PF steps, branch-free (see the forward loop)
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int t = t0 - j;
                double v[NK];
#pragma unroll
                for (int q = 0; q < NK; ++q) v[q] = prev[sidx + koff<NK>(q)];
                const double bt = lse_masked<NK>(v, mkd);
                outb[(size_t)t * Sp + sidx] = bt;
                cur[sidx] = bt + (double)lq[j];
                lq[j] = lpb[(size_t)max(t - PF, 0) * Sp + sidx];
                ASR_LDS_BARRIER();
                double* tmp = prev; prev = cur; cur = tmp;
            }
        }
#pragma unroll
        for (int j = 0; j < PF - 1; ++j) {
            const int t = t0 - j;
            if (t >= 0) {
                double v[NK];
#pragma unroll
                for (int q = 0; q < NK; ++q) v[q] = prev[sidx + koff<NK>(q)];
                const double bt = lse_masked<NK>(v, mkd);
                outb[(size_t)t * Sp + sidx] = bt;
                cur[sidx] = bt + (double)lq[j];
                ASR_LDS_BARRIER();
                double* tmp = prev; prev = cur; cur = tmp;
            }
        }
    } else if (!backward) {
        if (threadIdx.x == 0) prev[0] = 0.0;    // virtual alpha_{-1} = e_0  (asr/loss/gram_ctc.py:144)
        __syncthreads();
        for (int t = 0; t < xl; ++t) {
            const float* lpt = lpb + (size_t)t * Sp;
            double* ot = outb + (size_t)t * Sp;
            for (int s = threadIdx.x; s < Sp; s += blockDim.x) {
                const int mk = mask_s[s];
                double v[NK];
#pragma unroll
                for (int j = 0; j < NK; ++j) v[j] = prev[s - koff<NK>(j)];
                double a = lse_masked<NK>(v, mk);
                a += (double)lpt[s];
                cur[s] = a;
                ot[s] = a;
            }
            __syncthreads();
            double* tmp = prev; prev = cur; cur = tmp;
        }
        // prev = alpha_{xl-1}; final nodes: last blank, last unigram, last bigram (if alive)
        if (threadIdx.x == 0) {
            double v[3] = {-INFINITY, -INFINITY, -INFINITY};
            v[0] = prev[S - 1];
            if (S >= 2) v[1] = prev[S - 2];
            if (NK != 3 && S >= 3) v[2] = prev[S - 3];
            // dead nodes hold -inf already (their lp is -inf)
            const double tot = lse_masked<3>(v, 7);
            total[b] = tot;
            loss[b] = tot == -INFINITY ? 1e10f : (float)(-tot);
        }
    } else {
        // beta_{xl-1}[s] = 0 on final nodes
        for (int s = threadIdx.x; s < Sp; s += blockDim.x) {
            bool fin;
            if (NK == 3) fin = (s == S - 1) || (s == S - 2 && S >= 2);
            else fin = (s == S - 1) || (S >= 3 && (s == S - 2 || s == S - 3));
            fin = fin && pl[s] >= 0;
            const double bt = fin ? 0.0 : -INFINITY;
            outb[(size_t)(xl - 1) * Sp + s] = bt;
            prev[s] = bt + (double)lpb[(size_t)(xl - 1) * Sp + s];     // w_{xl-1}
        }
        __syncthreads();
        for (int t = xl - 2; t >= 0; --t) {
            const float* lpt = lpb + (size_t)t * Sp;
            double* ot = outb + (size_t)t * Sp;
            for (int s = threadIdx.x; s < Sp; s += blockDim.x) {
                double v[NK];
                int mk = 0;
#pragma unroll
                for (int j = 0; j < NK; ++j) {
                    const int k = koff<NK>(j);
                    v[j] = prev[s + k];
                    mk |= ((mask_s[s + k] >> j) & 1) << j;     // edge s -> s + k belongs to the destination's mask
                }
                const double bt = lse_masked<NK>(v, mk);
                ot[s] = bt;
                cur[s] = bt + (double)lpt[s];
            }
            __syncthreads();
            double* tmp = prev; prev = cur; cur = tmp;
        }
    }
}

// ------------------------------------------------------------------------------------------------ grad
constexpr int kOccChunk = 8192;

__global__ __launch_bounds__(256) void grad_kernel(const float* __restrict__ xs, const int* __restrict__ x_len,
                                                   const int* __restrict__ path_label,
                                                   const int* __restrict__ path_len, const float* __restrict__ lse_in,
                                                   const double* __restrict__ alpha, const double* __restrict__ beta,
                                                   const double* __restrict__ total, const float* __restrict__ gy,
                                                   int gy_per_utt, float scale, int T, int B, int V, int Sp,
                                                   float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* occ = reinterpret_cast<float*>(smem);
    const int row = blockIdx.x;
    const int t = row / B, b = row - t * B;
    const int xl = x_len ? min(x_len[b], T) : T;
    float* g = grad + (size_t)row * V;
    const bool vec = ((V & 3) == 0) && ((((uintptr_t)g) & 15) == 0) && ((((uintptr_t)(xs + (size_t)row * V)) & 15) == 0);
    if (t >= xl) {      // asr/loss/gram_ctc.py:296
        if (vec) {
            float4* g4 = reinterpret_cast<float4*>(g);
            for (int i = threadIdx.x; i < (V >> 2); i += blockDim.x) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (int i = threadIdx.x; i < V; i += blockDim.x) g[i] = 0.f;
        }
        return;
    }
    const float* x = xs + (size_t)row * V;
    const float lse = lse_in[row];
    const double tot = total[b];
    float sc = scale;
    if (gy) sc *= gy_per_utt ? gy[b] : gy[0];
    const int S = path_len[b];
    const int* pl = path_label + (size_t)b * Sp;
    const double* al = alpha + ((size_t)b * T + t) * Sp;
    const double* be = beta + ((size_t)b * T + t) * Sp;
    for (int v0 = 0; v0 < V; v0 += kOccChunk) {
        const int vn = min(kOccChunk, V - v0);
        for (int i = threadIdx.x; i < vn; i += blockDim.x) occ[i] = 0.f;
        __syncthreads();
        if (tot != -INFINITY) {
            for (int s = threadIdx.x; s < S; s += blockDim.x) {
                const int l = pl[s];
                if (l >= v0 && l < v0 + vn) {
                    const double e = al[s] + be[s] - tot;
                    if (e > -80.0) atomicAdd(&occ[l - v0], expf((float)e));
                }
            }
        }
        __syncthreads();
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(x + v0);
            const float4* o4 = reinterpret_cast<const float4*>(occ);
            float4* g4 = reinterpret_cast<float4*>(g + v0);
            for (int i = threadIdx.x; i < (vn >> 2); i += blockDim.x) {
                const float4 xv = x4[i];
                const float4 ov = o4[i];
                float4 r;
                r.x = (__expf(xv.x - lse) - ov.x) * sc;
                r.y = (__expf(xv.y - lse) - ov.y) * sc;
                r.z = (__expf(xv.z - lse) - ov.z) * sc;
                r.w = (__expf(xv.w - lse) - ov.w) * sc;
                g4[i] = r;
            }
        } else {
            for (int i = threadIdx.x; i < vn; i += blockDim.x) g[v0 + i] = (__expf(x[v0 + i] - lse) - occ[i]) * sc;
        }
        __syncthreads();
    }
}

__global__ void mean_kernel(const float* __restrict__ loss, int B, float* __restrict__ out) {
    __shared__ float scratch[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) s += loss[i];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[0] = s / (float)B;
}

}  // namespace ctc
}  // namespace asr

using namespace asr;
using namespace asr::ctc;

extern "C" size_t asr_ctc_workspace_bytes(int T, int B, int V, int Lmax, int gram) {
    (void)V;
    if (T <= 0 || B <= 0 || Lmax <= 0) return 0;
    return carve(nullptr, T, B, Lmax, gram).bytes;
}

extern "C" int asr_ctc_forward_lse(void* stream_, const float* xs, const int32_t* label_unigram,
                                   const int32_t* label_bigram, const int32_t* x_len, const int32_t* l_len, int T, int B,
                                   int V, int Lmax, int blank, float* loss_per_utt, float* loss_mean, void* workspace,
                                   size_t workspace_bytes, const float* row_lse) {
    if (!xs || !label_unigram || !loss_per_utt || !workspace) return ASR_ERR_BAD_ARG;
    if (T <= 0 || B <= 0 || V <= 0 || Lmax <= 0 || blank < 0 || blank >= V) return ASR_ERR_BAD_ARG;
    const int gram = label_bigram != nullptr;
    Workspace w = carve(workspace, T, B, Lmax, gram);
    if (workspace_bytes < w.bytes) return ASR_ERR_WORKSPACE;
    const int Sp = path_pad(Lmax, gram);
    const size_t lds = sizeof(double) * 2 * (Sp + 16) + sizeof(int) * (Sp + 8);
    if (lds > 150 * 1024) return ASR_ERR_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    if (gram)
        hipLaunchKernelGGL(prep_kernel<true>, dim3(B), dim3(256), 0, stream, label_unigram, label_bigram, l_len, Lmax, Sp,
                           V, blank, w.path_label, w.path_mask, w.path_len);
    else
        hipLaunchKernelGGL(prep_kernel<false>, dim3(B), dim3(256), 0, stream, label_unigram, label_bigram, l_len, Lmax,
                           Sp, V, blank, w.path_label, w.path_mask, w.path_len);
    ASR_LAUNCH_CHECK();
    hipLaunchKernelGGL(rows_kernel, dim3(row_lse ? (T * B + 3) / 4 : T * B), dim3(256), 0, stream, xs, x_len, w.path_label, T, B, V, Sp, w.lse, w.lp, row_lse);
    ASR_LAUNCH_CHECK();
    const int threads = Sp < 1024 ? Sp : 1024;
    if (gram) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)lattice_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lattice_kernel<7>, dim3(B, 2), dim3(threads), lds, stream, w.lp, x_len, w.path_label,
                           w.path_mask, w.path_len, T, B, Sp, w.alpha, w.beta, w.total, loss_per_utt);
    } else {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)lattice_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lattice_kernel<3>, dim3(B, 2), dim3(threads), lds, stream, w.lp, x_len, w.path_label,
                           w.path_mask, w.path_len, T, B, Sp, w.alpha, w.beta, w.total, loss_per_utt);
    }
    ASR_LAUNCH_CHECK();
    if (loss_mean) {
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, loss_per_utt, B, loss_mean);
        ASR_LAUNCH_CHECK();
    }
    return ASR_OK;
}

extern "C" int asr_ctc_forward(void* stream_, const float* xs, const int32_t* label_unigram,
                               const int32_t* label_bigram, const int32_t* x_len, const int32_t* l_len, int T, int B,
                               int V, int Lmax, int blank, float* loss_per_utt, float* loss_mean, void* workspace,
                               size_t workspace_bytes) {
    return asr_ctc_forward_lse(stream_, xs, label_unigram, label_bigram, x_len, l_len, T, B, V, Lmax, blank, loss_per_utt, loss_mean,
                               workspace, workspace_bytes, nullptr);
}

extern "C" int asr_ctc_backward(void* stream_, const float* xs, const int32_t* x_len, int T, int B, int V, int Lmax,
                                int gram, const float* gy, int gy_per_utt, float scale, float* grad,
                                const void* workspace, size_t workspace_bytes) {
    if (!xs || !grad || !workspace) return ASR_ERR_BAD_ARG;
    if (T <= 0 || B <= 0 || V <= 0 || Lmax <= 0) return ASR_ERR_BAD_ARG;
    Workspace w = carve(const_cast<void*>(workspace), T, B, Lmax, gram);
    if (workspace_bytes < w.bytes) return ASR_ERR_WORKSPACE;
    const int Sp = path_pad(Lmax, gram);
    const size_t lds = sizeof(float) * (size_t)(V < kOccChunk ? (int)align_up(V, 4) : kOccChunk);
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(grad_kernel, dim3(T * B), dim3(256), lds, stream, xs, x_len, w.path_label, w.path_len, w.lse,
                       w.alpha, w.beta, w.total, gy, gy_per_utt, scale, T, B, V, Sp, grad);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_ctc_loss_grad(void* stream, const float* xs, const int32_t* label_unigram,
                                 const int32_t* label_bigram, const int32_t* x_len, const int32_t* l_len, int T, int B,
                                 int V, int Lmax, int blank, float scale, float* loss_per_utt, float* loss_mean,
                                 float* grad, void* workspace, size_t workspace_bytes) {
    int rc = asr_ctc_forward(stream, xs, label_unigram, label_bigram, x_len, l_len, T, B, V, Lmax, blank, loss_per_utt,
                             loss_mean, workspace, workspace_bytes);
    if (rc != ASR_OK) return rc;
    return asr_ctc_backward(stream, xs, x_len, T, B, V, Lmax, label_bigram != nullptr, nullptr, 0, scale, grad,
                            workspace, workspace_bytes);
}
