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

// one v_exp_f32 + one v_rcp_f32 each (an IEEE division costs ~10 dependent VALU instructions, and the gate math runs on
// one wave per SIMD in the latency chain of a step)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(-2.0f * fabsf(x));                 // in (0, 1]
    const float t = 1.0f - 2.0f * e * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

constexpr int MT = 2;   // 16-row batch tiles per pass (32 utterances); larger batches loop

// Every step is latency-bound (a few hundred KB from L2 / Infinity Cache, ~100 MFMAs): all global loads of a step --
// the MFMA fragments of this wave's K slices AND the element-wise operands -- are issued up front, so a step pays one
// memory round trip instead of one per K slice.  KSW = K slices (of 32) per wave, unrolled at compile time.
template <int KSW>
__global__ __launch_bounds__(256) void fwd_step_kernel(const float* __restrict__ gi, const uint16_t* __restrict__ whh,
                                                       const float* __restrict__ bhh, float* __restrict__ hseq,
                                                       uint16_t* __restrict__ hseq16, float* __restrict__ gates, int T,
                                                       int B, int H, int ndir, int s, const float* __restrict__ hx) {
    // hx (ndir, B, H) float32 or NULL: the state in front of the first step (asr_gru_fwd_state); NULL = zeros, no product at s == 0
    __shared__ __attribute__((aligned(16))) float4 part[4 * MT * 3 * 64];
    const int d = blockIdx.y, j0 = blockIdx.x * 16;
    const int t = d == 0 ? s : T - 1 - s;
    const int tp = d == 0 ? t - 1 : t + 1;
    const bool from_hx = s == 0 && hx != nullptr;
    const bool first = s == 0 && hx == nullptr;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nks = H >> 5;
    const size_t hs = (size_t)ndir * H;
    for (int b0 = 0; b0 < B; b0 += 16 * MT) {
        // ---- element-wise operands of this thread's (b, j) pairs: issue the loads now, use them after the MFMAs
        float e_gi[MT][3], e_hp[MT], e_bh[MT][3];
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            const int idx = tid + q * 256;
            const int bl = idx >> 4, j = idx & 15;
            const int b = b0 + bl;
            const bool ok = b < B;
            const size_t rowi = (size_t)t * B + (ok ? b : 0);
            const float* gir = gi + rowi * (3 * hs) + (size_t)d * 3 * H + j0 + j;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                e_gi[q][g] = ok ? gir[g * H] : 0.f;
                e_bh[q][g] = bhh[(d * 3 + g) * H + j0 + j];
            }
            e_hp[q] = (ok && !first) ? (from_hx ? hx[((size_t)d * B + b) * H + j0 + j] : hseq[((size_t)tp * B + b) * hs + d * H + j0 + j]) : 0.f;
        }
        if (!first) {
            Frag a[KSW][MT], bb[KSW][3];
#pragma unroll
            for (int i = 0; i < KSW; ++i) {
                const int ks = w * KSW + i;
                const bool kok = ks < nks;
                const int k = ks * 32 + 8 * (lane >> 4);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int row = b0 + m * 16 + (lane & 15);
                    if (from_hx) {          // the given state, rounded to the 16-bit operand format as every later state is
                        a[i][m].u = make_uint4(0, 0, 0, 0);
                        if (kok && row < B) {
                            const float4* src = reinterpret_cast<const float4*>(hx + ((size_t)d * B + row) * H + k);
                            const float4 v0 = src[0], v1 = src[1];
                            a[i][m].u = make_uint4(pack_bf16x2(v0.x, v0.y), pack_bf16x2(v0.z, v0.w), pack_bf16x2(v1.x, v1.y), pack_bf16x2(v1.z, v1.w));
                        }
                    } else
                    a[i][m].u = (kok && row < B) ? *reinterpret_cast<const uint4*>(hseq16 + ((size_t)tp * B + row) * hs + d * H + k)
                                                 : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    bb[i][g].u = kok ? *reinterpret_cast<const uint4*>(whh + ((size_t)(d * 3 + g) * H + j0 + (lane & 15)) * H + k)
                                     : make_uint4(0, 0, 0, 0);
            }
            f32x4 acc[MT][3];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[m][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KSW; ++i)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][m].v, bb[i][g].v, acc[m][g], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    part[((w * MT + m) * 3 + g) * 64 + lane] = make_float4(acc[m][g][0], acc[m][g][1], acc[m][g][2], acc[m][g][3]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            const int idx = tid + q * 256;
            const int bl = idx >> 4, j = idx & 15;
            const int b = b0 + bl;
            if (b >= B) continue;
            const int m = bl >> 4, row = bl & 15;
            const int pl = (row >> 2) * 16 + j, pr = row & 3;
            float gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float sum = e_bh[q][g];
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
            const float r = sigmoidf_(e_gi[q][0] + gh[0]);
            const float z = sigmoidf_(e_gi[q][1] + gh[1]);
            const float n = tanhf_(e_gi[q][2] + r * gh[2]);
            const float h = (1.0f - z) * n + z * e_hp[q];
            hseq[rowi * hs + d * H + j0 + j] = h;
            hseq16[rowi * hs + d * H + j0 + j] = f32_to_bf16(h);
            float* gs = gates + (rowi * ndir + d) * 4 * H + j0 + j;
            gs[0] = r; gs[H] = z; gs[2 * H] = n; gs[3 * H] = gh[2];
        }
        __syncthreads();
    }
}

// backward step: dh_t = dy_t + carry + dgh_{next} W_hh ; gate gradients ; carry <- dh_t * z_t
template <int KSW>
__global__ __launch_bounds__(256) void bwd_step_kernel(const uint16_t* __restrict__ dy, const float* __restrict__ gates,
                                                       const float* __restrict__ hseq,
                                                       const uint16_t* __restrict__ whhT, uint16_t* __restrict__ dgi,
                                                       uint16_t* __restrict__ dgh, float* __restrict__ carry, int T,
                                                       int B, int H, int ndir, int s, const float* __restrict__ hx, int carry_init) {
    // hx: the state in front of the first forward step (NULL = zeros); carry_init: `carry` arrives holding the gradient of the final
    // state (dhy of asr_gru_bwd_state) instead of being undefined before the first step
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
        float e_dy[MT], e_c[MT], e_g[MT][4], e_hp[MT];
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            const int idx = tid + q * 256;
            const int bl = idx >> 4, j = idx & 15;
            const int b = b0 + bl;
            const bool ok = b < B;
            const size_t rowi = (size_t)t * B + (ok ? b : 0);
            e_dy[q] = ok ? bf16_to_f32(dy[rowi * H + j0 + j]) : 0.f;
            e_c[q] = (ok && (!first || carry_init)) ? carry[((size_t)d * B + b) * H + j0 + j] : 0.f;
            const float* gs = gates + (rowi * ndir + d) * 4 * H + j0 + j;
#pragma unroll
            for (int g = 0; g < 4; ++g) e_g[q][g] = ok ? gs[g * H] : 0.f;
            e_hp[q] = !ok ? 0.f : has_prev ? hseq[((size_t)tp * B + b) * hs + d * H + j0 + j] : (hx ? hx[((size_t)d * B + b) * H + j0 + j] : 0.f);
        }
        if (!first) {
            f32x4 acc[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int base = 0; base * 4 < nks; base += KSW) {      // one pass when 4 * KSW >= nks
                Frag a[KSW][MT], bb[KSW];
#pragma unroll
                for (int i = 0; i < KSW; ++i) {
                    const int ks = (base + i) * 4 + w;
                    const bool kok = ks < nks;
                    const int k = ks * 32 + 8 * (lane >> 4);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int row = b0 + m * 16 + (lane & 15);
                        a[i][m].u = (kok && row < B)
                                        ? *reinterpret_cast<const uint4*>(dgh + ((size_t)tn * B + row) * gs3 + (size_t)d * 3 * H + k)
                                        : make_uint4(0, 0, 0, 0);
                    }
                    bb[i].u = kok ? *reinterpret_cast<const uint4*>(whhT + ((size_t)d * H + j0 + (lane & 15)) * (3 * H) + k)
                                  : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < KSW; ++i)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][m].v, bb[i].v, acc[m], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
                part[(w * MT + m) * 64 + lane] = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MT; ++q) {
            const int idx = tid + q * 256;
            const int bl = idx >> 4, j = idx & 15;
            const int b = b0 + bl;
            if (b >= B) continue;
            const int m = bl >> 4, row = bl & 15;
            const int pl = (row >> 2) * 16 + j, pr = row & 3;
            const size_t rowi = (size_t)t * B + b;
            float dh = e_dy[q] + e_c[q];
            if (!first) {
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) {
                    const float4 v = part[(ww * MT + m) * 64 + pl];
                    dh += pr == 0 ? v.x : (pr == 1 ? v.y : (pr == 2 ? v.z : v.w));
                }
            }
            const float r = e_g[q][0], z = e_g[q][1], n = e_g[q][2], qq = e_g[q][3];
            const float hp = e_hp[q];
            const float dn = dh * (1.0f - z);
            const float dz = dh * (hp - n);
            const float dan = dn * (1.0f - n * n);
            const float daz = dz * z * (1.0f - z);
            const float dq = dan * r;
            const float dar = dan * qq * r * (1.0f - r);
            carry[((size_t)d * B + b) * H + j0 + j] = dh * z;
            uint16_t* gi_o = dgi + rowi * gs3 + (size_t)d * 3 * H + j0 + j;
            uint16_t* gh_o = dgh + rowi * gs3 + (size_t)d * 3 * H + j0 + j;
            const uint16_t ar = f32_to_bf16(dar), az = f32_to_bf16(daz);
            gi_o[0] = ar; gi_o[H] = az; gi_o[2 * H] = f32_to_bf16(dan);
            gh_o[0] = ar; gh_o[H] = az; gh_o[2 * H] = f32_to_bf16(dq);
        }
        __syncthreads();
    }
}

// ================================================================================================ persistent form
// One launch per layer: every workgroup keeps its W_hh slice in registers for all T steps and the workgroups of a
// direction hand h_t (forward) / dgh_t (backward) to each other through HBM-side memory inside the launch:
//   producer: payload stored write-through (sc1) -> every storing wave s_waitcnt vmcnt(0) -> workgroup barrier ->
//             ONE lane: agent-scope atomic add on the direction's step counter
//   consumer: ONE lane polls the counter with relaxed agent-scope (sc1) loads, bounded, -> workgroup barrier ->
//             every load of the payload is an sc1 buffer load to registers
// (MI355X_MICROARCH.md "Valid forms", first row of the sc1 table; placement independent).  One workgroup per CU is
// forced by the LDS request; a grid of at most 128 workgroups is always co-resident on 256 CUs.  Every spin is bounded:
// on time-out the workgroup raises the abort word, which all pollers watch, and the launch drains.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// 2^22 polls of >= 1 us each: seconds.  Nothing a launch can legitimately wait for takes that long -- a workgroup that finds
// its CU busy (a weight-gradient GEMM of the side stream; never a collective: asr/parallel.py joins those before a
// recurrence is queued) starts milliseconds late at worst -- but a launch that can NEVER be fully resident (partitioned
// device, another process holding CUs for good) still ends, raises the abort word and costs one dropped step.
constexpr unsigned kSpinLimit = 1u << 22;
// When to ask for the other workgroups' payload (s_sleep units of 64 cycles after this wave's own store / after the workgroup's
// store flag).  A poll issued at once is always too early in this symmetric machine -- the other producers' stores are still on
// their way to the L2 -- and the retry queues BEHIND it in the CU's memory pipeline: the hand-off then costs two round trips.  A
// poll timed to pass the L2 just after the stores land usually succeeds at once.  Forward kernel, T=1000, B=32, H=512, us per step:
// 0: 1.487, 3: 1.433, 6: 1.384, 10: 1.345, 11-12: 1.340, 14: 1.369, 18: 1.448, 26: 1.658 (three staggered polls in flight: 1.83).
// The partial-sum backward kernel did not gain while its first attempt followed 64 16-lane store instructions of the workgroup (round 3:
// 0: 1.541, 6: 1.545, 10: 1.639); with 16 full-wave stores (round 4) the poll goes out ~300 cycles earlier and wants the pause back:
// 0: 1.331, 3: 1.287, 6: 1.262, 8: 1.251, 10: 1.252, 12: 1.297, 15: 1.379 us per step.
#ifndef ASR_FIRST_POLL_DELAY
#define ASR_FIRST_POLL_DELAY 11
#endif
#ifndef ASR_BWD_POLL_DELAY
#define ASR_BWD_POLL_DELAY 8
#endif
#ifndef ASR_TOUCH
#define ASR_TOUCH 1
#endif
constexpr bool kTouch = ASR_TOUCH != 0;
#ifndef ASR_TOUCH_AHEAD
#define ASR_TOUCH_AHEAD 2
#endif
constexpr int kTouchAhead = ASR_TOUCH_AHEAD;
constexpr int kFirstPollDelay = ASR_FIRST_POLL_DELAY, kBwdPollDelay = ASR_BWD_POLL_DELAY;
constexpr int kPersistLds = 96 * 1024;
// The wide backward kernel asks for so much LDS that no GEMM workgroup (36..64 KB) fits beside it on a CU.  Sharing the
// CU paid while the recurrence waited on memory ("Co-residency"); once its step had become an
// instruction chain on the gate waves, a weight-gradient GEMM beside it cost 27 % per launch (22.4 vs 20.5 ms per train
// step), so the GEMMs of the side stream now run between the recurrences.
constexpr int kExclusiveLds = 132 * 1024;
constexpr size_t kShardBytes = 16 * 8 * 128;      // 16 recurrences x 8 shards x one 128-B line

#define ASR_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// returns false when the wait was abandoned (abort raised here or elsewhere)
__device__ __forceinline__ bool wait_counter(unsigned* counter, unsigned target, unsigned* abort_word) {
    unsigned spins = 0;
    for (;;) {
        if (__hip_atomic_load(counter, ASR_RLX_AGENT) >= target) return true;
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 63u) == 0u) {          // the abort word is looked at once per 64 polls: the poll itself stays one load
            if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) return false;
            if (spins > kSpinLimit) {
                __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT);
                return false;
            }
        }
    }
}

// depth (time steps) of the LDS operand rings the loader waves of the persistent kernels fill by LDS-DMA
constexpr int BIO_GD = 4;
#define ASR_RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP

// XCD-local hand-off: counter polled with an atomic executed in the XCD's L2 (a plain load could sit in the CU's L1)
__device__ __forceinline__ bool wait_counter_l2(unsigned* counter, unsigned target, unsigned* abort_word) {
    unsigned spins = 0;
    for (;;) {
        unsigned seen;      // returning atomic without a scope bit: executed in L2, never served from the L1
        asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(counter), "v"(0u) : "memory");
        if (seen >= target) return true;
        ++spins;
        if ((spins & 63u) == 0u) {
            if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) return false;
            if (spins > kSpinLimit) {
                __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT);
                return false;
            }
        }
    }
}

// rows r and r + 8 of every 16-lane row swap places (DPP row_ror:8), for all four dwords of a fragment
// volatile accesses to an LDS word THROUGH THE LDS ADDRESS SPACE: a cast to `volatile int*` makes the pointer generic and the
// access a flat_load / flat_store with sc0 sc1 followed by s_waitcnt vmcnt(0) -- on the gate waves that drained their own
// payload store before the next hand-off loads could be issued
typedef __attribute__((address_space(3))) volatile int lds_vint;
// 16 B out of LDS by inline asm (waits for that read only).  The compiler puts s_waitcnt vmcnt(0) in front of every LDS read of a
// kernel that also uses LDS-DMA (the DMA writes LDS and may alias) -- on the storer wave that is a wait for the acknowledgement of
// everything it has written so far, and for its pre-touch loads, before it may read what it is about to store.
typedef float f32x4_asm __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_asm lds_read16_raw(const void* p) {
    const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
    f32x4_asm v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ int lds_peek(const int* p) { return *(const lds_vint*)(const __attribute__((address_space(3))) int*)p; }
__device__ __forceinline__ void lds_poke(int* p, int v) { *(lds_vint*)(__attribute__((address_space(3))) int*)p = v; }

// value of lane ^ 1 through DPP quad_perm [1,0,3,2]: __shfl_xor compiles to ds_bpermute, an LDS round trip on the chain
// between the gate math and the payload store
// the pause in front of a first poll: one s_sleep with an immediate for the values the host hands out (a loop of s_sleep(1) costs
// the chain ~0.02 us more: 1.369 against 1.347 us per step), the loop for anything else (ASR_DEBUG gru_poll_delay)
__device__ __forceinline__ void poll_pause(int n) {
    if (n == 6) __builtin_amdgcn_s_sleep(6);
    else if (n == 3) __builtin_amdgcn_s_sleep(3);
    else if (n == 11) __builtin_amdgcn_s_sleep(11);
    else for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}

// the value of another lane through a DPP move (CTRL: 0x141 row_half_mirror = lane 7 - k of each eight, quad_perm 0x4E = k ^ 2, 0xB1 = k ^ 1)
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ unsigned lane_xor1_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ uint4 swap_half_rows(uint4 v) {
    uint4 r;
    r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x128, 0xF, 0xF, false);
    r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x128, 0xF, 0xF, false);
    r.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x128, 0xF, 0xF, false);
    r.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x128, 0xF, 0xF, false);
    return r;
}

// The step counter of a recurrence is kept in NSH shards on lines of their own (workgroup i adds to shard i % NSH): atomics on
// one address serialise in the L2, 32 of them per step were ~0.3 us of the chain.  The poller reads all four (sc1 loads:
// they bypass the L1 and are served by the L2, so they also see the L2-scope adds of the XCD-local form).
constexpr int NSH = 4;          // shards per recurrence (8 measured no better)
__device__ __forceinline__ unsigned* shard_base(unsigned* sync, int rec) {
    return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(sync) + 4096) + rec * NSH * 32;
}
template <bool L2ATOMIC>
__device__ __forceinline__ bool wait_shards(unsigned* base, int nwg, unsigned s, unsigned* abort_word) {
    unsigned spins = 0;
    for (;;) {
        unsigned c[NSH];
#pragma unroll
        for (int i = 0; i < NSH; ++i) {
            if (L2ATOMIC)       // XCD-local form: returning atomics without a scope bit, executed in the L2 that holds the counters
                asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=&v"(c[i]) : "v"(base + i * 32), "v"(0u) : "memory");
            else
                c[i] = __hip_atomic_load(base + i * 32, ASR_RLX_AGENT);
        }
        if (L2ATOMIC) {
#pragma unroll
            for (int i = 0; i < NSH; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(c[i]) :: "memory");
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NSH; ++i) ok = ok && c[i] >= (unsigned)((nwg + NSH - 1 - i) / NSH) * s;
        if (ok) return true;
        ++spins;
        if ((spins & 63u) == 0u) {
            if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) return false;
            if (spins > kSpinLimit) {
                __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT);
                return false;
            }
        }
    }
}

// XCD-local form: per-producer flags instead of a counter.  Workgroup i of a recurrence stores step + 1 into word i of a
// flag line (a plain store: it lands in the XCD's L2); the polling wave reads the whole line with ONE sc1 wave load (lane
// i -> word i, served by that L2) and votes.  No atomics on the chain: 2.58 -> 2.10 (fwd) and 2.95 -> 2.55 us (bwd).
// (With sc1 write-through flag stores the placement-free form got slower, 2.72 -> 2.96 us: it keeps the sharded counters.)
// Called by all 64 lanes of the polling wave; nwg <= 64.
__device__ __forceinline__ bool wait_flags(unsigned* flags, int nwg, unsigned target, unsigned* abort_word, int lane) {
    unsigned spins = 0;
    for (;;) {
        const unsigned v = lane < nwg ? __hip_atomic_load(flags + lane, ASR_RLX_AGENT) : 0xffffffffu;
        if (__ballot(v >= target) == ~0ull) return true;
        ++spins;
        if ((spins & 63u) == 0u) {
            if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) return false;
            if (spins > kSpinLimit) {
                if (lane == 0) __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT);
                return false;
            }
        }
    }
}
// the same for a subset of the producers (bit i of mask = workgroup i): a compute wave waits only for the workgroups whose
// units its K slices cover, and starts its loads without a workgroup barrier
__device__ __forceinline__ bool wait_flags_mask(unsigned* flags, int nflags, unsigned long long mask, unsigned target,
                                                unsigned* abort_word, int lane) {
    unsigned spins = 0;
    for (;;) {
        const unsigned v = lane < nflags ? __hip_atomic_load(flags + lane, ASR_RLX_AGENT) : 0xffffffffu;
        if ((__ballot(v >= target) & mask) == mask) return true;
        ++spins;
        if ((spins & 63u) == 0u) {
            if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) return false;
            if (spins > kSpinLimit) {
                if (lane == 0) __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT);
                return false;
            }
        }
    }
}
__device__ __forceinline__ void set_flag(unsigned* flag, unsigned v, bool local) {
    if (local) asm volatile("global_store_dword %0, %1, off" :: "v"(flag), "v"(v) : "memory");
    else __hip_atomic_store(flag, v, ASR_RLX_AGENT);
}

// XCD-local hand-off is a speed-up, not an assumption: the workgroups of a recurrence agree at kernel start whether they
// all sit on one XCD (HIP promises no placement).  Each registers its HW_REG_XCC_ID; after all have arrived every one
// reads the same verdict: local (plain payload stores that stay in the XCD's L2 + a counter kept by L2 atomics) or the
// placement-free protocol (sc1 write-through stores + agent-scope counter).  Payload loads are sc1 either way (they
// bypass the L1 and are served by the L2).  Control words: sync[960 + r] XCC id, [976 + r] mismatch, [992 + r] arrivals.
__device__ __forceinline__ int decide_local(unsigned* sync, int rec, int nwg, unsigned* abort_word, int forge) {
    unsigned xcc = (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u) + 1u;      // HW_REG_XCC_ID[3:0]
    if (forge == 1) xcc = 1u + (blockIdx.x >> 3 & 1u);                         // test hook: pretend a split placement
    else if (forge > 1) xcc = (unsigned)forge;                                 // (wide kernels pass 2 + slot parity)
    unsigned expect = 0u;
    if (!__hip_atomic_compare_exchange_strong(sync + 960 + rec, &expect, xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
        expect != xcc)
        __hip_atomic_store(sync + 976 + rec, 1u, ASR_RLX_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(sync + 992 + rec, 1u, ASR_RLX_AGENT);
    if (!wait_counter(sync + 992 + rec, (unsigned)nwg, abort_word)) return -1;
    return __hip_atomic_load(sync + 976 + rec, ASR_RLX_AGENT) == 0u ? 1 : 0;
}

// LDS ring slot of the backward loader (bytes): r, z, n, q, h_prev as [8 rows][16 units] f32 (512 B each), dy as
// [8 rows][16 units] bf16 (256 B).  Filled by LDS-DMA (global_load_lds_dwordx4: the wave's lanes write 16 B each,
// lane-linear from a wave-uniform base), so the loader holds no data registers.
constexpr int BIO_SLOT = 5 * 512 + 256;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
#define ASR_RAW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- wide form: 32 hidden units per workgroup, recurrences of 4 batch rows ------------------------------------------
// The hand-off load phase scales with the payload bytes per CU (rows x K x 2 B, whatever the instruction form: section 5
// of DESIGN.md).  Halving the rows per recurrence halves it; to stay within 256 workgroups each one then owns 32 units:
// 8 compute waves (K split in 8, both 16-unit tiles per wave, 48 weight registers as before) + loader + storer = 640
// threads.  The 16-row MFMA tile has 4 live rows, so the lanes of tile rows 4..15 fetch the next three K slices of the
// same rows (one load instruction = 4 slices = whole 128-B lines) and DPP row shifts bring them to rows 0..3.
__device__ __forceinline__ uint4 shl_rows(uint4 v, int n4) {      // lane i of every 16-lane row <- lane i + 4 n4 (zero beyond)
    uint4 r = v;
    if (n4 == 1) {
        r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x104, 0xF, 0xF, true); r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x104, 0xF, 0xF, true);
        r.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x104, 0xF, 0xF, true); r.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x104, 0xF, 0xF, true);
    } else if (n4 == 2) {
        r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x108, 0xF, 0xF, true); r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x108, 0xF, 0xF, true);
        r.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x108, 0xF, 0xF, true); r.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x108, 0xF, 0xF, true);
    } else if (n4 == 3) {
        r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x10C, 0xF, 0xF, true); r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x10C, 0xF, 0xF, true);
        r.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x10C, 0xF, 0xF, true); r.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x10C, 0xF, 0xF, true);
    }
    return r;
}

template <int KS8, bool LOCAL>
__global__ __launch_bounds__(640, 3) void bwd_wide_kernel(const uint16_t* __restrict__ dy, const float* __restrict__ gates,
                                                          const float* __restrict__ hseq, const uint16_t* __restrict__ whhT,
                                                          uint16_t* __restrict__ dgi, uint16_t* dgh, float* __restrict__ db_ih,
                                                          float* __restrict__ db_hh, unsigned* sync, int T, int B, int H, int ndir,
                                                          int forge) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* part = reinterpret_cast<float4*>(smem);                                   // [2 (step parity)][8 waves][2 tiles][64]
    char* opring = smem + 2 * 8 * 2 * 64 * 16;                                          // [BIO_GD][BIO_SLOT]: 5 x [4 rows][32 units] f32, dy bf16
    unsigned* oring = reinterpret_cast<unsigned*>(opring + BIO_GD * BIO_SLOT);       // [2][3: ar az an][4 rows][16 pairs]
    int* s_abort = reinterpret_cast<int*>(oring + 2 * 3 * 4 * 16);
    constexpr int rows = 4;
    const int nwg = H / 32;
    const int Gn = (B + rows - 1) / rows, nrec = Gn * ndir, nrec_pad = (nrec + 7) & ~7;
    // LOCAL: 1-D grid of nrec_pad x H/32 workgroups; ids are dealt round-robin over the 8 XCDs, so recurrence id % nrec_pad
    // sits on XCD id % 8 with all its workgroups (nrec_pad is a multiple of 8)
    const int rec = LOCAL ? (int)(blockIdx.x % nrec_pad) : (int)(blockIdx.z * gridDim.y + blockIdx.y);
    const int slot = LOCAL ? (int)(blockIdx.x / nrec_pad) : (int)blockIdx.x;
    if (LOCAL && rec >= nrec) return;
    const int d = rec / Gn, g = rec % Gn;
    const int j0 = slot * 32;
    const int b0 = g * rows, Bl = min(rows, B - b0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_compute = w < 8, is_loader = w == 8, is_storer = w == 9;
    const int nks = (3 * H) >> 5;
    const size_t hs = (size_t)ndir * H, gs3 = (size_t)ndir * 3 * H;
    unsigned* shards = shard_base(sync, rec);
    unsigned* my_shard = shards + (slot % NSH) * 32;
    unsigned* abort_word = sync + 1023;
    const __amdgpu_buffer_rsrc_t dghrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)dgh, 0, (int)((size_t)T * B * gs3 * 2), 0x00020000);
    const long long tstep = d == 0 ? -1 : 1;
    const int tfirst = d == 0 ? T - 1 : 0;

    // ---- loader (wave 8): THREE LDS-DMA instructions per step fill the 2816-byte slot image (every instruction of a step
    // costs the recurrence about 0.05 us, see fwd_persistent_io_kernel): lanes 0..63 gates r | z, lanes 0..63 gates n | q
    // (array = lane / 32, then (row, units 4 (lane % 8) ..) = lane % 32), lanes 0..31 h_prev + lanes 32..47 dy (row
    // (lane - 32) / 4, 8 bf16 each)
    const int lrow = (lane & 31) >> 3, lyrow = (lane - 32) >> 2;
    const float* lgp = gates + (((size_t)tfirst * B + b0 + (lrow < Bl ? lrow : 0)) * ndir + d) * 4 * H + j0 + (lane & 7) * 4 +
                       (size_t)(lane >> 5) * H;
    const float* lhp = hseq + ((long long)(d == 0 ? tfirst - 1 : tfirst + 1) * B + b0 + (lrow < Bl ? lrow : 0)) * (long long)hs +
                       (size_t)d * H + j0 + (lane & 7) * 4;
    const uint16_t* lyp = dy + ((size_t)tfirst * B + b0 + (lane >= 32 && lane < 48 && lyrow < Bl ? lyrow : 0)) * H + j0 + (lane & 3) * 8;
    const char* l2p = lane < 32 ? reinterpret_cast<const char*>(lhp) : reinterpret_cast<const char*>(lyp);
    const long long lgs = tstep * (long long)B * ndir * 4 * H;
    const long long l2s = lane < 32 ? tstep * (long long)B * (long long)hs * 4 : tstep * (long long)B * H * 2;      // bytes
    auto issue = [&](int sq) {
        if (sq < T) {
            char* sl = opring + (sq % BIO_GD) * BIO_SLOT;
            if (lrow < Bl) {
                __builtin_amdgcn_global_load_lds((glb_ptr_t)lgp, (lds_ptr_t)sl, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(lgp + 2 * (size_t)H), (lds_ptr_t)(sl + 1024), 16, 0, 0);
            }
            if (lane < 32 ? (lrow < Bl && sq < T - 1) : (lane < 48 && lyrow < Bl))
                __builtin_amdgcn_global_load_lds((glb_ptr_t)l2p, (lds_ptr_t)(sl + 2048), 16, 0, 0);
        }
        lgp += lgs; l2p += l2s;
    };
    // ---- storer (wave 9): dgi: 3 gates x 4 rows x 64 B = 48 pieces of 16 B
    auto store_step = [&](int sp) {
        const long long tq = tfirst + tstep * sp;
        const unsigned* src = oring + (size_t)(sp & 1) * 3 * 4 * 16;
        const int gsel = lane >> 4, row = (lane & 15) >> 2, c = lane & 3;
        if (lane < 48 && row < Bl)
            *reinterpret_cast<uint4*>(dgi + ((size_t)tq * B + b0 + row) * gs3 + (size_t)d * 3 * H + gsel * H + j0 + c * 8) =
                *reinterpret_cast<const uint4*>(src + (gsel * 4 + row) * 16 + c * 4);
    };
    if (is_loader) {
        for (int s0 = 0; s0 < BIO_GD - 1; ++s0) issue(s0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // weights: wave w owns the K slices [w KS8, (w + 1) KS8) for both 16-unit tiles
    Frag bb[KS8][2];
    if (is_compute) {
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < KS8; ++i) {
            const int ks = w * KS8 + i;
            const int k = ks * 32 + 8 * (lane >> 4);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bb[i][n].u = ks < nks ? *reinterpret_cast<const uint4*>(whhT + ((size_t)d * H + j0 + n * 16 + (lane & 15)) * (3 * H) + k)
                                      : make_uint4(0, 0, 0, 0);
        }
    }
    // gate phase on waves 2 and 3: thread (row (tid - 128) / 32, unit tid % 32)
    const int b = ((tid - 128) >> 5) & 3, u0 = tid & 31;
    const int j = j0 + (u0 & ~1);
    const bool gate_wave = tid >= 128 && tid < 256;
    const bool act = gate_wave && b < Bl;
    constexpr int kPoller = 128;
    float carry = 0.f, sb[4] = {0.f, 0.f, 0.f, 0.f};
    // producers (workgroups of 32 units) behind this wave's K slices: slice ks = columns [32 ks, 32 ks + 32) of (gate, unit)
    unsigned long long my_producers = 0ull;
#pragma unroll
    for (int i = 0; i < KS8; ++i) {
        const int ks = w * KS8 + i;
        if (ks < nks) my_producers |= 1ull << (((32 * ks) % H) >> 5);
    }
    if (tid == 0) {
        *s_abort = 0;
        s_abort[1] = 0;
        s_abort[2] = 0;
        s_abort[3] = 0;
        if (LOCAL) {
            const int v = decide_local(sync, rec, nwg, abort_word, (forge & 1) ? 2 + (slot & 1) : 0);
            if (v < 0) *s_abort = 1; else s_abort[1] = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ASR_RAW_BARRIER();
    const bool local = LOCAL && s_abort[1] != 0;
    // data polling (forge bit 8, see fwd_persistent_io_kernel): dgh arrives filled with 0xffff and is its own signal
    const bool dp = local && (forge & 8) != 0;

    // byte offsets inside dgh (< 2^31, can_persist) kept per lane and stepped by one (t) row: no 64-bit products on the chain
    constexpr int NLc = (KS8 + 3) / 4;
    const unsigned row_bytes = (unsigned)((size_t)B * gs3 * 2);
    unsigned frag_off[NLc];
    bool frag_on[NLc];
#pragma unroll
    for (int l = 0; l < NLc; ++l) {
        const int r16 = lane & 15, row = r16 & 3, i = 4 * l + (r16 >> 2), ks = w * KS8 + i;
        frag_on[l] = i < KS8 && ks < nks && row < Bl;
        frag_off[l] = (unsigned)((((size_t)b0 + row) * gs3 + (size_t)d * 3 * H + ks * 32 + 8 * (lane >> 4)) * 2);
    }
    const unsigned store_off = (unsigned)((((size_t)b0 + b) * gs3 + (size_t)d * 3 * H + j) * 2);
    // hand-off fragments: zeroed ONCE (a lane whose fragment is off never loads); `ahead` = the gate waves' first attempt for
    // the next step, issued right behind their own stores (see fwd_persistent_io_kernel)
    Frag acur[NLc], ahead[NLc];
#pragma unroll
    for (int l = 0; l < NLc; ++l) { acur[l].u = make_uint4(0, 0, 0, 0); ahead[l].u = make_uint4(0, 0, 0, 0); }
    bool have_ahead = false;
    auto fetch_row = [&](Frag (&f)[NLc], int tq) {
#pragma unroll
        for (int l = 0; l < NLc; ++l)
            if (frag_on[l]) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(dghrsrc, frag_off[l] + (unsigned)tq * row_bytes, 0, 16 /* sc1 */);
                f[l].u = make_uint4(v[0], v[1], v[2], v[3]);
            }
    };
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? T - 1 - s : s;
        const int tn = d == 0 ? t + 1 : t - 1;
        float rcr = 0.f;
        float dyy = 0.f, r = 0.f, z = 0.f, n = 0.f, qq = 0.f, hp = 0.f;
        if (gate_wave) {
            const char* sl = opring + (s % BIO_GD) * BIO_SLOT;
            const float* of = reinterpret_cast<const float*>(sl) + b * 32 + u0;
            r = of[0]; z = of[128]; n = of[256]; qq = of[384];
            hp = s < T - 1 ? of[512] : 0.f;
            dyy = bf16_to_f32(reinterpret_cast<const uint16_t*>(sl + 5 * 512)[b * 32 + u0]);
        }
        if (s > 0) {
            if (dp) {
            } else if (local) { // every compute wave waits for the producers of ITS K slices only, no workgroup barrier
                if (is_compute && !wait_flags_mask(shards, nwg, my_producers, (unsigned)s, abort_word, lane) && lane == 0) *s_abort = 1;
            } else {
                if (tid == kPoller && !wait_shards<false>(shards, nwg, (unsigned)s, abort_word)) *s_abort = 1;
                ASR_RAW_BARRIER();
            }
            if (is_compute) {
                f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
                constexpr int NL = (KS8 + 3) / 4;
                Frag (&a)[NL] = acur;
                unsigned spins = 0;
                auto fetch = [&]() { fetch_row(a, tn); };       // (first attempt outside the retry loop: see fwd_persistent_io_kernel)
                if (have_ahead) {
#pragma unroll
                    for (int l = 0; l < NL; ++l) a[l].u = ahead[l].u;
                } else {
                    if (dp) {       // pure compute waves: start when this workgroup's gate waves have stored (all are in step)
                        unsigned nap = 0;
                        while ((lds_peek(s_abort + 2) < s || lds_peek(s_abort + 3) < s) && !lds_peek(s_abort) && ++nap < kSpinLimit)
                            __builtin_amdgcn_s_sleep(1);
                    }
                    fetch();
                }
                have_ahead = false;
                while (dp) {
                    bool missing = false;
#pragma unroll
                    for (int l = 0; l < NL; ++l)
                        missing |= a[l].u.x == 0xffffffffu || a[l].u.y == 0xffffffffu || a[l].u.z == 0xffffffffu || a[l].u.w == 0xffffffffu;
                    if (__ballot(missing) == 0ull) break;
                    if ((++spins & 63u) == 0u) {
                        if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) { if (lane == 0) *s_abort = 1; break; }
                        if (spins > kSpinLimit) {
                            if (lane == 0) { __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT); *s_abort = 1; }
                            break;
                        }
                    }
                    fetch();
                }
#pragma unroll
                for (int i = 0; i < KS8; ++i) {
                    Frag f;
                    f.u = shl_rows(a[i >> 2].u, i & 3);
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn) acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.v, bb[i][nn].v, acc[nn], 0, 0, 0);
                }
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
                    if (lane < 16) part[((dp ? (s & 1) * 8 : 0) + w) * 2 * 64 + nn * 64 + lane] = make_float4(acc[nn][0], acc[nn][1], acc[nn][2], acc[nn][3]);     // live rows 0..3 only
            }
            ASR_RAW_BARRIER();
            if ((s & 15) == 0 && *s_abort) break;      // (every 16 steps: the LDS read sat on the chain behind the barrier; polls give up at once anyway)
            if (act) {      // tile rows 0..3 live in lanes 0..15 (column = lane), component = row
                // (a reader-side layout -- two float4 per thread, eight scalar writes per wave -- measured no better here)
                const float* pf = reinterpret_cast<const float*>(part + (dp ? (s & 1) * 8 * 2 * 64 : 0)) + ((u0 >> 4) * 64 + (u0 & 15)) * 4 + b;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) rcr += pf[ww * 2 * 256];
            }
        }
        if (is_loader) {        // slot s + 1 must have landed; the two younger issues (3 instructions each) stay in flight
            issue(s + BIO_GD - 1);
            if (s + BIO_GD - 1 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (s + BIO_GD - 2 < T) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (is_storer) {
            if (s > 0) store_step(s - 1);
        } else if (gate_wave) {
            const float dh = dyy + carry + rcr;
            const float dn = dh * (1.0f - z);
            const float dz = dh * (hp - n);
            const float dan = dn * (1.0f - n * n);
            const float daz = dz * z * (1.0f - z);
            const float dq = dan * r;
            const float dar = dan * qq * r * (1.0f - r);
            carry = dh * z;
            const uint16_t ar = f32_to_bf16(dar), az = f32_to_bf16(daz), an = f32_to_bf16(dan), aq = f32_to_bf16(dq);
            if (act) { sb[0] += bf16_to_f32(ar); sb[1] += bf16_to_f32(az); sb[2] += bf16_to_f32(an); sb[3] += bf16_to_f32(aq); }
            const unsigned m1 = (unsigned)ar | ((unsigned)az << 16), m2 = (unsigned)an | ((unsigned)aq << 16);
            const unsigned o1 = lane_xor1_u32(m1), o2 = lane_xor1_u32(m2);
            const bool odd = u0 & 1;
            const unsigned e1 = odd ? o1 : m1, d1 = odd ? m1 : o1, e2 = odd ? o2 : m2, d2 = odd ? m2 : o2;
            unsigned pr_ = (e1 & 0xffffu) | (d1 << 16), pz_ = (e1 >> 16) | (d1 & 0xffff0000u);
            unsigned pn_ = (e2 & 0xffffu) | (d2 << 16), pq_ = (e2 >> 16) | (d2 & 0xffff0000u);
            if (dp) {       // (NaN pairs of a diverged run) never the sentinel
                if (pr_ == 0xffffffffu) pr_ = 0x7fc07fc0u;
                if (pz_ == 0xffffffffu) pz_ = 0x7fc07fc0u;
                if (pq_ == 0xffffffffu) pq_ = 0x7fc07fc0u;
            }
            const unsigned ob = store_off + (unsigned)t * row_bytes;
            char* dghb = reinterpret_cast<char*>(dgh);
            __builtin_amdgcn_s_waitcnt(0x0F70);      // nothing of this wave is in flight here: clears the compiler's scoreboard
            unsigned* od = oring + (size_t)(s & 1) * 3 * 4 * 16 + b * 16 + (u0 >> 1);
            if (act && !odd) {
                if (local) {
                    __builtin_amdgcn_raw_buffer_store_b32(pr_, dghrsrc, ob, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(pz_, dghrsrc, ob + 2u * (unsigned)H, 0, 0);
                } else {
                    __hip_atomic_store(reinterpret_cast<unsigned*>(dghb + ob), pr_, ASR_RLX_AGENT);
                    __hip_atomic_store(reinterpret_cast<unsigned*>(dghb + ob + 2u * (unsigned)H), pz_, ASR_RLX_AGENT);
                }
                od[0] = pr_;
            }
            if (act && odd) {
                if (local) __builtin_amdgcn_raw_buffer_store_b32(pq_, dghrsrc, ob + 4u * (unsigned)H, 0, 0);
                else __hip_atomic_store(reinterpret_cast<unsigned*>(dghb + ob + 4u * (unsigned)H), pq_, ASR_RLX_AGENT);
                od[4 * 16] = pz_; od[2 * 4 * 16] = pn_;
            }
            if (dp && s + 1 < T) {      // row t is what step s + 1 reads
                if (lane == 0) lds_poke(s_abort + w, s + 1);       // (waves 2 and 3 -> words 2 and 3)
                fetch_row(ahead, t);
                have_ahead = true;
            }
            if (!dp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (dp) {
            if (s == 0) ASR_RAW_BARRIER();
            continue;
        }
        ASR_RAW_BARRIER();
        if (tid == kPoller) {
            if (local) set_flag(shards + slot, (unsigned)s + 1u, true);
            else __hip_atomic_fetch_add(my_shard, 1u, ASR_RLX_AGENT);
        }
    }
    if (is_loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASR_RAW_BARRIER();
    if (is_storer && !*s_abort) store_step(T - 1);
    if (act && db_ih && db_hh) {
        float* bi = db_ih + (size_t)d * 3 * H + j0 + u0;
        float* bh2 = db_hh + (size_t)d * 3 * H + j0 + u0;
        atomicAdd(bi, sb[0]); atomicAdd(bi + H, sb[1]); atomicAdd(bi + 2 * H, sb[2]);
        atomicAdd(bh2, sb[0]); atomicAdd(bh2 + H, sb[1]); atomicAdd(bh2 + 2 * H, sb[3]);
    }
}

// ------------------------------------------------------------------------------------------------ backward, partial sums
// bwd_wide_kernel hands the 3H gate gradients of a step to every workgroup of the recurrence (12 KB per workgroup and step at
// H = 512, 4 rows), each of which then forms ITS 32 columns of  dh = dgh . W_hh  -- and the time of a step grows by ~0.04 us
// per KB a workgroup has to fetch through the L2 (what-if build fetching 8 KB: 1.80 -> 1.64 us per step).  Here the product
// moves to the producer: a workgroup multiplies the 96 gate gradients it has just computed (its 32 units, gates r, z, q) by
// its 96 ROWS of W_hh and publishes the H partial sums  P[p][h'][row] = sum_{g in p's 96} dgh[row][g] W_hh[g][h']  in bf16
// (4 rows x H x 2 B = 4 KB written); a consumer fetches the 32 columns it owns from all H / 32 producers (256 B each: 4 KB
// at H = 512) and adds them up (float32).  Same number of MFMAs, a third of the hand-off bytes:
//   gate waves (2, 3):  poll-load P_{s-1} (sentinel 0xffff, as in bwd_wide_kernel): lane k of an eight-lane group the pieces of producers
//                       k, k + 8 for the group's unit pair -> lane-local sums -> reduce-scatter over the group with three DPP exchanges
//                       (round 4; until then: float32 in LDS, a barrier, a 16-way sum) -> gate math -> (ar, az, aq) as the MFMA A
//                       image in LDS + (ar, az, an, aq) for the storer
//   all 8 compute waves: barrier A (the step's only one) -> 3 K steps (gates) x H/128 column tiles of MFMA, the four live rows read
//                       into every 4-row group of the tile -> bf16 -> P_s and the re-arm as ONE 64-lane store each (plain stores when
//                       the recurrence sits on one XCD, write-through otherwise; polled with sc1 loads either way)
//   loader / storer waves as in bwd_wide_kernel, cued by an LDS word the gate wave raises when its poll has succeeded; the storer
//   also writes dgh (no longer the exchange medium, so it needs no sentinel fill: 196 MB of memset per launch less).
// The exchange buffer is a ring of four steps (a producer re-arms the slot of step s - 2 with the sentinel once its loads of
// step s - 1 have succeeded: every consumer has then finished with step s - 2), 4 x H/32 x H x 8 B per recurrence inside sync_ws.
constexpr int PS_RING = 4;
__host__ __device__ inline size_t ps_exchange_bytes(int nrec_pad, int H) { return (size_t)PS_RING * nrec_pad * (H / 32) * H * 8; }
constexpr size_t kPsOffset = 4096 + kShardBytes;      // behind the control words and the sharded counters

// G16: the saved gates arrive as IEEE half in the blocked layout (fwd_persistent_io_kernel<.., G16>): ONE LDS-DMA instruction brings
// r | z | n | q of a step (4 rows x 256 contiguous bytes) instead of two (8 x 128-B runs each).
template <int NT, bool LOCAL, bool G16 = false>
__global__ __launch_bounds__(640, 3) void bwd_ps_kernel(const uint16_t* __restrict__ dy, const float* __restrict__ gates,
                                                        const float* __restrict__ hseq, const uint16_t* __restrict__ whhT,
                                                        uint16_t* __restrict__ dgi, uint16_t* __restrict__ dgh, float* __restrict__ db_ih,
                                                        float* __restrict__ db_hh, unsigned* sync, int T, int B, int H, int ndir,
                                                        int forge, int boff, int Bn) {
    // (boff, Bn): the slab of batch rows this launch serves (see fwd_persistent_io_kernel)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RED_PITCH = 144;          // (the first 8 x 144 floats held the partial sums of the eight lane groups until round 4)
    // MFMA A image [3 gates][16 rows][32 units] bf16 = 3 KB, of which rows 0 .. 3 are written: every lane reads row lane % 4 (an LDS
    // broadcast), unconditionally -- three ds_reads back to back and ONE wait
    // (chunk lane / 16 of its row; rows are 64 B apart, so the four rows x four chunks of a ds_read_b128 lane group cover all 64 banks)
    uint16_t* aimg = reinterpret_cast<uint16_t*>(smem + 8 * RED_PITCH * 4);
    char* opring = smem + 8 * RED_PITCH * 4 + 3072;                                     // [BIO_GD][BIO_SLOT] (as bwd_wide_kernel)
    unsigned* oring = reinterpret_cast<unsigned*>(opring + BIO_GD * BIO_SLOT);         // [2][4: ar az an aq][4 rows][16 pairs]
    int* s_abort = reinterpret_cast<int*>(oring + 2 * 4 * 4 * 16);
    constexpr int rows = 4;
    const int nwg = H / 32;
    const int Gn = (Bn + rows - 1) / rows, nrec = Gn * ndir, nrec_pad = (nrec + 7) & ~7;
    const int rec = LOCAL ? (int)(blockIdx.x % nrec_pad) : (int)(blockIdx.z * gridDim.y + blockIdx.y);
    const int slot = LOCAL ? (int)(blockIdx.x / nrec_pad) : (int)blockIdx.x;
    if (LOCAL && rec >= nrec) return;
    const int d = rec / Gn, g = rec % Gn;
    const int j0 = slot * 32;
    const int b0 = boff + g * rows, Bl = min(rows, boff + Bn - b0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_compute = w < 8, is_loader = w == 8, is_storer = w == 9;
    const size_t hs = (size_t)ndir * H, gs3 = (size_t)ndir * 3 * H;
    unsigned* abort_word = sync + 1023;
    // exchange ring of this recurrence: [PS_RING][nwg producers][H columns][4 rows] bf16
    char* xbase = reinterpret_cast<char*>(sync) + kPsOffset + (size_t)rec * PS_RING * nwg * H * 8;
    const unsigned xslot_bytes = (unsigned)(nwg * H * 8);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, (int)(PS_RING * xslot_bytes), 0x00020000);
    const long long tstep = d == 0 ? -1 : 1;
    const int tfirst = d == 0 ? T - 1 : 0;

    // ---- loader (wave 8): as bwd_wide_kernel
    const int lrow = (lane & 31) >> 3, lyrow = (lane - 32) >> 2;
    const float* lgp = gates + (((size_t)tfirst * B + b0 + (lrow < Bl ? lrow : 0)) * ndir + d) * 4 * H + j0 + (lane & 7) * 4 +
                       (size_t)(lane >> 5) * H;
    // G16 (blocked layout [T*B][ndir][H / 16][4 gates][16 units] half): the 32 units of this workgroup are two adjacent blocks = 256
    // contiguous bytes per row: lane = (16-B piece lane / 4, row lane % 4), so that the LDS image (lane-linear) is [piece][row]: the four
    // rows of a piece are 16 B apart and a gate wave -- whose lanes span all four rows since the DPP reduction -- reads 64 different
    // halves out of 128 contiguous bytes per gate (in the [row][piece] image rows were 256 B apart: a 4-way bank conflict per read)
    const int lrow16 = lane & 3;
    const uint16_t* lgp16 = reinterpret_cast<const uint16_t*>(gates) + ((((size_t)tfirst * B + b0 + (lrow16 < Bl ? lrow16 : 0)) * ndir + d) * (H >> 4) +
                                                                        (j0 >> 4)) * 64 + (lane >> 2) * 8;
    // h_{t-1}: likewise [piece of 4 units][row] float32
    const int hrow = lane & 3;
    const float* lhp = hseq + ((long long)(d == 0 ? tfirst - 1 : tfirst + 1) * B + b0 + (hrow < Bl ? hrow : 0)) * (long long)hs +
                       (size_t)d * H + j0 + ((lane & 31) >> 2) * 4;
    const uint16_t* lyp = dy + ((size_t)tfirst * B + b0 + (lane >= 32 && lane < 48 && lyrow < Bl ? lyrow : 0)) * H + j0 + (lane & 3) * 8;
    const char* l2p = lane < 32 ? reinterpret_cast<const char*>(lhp) : reinterpret_cast<const char*>(lyp);
    const long long lgs = tstep * (long long)B * ndir * 4 * H;
    const long long l2s = lane < 32 ? tstep * (long long)B * (long long)hs * 4 : tstep * (long long)B * H * 2;      // bytes
    auto issue = [&](int sq) {              // called with sq = 0, 1, 2, ... in order
        if (sq < T) {
            char* sl = opring + (sq % BIO_GD) * BIO_SLOT;
            if (G16) {
                if (lrow16 < Bl) __builtin_amdgcn_global_load_lds((glb_ptr_t)lgp16, (lds_ptr_t)sl, 16, 0, 0);
            } else if (lrow < Bl) {
                __builtin_amdgcn_global_load_lds((glb_ptr_t)lgp, (lds_ptr_t)sl, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(lgp + 2 * (size_t)H), (lds_ptr_t)(sl + 1024), 16, 0, 0);
            }
            if (lane < 32 ? (hrow < Bl && sq < T - 1) : (lane < 48 && lyrow < Bl))
                __builtin_amdgcn_global_load_lds((glb_ptr_t)l2p, (lds_ptr_t)(sl + 2048), 16, 0, 0);
        }
        lgp += lgs; lgp16 += lgs; l2p += l2s;
    };
    // ---- storer (wave 9): dgi = (ar, az, an) and dgh = (ar, az, aq): 3 gates x 4 rows x 64 B = 48 pieces of 16 B each
    auto store_step = [&](int sp) {
        const long long tq = tfirst + tstep * sp;
        const unsigned* src = oring + (size_t)(sp & 1) * 4 * 4 * 16;
        const int gsel = lane >> 4, row = (lane & 15) >> 2, c = lane & 3;
        if (lane < 48 && row < Bl) {
            const size_t off = ((size_t)tq * B + b0 + row) * gs3 + (size_t)d * 3 * H + gsel * H + j0 + c * 8;
            // (raw LDS reads: see lds_read16_raw)
            *reinterpret_cast<f32x4_asm*>(dgi + off) = lds_read16_raw(src + (gsel * 4 + row) * 16 + c * 4);
            *reinterpret_cast<f32x4_asm*>(dgh + off) = lds_read16_raw(src + ((gsel == 2 ? 3 : gsel) * 4 + row) * 16 + c * 4);
        }
    };
    if (is_loader) {
        for (int s0 = 0; s0 < BIO_GD; ++s0) issue(s0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ---- weights: wave w owns the dh columns [w H/8, (w + 1) H/8) = NT tiles of 16; K step gg = gate gg of this workgroup's units
    Frag bb[3][NT];
    const int ncol0 = w * (H / 8);
    if (is_compute) {
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int gg = 0; gg < 3; ++gg)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                bb[gg][n].u = *reinterpret_cast<const uint4*>(whhT + ((size_t)d * H + ncol0 + n * 16 + (lane & 15)) * (3 * H) + (size_t)gg * H + j0 +
                                                              8 * (lane >> 4));
    }
    // gate phase on waves 2 and 3.  Gate lane gl = tid - 128 belongs to the eight-lane group gl / 8 = unit pair `seg` and is thread
    // (row gl % 4, unit 2 seg + (gl / 4) % 2): the eight lanes of a group fetch the 16 producers' pieces of THEIR unit pair (two each) and
    // reduce-scatter them among themselves with DPP moves (see below) -- every lane ends up with the exchanged sum of its own (row, unit)
    // without the LDS image and the workgroup barrier that stood between the poll and the gate math
    const int b = (tid - 128) & 3, u0 = ((((tid - 128) >> 3) & 15) << 1) | (((tid - 128) >> 2) & 1);
    const bool gate_wave = tid >= 128 && tid < 256;
    const bool act = gate_wave && b < Bl;
    float carry = 0.f, sb[4] = {0.f, 0.f, 0.f, 0.f};
    if (tid == 0) {
        *s_abort = 0;
        s_abort[1] = 0;
        s_abort[2] = 0;
        if (LOCAL) {
            const int v = decide_local(sync, rec, nwg, abort_word, (forge & 1) ? 2 + (slot & 1) : 0);
            if (v < 0) *s_abort = 1; else s_abort[1] = v;
        }
    }
    for (int i = tid; i < 768; i += 640) reinterpret_cast<unsigned*>(aimg)[i] = 0u;              // (rows >= Bl stay zero)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ASR_RAW_BARRIER();
    const bool local = LOCAL && s_abort[1] != 0;
    // -DASR_STAMP: cycles per phase and wave (s_memtime), read back by tools/stamp_gru_ps.py
#ifdef ASR_STAMP
    unsigned long long ps_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ps_last = __builtin_amdgcn_s_memtime();
#define ASR_PS(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ps_acc[i] += n_ - ps_last; ps_last = n_; }
#else
#define ASR_PS(i)
#endif

    // hand-off loads of the gate waves: 128 lanes x 2 pieces of 16 B = H/32 (<= 16.. 32) producers x 256 B.  Lane gl (group seg = gl / 8,
    // k = gl % 8) fetches producers k, k + 8, ... : columns j0 + 2 seg, + 1 (x 4 rows).  (H = 1024: 32 producers -> 4 pieces per lane.)
    constexpr int NP = NT <= 4 ? 2 : 4;
    const int gl = tid - 128;                                   // 0 .. 127 on the gate waves
    unsigned poff[NP];
    bool pon[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int pp = (gl & 7) + 8 * i, seg = (gl >> 3) & 15;
        pon[i] = gate_wave && pp < nwg;
        poff[i] = (unsigned)(((size_t)pp * H + j0 + 2 * seg) * 8);
    }
    uint4 pcur[NP], pahead[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { pcur[i] = make_uint4(0, 0, 0, 0); pahead[i] = make_uint4(0, 0, 0, 0); }
    auto fetch_p = [&](uint4 (&f)[NP], int ring) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (pon[i]) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, poff[i] + (unsigned)ring * xslot_bytes, 0, 16 /* sc1 */);
                f[i] = make_uint4(v[0], v[1], v[2], v[3]);
            }
    };
    // this wave's stores into the ring: tile n, lanes 0..15 (column ncol0 + 16 n + lane), 4 rows = 8 B
    const unsigned soff = (unsigned)(((size_t)slot * H + ncol0 + (lane & 15)) * 8);

    for (int s = 0; s < T; ++s) {
        float rcr = 0.f;
        float dyy = 0.f, r = 0.f, z = 0.f, n = 0.f, qq = 0.f, hp = 0.f, dyc = 0.f, f_an = 0.f, f_az = 0.f, f_ar = 0.f;
        if (gate_wave) {
            const char* sl = opring + (s % BIO_GD) * BIO_SLOT;
            const float* of = reinterpret_cast<const float*>(sl) + b * 32 + u0;
            if (G16) {
                // slot image [16 pieces: (block, gate, unit half)][4 rows][8 units] half
                const _Float16* oh = reinterpret_cast<const _Float16*>(sl) + ((((u0 >> 4) * 8 + ((u0 & 15) >> 3)) * 4 + b) << 3) + (u0 & 7);
                r = (float)oh[0]; z = (float)oh[64]; n = (float)oh[128]; qq = (float)oh[192];
            } else {
                r = of[0]; z = of[128]; n = of[256]; qq = of[384];
            }
            hp = s < T - 1 ? reinterpret_cast<const float*>(sl)[512 + (((u0 >> 2) * 4 + b) << 2) + (u0 & 3)] : 0.f;
            dyy = bf16_to_f32(reinterpret_cast<const uint16_t*>(sl + 5 * 512)[b * 32 + u0]);
            dyc = dyy + carry;                          // dh = dy + z dh' + (the exchanged sum)
            f_an = (1.0f - z) * (1.0f - n * n);         // dan = dh (1 - z)(1 - n^2)
            f_az = (hp - n) * z * (1.0f - z);           // daz = dh (h' - n) z (1 - z)
            f_ar = qq * r * (1.0f - r);                 // dar = dan q r (1 - r)
            ASR_PS(0)
            if (s > 0) {
                // P_{s-1}: the first attempt was issued right behind this wave's own stores of the previous step
#pragma unroll
                for (int i = 0; i < NP; ++i) pcur[i] = pahead[i];
                unsigned spins = 0;
                for (;;) {
                    bool missing = false;
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        missing |= pcur[i].x == 0xffffffffu || pcur[i].y == 0xffffffffu || pcur[i].z == 0xffffffffu || pcur[i].w == 0xffffffffu;
                    if (__ballot(missing) == 0ull) break;
                    if ((++spins & 63u) == 0u) {
                        if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) { if (lane == 0) *s_abort = 1; break; }
                        if (spins > kSpinLimit) {
                            if (lane == 0) { __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT); *s_abort = 1; }
                            break;
                        }
                    }
                    fetch_p(pcur, (s - 1) & (PS_RING - 1));
                }
                ASR_PS(1)
#ifdef ASR_STAMP
                ps_acc[7] += spins + 1;
#endif
                if (tid == 128) lds_poke(s_abort + 2, s);           // (the storer's cue)
                // lane-local sum over this lane's producers, then [group = gl / 16][row][unit parity][seg] float32 in LDS
                float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const unsigned wds[4] = {pcur[i].x, pcur[i].y, pcur[i].z, pcur[i].w};      // unit e = word / 2: rows 2 (word % 2), + 1
                    if (pon[i]) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            acc8[2 * k] += bf16_to_f32((uint16_t)(wds[k] & 0xffffu));
                            acc8[2 * k + 1] += bf16_to_f32((uint16_t)(wds[k] >> 16));
                        }
                    }
                }
                // acc8[j], j = 4 (unit parity) + row, is lane j's (of this eight-lane group) share held by this lane: reduce-scatter over the
                // group in three DPP exchanges -- partner 7 - k (row_half_mirror) hands over the half it does not keep, then k ^ 2 and k ^ 1
                // (quad_perm) -- 7 additions and 14 selects on the wave instead of 8 LDS writes, a workgroup barrier and 8 LDS reads
                const bool lo4 = (gl & 4) == 0, lo2 = (gl & 2) == 0, lo1 = (gl & 1) == 0;
                float k4[4], k2[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) k4[i] = (lo4 ? acc8[i] : acc8[4 + i]) + dpp_mov_f32<0x141>(lo4 ? acc8[4 + i] : acc8[i]);
#pragma unroll
                for (int i = 0; i < 2; ++i) k2[i] = (lo2 ? k4[i] : k4[2 + i]) + dpp_mov_f32<0x4E>(lo2 ? k4[2 + i] : k4[i]);
                rcr = (lo1 ? k2[0] : k2[1]) + dpp_mov_f32<0xB1>(lo1 ? k2[1] : k2[0]);
            }
        }
        ASR_PS(2)
        ASR_PS(3)
        if (gate_wave) {
            // (the factors that do not depend on dh were formed at the top of the step, under the hand-off's round trip)
            const float dh = dyc + rcr;
            const float dan = dh * f_an;
            const float daz = dh * f_az;
            const float dq = dan * r;
            const float dar = dan * f_ar;
            carry = dh * z;
            const uint16_t ar = f32_to_bf16(dar), az = f32_to_bf16(daz), an = f32_to_bf16(dan), aq = f32_to_bf16(dq);
            if (act) { sb[0] += bf16_to_f32(ar); sb[1] += bf16_to_f32(az); sb[2] += bf16_to_f32(an); sb[3] += bf16_to_f32(aq); }
            if (act) {
                // MFMA A image [gate][row][unit] and the storer's [array][row][unit] (pairs of units per dword)
                aimg[(0 * 16 + b) * 32 + u0] = ar; aimg[(1 * 16 + b) * 32 + u0] = az; aimg[(2 * 16 + b) * 32 + u0] = aq;
                uint16_t* od = reinterpret_cast<uint16_t*>(oring + (size_t)(s & 1) * 4 * 4 * 16) + b * 32 + u0;
                od[0] = ar; od[128] = az; od[256] = an; od[384] = aq;
            }
        }
        ASR_PS(4)
        ASR_RAW_BARRIER();                  // (A) the gate gradients of this step are in LDS -- the step's only workgroup barrier
        ASR_PS(5)
        if ((s & 15) == 0 && lds_peek(s_abort)) break;
        if (is_loader) {
            // the slot step s read (before barrier A) is refilled once the gate waves hold the exchanged sums of step s + 1 (the storer's cue
            // below: an LDS-DMA in front of the payload stores and polls costs the chain ~0.05 us); then wait until step s + 2 has landed,
            // which is read behind A(s + 1)
            if (s + 1 < T) {
                unsigned nap = 0;
                while (lds_peek(s_abort + 2) < s + 1 && !lds_peek(s_abort) && ++nap < kSpinLimit) __builtin_amdgcn_s_sleep(1);
            }
            issue(s + BIO_GD);
            const int left = T - 1 - (s + 2);
            // (in flight afterwards: the issues of the two youngest steps, 3 LDS-DMA instructions each -- 2 with half gates)
            if (left >= BIO_GD - 2) { if (G16) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
            else if (left == 1) { if (G16) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (is_storer) {
            // this step's gate gradients (staging slot s % 2: written again behind A(s + 2)) -- once the gate waves hold the exchanged sums
            // of the NEXT step, i.e. when the payload / re-arm stores and the polls of the compute waves have left the CU's memory
            // pipeline (where barrier R used to put these stores)
            if (s + 1 < T) {
                unsigned nap = 0;
                while (lds_peek(s_abort + 2) < s + 1 && !lds_peek(s_abort) && ++nap < kSpinLimit) __builtin_amdgcn_s_sleep(1);
            }
            store_step(s);
        }
        if (is_compute && s + 1 < T) {
            f32x4 acc[NT];
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) acc[nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
            Frag a[3];
            // every 4-row group of the 16-row MFMA tile reads the SAME four live rows (an LDS broadcast read: rows lane % 4, chunk lane / 16),
            // so every lane group q = lane / 16 holds a copy of every tile's result and stores tile q (+ 4, ...): ONE 64-lane store instruction
            // per four tiles instead of four 16-lane ones, for the payload and for the re-arm.  The CU's memory pipeline took 64 store
            // instructions per step from the eight compute waves; 16 now.  Same box, T=1000, B=32, H=512: 1.352 -> 1.255 us per step together
            // with the DPP reduction of the gate waves and the first poll 8 x 64 cycles behind the stores (either change alone: 1.38 / 1.28).
            const int aq = lane >> 4;
#pragma unroll
            for (int gg = 0; gg < 3; ++gg) a[gg].u = *reinterpret_cast<const uint4*>(aimg + (gg * 16 + (lane & 3)) * 32 + 8 * aq);
#pragma unroll
            for (int gg = 0; gg < 3; ++gg)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[gg].v, bb[gg][nn].v, acc[nn], 0, 0, 0);
            const unsigned ring_off = (unsigned)(s & (PS_RING - 1)) * xslot_bytes, rearm_off = (unsigned)((s + 2) & (PS_RING - 1)) * xslot_bytes;
            {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const int q = lane >> 4;
                constexpr int NG = (NT + 3) / 4;
#pragma unroll
                for (int g4 = 0; g4 < NG; ++g4) {
                    unsigned lo = 0u, hi = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (g4 * 4 + j < NT) {
                            const unsigned l = pack_bf16x2(acc[g4 * 4 + j][0], acc[g4 * 4 + j][1]), h2 = pack_bf16x2(acc[g4 * 4 + j][2], acc[g4 * 4 + j][3]);
                            lo = q == j ? l : lo;
                            hi = q == j ? h2 : hi;
                        }
                    }
                    if (lo == 0xffffffffu) lo = 0x7fc07fc0u;            // (NaN pairs of a diverged run) never the sentinel
                    if (hi == 0xffffffffu) hi = 0x7fc07fc0u;
                    // the slot of step s - 2 is re-armed behind it (every consumer finished with it before it could produce the P_{s-1} this
                    // workgroup has just consumed); the slots of steps s + 1, s + 3 are armed (launch fill / earlier re-arm)
                    const bool on = g4 * 4 + q < NT;
                    const unsigned o = soff + (unsigned)(g4 * 4 + q) * 128u;
                    if (on) {
                        if (local) {
                            __builtin_amdgcn_raw_buffer_store_b64((u32x2){lo, hi}, xrsrc, o + ring_off, 0, 0);
                            if (s >= 2) __builtin_amdgcn_raw_buffer_store_b64((u32x2){0xffffffffu, 0xffffffffu}, xrsrc, o + rearm_off, 0, 0);
                        } else {
                            __hip_atomic_store(reinterpret_cast<unsigned*>(xbase + o + ring_off), lo, ASR_RLX_AGENT);
                            __hip_atomic_store(reinterpret_cast<unsigned*>(xbase + o + ring_off + 4u), hi, ASR_RLX_AGENT);
                            if (s >= 2) {
                                __hip_atomic_store(reinterpret_cast<unsigned*>(xbase + o + rearm_off), 0xffffffffu, ASR_RLX_AGENT);
                                __hip_atomic_store(reinterpret_cast<unsigned*>(xbase + o + rearm_off + 4u), 0xffffffffu, ASR_RLX_AGENT);
                            }
                        }
                    }
                }
            }
            // P_s: the first attempt, kBwdPollDelay after this wave's own stores (see kFirstPollDelay)
            if (gate_wave) {
                if (kBwdPollDelay > 0) __builtin_amdgcn_s_sleep(kBwdPollDelay);
                fetch_p(pahead, s & (PS_RING - 1));
            }
        }
        ASR_PS(6)
    }
    if (is_loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASR_RAW_BARRIER();
    if (act && db_ih && db_hh) {
        float* bi = db_ih + (size_t)d * 3 * H + j0 + u0;
        float* bh2 = db_hh + (size_t)d * 3 * H + j0 + u0;
        atomicAdd(bi, sb[0]); atomicAdd(bi + H, sb[1]); atomicAdd(bi + 2 * H, sb[2]);
        atomicAdd(bh2, sb[0]); atomicAdd(bh2 + H, sb[1]); atomicAdd(bh2 + 2 * H, sb[3]);
    }
#ifdef ASR_STAMP
    if (blockIdx.x < 32 && (blockIdx.x & 15) == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(sync + 1024)[(((blockIdx.x >> 4) * 10) + w) * 8 + i] = ps_acc[i];
#endif
}

// The default forward kernel: 16 hidden units per workgroup, batch rows in independent recurrences of at
// most 8 rows, 4 compute waves (K = H split in 4) + loader (gi ring) + storer (f32 state and the four saved gate arrays,
// one step behind).  The exchanged payload is the bf16 h row (hseq16), written sc1 by the gate threads themselves.
// Raw barriers (s_barrier + lgkmcnt(0)): __syncthreads() would also wait for the loader's LDS-DMA in flight.
// GI16: the input projections arrive in bf16 (asr_gemm_nt with a bf16 output: half the bytes written by the projection and
// read here, ONE LDS-DMA instruction per step instead of two): ring slot = [3 gates][8 rows][16 units] bf16 = 768 B.
// G16: the saved gates (r, z, n, q) go to memory as IEEE half (r, z, n lie in [-1, 1], q = gh_n is O(1): 11 significant bits, i.e.
// 2^-11 relative rounding -- eight times finer than the bf16 every activation of the path carries) instead of float32: the storer
// wave converts on its way from the LDS staging ring to memory (the gate waves, i.e. the step's critical chain, are untouched),
// 2 store instructions per step instead of 3 and half the gate bytes in the CU's memory queue beside the hand-off (what-if builds of
// round 2: -5 % per recurrence).  The backward kernel that reads them is bwd_ps_kernel<.., G16 = true>.
// MINW: waves per SIMD the register budget has to allow (__launch_bounds__): 3 (<= 168 VGPRs; the kernel takes 130) for the default, one
// workgroup per CU; 4 (<= 128) for the four-row what-if of fwd_io_launch, where TWO workgroups share a CU -- a 6-wave workgroup lands on
// the SIMDs as 2, 2, 1, 1, two of them as up to 4, 4, 2, 2, and at 136 allocated registers a fourth wave does not fit a SIMD: the second
// workgroup of every CU then waits until the first has LEFT, which a persistent recurrence never does (measured: every workgroup with
// id >= 256 arrived only after the first 256 had given up).
template <int KSW, bool LOCAL, bool GI16, bool RING, bool G16 = false, int MINW = 3>
__global__ __launch_bounds__(384, MINW) void fwd_persistent_io_kernel(const void* __restrict__ gi_, const uint16_t* __restrict__ whh,
                                                                const float* __restrict__ bhh, float* __restrict__ hseq,
                                                                uint16_t* hseq16, float* __restrict__ gates, unsigned* sync,
                                                                int T, int B, int H, int ndir, int rows, int forge, int boff, int Bn) {
    // (boff, Bn): this launch serves the batch rows [boff, boff + Bn) of the B rows a time step holds -- batches beyond the resident
    // limit run as consecutive slabs of <= 32 rows (utterances are independent: asr_gru_fwd)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* part = reinterpret_cast<float4*>(smem);                                   // [2 (step parity)][4 waves][3 gates][64]
    float* opring = reinterpret_cast<float*>(part + 2 * 4 * 3 * 64);                  // [BIO_GD][3 gates][8 rows][16 units]
    float* oring = opring + BIO_GD * 3 * 8 * 16;                                      // [2][5: h r z n q][8 rows][16 units]
    int* s_abort = reinterpret_cast<int*>(oring + 2 * 5 * 8 * 16);
    const int G_ = LOCAL ? (Bn + rows - 1) / rows : (int)gridDim.y;
    // (MINW == 4: sixteen recurrences of four rows, two workgroups per CU -- the what-if of fwd_io_launch)
    constexpr int recbits = MINW == 4 ? 4 : 3;
    const int rec = LOCAL ? (int)(blockIdx.x & ((1 << recbits) - 1)) : (int)(blockIdx.z * gridDim.y + blockIdx.y);
    if (LOCAL && rec >= G_ * ndir) return;
    const int d = rec / G_, g = rec % G_;
    const int j0 = (LOCAL ? (int)(blockIdx.x >> recbits) : (int)blockIdx.x) * 16, nwg = H / 16;
    const int b0 = boff + g * rows, Bl = min(rows, boff + Bn - b0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_compute = w < 4, is_loader = w == 4, is_storer = w == 5;
    const int nks = H >> 5;
    constexpr bool PAIRED = KSW % 2 == 0;
    const size_t hs = (size_t)ndir * H;
    unsigned* shards = shard_base(sync, rec);
    unsigned* my_shard = shards + ((j0 >> 4) % NSH) * 32;
    unsigned* abort_word = sync + 1023;
    const __amdgpu_buffer_rsrc_t h16rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)hseq16, 0, (int)((size_t)T * B * hs * 2), 0x00020000);
    const long long tstep = d == 0 ? 1 : -1;
    const int tfirst = d == 0 ? 0 : T - 1;
    const size_t row_bytes_ = (size_t)B * hs * 2;           // one time step of hseq16

    // loader (wave 4), LDS-DMA straight into the ring slot (lane-linear image = [3 gates][8 rows][16 units] f32), no data
    // registers: TWO instructions per step -- all 64 lanes fetch gates r | z (lane / 32), lanes 0..31 gate n; lane % 32 =
    // (row, units 4 (lane % 4) ..).  Every LDS-DMA instruction of a step costs the recurrence about 0.05 us (measured with
    // three, one and none): the hand-off loads of the compute waves return behind it.
    const float* gi = reinterpret_cast<const float*>(gi_);
    const uint16_t* gi16 = reinterpret_cast<const uint16_t*>(gi_);
    // the gate thread's (row, unit): lane l of gate wave g (= 0, 1) is row 4 g + (l / 2) % 4, unit 2 (l / 8) + l % 2 -- units u, u + 1 stay
    // lanes l, l ^ 1 (the payload pair), and the partial-sum reads of a 32-lane group (dword 4 u + row % 4) cover 32 consecutive dwords:
    // with (row l / 16, unit l % 16) units u and u + 8 met in one bank, a 2-way conflict on each of the step's twelve ds_read_b32
    const int b_ = (((tid - 128) >> 6) << 2) | (((tid - 128) >> 1) & 3), u_ = ((((tid - 128) >> 3) & 7) << 1) | (tid & 1);
    // GI16: lane = (gate lane / 16, row (lane % 16) / 2, units 8 (lane % 2) ..), lanes 0..47
    const int lrow = GI16 ? (lane & 15) >> 1 : (lane & 31) >> 2;
    const float* lgp = gi + ((size_t)tfirst * B + b0 + (lrow < Bl ? lrow : 0)) * (3 * hs) + (size_t)d * 3 * H + j0 + (lane & 3) * 4 +
                       (size_t)(lane >> 5) * H;
    const uint16_t* lgp16 = gi16 + ((size_t)tfirst * B + b0 + (lrow < Bl ? lrow : 0)) * (3 * hs) + (size_t)d * 3 * H + j0 + (lane & 1) * 8 +
                            (size_t)(lane >> 4) * H;
    const long long lstride = tstep * (long long)B * 3 * (long long)hs;
    auto issue = [&](int sq) {
        if (GI16) {
            if (sq < T && lane < 48 && lrow < Bl) {
                char* slot = reinterpret_cast<char*>(opring) + (sq % BIO_GD) * (3 * 512);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)lgp16, (lds_ptr_t)slot, 16, 0, 0);
            }
            lgp16 += lstride;
        } else {
            if (sq < T && lrow < Bl) {
                char* slot = reinterpret_cast<char*>(opring) + (sq % BIO_GD) * (3 * 512);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)lgp, (lds_ptr_t)slot, 16, 0, 0);
                if (lane < 32) __builtin_amdgcn_global_load_lds((glb_ptr_t)(lgp + 2 * (size_t)H), (lds_ptr_t)(slot + 1024), 16, 0, 0);
            }
            lgp += lstride;
        }
    };
    // this step's (r, z, n) input projections of gate thread (b, u) out of ring slot `sl`
    auto read_gi = [&](int sl, float& gr, float& gz, float& gn) {
        if (GI16) {
            const uint16_t* o16 = reinterpret_cast<const uint16_t*>(opring) + sl * (3 * 256) + b_ * 16 + u_;
            gr = bf16_to_f32(o16[0]); gz = bf16_to_f32(o16[128]); gn = bf16_to_f32(o16[256]);
        } else {
            const float* osrc = opring + (size_t)sl * 3 * 8 * 16 + b_ * 16 + u_;
            gr = osrc[0]; gz = osrc[128]; gn = osrc[256];
        }
    };
    // storer: 5 arrays x 8 rows x 64 B = 160 pieces: piece p = lane + 64 i (i < 3, p < 160): array p / 32, row (p % 32) / 4
    // G16: the state's 32 pieces as they are, then 4 gate arrays x 8 rows x 32 B of halves = 64 pieces (two LDS reads each)
    auto store_step = [&](int sp) {
        const long long tq = tfirst + tstep * sp;
        const float* src = oring + (size_t)(sp & 1) * 5 * 8 * 16;
        if (G16) {
            uint16_t* gates16 = reinterpret_cast<uint16_t*>(gates);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pp = lane + 64 * i;
                if (pp < 32) {
                    const int row = pp >> 2, c4 = (pp & 3) * 4;
                    if (row < Bl)
                        *reinterpret_cast<f32x4_asm*>(hseq + ((size_t)tq * B + b0 + row) * hs + (size_t)d * H + j0 + c4) = lds_read16_raw(src + pp * 4);
                } else if (pp < 96) {
                    // blocked layout [T*B][ndir][H / 16][4 gates][16 units]: this workgroup's r | z | n | q of a row are ONE 128-B line
                    // (in the [4][H] layout they were four 32-B pieces of four lines, each completed by three other workgroups).
                    // The gate waves leave them in LDS as that line already (put_state: IEEE half, [8 rows][4 gates][16 units], row pitch 144 B): a
                    // piece is one conflict-free 16-B read -- the float32 [gate][row][unit] image cost two 4-way conflicted reads and
                    // four conversions per piece on this wave.
                    const int q = pp - 32, row = q >> 3;
                    if (row < Bl) {
                        // (rows of the staging image are 144 B apart: see put_state)
                        const f32x4_asm v = lds_read16_raw(reinterpret_cast<const char*>(src + 128) + row * 144 + (q & 7) * 16);
                        *reinterpret_cast<f32x4_asm*>(gates16 + ((((size_t)tq * B + b0 + row) * ndir + d) * (H >> 4) + (j0 >> 4)) * 64 + (q & 7) * 8) = v;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pp = lane + 64 * i, arr = pp >> 5, row = (pp & 31) >> 2, c4 = (pp & 3) * 4;
            if (pp < 160 && row < Bl) {
                const size_t rowi = (size_t)tq * B + b0 + row;
                float* dst = arr == 0 ? hseq + rowi * hs + (size_t)d * H + j0 + c4
                                      : gates + (rowi * ndir + d) * 4 * H + (size_t)(arr - 1) * H + j0 + c4;
                *reinterpret_cast<f32x4_asm*>(dst) = lds_read16_raw(src + pp * 4);
            }
        }
    };
    // RING: the storer also writes the bf16 sequence (8 rows x 16 units: lanes 0..15 write 16 B each)
    auto store_h16 = [&](int sp) {
        if (lane < 16 && (lane >> 1) < Bl) {
            const long long tq = tfirst + tstep * sp;
            const float* src = oring + (size_t)(sp & 1) * 5 * 8 * 16 + (lane >> 1) * 16 + (lane & 1) * 8;
            const f32x4_asm v0 = lds_read16_raw(src), v1 = lds_read16_raw(src + 4);
            uint4 pk;
            pk.x = pack_bf16x2(v0[0], v0[1]); pk.y = pack_bf16x2(v0[2], v0[3]);
            pk.z = pack_bf16x2(v1[0], v1[1]); pk.w = pack_bf16x2(v1[2], v1[3]);
            *reinterpret_cast<uint4*>(hseq16 + ((size_t)tq * B + b0 + (lane >> 1)) * hs + (size_t)d * H + j0 + (lane & 1) * 8) = pk;
        }
    };
    // gate thread (row bb_, unit uu_) -> the storer's staging slot of this step's parity: h float32 [8][16]; the saved gates float32
    // [4 gates][8][16] -- or, G16, IEEE half as the memory line [8 rows][4 gates][16 units] (the conversions sit behind the payload
    // store, inside the pause before the next poll)
    auto put_state = [&](int parity, int bb_, int uu_, float h, float r, float z, float n, float q) {
        float* od = oring + (size_t)parity * 5 * 8 * 16;
        od[bb_ * 16 + uu_] = h;
        if (G16) {
            // the memory line [row][4 gates][16 units] with a row pitch of 144 instead of 128 B: the 4 rows x 8 units a 32-lane group stores
            // per gate then hit 16 different banks (at 128 B all four rows met in one bank: 4-way on each of the four stores); the storer's
            // pieces stay 16 contiguous bytes
            _Float16* oh = reinterpret_cast<_Float16*>(od + 128) + bb_ * 72 + uu_;
            oh[0] = (_Float16)r; oh[16] = (_Float16)z; oh[32] = (_Float16)n; oh[48] = (_Float16)q;
        } else {
            float* og = od + 128 + bb_ * 16 + uu_;
            og[0] = r; og[128] = z; og[256] = n; og[384] = q;
        }
    };
    if (is_loader) {
        for (int s0 = 0; s0 < BIO_GD - 1; ++s0) issue(s0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    Frag bb[KSW][3];
    if (is_compute) {
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            const int ks = PAIRED ? (((i >> 1) * 4 + w) * 2 + (i & 1)) : i * 4 + w;      // adjacent slice pairs per load
            const int k = ks * 32 + 8 * (lane >> 4);
#pragma unroll
            for (int gg = 0; gg < 3; ++gg)
                bb[i][gg].u = ks < nks ? *reinterpret_cast<const uint4*>(whh + ((size_t)(d * 3 + gg) * H + j0 + (lane & 15)) * H + k)
                                       : make_uint4(0, 0, 0, 0);
        }
    }
    // gate phase on waves 2 and 3 (no I/O wave on their SIMDs): thread (row (tid - 128) / 16, unit tid % 16)
    const int b = b_ & 7, u = u_;
    const bool act = tid >= 128 && tid < 256 && b < Bl;
    constexpr int kPoller = 128;
    float bh[3] = {0.f, 0.f, 0.f};
    if (tid >= 128 && tid < 256) {
#pragma unroll
        for (int gg = 0; gg < 3; ++gg) bh[gg] = bhh[(d * 3 + gg) * H + j0 + u];
    }
    float hprev = 0.f;
    // producers (workgroups of 16 units) behind this wave's K slices
    unsigned long long my_producers = 0ull;
    if (PAIRED) {
#pragma unroll
        for (int i2 = 0; i2 < KSW / 2; ++i2) {
            const int p2 = i2 * 4 + w;                          // slices 2 p2, 2 p2 + 1 = units [64 p2, 64 p2 + 64)
            for (int q = 0; q < 4; ++q) if (4 * p2 + q < nwg) my_producers |= 1ull << (4 * p2 + q);
        }
    } else {
        my_producers = nwg >= 64 ? ~0ull : ((1ull << nwg) - 1ull);
    }
    if (tid == 0) {
        *s_abort = 0;
        s_abort[1] = 0;
        s_abort[2] = 0;
        s_abort[3] = 0;
        if (LOCAL) {
            const int v = decide_local(sync, rec, nwg, abort_word, forge & 1);
            if (v < 0) *s_abort = 1; else s_abort[1] = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ASR_RAW_BARRIER();
    const bool local = LOCAL && s_abort[1] != 0;
    // data polling (forge bit 8, XCD-local form with paired loads): hseq16 arrives filled with 0xffff; a compute wave
    // simply loads its K slices until no sentinel dword is left -- no flags, no drain of the payload stores before a
    // signal, and one barrier per step (partials double-buffered by step parity)
    const bool dp = local && PAIRED && (forge & 8) != 0;
    const int poll_delay = (forge >> 8) & 0xff;          // s_sleep(1) units between a wave's own store / flag and its first poll (host: fwd_poll_delay)
#ifdef ASR_STAMP
    unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define ASR_ST(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_last; st_last = n_; }
#else
#define ASR_ST(i)
#endif

    // hand-off loads of one step (PAIRED form): fragment i2 = K slices 2 (4 i2 + w), + 1 of row tq
    constexpr int NA = (KSW + 1) / 2;
    // byte offsets inside hseq16 (< 2^31, can_persist) are kept per lane and stepped by one (t) row: no 64-bit products
    // on the chain
    const unsigned row_bytes = (unsigned)((size_t)B * hs * 2);
    unsigned frag_off[NA];
    bool frag_on[NA];
#pragma unroll
    for (int i2 = 0; i2 < KSW / 2; ++i2) {
        const int r16 = lane & 15, row = r16 & 7;
        const int ks = ((i2 * 4 + w) * 2) + (r16 >> 3);
        frag_on[i2] = ks < nks && row < Bl;
        frag_off[i2] = (unsigned)((((size_t)b0 + row) * hs + (size_t)d * H + ks * 32 + 8 * (lane >> 4)) * 2);
    }
    // (no zeroing here: a lane whose fragment is off never loads, so the zeros its registers start with stay; writing them
    // again each step made the compiler wait vmcnt(0) -- for the h store just issued -- in front of these loads)
    auto fetch_row = [&](Frag (&f)[NA], int tq) {
#pragma unroll
        for (int i2 = 0; i2 < KSW / 2; ++i2) {
            if (frag_on[i2]) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(h16rsrc, frag_off[i2] + (unsigned)tq * row_bytes, 0, 16 /* sc1 */);
                f[i2].u = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    const unsigned store_off = (unsigned)((((size_t)b0 + b) * hs + (size_t)d * H + j0 + u) * 2);
    // RING: the polled payload lives in a 4-slot ring of this recurrence inside sync_ws ([slot][8 rows][H] bf16, behind the control
    // words, the place of the backward kernel's partial-sum ring): 32 KB of L2-resident lines per recurrence, written and read every
    // fourth step, instead of a fresh 8 KB of the (T, B, ndir H) sequence per step, which has to be filled with sentinels before the
    // launch and pre-touched.  A producer re-arms its piece of slot s - 2 behind its store of step s (every consumer has read slot
    // s - 2: it needed it to produce the h_{s-1} this workgroup has consumed).  The bf16 sequence itself (operand of the dW_hh
    // product) is then written by the storer, off the chain.
    const unsigned slot_bytes = (unsigned)(8 * H * 2);
    const __amdgpu_buffer_rsrc_t ringrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<char*>(sync) + kPsOffset + (size_t)rec * PS_RING * slot_bytes), 0, (int)(PS_RING * slot_bytes), 0x00020000);
    unsigned ring_frag_off[NA];
#pragma unroll
    for (int i2 = 0; i2 < KSW / 2; ++i2) {
        const int r16 = lane & 15, row = r16 & 7;
        const int ks = ((i2 * 4 + w) * 2) + (r16 >> 3);
        // producer-major slots: [workgroup (16 units)][8 rows][16 units], so that a producer's store instruction writes its 256 B as
        // two whole lines (in the (T, B, H) sequence a workgroup's piece is 32 B of every row's line)
        const int kk = ks * 32 + 8 * (lane >> 4);
        ring_frag_off[i2] = (unsigned)(((kk >> 4) * 128 + row * 16 + (kk & 15)) * 2);
    }
#ifdef ASR_EXP_REDUNDANT
    // what-if: every compute wave also fetches (and waits for) the other ASR_EXP_REDUNDANT waves' K slices -- the L2 -> CU volume of a form
    // in which each wave needs the whole state row
    Frag xtra[3][NA];
    unsigned xoff[3][NA];
#pragma unroll
    for (int ww = 0; ww < 3; ++ww)
#pragma unroll
        for (int i2 = 0; i2 < KSW / 2; ++i2) {
            const int r16 = lane & 15, row = r16 & 7, w2 = (w + ww + 1) & 3;
            const int ks = ((i2 * 4 + w2) * 2) + (r16 >> 3);
            const int kk = ks * 32 + 8 * (lane >> 4);
            xoff[ww][i2] = (unsigned)(((kk >> 4) * 128 + row * 16 + (kk & 15)) * 2);
            xtra[ww][i2].u = make_uint4(0, 0, 0, 0);
        }
#endif
    auto fetch_ring = [&](Frag (&f)[NA], int slot) {
#pragma unroll
        for (int i2 = 0; i2 < KSW / 2; ++i2) {
            if (frag_on[i2]) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ringrsrc, ring_frag_off[i2] + (unsigned)slot * slot_bytes, 0, 16 /* sc1 */);
                f[i2].u = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
#ifdef ASR_EXP_REDUNDANT
#pragma unroll
        for (int ww = 0; ww < ASR_EXP_REDUNDANT; ++ww)
#pragma unroll
            for (int i2 = 0; i2 < KSW / 2; ++i2) {
                if (frag_on[i2]) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ringrsrc, xoff[ww][i2] + (unsigned)slot * slot_bytes, 0, 16 /* sc1 */);
                    xtra[ww][i2].u = make_uint4(v[0], v[1], v[2], v[3]);
                }
            }
#endif
    };
    const unsigned ring_store_off = (unsigned)(((j0 >> 4) * 128 + b * 16 + u) * 2);
    Frag ahead[NA], acur[NA];   // ahead: (gate waves) the next step's first attempt, issued right behind their own h store
#pragma unroll
    for (int i2 = 0; i2 < NA; ++i2) { ahead[i2].u = make_uint4(0, 0, 0, 0); acur[i2].u = make_uint4(0, 0, 0, 0); }
    bool have_ahead = false;

#ifndef ASR_STAMP
    // ---------------------------------------------------------------------------------------------------------------
    // Role-specialised time loops for the default form (XCD-local, payload polled, paired loads).  The generic loop below
    // serves every hand-off form with one body, so each wave walks a maze of exec-mask and scalar branches per step
    // (rocprofv3: ~104 SALU + 87 VALU instructions per wave and step, most of them control) and the step had become an
    // instruction chain.  Here every role runs its own loop: loader / storer / pure compute waves / compute + gate waves,
    // ONE s_barrier per step (beta_s: "the partial products of step s are in LDS").  The arithmetic, the LDS layouts and the
    // hand-off are those of the generic loop: results are bit-identical.  The gi ring is re-indexed: after beta_s the
    // loader refills the slot step s has just read with step s + BIO_GD and waits until step s + 2 has landed, which the
    // gate waves read after beta_{s+1} -- a barrier now separates every landing check from the read it covers.
    if (dp) {
        constexpr int GD = BIO_GD;
        // -DASR_STAMP_DP: cycles per phase of the compute waves (read back by tools/stamp_gru_fwd_dp.py)
#ifdef ASR_STAMP_DP
        unsigned long long pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_last = __builtin_amdgcn_s_memtime();
#define ASR_PF(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_acc[i] += n_ - pf_last; pf_last = n_; }
#else
#define ASR_PF(i)
#endif
        if (is_loader) {
            issue(GD - 1);                                          // (the prologue issued steps 0 .. GD - 2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int s = 0; s < T; ++s) {
                ASR_PF(2)
                ASR_RAW_BARRIER();
                ASR_PF(3)
                if ((s & 15) == 0 && lds_peek(s_abort)) break;
                issue(s + GD);                                       // slot s % GD: read at the top of step s, before beta_s
                // in flight afterwards: the issues of steps s + 3 .. s + GD that exist (two LDS-DMA instructions each)
                const int left = T - 1 - (s + 2);
                if (left >= GD - 2) { if (GI16) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
                else if (left == 1) { if (GI16) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#ifdef ASR_STAMP_DP
            if (blockIdx.x < 16 && (blockIdx.x & 7) == 0 && lane == 0)
                for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(sync + 1024)[64 + (((blockIdx.x >> 3) * 2) + 0) * 8 + i] = pf_acc[i];
#endif
        } else if (is_storer) {
            // Pre-touch: the payload lines of step s + 2 (still all sentinel) are pulled into this XCD's L2 two steps before their
            // producers write them, so that partial-line stores and polls find the line resident instead of fetching the sentinel
            // line from memory (the launch's 65 MB sentinel fill does not stay in the L2s, and lines written from another XCD never
            // were in this one): 1.491 -> 1.469 us per step.  Two lines per workgroup (64 lines per row group and step); the load's
            // result is never used and never waited for inside the loop: inline asm, so that no waitcnt is scheduled around it.
            unsigned touched = 0;
            const int wg = j0 >> 4, tl = 2 * wg + (lane & 1), trow = tl >> 3;
            const char* tbase = reinterpret_cast<const char*>(hseq16) + (((size_t)b0 + (trow < Bl ? trow : 0)) * hs + (size_t)d * H) * 2 + (size_t)(tl & 7) * 128;
            const bool toucher = lane < 2 && trow < Bl && (size_t)(tl & 7) * 128 < (size_t)H * 2;
            for (int s = 0; s < T; ++s) {
                ASR_PF(2)
                ASR_RAW_BARRIER();
                ASR_PF(3)
                if ((s & 15) == 0 && lds_peek(s_abort)) break;
                if (!RING && kTouch && toucher && s + kTouchAhead < T) {
                    const int t2 = d == 0 ? s + kTouchAhead : T - 1 - kTouchAhead - s;
                    const char* tp_ = tbase + (size_t)t2 * row_bytes_;
                    asm volatile("global_load_dword %0, %1, off sc1" : "+v"(touched) : "v"(tp_) : "memory");
                }
                if (s > 0) { store_step(s - 1); if (RING) store_h16(s - 1); }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ASR_STAMP_DP
            if (blockIdx.x < 16 && (blockIdx.x & 7) == 0 && lane == 0)
                for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(sync + 1024)[64 + (((blockIdx.x >> 3) * 2) + 1) * 8 + i] = pf_acc[i];
#endif
            if (touched == 0x12345678u && lane == 63) lds_poke(s_abort + 3, 0);      // (keeps the register alive; never true for bf16 pairs of a GRU state)
        } else {
            const bool gate_wave = w >= 2;
            int slot = 0;
            for (int s = 0; s < T; ++s) {
                const int t = d == 0 ? s : T - 1 - s;
                float gr = 0.f, gz = 0.f, gn = 0.f;
                if (gate_wave) {                                    // this step's gi: in the ring, checked two barriers ago
                    read_gi(slot, gr, gz, gn);
                    slot = slot == GD - 1 ? 0 : slot + 1;
                }
                ASR_PF(0)
                if (s > 0) {
                    const int tp = d == 0 ? t - 1 : t + 1;
                    if (gate_wave) {
#pragma unroll
                        for (int i2 = 0; i2 < NA; ++i2) acur[i2].u = ahead[i2].u;
                    } else {
                        unsigned nap = 0;
                        // (timing this first poll from the step's barrier instead of the gate waves' flag measured worse: 1.35-1.38
                        // against 1.33 us per step)
                        while ((lds_peek(s_abort + 2) < s || lds_peek(s_abort + 3) < s) && !lds_peek(s_abort) && ++nap < kSpinLimit)
                            __builtin_amdgcn_s_sleep(1);
                        poll_pause(poll_delay);
                        if (RING) fetch_ring(acur, (s - 1) & (PS_RING - 1)); else fetch_row(acur, tp);
                    }
                    unsigned spins = 0;
                    for (;;) {
                        bool missing = false;
#pragma unroll
                        for (int i2 = 0; i2 < KSW / 2; ++i2)
                            missing |= acur[i2].u.x == 0xffffffffu || acur[i2].u.y == 0xffffffffu || acur[i2].u.z == 0xffffffffu || acur[i2].u.w == 0xffffffffu;
#ifdef ASR_EXP_REDUNDANT
#pragma unroll
                        for (int ww = 0; ww < ASR_EXP_REDUNDANT; ++ww)
#pragma unroll
                            for (int i2 = 0; i2 < KSW / 2; ++i2)
                                missing |= xtra[ww][i2].u.x == 0xffffffffu || xtra[ww][i2].u.y == 0xffffffffu || xtra[ww][i2].u.z == 0xffffffffu || xtra[ww][i2].u.w == 0xffffffffu;
#endif
                        if (__ballot(missing) == 0ull) break;
                        if ((++spins & 63u) == 0u) {
                            if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) { if (lane == 0) *s_abort = 1; break; }
                            if (spins > kSpinLimit) {
                                if (lane == 0) { __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT); *s_abort = 1; }
                                break;
                            }
                        }
                        if (RING) fetch_ring(acur, (s - 1) & (PS_RING - 1)); else fetch_row(acur, tp);
                    }
                    ASR_PF(1)
#ifdef ASR_STAMP_DP
                    pf_acc[7] += spins + 1;
#endif
                    f32x4 acc[3];
#pragma unroll
                    for (int gg = 0; gg < 3; ++gg) acc[gg] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i2 = 0; i2 < KSW / 2; ++i2) {
                        Frag a1;
                        a1.u = swap_half_rows(acur[i2].u);
#pragma unroll
                        for (int gg = 0; gg < 3; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[i2].v, bb[2 * i2][gg].v, acc[gg], 0, 0, 0);
#pragma unroll
                        for (int gg = 0; gg < 3; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v, bb[2 * i2 + 1][gg].v, acc[gg], 0, 0, 0);
                    }
                    // K-split partial products -> LDS, image [parity][gate][wave][32 live lanes] float4: a lane's four rows of one gate are
                    // ONE ds_write_b128 (consecutive lanes, consecutive 16 B: conflict-free), three stores per wave and step.  (The
                    // reader-major image [gate][row][unit][wave] took twelve ds_write_b32 per wave at a 4-way bank conflict -- the
                    // 0.42 conflict fraction of profiles/r03_pmc_sq.csv -- to save the gate threads nine reads.)
                    if (lane < 32) {
                        float4* pw = part + (s & 1) * 3 * 128 + w * 32 + lane;
#pragma unroll
                        for (int gg = 0; gg < 3; ++gg) pw[gg * 128] = make_float4(acc[gg][0], acc[gg][1], acc[gg][2], acc[gg][3]);
                    }
                }
                ASR_PF(2)
                ASR_RAW_BARRIER();
                ASR_PF(3)
                if ((s & 15) == 0 && lds_peek(s_abort)) break;
                if (gate_wave) {
                    float gh0 = bh[0], gh1 = bh[1], gh2 = bh[2];
                    if (s > 0 && act) {
                        // (row b, unit u) of wave ww: float (b & 3) of lane (b >> 2) * 16 + u -- a gate wave's 64 threads read 64
                        // consecutive dwords per (gate, wave): conflict-free ds_read_b32, twelve of them in flight together
                        const float* pf = reinterpret_cast<const float*>(part + (s & 1) * 3 * 128) + ((b >> 2) * 16 + u) * 4 + (b & 3);
                        const float a0 = pf[0], a1 = pf[128], a2 = pf[256], a3 = pf[384];
                        const float c0 = pf[512], c1 = pf[640], c2 = pf[768], c3 = pf[896];
                        const float e0 = pf[1024], e1 = pf[1152], e2 = pf[1280], e3 = pf[1408];
                        gh0 += (a0 + a1) + (a2 + a3);
                        gh1 += (c0 + c1) + (c2 + c3);
                        gh2 += (e0 + e1) + (e2 + e3);
                    }
                    const float r = sigmoidf_(gr + gh0);
                    const float z = sigmoidf_(gz + gh1);
                    const float n = tanhf_(gn + r * gh2);
                    const float h = (1.0f - z) * n + z * hprev;
                    hprev = h;
                    const unsigned mine = (unsigned)f32_to_bf16(h);
                    const unsigned other = lane_xor1_u32(mine);
                    ASR_PF(4)
                    __builtin_amdgcn_s_waitcnt(0x0F70);              // nothing of this wave is in flight (see the generic loop)
                    if (act && !(u & 1)) {
                        unsigned packed = mine | (other << 16);
                        if (packed == 0xffffffffu) packed = 0x7fc07fc0u;
                        if (RING) {
                            __builtin_amdgcn_raw_buffer_store_b32(packed, ringrsrc, ring_store_off + (unsigned)(s & (PS_RING - 1)) * slot_bytes, 0, 0);
                            if (s >= 2) __builtin_amdgcn_raw_buffer_store_b32(0xffffffffu, ringrsrc, ring_store_off + (unsigned)((s - 2) & (PS_RING - 1)) * slot_bytes, 0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b32(packed, h16rsrc, store_off + (unsigned)t * row_bytes, 0, 0);
                        }
                    }
                    if (s + 1 < T && lane == 0) lds_poke(s_abort + w, s + 1);
                    if (act) put_state(s & 1, b, u, h, r, z, n, gh2);
                    if (s + 1 < T) {
                        poll_pause(poll_delay);
                        if (RING) fetch_ring(ahead, s & (PS_RING - 1)); else fetch_row(ahead, t);
                    }
                    ASR_PF(5)
                }
            }
#ifdef ASR_STAMP_DP
            if (blockIdx.x < 16 && (blockIdx.x & 7) == 0 && lane == 0)
                for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long*>(sync + 1024)[(((blockIdx.x >> 3) * 4) + w) * 8 + i] = pf_acc[i];
#endif
        }
    } else
#endif
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        const int tp = d == 0 ? t - 1 : t + 1;
        float gh[3] = {bh[0], bh[1], bh[2]};
        // this step's gi (in the ring since at least three steps ago) is read before the wait, off the chain
        float gr = 0.f, gz = 0.f, gn = 0.f;
        if (tid >= 128 && tid < 256) read_gi(s % BIO_GD, gr, gz, gn);
        if (s > 0) {
            if (dp) {
            } else if (local) { // one line of per-producer flags; every compute wave polls it for ITS producers and goes on
                if (is_compute && !wait_flags_mask(shards, nwg, my_producers, (unsigned)s, abort_word, lane) && lane == 0) *s_abort = 1;
            } else {
                if (tid == kPoller && !wait_shards<false>(shards, nwg, (unsigned)s, abort_word))
                    *s_abort = 1;   // placement-free form: sharded agent-scope counters (flag stores to one line were slower there)
                ASR_RAW_BARRIER();
            }
            ASR_ST(1)
            if (is_compute) {
                f32x4 acc[3];
#pragma unroll
                for (int gg = 0; gg < 3; ++gg) acc[gg] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (PAIRED) {       // two K slices per load instruction (the 16-row MFMA tile has 8 idle rows: their lanes fetch the next slice)
                    Frag (&a)[NA] = acur;
                    unsigned spins = 0;
                    // the first attempt is issued outside the retry loop: at a loop header the compiler waits vmcnt(0) for
                    // the registers it is about to reload, which on the gate waves also drains their own h store (gfx9
                    // has one counter for loads and stores) -- 0.29 us per step in front of the loads
                    auto fetch = [&]() { fetch_row(a, tp); };
                    if (have_ahead) {
#pragma unroll
                        for (int i2 = 0; i2 < NA; ++i2) a[i2].u = ahead[i2].u;
                    } else {
                        // the pure compute waves would poll from the barrier on and waste one attempt per step (measured:
                        // 1.01 retries) while the producers are still in their gate phase: they wait on an LDS word that
                        // the gate waves of THIS workgroup raise when they store (all workgroups are in step), then ask
                        if (dp) {
                            unsigned nap = 0;
                            while ((lds_peek(s_abort + 2) < s || lds_peek(s_abort + 3) < s) && !lds_peek(s_abort) && ++nap < kSpinLimit)
                                __builtin_amdgcn_s_sleep(1);
                        }
                        fetch();
                    }
                    have_ahead = false;
                    while (dp) {
                        // (one attempt at a time: a second one in flight made both slower, 1.78 -> 1.99 us per step)
                        bool missing = false;
#pragma unroll
                        for (int i2 = 0; i2 < KSW / 2; ++i2)
                            missing |= a[i2].u.x == 0xffffffffu || a[i2].u.y == 0xffffffffu || a[i2].u.z == 0xffffffffu || a[i2].u.w == 0xffffffffu;
                        if (__ballot(missing) == 0ull) break;
                        if ((++spins & 63u) == 0u) {
                            if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) { if (lane == 0) *s_abort = 1; break; }
                            if (spins > kSpinLimit) {
                                if (lane == 0) { __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT); *s_abort = 1; }
                                break;
                            }
                        }
                        fetch();
                    }
#ifdef ASR_STAMP
                    st_acc[11] += spins + 1;      // attempts
#endif
                    ASR_ST(2)
#pragma unroll
                    for (int i2 = 0; i2 < KSW / 2; ++i2) {
                        Frag a1;
                        a1.u = swap_half_rows(a[i2].u);
#pragma unroll
                        for (int gg = 0; gg < 3; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i2].v, bb[2 * i2][gg].v, acc[gg], 0, 0, 0);
#pragma unroll
                        for (int gg = 0; gg < 3; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v, bb[2 * i2 + 1][gg].v, acc[gg], 0, 0, 0);
                    }
                } else {
                Frag a[KSW];
#pragma unroll
                for (int i = 0; i < KSW; ++i) {
                    const int ks = i * 4 + w;
                    const int k = ks * 32 + 8 * (lane >> 4);
                    const int row = lane & 15;
                    a[i].u = make_uint4(0, 0, 0, 0);
                    if (ks < nks && row < Bl) {       // lanes of the empty MFMA rows are masked off
                        const unsigned off = (unsigned)((((size_t)tp * B + b0 + row) * hs + (size_t)d * H + k) * 2);
                        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(h16rsrc, off, 0, 16 /* sc1: bypasses the L1, served by the L2 */);
                        a[i].u = make_uint4(v[0], v[1], v[2], v[3]);
                    }
                }
#ifdef ASR_STAMP
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                ASR_ST(2)
#pragma unroll
                for (int i = 0; i < KSW; ++i)
#pragma unroll
                    for (int gg = 0; gg < 3; ++gg) acc[gg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].v, bb[i][gg].v, acc[gg], 0, 0, 0);
                }
                if (lane < 32) {        // (image: see the role-specialised loop)
                    float4* pw = part + (dp ? (s & 1) * 3 * 128 : 0) + w * 32 + lane;
#pragma unroll
                    for (int gg = 0; gg < 3; ++gg) pw[gg * 128] = make_float4(acc[gg][0], acc[gg][1], acc[gg][2], acc[gg][3]);
                }
            }
            ASR_ST(3)
            ASR_RAW_BARRIER();
            ASR_ST(4)
            if ((s & 15) == 0 && *s_abort) break;      // (every 16 steps: the LDS read sat on the chain behind the barrier; polls give up at once anyway)
            if (act) {
                // one scalar LDS read per partial: selecting a component of a float4 by a runtime index compiles to a
                // nest of divergent branches around narrow reads (measured 0.95 us per step) -- here the index is part of the address
                const float* pf = reinterpret_cast<const float*>(part + (dp ? (s & 1) * 3 * 128 : 0)) + ((b >> 2) * 16 + u) * 4 + (b & 3);
#pragma unroll
                for (int gg = 0; gg < 3; ++gg) gh[gg] += (pf[gg * 512] + pf[gg * 512 + 128]) + (pf[gg * 512 + 256] + pf[gg * 512 + 384]);
            }
            ASR_ST(9)
        }
        if (is_loader) {
            // slot s + 1 (issued two steps ago) must have landed before this step's barrier; the two younger issues, two
            // LDS-DMA instructions each, stay in flight.  (vmcnt(0) here waited for the load issued ONE step ago, i.e. for a
            // fresh HBM round trip, and made the loader the last wave at the barrier: 0.25 us per step.)
            issue(s + BIO_GD - 1);
            if (s + BIO_GD - 1 < T) { if (GI16) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else if (s + BIO_GD - 2 < T) { if (GI16) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (is_storer) {
            if (s > 0) store_step(s - 1);
        } else if (tid >= 128) {
            const float r = sigmoidf_(gr + gh[0]);
            const float z = sigmoidf_(gz + gh[1]);
            const float n = tanhf_(gn + r * gh[2]);
            const float h = (1.0f - z) * n + z * hprev;
            hprev = h;
            ASR_ST(5)
            const unsigned mine = (unsigned)f32_to_bf16(h);
            const unsigned other = lane_xor1_u32(mine);
            // nothing of this wave is in flight here (its loads fed the MFMAs); saying so keeps the compiler from waiting
            // vmcnt(0) -- i.e. for the store below -- when it next writes a register that once was a load destination
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (act) {
                if (!(u & 1)) {
                    unsigned packed = mine | (other << 16);
                    if (packed == 0xffffffffu) packed = 0x7fc07fc0u;      // (two NaNs of a diverged run) never the sentinel
                    const unsigned ob = store_off + (unsigned)t * row_bytes;
                    if (local) __builtin_amdgcn_raw_buffer_store_b32(packed, h16rsrc, ob, 0, 0);
                    else __hip_atomic_store(reinterpret_cast<unsigned*>(reinterpret_cast<char*>(hseq16) + ob), packed, ASR_RLX_AGENT);     // sc1 payload
                }
            }
            if (PAIRED && dp && s + 1 < T) {     // row t is what step s + 1 reads: ask for it before the LDS bookkeeping below
                if (lane == 0) lds_poke(s_abort + w, s + 1);       // (waves 2 and 3 -> words 2 and 3)
                fetch_row(ahead, t);
                have_ahead = true;
            }
            if (act) put_state(s & 1, b, u, h, r, z, n, gh[2]);
            ASR_ST(6)
            if (!dp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ASR_ST(7)
        }
        if (dp) {
            if (s == 0) ASR_RAW_BARRIER();      // step 0 has no barrier of its own (no partial products)
            continue;
        }
        ASR_RAW_BARRIER();
        ASR_ST(8)
        if (tid == kPoller) {
            if (local) set_flag(shards + (j0 >> 4), (unsigned)s + 1u, true);
            else __hip_atomic_fetch_add(my_shard, 1u, ASR_RLX_AGENT);
        }
    }
#ifdef ASR_STAMP
    if (blockIdx.x < 8 && lane == 0)
        for (int i = 0; i < 12; ++i) reinterpret_cast<unsigned long long*>(sync + 1024)[((blockIdx.x * 6) + w) * 12 + i] = st_acc[i];
#endif
    if (is_loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup
    ASR_RAW_BARRIER();
    if (is_storer && !*s_abort) store_step(T - 1);
}

// Forward wide form (see bwd_wide_kernel): 32 units x 4-row recurrences, 8 compute waves (K = H split in 8, all 3 gates x 2
// unit tiles per wave), loader, storer.
template <int KS8, bool LOCAL>
__global__ __launch_bounds__(640, 3) void fwd_wide_kernel(const float* __restrict__ gi, const uint16_t* __restrict__ whh,
                                                          const float* __restrict__ bhh, float* __restrict__ hseq,
                                                          uint16_t* hseq16, float* __restrict__ gates, unsigned* sync, int T,
                                                          int B, int H, int ndir, int forge) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* part = reinterpret_cast<float4*>(smem);                                   // [2 (step parity)][8 waves][6 tiles: gate x unit half][16 live lanes]
    float* opring = reinterpret_cast<float*>(part + 2 * 8 * 6 * 16);                      // [BIO_GD][3 gates][4 rows][32 units]
    float* oring = opring + BIO_GD * 3 * 128;                                         // [2][5: h r z n q][4 rows][32 units]
    int* s_abort = reinterpret_cast<int*>(oring + 2 * 5 * 128);
    constexpr int rows = 4;
    const int nwg = H / 32;
    const int Gn = (B + rows - 1) / rows, nrec = Gn * ndir, nrec_pad = (nrec + 7) & ~7;
    const int rec = LOCAL ? (int)(blockIdx.x % nrec_pad) : (int)(blockIdx.z * gridDim.y + blockIdx.y);
    const int slot = LOCAL ? (int)(blockIdx.x / nrec_pad) : (int)blockIdx.x;
    if (LOCAL && rec >= nrec) return;
    const int d = rec / Gn, g = rec % Gn;
    const int j0 = slot * 32;
    const int b0 = g * rows, Bl = min(rows, B - b0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool is_compute = w < 8, is_loader = w == 8, is_storer = w == 9;
    const int nks = H >> 5;
    const size_t hs = (size_t)ndir * H;
    unsigned* shards = shard_base(sync, rec);
    unsigned* my_shard = shards + (slot % NSH) * 32;
    unsigned* abort_word = sync + 1023;
    const __amdgpu_buffer_rsrc_t h16rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)hseq16, 0, (int)((size_t)T * B * hs * 2), 0x00020000);
    const long long tstep = d == 0 ? 1 : -1;
    const int tfirst = d == 0 ? 0 : T - 1;

    // loader (wave 8): lanes 0..31 fetch (row lane / 8, units 4 (lane % 8) ..) of the three gi gate arrays
    const int lrow = lane >> 3;
    const float* lgp = gi + ((size_t)tfirst * B + b0 + (lrow < Bl ? lrow : 0)) * (3 * hs) + (size_t)d * 3 * H + j0 + (lane & 7) * 4;
    const long long lstride = tstep * (long long)B * 3 * (long long)hs;
    auto issue = [&](int sq) {
        if (sq < T && lane < 32 && lrow < Bl) {
            char* sl = reinterpret_cast<char*>(opring) + (sq % BIO_GD) * (3 * 512);
#pragma unroll
            for (int gg = 0; gg < 3; ++gg)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(lgp + (size_t)gg * H), (lds_ptr_t)(sl + gg * 512), 16, 0, 0);
        }
        lgp += lstride;
    };
    // storer (wave 9): 5 arrays x 4 rows x 128 B = 160 pieces: piece p = lane + 64 i: array p / 32, row (p % 32) / 8
    auto store_step = [&](int sp) {
        const long long tq = tfirst + tstep * sp;
        const float* src = oring + (size_t)(sp & 1) * 5 * 128;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int pp = lane + 64 * i, arr = pp >> 5, row = (pp & 31) >> 3, c4 = (pp & 7) * 4;
            if (pp < 160 && row < Bl) {
                const size_t rowi = (size_t)tq * B + b0 + row;
                float* dst = arr == 0 ? hseq + rowi * hs + (size_t)d * H + j0 + c4
                                      : gates + (rowi * ndir + d) * 4 * H + (size_t)(arr - 1) * H + j0 + c4;
                *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src + pp * 4);
            }
        }
    };
    if (is_loader) {
        for (int s0 = 0; s0 < BIO_GD - 1; ++s0) issue(s0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // weights: wave w owns K slices [w KS8, (w + 1) KS8) of all six 16-column tiles (gate gg, unit half n: tile 2 gg + n)
    Frag bb[KS8][6];
    if (is_compute) {
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int i = 0; i < KS8; ++i) {
            const int ks = w * KS8 + i;
            const int k = ks * 32 + 8 * (lane >> 4);
#pragma unroll
            for (int tl = 0; tl < 6; ++tl)
                bb[i][tl].u = ks < nks ? *reinterpret_cast<const uint4*>(whh + ((size_t)(d * 3 + (tl >> 1)) * H + j0 + (tl & 1) * 16 + (lane & 15)) * H + k)
                                       : make_uint4(0, 0, 0, 0);
        }
    }
    const int b = ((tid - 128) >> 5) & 3, u = tid & 31;
    const bool gate_wave = tid >= 128 && tid < 256;
    const bool act = gate_wave && b < Bl;
    constexpr int kPoller = 128;
    float bh[3] = {0.f, 0.f, 0.f};
    if (gate_wave) {
#pragma unroll
        for (int gg = 0; gg < 3; ++gg) bh[gg] = bhh[(d * 3 + gg) * H + j0 + u];
    }
    float hprev = 0.f;
    unsigned long long my_producers = 0ull;       // workgroups (32 units each) behind this wave's K slices
#pragma unroll
    for (int i = 0; i < KS8; ++i) {
        const int ks = w * KS8 + i;
        if (ks < nks) my_producers |= 1ull << ks;
    }
    if (tid == 0) {
        *s_abort = 0;
        s_abort[1] = 0;
        if (LOCAL) {
            const int v = decide_local(sync, rec, nwg, abort_word, (forge & 1) ? 2 + (slot & 1) : 0);
            if (v < 0) *s_abort = 1; else s_abort[1] = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    ASR_RAW_BARRIER();
    const bool local = LOCAL && s_abort[1] != 0;
    const bool dp = local && (forge & 8) != 0;        // payload polled as its own signal (see fwd_persistent_io_kernel)

    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        const int tp = d == 0 ? t - 1 : t + 1;
        float gh[3] = {bh[0], bh[1], bh[2]};
        float gr = 0.f, gz = 0.f, gn = 0.f;
        if (gate_wave) {
            const float* osrc = opring + (size_t)(s % BIO_GD) * 3 * 128 + b * 32 + u;
            gr = osrc[0]; gz = osrc[128]; gn = osrc[256];
        }
        if (s > 0) {
            if (dp) {
            } else if (local) { // every compute wave waits for the producers of ITS K slices only, no workgroup barrier
                if (is_compute && !wait_flags_mask(shards, nwg, my_producers, (unsigned)s, abort_word, lane) && lane == 0) *s_abort = 1;
            } else {
                if (tid == kPoller && !wait_shards<false>(shards, nwg, (unsigned)s, abort_word)) *s_abort = 1;
                ASR_RAW_BARRIER();
            }
            if (is_compute) {
                f32x4 acc[6];
#pragma unroll
                for (int tl = 0; tl < 6; ++tl) acc[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
                constexpr int NL = (KS8 + 3) / 4;
                Frag a[NL];
                const int r16 = lane & 15, row = r16 & 3, sl4 = r16 >> 2;
                unsigned spins = 0;
                auto fetch = [&]() {        // (first attempt outside the retry loop: see fwd_persistent_io_kernel)
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        const int i = 4 * l + sl4, ks = w * KS8 + i;
                        a[l].u = make_uint4(0, 0, 0, 0);
                        if (i < KS8 && ks < nks && row < Bl) {
                            const unsigned off = (unsigned)((((size_t)tp * B + b0 + row) * hs + (size_t)d * H + ks * 32 + 8 * (lane >> 4)) * 2);
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(h16rsrc, off, 0, 16 /* sc1 */);
                            a[l].u = make_uint4(v[0], v[1], v[2], v[3]);
                        }
                    }
                };
                fetch();
                while (dp) {
                    bool missing = false;
#pragma unroll
                    for (int l = 0; l < NL; ++l)
                        missing |= a[l].u.x == 0xffffffffu || a[l].u.y == 0xffffffffu || a[l].u.z == 0xffffffffu || a[l].u.w == 0xffffffffu;
                    if (__ballot(missing) == 0ull) break;
                    if ((++spins & 63u) == 0u) {
                        if (__hip_atomic_load(abort_word, ASR_RLX_AGENT) != 0u) { if (lane == 0) *s_abort = 1; break; }
                        if (spins > kSpinLimit) {
                            if (lane == 0) { __hip_atomic_store(abort_word, 1u, ASR_RLX_AGENT); *s_abort = 1; }
                            break;
                        }
                    }
                    fetch();
                }
#pragma unroll
                for (int i = 0; i < KS8; ++i) {
                    Frag f;
                    f.u = shl_rows(a[i >> 2].u, i & 3);
#pragma unroll
                    for (int tl = 0; tl < 6; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.v, bb[i][tl].v, acc[tl], 0, 0, 0);
                }
#pragma unroll
                for (int tl = 0; tl < 6; ++tl)
                    if (lane < 16) part[(((dp ? (s & 1) * 8 : 0) + w) * 6 + tl) * 16 + lane] = make_float4(acc[tl][0], acc[tl][1], acc[tl][2], acc[tl][3]);     // live rows 0..3 only
            }
            ASR_RAW_BARRIER();
            if ((s & 15) == 0 && *s_abort) break;      // (every 16 steps: the LDS read sat on the chain behind the barrier; polls give up at once anyway)
            if (act) {
                const float* pf = reinterpret_cast<const float*>(part + (dp ? (s & 1) * 8 * 6 * 16 : 0)) + ((u >> 4) * 16 + (u & 15)) * 4 + b;
#pragma unroll
                for (int gg = 0; gg < 3; ++gg)
#pragma unroll
                    for (int ww = 0; ww < 8; ++ww) gh[gg] += pf[(ww * 6 + gg * 2) * 64];
            }
        }
        if (is_loader) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            issue(s + BIO_GD - 1);
        } else if (is_storer) {
            if (s > 0) store_step(s - 1);
        } else if (gate_wave) {
            const float r = sigmoidf_(gr + gh[0]);
            const float z = sigmoidf_(gz + gh[1]);
            const float n = tanhf_(gn + r * gh[2]);
            const float h = (1.0f - z) * n + z * hprev;
            hprev = h;
            const unsigned mine = (unsigned)f32_to_bf16(h);
            const unsigned other = lane_xor1_u32(mine);
            if (act) {
                if (!(u & 1)) {
                    unsigned packed = mine | (other << 16);
                    if (packed == 0xffffffffu) packed = 0x7fc07fc0u;      // never the sentinel
                    const size_t o = ((size_t)t * B + b0 + b) * hs + (size_t)d * H + j0 + u;
                    if (local) __builtin_amdgcn_raw_buffer_store_b32(packed, h16rsrc, (unsigned)(o * 2), 0, 0);
                    else __hip_atomic_store(reinterpret_cast<unsigned*>(hseq16 + o), packed, ASR_RLX_AGENT);
                }
                float* od = oring + (size_t)(s & 1) * 5 * 128 + b * 32 + u;
                od[0] = h; od[128] = r; od[256] = z; od[384] = n; od[512] = gh[2];
            }
            if (!dp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (dp) {
            if (s == 0) ASR_RAW_BARRIER();
            continue;
        }
        ASR_RAW_BARRIER();
        if (tid == kPoller) {
            if (local) set_flag(shards + slot, (unsigned)s + 1u, true);
            else __hip_atomic_fetch_add(my_shard, 1u, ASR_RLX_AGENT);
        }
    }
    if (is_loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASR_RAW_BARRIER();
    if (is_storer && !*s_abort) store_step(T - 1);
}

// y = hf + hb (or a copy for one direction): f32 state -> bf16 layer output; rows beyond an utterance's length are zero
// gradient of the initial state (asr_gru_bwd_state): dhx = dh_{first} z_{first} (what the sweep left in `carry`) + dgh_{first} W_hh
__global__ __launch_bounds__(256) void dhx_kernel(const uint16_t* __restrict__ dgh, const uint16_t* __restrict__ whhT, const float* __restrict__ carry,
                                                  float* __restrict__ dhx, int T, int B, int H, int ndir) {
    const long long idx = blockIdx.x * 256ll + threadIdx.x, total = (long long)ndir * B * H;
    if (idx >= total) return;
    const int d = (int)(idx / ((long long)B * H)), b = (int)((idx / H) % B), j = (int)(idx % H);
    const long long t0 = d == 0 ? 0 : T - 1;
    const uint4* g = reinterpret_cast<const uint4*>(dgh + ((size_t)t0 * B + b) * ((size_t)ndir * 3 * H) + (size_t)d * 3 * H);
    const uint4* w = reinterpret_cast<const uint4*>(whhT + ((size_t)d * H + j) * (3 * (size_t)H));
    float acc = carry[idx];
    for (int k8 = 0; k8 < (3 * H) / 8; ++k8) {
        const uint4 gv = g[k8], wv = w[k8];
        const unsigned gs[4] = {gv.x, gv.y, gv.z, gv.w}, ws[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            acc += bf16_to_f32((uint16_t)(gs[e] & 0xffff)) * bf16_to_f32((uint16_t)(ws[e] & 0xffff)) +
                   bf16_to_f32((uint16_t)(gs[e] >> 16)) * bf16_to_f32((uint16_t)(ws[e] >> 16));
    }
    dhx[idx] = acc;
}

__global__ void merge_dirs_kernel(const float* __restrict__ hseq, uint16_t* __restrict__ y, long long rows, int H,
                                  int ndir, const int* __restrict__ x_len, int B) {
    const long long n = rows * H;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / H;
        const int j = (int)(i - r * H);
        float v = hseq[r * ndir * H + j];
        if (ndir == 2) v += hseq[r * ndir * H + H + j];
        if (x_len && (int)(r / B) >= x_len[(int)(r % B)]) v = 0.f;
        y[i] = f32_to_bf16(v);
    }
}
// the same, eight units per thread (H % 8 == 0, aligned): 2 x 2 16-B loads and one 16-B store instead of eight 4-B / 2-B ones
__global__ void merge_dirs_vec_kernel(const float* __restrict__ hseq, uint16_t* __restrict__ y, long long rows, int H,
                                      int ndir, const int* __restrict__ x_len, int B) {
    const int h8 = H >> 3;
    const long long n = rows * h8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / h8;
        const int j = (int)(i - r * h8) * 8;
        const float4* p = reinterpret_cast<const float4*>(hseq + r * ndir * H + j);
        float4 a = p[0], b = p[1];
        if (ndir == 2) {
            const float4* q = reinterpret_cast<const float4*>(hseq + r * ndir * H + H + j);
            const float4 c = q[0], d = q[1];
            a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
            b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
        }
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16);
        o.y = (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16);
        o.z = (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16);
        o.w = (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16);
        if (x_len && (int)(r / B) >= x_len[(int)(r % B)]) o = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(y + r * H + j) = o;
    }
}

// ---- per-utterance lengths (chainer.links.NStepBiGRU runs every sequence over its own length: asr/nn/nn.py:3)
// Row b is live for t < x_len[b].  Beyond that the state must stay frozen -- so that the reverse direction, which meets the
// padding first, arrives at t = x_len[b] - 1 with the zero state it would start from -- and no gradient may pass.  Instead of a
// length test on the latency chain of every recurrence kernel, the update gate is pinned to exactly 1 on those rows:
// z = sigmoid(gi_z + gh_z) with gi_z = +10^4 is 1.0f (exp underflows to 0, rcp(1) = 1), hence h' = (1 - z) n + z h = h bit for
// bit, and in the backward pass every gate gradient carries a factor (1 - z) = 0 while dh passes through unchanged (dh z).
// One workgroup per (t, b) row; live rows leave at once.
template <typename GT>
__global__ __launch_bounds__(256) void pin_update_gate_kernel(GT* __restrict__ gi, const int* __restrict__ x_len, int B, int H, int ndir) {
    const long long row = blockIdx.x;
    const int t = (int)(row / B), b = (int)(row % B);
    if (t < x_len[b]) return;
    GT* g = gi + row * (long long)ndir * 3 * H;
    for (int i = threadIdx.x; i < ndir * H; i += blockDim.x) {
        const int d = i / H, j = i - d * H;
        if (sizeof(GT) == 2) g[(size_t)d * 3 * H + H + j] = (GT)f32_to_bf16(1.0e4f);
        else g[(size_t)d * 3 * H + H + j] = (GT)1.0e4f;
    }
}

// the layer output is zero (a constant) beyond an utterance's length: whatever gradient arrives there is dropped
__global__ void mask_rows_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, const int* __restrict__ x_len,
                                 long long rows, int B, int W8) {
    const long long n = rows * W8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / W8;
        const bool live = (int)(r / B) < x_len[(int)(r % B)];
        reinterpret_cast<uint4*>(dst)[i] = live ? reinterpret_cast<const uint4*>(src)[i] : make_uint4(0, 0, 0, 0);
    }
}

}  // namespace gru
}  // namespace asr

using namespace asr;
using namespace asr::gru;

extern "C" size_t asr_gru_sync_bytes(int B, int H, int ndir) {
    const int nrec_pad = (ndir * ((B + 3) / 4) + 7) & ~7;
    const size_t ps = ps_exchange_bytes(nrec_pad, H);                  // the partial-sum ring (bwd_ps_kernel)
    const size_t ring = (size_t)16 * PS_RING * 8 * H * 2;              // the forward kernel's hand-off ring (16 recurrences: its four-row what-if)
    return 4096 + kShardBytes + (ps > ring ? ps : ring);               // control words, sharded counters, the larger exchange area
}

static bool valid_mode(int mode) { return mode == 0 || mode == 1 || mode == 2 || mode == 4 || (mode >= 7 && mode <= 10); }
static int check_dims(int T, int B, int H, int ndir) {
    if (T <= 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return ASR_ERR_BAD_ARG;
    if (H % 32) return ASR_ERR_UNSUPPORTED;       // MFMA K step and 16-unit workgroup slices
    return ASR_OK;
}

// control words [0, 4092) and the area behind them are zeroed by every call; the abort word (int 1023) is NOT: it is
// sticky, so that a caller reusing one sync_ws sees a failed launch later (and every later launch gives up at once)
// (one kernel: two hipMemsetAsync calls became three 5.6 us fill launches in front of every recurrence)
__global__ void clear_sync_kernel(uint4* __restrict__ ws, unsigned n16) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    if (i == 255) {         // bytes [4080, 4096): keep int 1023
        unsigned* w = (unsigned*)(ws + 255);
        w[0] = 0; w[1] = 0; w[2] = 0;
        return;
    }
    ws[i] = make_uint4(0, 0, 0, 0);
}

static bool clear_sync(void* sync_ws, size_t extra, hipStream_t st) {
    const unsigned n16 = (unsigned)((4096 + extra + 15) / 16);
    hipLaunchKernelGGL(clear_sync_kernel, dim3((n16 + 255) / 256), dim3(256), 0, st, (uint4*)sync_ws, n16);
    return hipGetLastError() == hipSuccess;
}

// the same launch also writes the 0xffff sentinels of a payload-polled hand-off area (fill, 16-B aligned, a multiple of 16 bytes):
// the first blocks clear the control words, the others stride over the fill -- one launch in front of a recurrence instead of a
// kernel and a memset (5-12 us each, eight times per train step)
__global__ void clear_sync_fill_kernel(uint4* __restrict__ ws, unsigned n16, unsigned clear_blocks, uint4* __restrict__ fill, size_t f16) {
    if (blockIdx.x < clear_blocks) {
        const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= n16) return;
        if (i == 255) {
            unsigned* w = (unsigned*)(ws + 255);
            w[0] = 0; w[1] = 0; w[2] = 0;
            return;
        }
        ws[i] = make_uint4(0, 0, 0, 0);
        return;
    }
    const size_t stride = (size_t)(gridDim.x - clear_blocks) * blockDim.x;
    for (size_t i = (size_t)(blockIdx.x - clear_blocks) * blockDim.x + threadIdx.x; i < f16; i += stride)
        fill[i] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
}

static bool clear_sync_fill(void* sync_ws, size_t extra, void* fill, size_t fill_bytes, hipStream_t st) {
    if (!fill || fill_bytes == 0) return clear_sync(sync_ws, extra, st);
    if ((((uintptr_t)fill) & 15) || (fill_bytes & 15)) {
        return clear_sync(sync_ws, extra, st) && hipMemsetAsync(fill, 0xff, fill_bytes, st) == hipSuccess;
    }
    const unsigned n16 = (unsigned)((4096 + extra + 15) / 16), cb = (n16 + 255) / 256;
    const size_t f16 = fill_bytes / 16;
    size_t fb = (f16 + 256 * 4 - 1) / (256 * 4);         // ~4 stores per thread
    if (fb > 2048) fb = 2048;
    if (fb < 1) fb = 1;
    hipLaunchKernelGGL(clear_sync_fill_kernel, dim3(cb + (unsigned)fb), dim3(256), 0, st, (uint4*)sync_ws, n16, cb, (uint4*)fill, f16);
    return hipGetLastError() == hipSuccess;
}

// compute units of the current device (cached per device ordinal; 0 if the query fails)
static int device_cus() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = -1;
        cached[dev] = n > 0 ? n : -1;
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}

static bool can_persist(int T, int B, int H, int ndir, int mode, const void* sync_ws) {
    if (mode == 1 || !sync_ws) return false;
    if (B > 16 * MT || (H / 16) * ndir > 128 || H > 1024) return false;
    if ((size_t)T * B * ndir * 3 * H * 2 >= ((size_t)1 << 31)) return false;
    // every workgroup of a persistent launch asks for most of a CU's LDS and must be resident together with all the others:
    // the largest grid any of the forms below builds for these sizes has to fit the device's CUs (a partitioned device --
    // CPX mode, a CU mask -- reports fewer), otherwise the per-step launches serve.  Which XCD a workgroup lands on is NOT
    // assumed: the XCD-local hand-off is chosen inside the launch and falls back to the placement-free one (decide_local).
    const int rec8 = (ndir * ((B + 7) / 8) + 7) & ~7, rec4 = (ndir * ((B + 3) / 4) + 7) & ~7;
    const int narrow = rec8 * (H / 16), wide = rec4 * ((H + 31) / 32);
    const int need = narrow > wide ? narrow : wide;
    const int cus = device_cus();
    if (cus > 0 && need > cus) return false;
    return true;
}

// first-poll delay of the forward hand-off in s_sleep(1) units (~70 cycles each), see kFirstPollDelay: measured per H (the number of
// producers of a recurrence, H / 16, shifts the moment the last store lands); ASR_DEBUG gru_poll_delay overrides
static int fwd_poll_delay(int H, bool ring) {
    static int env = -2;
    if (env == -2) env = debug_flag("gru_poll_delay", -1);
    if (env >= 0) return env > 255 ? 255 : env;
    // T=1000, B=32, us per step at 0 / best.  Ring form (whole-line stores into L2-resident slots): H=512 1.30 / 1.23 (5-6),
    // H=384 1.20 / 1.18 (3), H=256 1.03 / 1.00 (3), H=128 1.20 flat.  Sequence form (32-B pieces of fresh lines): H=512 1.49 / 1.34 (11),
    // H=384 1.41 / 1.39 (2-4), H=256 1.156 / 1.137 (2-4), H=128 1.24 / 1.21 (6-12)
    // (round 4, after the partial-product image changed, H = 512: 0 1.225, 3 1.203, 4 1.188, 5 1.185, 6 1.189, 7 1.203, 8 1.225, 9 1.255 us per
    // step -- flat between 4 and 6: 6 stays)
    if (ring) return H >= 512 ? 6 : 3;
    return H >= 512 ? kFirstPollDelay : (H >= 256 ? 3 : 6);
}

// which forward kernel family serves a call: 0 one launch per time step, 1 wide (32 units x 4-row recurrences), 2 the 16-unit x 8-row
// kernel (the default at 16 < B <= 32, and of every slab of a larger batch)
static int fwd_family(int T, int B, int H, int ndir, int mode, const void* sync_ws) {
    if (!can_persist(T, B, H, ndir, mode, sync_ws)) return 0;
    // (until round 4 the wide form also served 4 < B <= 16 in the default modes -- "a 4-row recurrence fetches half the hand-off bytes" --
    // but the 8-row kernel has since gained the ring hand-off, bf16 input projections and half gates: T=1000, H=512, us per step,
    // wide / 8-row: B=16 1.64 / 1.18, B=12 1.58 / 1.17, B=8 1.46 / 1.16, B=5 1.42 / 1.16.  The wide form is the placement-free mode 2's.)
    if (mode == 2 && H >= 128 && ndir * ((B + 3) / 4) <= 16 && (H / 32 + 7) / 8 <= 4) return 1;
    return 2;
}

// the default pair -- forward with the L2-resident ring hand-off, backward with the partial-sum exchange -- can keep the saved gates
// in IEEE half (kernel comments: G16); every other kernel form reads / writes float32 gates
static bool fwd_ring_form(int T, int B, int H, int ndir, int mode, const void* sync_ws) {
    if (fwd_family(T, B, H, ndir, mode, sync_ws) != 2) return false;
    static int ring_env = -1;
    if (ring_env < 0) ring_env = debug_flag("fwd_ring", 1);
    const int Gio = (B + 7) / 8, ksw = (H / 32 + 3) / 4;
    return (mode == 0 || mode == 8) && ndir * Gio <= 8 && ksw >= 2 && ring_env != 0;
}
static bool bwd_ps_form(int T, int B, int H, int ndir, int mode, const void* sync_ws) {
    const int ksw = (3 * H / 32 + 3) / 4;
    return can_persist(T, B, H, ndir, mode, sync_ws) && ksw <= 12 && (mode == 0 || mode == 8 || mode == 9 || mode == 10) && H % 128 == 0 &&
           H <= 1024 && ndir * ((B + 3) / 4) <= 16;
}
// Any batch size on the fast path.  The persistent kernels are resident on every CU at 32 utterances (8 recurrences x 32 workgroups
// forward, 16 x 16 backward), and a recurrence's step time does not depend on how many of its rows are live.  A larger batch --
// the reference trains with 128 per bucket, shrinking by 16 (run/ctc/cnn/train.py:72-73,165,208-209) -- runs as consecutive slabs of
// <= 32 rows through the default kernel pair: utterances are independent, so the results are those of one launch; B = 64 costs two
// recurrences' time instead of the 8 x of the per-step launches.  0 = no slabs (the batch fits, another kernel form was asked
// for, or the 32-bit offsets inside the hand-off buffers would not cover the whole batch).
static int slab_rows(int T, int B, int H, int ndir, int mode, const void* sync_ws) {
    constexpr int kSlab = 16 * MT;
    if (B <= kSlab || !(mode == 0 || mode == 8)) return 0;
    if ((size_t)T * B * ndir * 3 * H * 2 >= ((size_t)1 << 31)) return 0;
    // the 16-unit x 8-row forward kernel (its ring form, or -- K split too shallow for it, H < 256 -- its flag form, which fills no
    // sentinels into the whole sequence) and the partial-sum backward kernel: both take (boff, Bn)
    if (fwd_family(T, kSlab, H, ndir, mode, sync_ws) != 2 || !bwd_ps_form(T, kSlab, H, ndir, mode, sync_ws)) return 0;
    if (!fwd_ring_form(T, kSlab, H, ndir, mode, sync_ws) && (H / 32 + 3) / 4 >= 2) return 0;       // (ASR_DEBUG fwd_ring=0: sequence polling)
    return kSlab;
}

extern "C" int asr_gru_gates_f16_ok(int T, int B, int H, int ndir, int mode) {
    if (check_dims(T, B, H, ndir) != ASR_OK) return 0;
    static int dummy;
    const int fmode = (mode == 9 || mode == 10) ? 0 : mode;
    if (const int slab = slab_rows(T, B, H, ndir, mode, &dummy)) return fwd_ring_form(T, slab, H, ndir, fmode, &dummy) ? 1 : 0;
    return fwd_ring_form(T, B, H, ndir, fmode, &dummy) && bwd_ps_form(T, B, H, ndir, mode, &dummy) ? 1 : 0;
}

extern "C" int asr_gru_fwd_accepts_bf16_gi(int T, int B, int H, int ndir, int mode) {
    if (check_dims(T, B, H, ndir) != ASR_OK) return 0;
    if (mode == 9 || mode == 10) mode = 0;
    static int dummy;
    if (slab_rows(T, B, H, ndir, mode, &dummy)) return 1;
    return fwd_family(T, B, H, ndir, mode, &dummy) == 2 ? 1 : 0;
}

// The 16-unit x 8-row forward kernel (family 2 of fwd_family: the default) on the batch rows [boff, boff + Bn) of a (T, B, ..) problem.
static int fwd_io_launch(hipStream_t st, void* gi_any, int gi_bf16, const void* whh_bf16, const float* bhh, float* hseq, void* hseq_bf16,
                     float* gates, void* sync_ws, int T, int B, int H, int ndir, int mode, int gates_f16, int boff, int Bn) {
    const int ksw = (H / 32 + 3) / 4;
    // ASR_DEBUG gru_fwd_rows=4 (what-if of round 5, VERDICT r4 next 2: "hide the hand-off of one half slab behind the compute of the
    // other"): recurrences of FOUR rows, sixteen of them, 2 x 256 workgroups that ask for 72 KB of LDS so that exactly two share a CU --
    // the hardware then interleaves two independent half slabs per CU, each with its own waves (an upper bound for what one workgroup
    // alternating between two half slabs could reach: there the halves would also queue for the same four compute waves).  Needs the
    // full slab (all sixteen recurrences live: a grid whose workgroups partly leave at once starves the rest of dispatch slots), the
    // ring hand-off, half gates and bf16 input projections (the default kernel instance).  Measured (tools/time_gru_fwd_rows.py,
    // profiles/r05_gru_fwd_half_slabs_whatif.txt): 1.33 us per time step against 1.27 for the 8-row form, outputs bit-identical.
    static const int rows_env = debug_flag("gru_fwd_rows", 8);
    const bool rows4 = rows_env == 4 && (mode == 0 || mode == 8) && ndir * ((Bn + 3) / 4) == 16 && ksw == 4 && gi_bf16 && gates_f16;
    const int io_rows = rows4 ? 4 : 8, io_lds = rows4 ? 72 * 1024 : kPersistLds;
    const int nrecmax = rows4 ? 16 : 8;
    const int Gio = (Bn + io_rows - 1) / io_rows;
    const bool local = (mode == 0 || mode == 4 || mode == 7 || mode == 8) && ndir * Gio <= nrecmax;     // try the XCD-local hand-off
    // data polling (kernel comment): the default of the XCD-local form; mode 4 keeps the flag line for comparison
    const int forge = (mode == 7 ? 1 : (local && (mode == 0 || mode == 8) && ksw >= 2 ? 8 : 0));
    static int ring_env = -1;
    if (ring_env < 0) ring_env = debug_flag("fwd_ring", 1);      // (0: the payload is polled in the bf16 sequence itself)
    const bool use_ring = local && (forge & 8) && ring_env;
    if (rows4 && !use_ring) return ASR_ERR_UNSUPPORTED;
    if (gates_f16 && !use_ring) return ASR_ERR_UNSUPPORTED;
    if (!use_ring && (forge & 8) && (boff != 0 || Bn != B)) return ASR_ERR_UNSUPPORTED;      // (slabs: not the form that fills / polls the whole sequence)
    if (use_ring) {
        if (!clear_sync_fill(sync_ws, kShardBytes, (char*)sync_ws + kPsOffset, (size_t)nrecmax * PS_RING * 8 * H * 2, st)) return ASR_ERR_LAUNCH;
    } else if (!clear_sync_fill(sync_ws, kShardBytes, (forge & 8) ? hseq_bf16 : nullptr, (size_t)T * B * ndir * H * 2, st)) return ASR_ERR_LAUNCH;
    const int forge_k = forge | (fwd_poll_delay(H, use_ring) << 8);
    const dim3 igrid = local ? dim3(nrecmax * (H / 16)) : dim3(H / 16, Gio, ndir), iblock(384);
    if (rows4) {
        (void)hipFuncSetAttribute((const void*)fwd_persistent_io_kernel<4, true, true, true, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLds);
        hipLaunchKernelGGL((fwd_persistent_io_kernel<4, true, true, true, true, 4>), igrid, iblock, io_lds, st, gi_any, (const uint16_t*)whh_bf16, bhh, hseq,
                           (uint16_t*)hseq_bf16, gates, (unsigned*)sync_ws, T, B, H, ndir, io_rows, forge_k, boff, Bn);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
#define ASR_FWDIO_(K, L, G, R)                                                                                            \
    do {                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)fwd_persistent_io_kernel<K, L, G, R>, hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLds); \
        hipLaunchKernelGGL((fwd_persistent_io_kernel<K, L, G, R>), igrid, iblock, io_lds, st, gi_any, (const uint16_t*)whh_bf16, bhh, hseq, \
                           (uint16_t*)hseq_bf16, gates, (unsigned*)sync_ws, T, B, H, ndir, io_rows, L ? forge_k : 0, boff, Bn);         \
    } while (0)
#define ASR_FWDIO16_(K, G)                                                                                                \
    do {                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)fwd_persistent_io_kernel<K, true, G, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLds); \
        hipLaunchKernelGGL((fwd_persistent_io_kernel<K, true, G, true, true>), igrid, iblock, io_lds, st, gi_any, (const uint16_t*)whh_bf16, bhh, hseq, \
                           (uint16_t*)hseq_bf16, gates, (unsigned*)sync_ws, T, B, H, ndir, io_rows, forge_k, boff, Bn);                 \
    } while (0)
#define ASR_FWDIO(K)                                                                                                      \
    do {                                                                                                                  \
        if (use_ring && gates_f16) { if (gi_bf16) ASR_FWDIO16_(K, true); else ASR_FWDIO16_(K, false); }                   \
        else if (use_ring) { if (gi_bf16) ASR_FWDIO_(K, true, true, true); else ASR_FWDIO_(K, true, false, true); }       \
        else if (local) { if (gi_bf16) ASR_FWDIO_(K, true, true, false); else ASR_FWDIO_(K, true, false, false); }        \
        else { if (gi_bf16) ASR_FWDIO_(K, false, true, false); else ASR_FWDIO_(K, false, false, false); }                 \
    } while (0)
    if (ksw <= 1) ASR_FWDIO(1); else if (ksw <= 2) ASR_FWDIO(2); else if (ksw <= 4) ASR_FWDIO(4); else ASR_FWDIO(8);
#undef ASR_FWDIO_
#undef ASR_FWDIO16_
#undef ASR_FWDIO
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_gru_fwd(void* stream, void* gi_any, int gi_bf16, const void* whh_bf16, const float* bhh, float* hseq,
                           void* hseq_bf16, void* gates_any, void* y_bf16, int T, int B, int H, int ndir, void* sync_ws,
                           int mode, const int* x_len, int gates_f16) {
    if (ASR_ACT_IS_F16) return ASR_ERR_UNSUPPORTED;      // (the recurrences handle bfloat16 bits directly: common.hpp)
    float* gates = reinterpret_cast<float*>(gates_any);
    if (!gi_any || !whh_bf16 || !bhh || !hseq || !hseq_bf16 || !gates) return ASR_ERR_BAD_ARG;
    const int rc = check_dims(T, B, H, ndir);
    if (rc != ASR_OK) return rc;
    if (!valid_mode(mode)) return ASR_ERR_BAD_ARG;
    if (mode == 9 || mode == 10) mode = 0;      // (those select backward kernels)
    const int slab = slab_rows(T, B, H, ndir, mode, sync_ws);
    if (gates_f16 && !fwd_ring_form(T, slab ? slab : B, H, ndir, mode, sync_ws)) return ASR_ERR_UNSUPPORTED;       // (ask asr_gru_gates_f16_ok)
    const float* gi = reinterpret_cast<const float*>(gi_any);
    if (gi_bf16 && !slab && fwd_family(T, B, H, ndir, mode, sync_ws) != 2) return ASR_ERR_UNSUPPORTED;      // (ask asr_gru_fwd_accepts_bf16_gi)
    hipStream_t st = (hipStream_t)stream;
    if (x_len) {        // pin the update gate of the rows beyond each utterance's length (see pin_update_gate_kernel)
        if ((long long)T * B > 0x7fffffffLL) return ASR_ERR_UNSUPPORTED;
        if (gi_bf16) hipLaunchKernelGGL(pin_update_gate_kernel<uint16_t>, dim3((unsigned)(T * B)), dim3(256), 0, st, (uint16_t*)gi_any, x_len, B, H, ndir);
        else hipLaunchKernelGGL(pin_update_gate_kernel<float>, dim3((unsigned)(T * B)), dim3(256), 0, st, (float*)gi_any, x_len, B, H, ndir);
        ASR_LAUNCH_CHECK();
    }
    const dim3 grid(H / 16, ndir), block(256);
    const int ksw = (H / 32 + 3) / 4;
    if (ksw > 8) return ASR_ERR_UNSUPPORTED;      // H <= 1024
    const bool persist = can_persist(T, B, H, ndir, mode, sync_ws);
    if (mode >= 2 && !persist) return ASR_ERR_UNSUPPORTED;
    const int family = fwd_family(T, B, H, ndir, mode, sync_ws);
    // (the wide form serves the placement-free mode 2; see fwd_family)
    if (slab) {         // consecutive slabs of <= 32 rows through the default kernel (see slab_rows); the launches share sync_ws in stream order
        for (int boff = 0; boff < B; boff += slab) {
            const int rc2 = fwd_io_launch(st, gi_any, gi_bf16, whh_bf16, bhh, hseq, hseq_bf16, gates, sync_ws, T, B, H, ndir, mode, gates_f16, boff,
                                          B - boff < slab ? B - boff : slab);
            if (rc2 != ASR_OK) return rc2;
        }
    } else if (family == 1) {
        // wide form (32 units x 4-row recurrences).  Measured at T=1000, B=32, H=512: with the XCD-local hand-off it ties
        // the 16-unit x 8-row kernel (2.11 vs 2.12 us: the smaller payload is paid back in the 8-wave reduction), with the
        // placement-free hand-off it wins (2.47 vs 2.72 us) -- so it serves mode 2 only; the backward pass is wide in both.
        const int Gw = (B + 3) / 4, nrec = ndir * Gw, nrec_pad = (nrec + 7) & ~7;
        const int ks8 = (H / 32 + 7) / 8;
        if (!clear_sync(sync_ws, kShardBytes, st)) return ASR_ERR_LAUNCH;
        const bool local = mode == 0 || mode == 4 || mode == 7 || mode == 8;
        const int forge = (mode == 7 ? 1 : (mode == 0 || mode == 8 ? 8 : 0));
        if ((forge & 8) && hipMemsetAsync(hseq_bf16, 0xff, (size_t)T * B * ndir * H * 2, st) != hipSuccess) return ASR_ERR_LAUNCH;
        const dim3 wgrid = local ? dim3(nrec_pad * (H / 32)) : dim3(H / 32, Gw, ndir), wblock(640);
#define ASR_FWDW(K)                                                                                                       \
    do {                                                                                                                  \
        if (local) {                                                                                                      \
            (void)hipFuncSetAttribute((const void*)fwd_wide_kernel<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLds); \
            hipLaunchKernelGGL((fwd_wide_kernel<K, true>), wgrid, wblock, kPersistLds, st, gi, (const uint16_t*)whh_bf16, bhh, hseq,  \
                               (uint16_t*)hseq_bf16, gates, (unsigned*)sync_ws, T, B, H, ndir, forge);                        \
        } else {                                                                                                          \
            (void)hipFuncSetAttribute((const void*)fwd_wide_kernel<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLds); \
            hipLaunchKernelGGL((fwd_wide_kernel<K, false>), wgrid, wblock, kPersistLds, st, gi, (const uint16_t*)whh_bf16, bhh, hseq, \
                               (uint16_t*)hseq_bf16, gates, (unsigned*)sync_ws, T, B, H, ndir, 0);                            \
        }                                                                                                                 \
    } while (0)
        if (ks8 <= 1) ASR_FWDW(1); else if (ks8 <= 2) ASR_FWDW(2); else ASR_FWDW(4);
#undef ASR_FWDW
    } else if (family == 2) {
        const int rc2 = fwd_io_launch(st, gi_any, gi_bf16, whh_bf16, bhh, hseq, hseq_bf16, gates, sync_ws, T, B, H, ndir, mode, gates_f16, 0, B);
        if (rc2 != ASR_OK) return rc2;
    } else
    for (int s = 0; s < T; ++s) {
#define ASR_FWD(K) hipLaunchKernelGGL(fwd_step_kernel<K>, grid, block, 0, st, gi, (const uint16_t*)whh_bf16, bhh, hseq, \
                                      (uint16_t*)hseq_bf16, gates, T, B, H, ndir, s, (const float*)nullptr)
        if (ksw <= 1) ASR_FWD(1); else if (ksw <= 2) ASR_FWD(2); else if (ksw <= 4) ASR_FWD(4); else ASR_FWD(8);
#undef ASR_FWD
    }
    ASR_LAUNCH_CHECK();
    if (y_bf16) {
        const bool vec = (H & 7) == 0 && ((((uintptr_t)hseq) | ((uintptr_t)y_bf16)) & 15) == 0;
        const long long n = (long long)T * B * (vec ? H / 8 : H);
        long long g = (n + 255) / 256;
        if (g > 4096) g = 4096;
        if (vec)
            hipLaunchKernelGGL(merge_dirs_vec_kernel, dim3((unsigned)g), dim3(256), 0, st, hseq, (uint16_t*)y_bf16,
                               (long long)T * B, H, ndir, x_len, B);
        else
            hipLaunchKernelGGL(merge_dirs_kernel, dim3((unsigned)g), dim3(256), 0, st, hseq, (uint16_t*)y_bf16,
                               (long long)T * B, H, ndir, x_len, B);
        ASR_LAUNCH_CHECK();
    }
    return ASR_OK;
}

extern "C" int asr_gru_bwd(void* stream, const void* dy_bf16, const void* gates_any, const float* hseq,
                           const void* whhT_bf16, void* dgi_bf16, void* dgh_bf16, float* carry_ws, float* db_ih,
                           float* db_hh, int T, int B, int H, int ndir, void* sync_ws, int mode, const int* x_len,
                           void* dy_ws, int gates_f16) {
    if (ASR_ACT_IS_F16) return ASR_ERR_UNSUPPORTED;      // (the recurrences handle bfloat16 bits directly: common.hpp)
    if (!dy_bf16 || !gates_any || !hseq || !whhT_bf16 || !dgi_bf16 || !dgh_bf16 || !carry_ws) return ASR_ERR_BAD_ARG;
    const float* gates = reinterpret_cast<const float*>(gates_any);
    if (!valid_mode(mode)) return ASR_ERR_BAD_ARG;
    if (gates_f16 && !((bwd_ps_form(T, B, H, ndir, mode, sync_ws) || slab_rows(T, B, H, ndir, mode, sync_ws)) && db_ih && db_hh)) return ASR_ERR_UNSUPPORTED;
    const int rc = check_dims(T, B, H, ndir);
    if (rc != ASR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (x_len) {        // the forward pass pinned z = 1 beyond the lengths (no gate gradient, dh passes); drop dy there
        if (!dy_ws || (H % 8)) return ASR_ERR_BAD_ARG;
        const long long n = (long long)T * B * (H / 8);
        long long g = (n + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, (const uint16_t*)dy_bf16, (uint16_t*)dy_ws, x_len,
                           (long long)T * B, B, H / 8);
        ASR_LAUNCH_CHECK();
        dy_bf16 = dy_ws;
    }
    const dim3 grid(H / 16, ndir), block(256);
    const int ksw = (3 * H / 32 + 3) / 4;
    const bool persist = can_persist(T, B, H, ndir, mode, sync_ws) && ksw <= 12;
    if (mode >= 2 && !persist) return ASR_ERR_UNSUPPORTED;
    // partial-sum exchange (bwd_ps_kernel): the default where it applies (modes 0 / 8; 9 asks for it, 10 forges a split
    // placement so that its placement-free stores are exercised); measured 1.78 -> 1.59 us per step at T=1000, B=32, H=512
    const int slab = (db_ih && db_hh) ? slab_rows(T, B, H, ndir, mode, sync_ws) : 0;       // (see slab_rows: batches beyond 32 rows)
    if (slab || (persist && (mode == 0 || mode == 8 || mode == 9 || mode == 10) && db_ih && db_hh && H % 128 == 0 && H <= 1024 &&
                 ndir * ((B + 3) / 4) <= 16)) {
        for (int boff = 0; boff < B; boff += slab ? slab : B) {
            const int Bn = slab ? (B - boff < slab ? B - boff : slab) : B;
            const int Gw = (Bn + 3) / 4, nrec = ndir * Gw, nrec_pad = (nrec + 7) & ~7;
            if (!clear_sync_fill(sync_ws, kShardBytes, (char*)sync_ws + kPsOffset, ps_exchange_bytes(nrec_pad, H), st)) return ASR_ERR_LAUNCH;
            const dim3 wgrid(nrec_pad * (H / 32)), wblock(640);
#define ASR_BWDPS_(NT_, G)                                                                                                \
    do {                                                                                                                  \
        (void)hipFuncSetAttribute((const void*)bwd_ps_kernel<NT_, true, G>, hipFuncAttributeMaxDynamicSharedMemorySize, kExclusiveLds); \
        hipLaunchKernelGGL((bwd_ps_kernel<NT_, true, G>), wgrid, wblock, kExclusiveLds, st, (const uint16_t*)dy_bf16, gates, hseq, \
                           (const uint16_t*)whhT_bf16, (uint16_t*)dgi_bf16, (uint16_t*)dgh_bf16, db_ih, db_hh, (unsigned*)sync_ws, \
                           T, B, H, ndir, mode == 10 ? 1 : 0, boff, Bn);                                                  \
    } while (0)
#define ASR_BWDPS(NT_)                                                                                                    \
    do {                                                                                                                  \
        if (gates_f16) ASR_BWDPS_(NT_, true); else ASR_BWDPS_(NT_, false);                                                \
    } while (0)
            switch (H / 128) {
                case 1: ASR_BWDPS(1); break;
                case 2: ASR_BWDPS(2); break;
                case 3: ASR_BWDPS(3); break;
                case 4: ASR_BWDPS(4); break;
                case 8: ASR_BWDPS(8); break;
                default: return ASR_ERR_UNSUPPORTED;
            }
#undef ASR_BWDPS
#undef ASR_BWDPS_
            ASR_LAUNCH_CHECK();
        }
        return ASR_OK;
    }
    // wide form (32 units x 4-row recurrences): every other persistent case
    if (persist && db_ih && db_hh && H >= 64 && ndir * ((B + 3) / 4) <= 16) {
        const int Gw = (B + 3) / 4, nrec = ndir * Gw, nrec_pad = (nrec + 7) & ~7;
        const int ks8 = (3 * H / 32 + 7) / 8;
        if (ks8 <= 6) {
            if (!clear_sync(sync_ws, kShardBytes, st)) return ASR_ERR_LAUNCH;
            const bool local = mode == 0 || mode == 4 || mode == 7 || mode == 8;
            const int forge = (mode == 7 ? 1 : (mode == 0 || mode == 8 ? 8 : 0));     // 8: data polling (mode 4: flag line)
            if ((forge & 8) && hipMemsetAsync(dgh_bf16, 0xff, (size_t)T * B * ndir * 3 * H * 2, st) != hipSuccess) return ASR_ERR_LAUNCH;
            const dim3 wgrid = local ? dim3(nrec_pad * (H / 32)) : dim3(H / 32, Gw, ndir), wblock(640);
#define ASR_BWDW(K)                                                                                                       \
    do {                                                                                                                  \
        if (local) {                                                                                                      \
            (void)hipFuncSetAttribute((const void*)bwd_wide_kernel<K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kExclusiveLds); \
            hipLaunchKernelGGL((bwd_wide_kernel<K, true>), wgrid, wblock, kExclusiveLds, st, (const uint16_t*)dy_bf16, gates, hseq, \
                               (const uint16_t*)whhT_bf16, (uint16_t*)dgi_bf16, (uint16_t*)dgh_bf16, db_ih, db_hh,                \
                               (unsigned*)sync_ws, T, B, H, ndir, forge);                                                         \
        } else {                                                                                                          \
            (void)hipFuncSetAttribute((const void*)bwd_wide_kernel<K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLds); \
            hipLaunchKernelGGL((bwd_wide_kernel<K, false>), wgrid, wblock, kPersistLds, st, (const uint16_t*)dy_bf16, gates, hseq, \
                               (const uint16_t*)whhT_bf16, (uint16_t*)dgi_bf16, (uint16_t*)dgh_bf16, db_ih, db_hh,                \
                               (unsigned*)sync_ws, T, B, H, ndir, 0);                                                             \
        }                                                                                                                 \
    } while (0)
            if (ks8 <= 2) ASR_BWDW(2); else if (ks8 <= 3) ASR_BWDW(3); else if (ks8 <= 4) ASR_BWDW(4); else if (ks8 <= 5) ASR_BWDW(5); else ASR_BWDW(6);
#undef ASR_BWDW
            ASR_LAUNCH_CHECK();
            return ASR_OK;
        }
    }
    if (mode >= 2) return ASR_ERR_UNSUPPORTED;       // (no persistent kernel serves this shape; modes >= 2 do not fall back)
    if (ksw > 12) return ASR_ERR_UNSUPPORTED;        // (the per-step kernel holds at most 12 K steps per wave: H <= 512)
    for (int s = 0; s < T; ++s) {
#define ASR_BWD(K) hipLaunchKernelGGL(bwd_step_kernel<K>, grid, block, 0, st, (const uint16_t*)dy_bf16, gates, hseq, \
                                      (const uint16_t*)whhT_bf16, (uint16_t*)dgi_bf16, (uint16_t*)dgh_bf16, carry_ws, T, B, H, ndir, s, (const float*)nullptr, 0)
        if (ksw <= 2) ASR_BWD(2); else if (ksw <= 6) ASR_BWD(6); else ASR_BWD(12);
#undef ASR_BWD
    }
    ASR_LAUNCH_CHECK();
    // bias gradients (the persistent kernels sum them in registers; here a column sum over all (t, b) rows)
    if (db_ih) {
        const int rc2 = asr_colsum_acc(stream, dgi_bf16, 1, (long long)T * B, ndir * 3 * H, ndir * 3 * H, db_ih);
        if (rc2 != ASR_OK) return rc2;
    }
    if (db_hh) {
        const int rc2 = asr_colsum_acc(stream, dgh_bf16, 1, (long long)T * B, ndir * 3 * H, ndir * 3 * H, db_hh);
        if (rc2 != ASR_OK) return rc2;
    }
    return ASR_OK;
}

// ---------------------------------------------------------------------------------------------- a layer with a given initial state
// chainer.links.NStepGRU / NStepBiGRU's hx and hy (asr/nn/nn.py:3 exports them; no recipe of the reference hands a state to a GRU --
// its SRU model carries one, run/ctc/sru/model.py:105-122): the same arithmetic on the one-launch-per-time-step kernels, which take the
// state in front of the first step from `hx` instead of zeros.  The final state is rows of hseq (direction 0: t = T - 1, direction 1:
// t = 0; with x_len the frozen state of a shorter utterance), so there is no hy argument.
extern "C" int asr_gru_fwd_state(void* stream, const float* gi, const void* whh_bf16, const float* bhh, const float* hx, float* hseq,
                                 void* hseq_bf16, float* gates, void* y_bf16, int T, int B, int H, int ndir, const int* x_len) {
    if (ASR_ACT_IS_F16) return ASR_ERR_UNSUPPORTED;
    if (!gi || !whh_bf16 || !bhh || !hseq || !hseq_bf16 || !gates) return ASR_ERR_BAD_ARG;
    const int rc = check_dims(T, B, H, ndir);
    if (rc != ASR_OK) return rc;
    const int ksw = (H / 32 + 3) / 4;
    if (ksw > 8 || (H % 8)) return ASR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (x_len) {
        if ((long long)T * B > 0x7fffffffLL) return ASR_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(pin_update_gate_kernel<float>, dim3((unsigned)(T * B)), dim3(256), 0, st, (float*)gi, x_len, B, H, ndir);
        ASR_LAUNCH_CHECK();
    }
    const dim3 grid(H / 16, ndir), block(256);
    for (int s = 0; s < T; ++s) {
#define ASR_FWD(K) hipLaunchKernelGGL(fwd_step_kernel<K>, grid, block, 0, st, gi, (const uint16_t*)whh_bf16, bhh, hseq, \
                                      (uint16_t*)hseq_bf16, gates, T, B, H, ndir, s, hx)
        if (ksw <= 1) ASR_FWD(1); else if (ksw <= 2) ASR_FWD(2); else if (ksw <= 4) ASR_FWD(4); else ASR_FWD(8);
#undef ASR_FWD
    }
    ASR_LAUNCH_CHECK();
    if (y_bf16) {
        const bool vec = ((((uintptr_t)hseq) | ((uintptr_t)y_bf16)) & 15) == 0;
        const long long n = (long long)T * B * (vec ? H / 8 : H);
        long long g = (n + 255) / 256;
        if (g > 4096) g = 4096;
        if (vec) hipLaunchKernelGGL(merge_dirs_vec_kernel, dim3((unsigned)g), dim3(256), 0, st, hseq, (uint16_t*)y_bf16, (long long)T * B, H, ndir, x_len, B);
        else hipLaunchKernelGGL(merge_dirs_kernel, dim3((unsigned)g), dim3(256), 0, st, hseq, (uint16_t*)y_bf16, (long long)T * B, H, ndir, x_len, B);
        ASR_LAUNCH_CHECK();
    }
    return ASR_OK;
}

extern "C" int asr_gru_bwd_state(void* stream, const void* dy_bf16, const float* gates, const float* hseq, const float* hx, const float* dhy,
                                 const void* whhT_bf16, void* dgi_bf16, void* dgh_bf16, float* carry_ws, float* db_ih, float* db_hh,
                                 float* dhx, int T, int B, int H, int ndir, const int* x_len, void* dy_ws) {
    if (ASR_ACT_IS_F16) return ASR_ERR_UNSUPPORTED;
    if (!dy_bf16 || !gates || !hseq || !whhT_bf16 || !dgi_bf16 || !dgh_bf16 || !carry_ws) return ASR_ERR_BAD_ARG;
    const int rc = check_dims(T, B, H, ndir);
    if (rc != ASR_OK) return rc;
    const int ksw = (3 * H / 32 + 3) / 4;
    if (ksw > 12 || (H % 8)) return ASR_ERR_UNSUPPORTED;        // (the per-step kernel holds at most 12 K steps per wave: H <= 512)
    hipStream_t st = (hipStream_t)stream;
    if (x_len) {
        if (!dy_ws) return ASR_ERR_BAD_ARG;
        const long long n = (long long)T * B * (H / 8);
        long long g = (n + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)g), dim3(256), 0, st, (const uint16_t*)dy_bf16, (uint16_t*)dy_ws, x_len, (long long)T * B, B, H / 8);
        ASR_LAUNCH_CHECK();
        dy_bf16 = dy_ws;
    }
    if (dhy && hipMemcpyAsync(carry_ws, dhy, (size_t)ndir * B * H * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return ASR_ERR_LAUNCH;
    const dim3 grid(H / 16, ndir), block(256);
    for (int s = 0; s < T; ++s) {
#define ASR_BWD(K) hipLaunchKernelGGL(bwd_step_kernel<K>, grid, block, 0, st, (const uint16_t*)dy_bf16, gates, hseq, (const uint16_t*)whhT_bf16, \
                                      (uint16_t*)dgi_bf16, (uint16_t*)dgh_bf16, carry_ws, T, B, H, ndir, s, hx, dhy ? 1 : 0)
        if (ksw <= 2) ASR_BWD(2); else if (ksw <= 6) ASR_BWD(6); else ASR_BWD(12);
#undef ASR_BWD
    }
    ASR_LAUNCH_CHECK();
    if (dhx) {
        const long long total = (long long)ndir * B * H;
        hipLaunchKernelGGL(dhx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint16_t*)dgh_bf16, (const uint16_t*)whhT_bf16,
                           (const float*)carry_ws, dhx, T, B, H, ndir);
        ASR_LAUNCH_CHECK();
    }
    if (db_ih) {
        const int rc2 = asr_colsum_acc(stream, dgi_bf16, 1, (long long)T * B, ndir * 3 * H, ndir * 3 * H, db_ih);
        if (rc2 != ASR_OK) return rc2;
    }
    if (db_hh) {
        const int rc2 = asr_colsum_acc(stream, dgh_bf16, 1, (long long)T * B, ndir * 3 * H, ndir * 3 * H, db_hh);
        if (rc2 != ASR_OK) return rc2;
    }
    return ASR_OK;
}
