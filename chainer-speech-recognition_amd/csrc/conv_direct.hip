// Direct convolution with the activation block resident in LDS (gfx950).
//
// The implicit-GEMM kernels of gemm.hip (gemm_nt256p_kernel<.., CONV>) fetch a 256-row tile of the activation operand from the L2 once per
// tap: a (3, 5) filter reads every activation row 15 times, and a product with few output columns (backward-data of the 64 -> 128
// convolution: N = 64) is bound by that fill -- 983 KB of activations per 256 x 64 tile, 0.21 of its roofline.
// Here a workgroup owns ONE utterance b, Tt consecutive time steps and all Hr output heights; the activations that tile can touch --
// (Tt + KW - 1) time steps x (Hr + KH - 1) heights x Cs channels -- are loaded into LDS once (rows outside the tensor as zeros), and every
// K step reads its MFMA A fragments from that block at the tap's offset.  Only the weight tiles (16 NJ columns x 32 k per step) still
// stream: global -> registers two steps ahead -> a two-stage LDS ring.  No LDS-DMA anywhere in this kernel: with one present the compiler
// puts s_waitcnt vmcnt(0) in front of every LDS read (the DMA may alias it), which waited for the weight tile just asked for -- a full L2
// round trip per K step (first version: 170 us where the implicit-GEMM kernel takes 134).
//
// Same arithmetic and operand formats as asr_conv_nt (gemm.hip): out[(t B + b) Hr + h][n] = bias[n] + sum_k A[..][k] W[n][k],
// k = (kh KW + kw) Cs + ci, A = x[t + sgn (kw - pt)][b][h + sgn (kh - ph)][ci] or zero outside the tensor; float32 accumulation, bf16
// output.  Replaces chainer.functions.convolution_2d forward / backward-data on the path asr/nn/convolution_2d.py:70-118.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "common.hpp"

namespace asr {
namespace convd {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
union Frag {
    bf16x8 v;
    uint4 u;
};
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Desc {
    int B, Hs, Cs, Ts, KH, KW, ph, pt, sgn, Hr, Tr;
    int Tt, TB, HB, dtmin, dhmin;       // tile: time steps per workgroup, block extents, smallest time / height offset of a tap
    int N, K, tiles_t, tiles_n, npos;
    int Cb, npass;                      // channels of the resident block (<= 128) and passes over the channels (Cs / Cb)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NSTG = 3;         // LDS stages of the weight ring (three more steps are in registers)

// NJ: 16-column MFMA tiles per wave = tile width / 16 (4 -> 64 columns for N <= 64, 8 -> 128).  Four waves, each 16 MI rows x all columns:
// MI = 2 -> tiles of 128 rows, two workgroups per CU (a workgroup alone on its CU -- MI = 4, 256 rows, ~100 KB of LDS -- has nothing to
// cover its LDS latency and barrier with: 880 cycles per K step of 256 MFMA cycles).
// LOG_NCH: log2(Cs / 8), the 16-B chunks of a position (Cs = 32, 64, 128, 256).
// KS: 32-wide K steps per barrier (2 where the tile is 64 columns wide: 8 MFMAs per wave between two barriers were too few).
// MULTI: more than 128 input channels, i.e. several passes over the channels with the accumulators kept (they then live through the block
// loads: 169 instead of 117 registers for the 64-column form, two workgroups per CU instead of four -- so the one-pass kernels stay apart).
template <int NJ, int LOG_NCH, int MI, int KS, bool MULTI>
__global__ __launch_bounds__(256, MI <= 3 ? 2 : 1) void conv_direct_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ W,
                                                             uint16_t* __restrict__ out, const float* __restrict__ bias, Desc d, unsigned x_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCH = 1 << LOG_NCH, TNW = 16 * NJ, B_HALF = TNW * 64, B_STAGE = KS * B_HALF, BP = NJ / 4;
    constexpr int R = KS == 2 ? 4 : 6;       // register sets of weight tiles in flight
    constexpr int POS_BYTES = NCH * 16;
    // 16-B chunk c of position p lives at slot c ^ swz(p): sixteen consecutive positions x one chunk index then cover all sixteen 16-B
    // bank groups of the LDS (a position is POS_BYTES = 64 .. 512 B: unswizzled, positions 256 / POS_BYTES apart met in the same banks)
    auto swz = [](int p) { return LOG_NCH >= 4 ? (p & 15) : ((p >> (4 - LOG_NCH)) & (NCH - 1)); };
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ablk_bytes = ((d.npos * POS_BYTES + 1023) >> 10) << 10;
    char* Ab = smem;
    char* Bs = smem + ablk_bytes;
    int bid = blockIdx.x;
    const int tn = bid % d.tiles_n; bid /= d.tiles_n;
    const int b = bid % d.B;
    const int t0 = (bid / d.B) * d.Tt;
    const int Mt = min(d.Tt, d.Tr - t0) * d.Hr;                 // live rows of this tile
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, -1, 0x00020000);

    // ---- weight tiles: LDS row rho of a stage holds output column NJ (rho % 16) + rho / 16 of the tile (see the epilogue), 64 B per row,
    // chunk q at position q ^ g4(row) (the swizzle of gemm.hip's NT kernels)
    auto g4 = [](int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; };
    const int lrow = tid >> 2, lchunk = ((tid & 3) ^ g4(lrow)) * 16;
    unsigned ob[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int rho = i * 64 + lrow;
        const int col = tn * TNW + NJ * (rho & 15) + (rho >> 4);
        ob[i] = __umul24((unsigned)min(col, d.N - 1), (unsigned)(d.K * 2)) + lchunk;
    }
    // K steps of 32 KS columns per channel pass (the host chooses KS = 2 only where that is even).  With more than 128 input channels the
    // block holds Cb = 128 of them and the tile is computed in Cs / Cb passes (block reloaded, accumulators kept): sub-step u of pass cp is
    // tap u / (Cb / 32), channels cp Cb + 32 (u % (Cb / 32)) ..  -- column (tap Cs + that) of W
    const int nk = (d.K / d.npass >> 5) / KS;
    const int spt_log = LOG_NCH - 2;           // log2(Cb / 32)
    int cp = 0;                                // the channel pass
    // weight tiles in flight (register sets): every workgroup of an XCD asks the same few KB of its L2 for the same tile at the same time
    // and the answer takes ~1800 cycles, several K steps of this workgroup (with two steps of distance a K step took 880 cycles, not 300)
    u32x4 breg[R][KS * BP];
    auto load_b = [&](u32x4 (&regs)[KS * BP], int ks) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int i = 0; i < BP; ++i)
            {
                const int u = ks * KS + kk;
                const int kofs = ((u >> spt_log) * d.Cs + cp * d.Cb + ((u & ((1 << spt_log) - 1)) << 5)) * 2;
                regs[kk * BP + i] = ks < nk ? __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, ob[i], kofs, 0) : (u32x4){0u, 0u, 0u, 0u};
            }
    };
    auto write_b = [&](const u32x4 (&regs)[KS * BP], int stage) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int i = 0; i < BP; ++i) *reinterpret_cast<u32x4*>(Bs + stage * B_STAGE + kk * B_HALF + (i * 256 + tid) * 16) = regs[kk * BP + i];
    };
    // ---- fragment addresses
    const int q = lane >> 4, r = lane & 15;
    int pbase[MI];              // block position of this lane's row of M tile i at the tap (dtmin, dhmin)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int m = wid * (16 * MI) + i * 16 + r;
        if (m >= Mt) m = 0;
        const int tl = m / d.Hr, h = m - tl * d.Hr;
        pbase[i] = tl * d.HB + h;
    }
    const int boff0 = r * 64 + ((q ^ g4(r)) << 4);
    f32x4 acc[MI][NJ];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    if (MULTI) zero_acc();

    for (cp = 0; cp < (MULTI ? d.npass : 1); ++cp) {
    #pragma unroll
        for (int j = 0; j < R; ++j) load_b(breg[j], j);
        // ---- the activation block: chunk index idx = p NCH + slot lives at LDS byte idx 16; UF loads in flight per thread (with four, the
        // 88 KB block of a 128-channel tile took five dependent round trips to memory: 20 of a tile's 31 us)
        {
            constexpr int UF = 12;
            const int nchunks = ablk_bytes >> 4;
            for (int base = tid; base < nchunks; base += UF * 256) {
                u32x4 v[UF];
    #pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const int idx = base + u * 256;
                    const int p = idx >> LOG_NCH, slot = idx & (NCH - 1);
                    const int c = slot ^ swz(p);
                    const int tb = p / d.HB, hb = p - tb * d.HB;
                    const int t = t0 + d.dtmin + tb, h = d.dhmin + hb;
                    const bool ok = idx < nchunks && p < d.npos && (unsigned)t < (unsigned)d.Ts && (unsigned)h < (unsigned)d.Hs;
                    const unsigned off = (unsigned)((((size_t)t * d.B + b) * d.Hs + h) * d.Cs + cp * d.Cb + c * 8) * 2u;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? off : 0xfffffff0u, 0, 0);       // (beyond num_records: zeros)
                }
    #pragma unroll
                for (int u = 0; u < UF; ++u)
                    if (base + u * 256 < nchunks) *reinterpret_cast<u32x4*>(Ab + (size_t)(base + u * 256) * 16) = v[u];
            }
        }
        write_b(breg[0], 0);
        write_b(breg[1], 1);
        load_b(breg[0], R);

        if (!MULTI) zero_acc();
        __syncthreads();
        int kh = 0, kw = 0, ci = 0;
        // fragments of one K step: per 32-wide sub-step the MI row tiles of the resident block at the tap's offset + the NJ column tiles of an LDS stage
        Frag fa[2][KS * MI], fb[2][KS * NJ];
        auto fetch = [&](Frag (&a)[KS * MI], Frag (&bq)[KS * NJ], int stage) {
    #pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                // tap of the sub-step -> offset inside the block; empty taps behind KH KW (zero weights) read tap 0
                const int dt = d.sgn * (kw - d.pt) - d.dtmin, dh = d.sgn * (kh - d.ph) - d.dhmin;
                const int dp = kh < d.KH ? dt * d.HB + dh : 0;
                const int c0 = (ci >> 3) + q;
                const char* Bb = Bs + stage * B_STAGE + kk * B_HALF;
    #pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int p = pbase[i] + dp;
                    a[kk * MI + i].u = *reinterpret_cast<const uint4*>(Ab + p * POS_BYTES + ((c0 ^ swz(p)) << 4));
                }
    #pragma unroll
                for (int j = 0; j < NJ; ++j) bq[kk * NJ + j].u = *reinterpret_cast<const uint4*>(Bb + boff0 + j * 1024);
                ci += 32;
                if (ci >= d.Cb) { ci = 0; if (++kw == d.KW) { kw = 0; ++kh; } }
            }
        };
        fetch(fa[0], fb[0], 0);
        // K step ks (P = ks % 12): the weight tile of step ks + R + 1 is asked for; the MFMAs run on fragments fetched one step ago; the tile of
        // step ks + 2 (asked for R - 1 steps ago) goes to LDS stage (ks + 2) % 3; the fragments of step ks + 1 -- its stage was written one step
        // ago and published by that step's barrier -- are fetched BEFORE this step's barrier, so that their LDS latency and the barrier
        // overlap the MFMAs still in the pipe.
        auto kstep = [&](int ks, auto pp) {
            constexpr int P = decltype(pp)::value, P3 = P % 3, P2 = P % 2;
            load_b(breg[(P + 1) % R], ks + R + 1);
    #pragma unroll
            for (int kk = 0; kk < KS; ++kk)
    #pragma unroll
                for (int j = 0; j < NJ; ++j)
    #pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i][j] = ASR_MFMA_16x16x32(fa[P2][kk * MI + i].v, fb[P2][kk * NJ + j].v, acc[i][j]);
            write_b(breg[(P + 2) % R], (P3 + 2) % 3);
            if (ks + 1 < nk) fetch(fa[P2 ^ 1], fb[P2 ^ 1], (P3 + 1) % 3);
            __syncthreads();
        };
        for (int ks = 0; ks < nk; ks += 12) {
            kstep(ks, std::integral_constant<int, 0>());
            if (ks + 1 < nk) kstep(ks + 1, std::integral_constant<int, 1>());
            if (ks + 2 < nk) kstep(ks + 2, std::integral_constant<int, 2>());
            if (ks + 3 < nk) kstep(ks + 3, std::integral_constant<int, 3>());
            if (ks + 4 < nk) kstep(ks + 4, std::integral_constant<int, 4>());
            if (ks + 5 < nk) kstep(ks + 5, std::integral_constant<int, 5>());
            if (ks + 6 < nk) kstep(ks + 6, std::integral_constant<int, 6>());
            if (ks + 7 < nk) kstep(ks + 7, std::integral_constant<int, 7>());
            if (ks + 8 < nk) kstep(ks + 8, std::integral_constant<int, 8>());
            if (ks + 9 < nk) kstep(ks + 9, std::integral_constant<int, 9>());
            if (ks + 10 < nk) kstep(ks + 10, std::integral_constant<int, 10>());
            if (ks + 11 < nk) kstep(ks + 11, std::integral_constant<int, 11>());
        }
    }
    // ---- epilogue: acc[i][j][reg] = out[row wid 64 + 16 i + 4 q + reg][column tn TNW + NJ r + j]: NJ consecutive columns per lane and row
    const int col = tn * TNW + NJ * r;
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bv[j] = (bias && col + j < d.N) ? bias[col + j] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = wid * (16 * MI) + i * 16 + 4 * q + reg;
            if (m >= Mt) continue;
            const int tl = m / d.Hr, h = m - tl * d.Hr;
            uint16_t* dst = out + (((size_t)(t0 + tl) * d.B + b) * d.Hr + h) * d.N + col;
            if (col + NJ - 1 < d.N) {
                if (NJ == 8) {
                    uint4 pk;
                    pk.x = pack_bf16x2(acc[i][0][reg] + bv[0], acc[i][1][reg] + bv[1]);
                    pk.y = pack_bf16x2(acc[i][2][reg] + bv[2], acc[i][3][reg] + bv[3]);
                    pk.z = pack_bf16x2(acc[i][NJ == 8 ? 4 : 0][reg] + bv[NJ == 8 ? 4 : 0], acc[i][NJ == 8 ? 5 : 0][reg] + bv[NJ == 8 ? 5 : 0]);
                    pk.w = pack_bf16x2(acc[i][NJ == 8 ? 6 : 0][reg] + bv[NJ == 8 ? 6 : 0], acc[i][NJ == 8 ? 7 : 0][reg] + bv[NJ == 8 ? 7 : 0]);
                    *reinterpret_cast<uint4*>(dst) = pk;
                } else {
                    uint2 pk;
                    pk.x = pack_bf16x2(acc[i][0][reg] + bv[0], acc[i][1][reg] + bv[1]);
                    pk.y = pack_bf16x2(acc[i][2][reg] + bv[2], acc[i][3][reg] + bv[3]);
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (col + j < d.N) dst[j] = f32_to_bf16(acc[i][j][reg] + bv[j]);
            }
        }
}

}  // namespace convd
}  // namespace asr

using namespace asr;
using namespace asr::convd;

// 1 when asr_conv_direct_nt serves the shape (asr_conv_nt asks before falling back to the implicit-GEMM kernels)
extern "C" int asr_conv_direct_ok(int Ts, int B, int Hs, int Cs, int KH, int KW, int Tr, int Hr, int N, int K, int out_bf16) {
    if (!out_bf16 || (Cs != 32 && Cs != 64 && Cs != 128 && Cs != 256 && Cs != 512) || (N % 8) || (K % 32) || (K % Cs) || K < KH * KW * Cs) return 0;
    if (Hr > 128 || Hr <= 0 || KH > 8 || KW > 8 || Tr <= 0) return 0;
    if ((unsigned long long)Ts * B * Hs * Cs * 2 >= 0xfffffff0ull || (unsigned long long)N * K * 2 >= (1ull << 31)) return 0;
    const int Cb = Cs < 128 ? Cs : 128;
    const int Tt = (Cb <= 64 ? 192 : 128) / Hr;        // (see asr_conv_direct_nt)
    if (Tt < 1) return 0;
    const size_t lds = (((size_t)(Tt + KW - 1) * (Hr + KH - 1) * Cb * 2 + 1023) & ~(size_t)1023) + (size_t)NSTG * (N <= 64 ? 2 * 64 : 128) * 64;
    return lds <= 150 * 1024 ? 1 : 0;
}

extern "C" int asr_conv_direct_nt(void* stream_, const void* x, const void* W, int ldw, void* out, const float* bias, int Ts, int B, int Hs,
                                  int Cs, int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N) {
    if (!x || !W || !out || (sgn != 1 && sgn != -1)) return ASR_ERR_BAD_ARG;
    if (!asr_conv_direct_ok(Ts, B, Hs, Cs, KH, KW, Tr, Hr, N, ldw, 1)) return ASR_ERR_UNSUPPORTED;
    if (((((uintptr_t)x) | ((uintptr_t)W) | ((uintptr_t)out)) & 15) || (bias && (((uintptr_t)bias) & 15))) return ASR_ERR_UNSUPPORTED;
    Desc d;
    d.B = B; d.Hs = Hs; d.Cs = Cs; d.Ts = Ts; d.KH = KH; d.KW = KW; d.ph = pad_h; d.pt = pad_t; d.sgn = sgn; d.Hr = Hr; d.Tr = Tr;
    // rows of a tile: 128 (two 16-row MFMA tiles per wave) with 128-channel blocks, 192 (three) up to 64 channels, where the block is small
    // enough for two to four workgroups per CU either way: 64 -> 128 channels forward 132 -> 110 us, 64 -> 64 68 -> 63, 32 -> 64 117 -> 103
    // (with 128 channels the larger block leaves one workgroup per CU: 128 -> 256 486 -> 596)
    const bool mi3 = Cs <= 64;
    d.Tt = (mi3 ? 192 : 128) / Hr;
    d.TB = d.Tt + KW - 1; d.HB = Hr + KH - 1;
    d.dtmin = sgn > 0 ? -pad_t : pad_t - (KW - 1);
    d.dhmin = sgn > 0 ? -pad_h : pad_h - (KH - 1);
    d.N = N; d.K = ldw;
    d.tiles_t = (Tr + d.Tt - 1) / d.Tt;
    const bool narrow = N <= 64;
    d.tiles_n = (N + (narrow ? 63 : 127)) / (narrow ? 64 : 128);
    d.npos = d.TB * d.HB;
    d.Cb = Cs < 128 ? Cs : 128;
    d.npass = Cs / d.Cb;
    // two 32-wide K steps per barrier for the 64-column tiles whose block (128 channels) allows two workgroups per CU anyway: 128 -> 64
    // channels 178 -> 164 us; with 64 channels the 117 registers of the one-step form keep four workgroups on a CU and the 177 of the
    // two-step form two (64 -> 64: 68 -> 87 us)
    const bool ks2 = narrow && Cs >= 128 && ((ldw / d.npass >> 5) % 2) == 0;
    const size_t lds = (((size_t)d.npos * d.Cb * 2 + 1023) & ~(size_t)1023) + (size_t)NSTG * (narrow ? (ks2 ? 2 : 1) * 64 : 128) * 64;
    const unsigned x_bytes = (unsigned)((unsigned long long)Ts * B * Hs * Cs * 2);
    const long long grid = (long long)d.tiles_t * B * d.tiles_n;
    if (grid > 0x7fffffffLL) return ASR_ERR_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
#define ASR_CD(NJ_, L_, MI_, KS_, MU_)                                                                                                \
    do {                                                                                                                             \
        static bool attr_ = false;                                                                                                   \
        if (!attr_) {                                                                                                                \
            (void)hipFuncSetAttribute((const void*)conv_direct_kernel<NJ_, L_, MI_, KS_, MU_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
            attr_ = true;                                                                                                            \
        }                                                                                                                            \
        hipLaunchKernelGGL((conv_direct_kernel<NJ_, L_, MI_, KS_, MU_>), dim3((unsigned)grid), dim3(256), lds, stream, (const uint16_t*)x, (const uint16_t*)W, \
                           (uint16_t*)out, bias, d, x_bytes);                                                                        \
    } while (0)
#define ASR_CDL(NJ_, KS_)                                                                                                            \
    do {                                                                                                                             \
        if (Cs == 32) ASR_CD(NJ_, 2, 3, KS_, false); else if (Cs == 64) ASR_CD(NJ_, 3, 3, KS_, false);                                \
        else if (Cs == 128) ASR_CD(NJ_, 4, 2, KS_, false); else ASR_CD(NJ_, 4, 2, KS_, true);                                         \
    } while (0)
    if (narrow) { if (ks2) ASR_CDL(4, 2); else ASR_CDL(4, 1); } else ASR_CDL(8, 1);
#undef ASR_CDL
#undef ASR_CD
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
