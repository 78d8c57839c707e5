// NT GEMM, 256 x 256 tile, eight waves in two staggered groups: C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]), bf16 operands, K % 64 == 0.
//
// The one-barrier-per-K-step kernels of gemm.hip leave the matrix pipes 20 - 28 % busy (profiles/r04_pmc_sq.json): every K step a wave
// reads its fragments, issues its MFMAs and then waits for the next stage, and what hides that wait is only the other workgroups of the CU.
// This kernel is the "8-phase" schedule of the CDNA guide (cdna_hip_programming.md section 5), written for this library's conventions:
//
//  * ONE workgroup of 8 waves per CU: 2 (M) x 4 (N), a wave owns 128 x 64 of the tile = 8 x 4 MFMA tiles (128 accumulator registers).
//    Waves i and i + 4 share a SIMD; the two M-halves (wm = 0 / 1) run ONE BARRIER APART, so that on every SIMD one wave is in its
//    compute segment (16 MFMAs) while its partner is in its load segment (fragment reads + one LDS-DMA half tile): the matrix pipe of a
//    SIMD is handed from one wave to the other at every barrier.
//  * A K step of 64 is FOUR phases, one quadrant of the wave's tile each: (A half 0, B half 0), (A0, B1), (A1, B1), (A1, B0); a phase
//    reads only the half it has not yet in registers (12 / 4 / 8 / 0 ds_read_b128: B half 0 stays in registers for its second use).
//    phase := [fragment reads; LDS-DMA of one half tile; s_waitcnt vmcnt(8); s_barrier] [lgkmcnt(0); 16 MFMAs at raised priority; s_barrier]
//  * LDS: 2 (parity of the K step) x {A0, A1, B0, B1} x 16 KiB = 128 KiB.  A "half tile" is 128 rows x 64 k: for A the 64 rows of each
//    M-half's quadrant row block, for B the 32 columns of each of the four N-waves' quadrant column block -- the loader picks its source
//    rows freely (LDS-DMA: lane-linear destination, per-lane source), so LDS row rho of a half tile simply IS the row the reader wants.
//    Rows are 128 B; 16-B chunk c of row rho sits at position c ^ (rho & 7) (source-side swizzle, the same XOR on the read).
//  * Four half tiles (8 LDS-DMA instructions per thread) are always in flight: the half tile issued in phase p is read six (A0) or five
//    phases later, and `vmcnt(8)` at the end of every load segment retires exactly the half tile the NEXT phase reads -- never a
//    vmcnt(0) in the loop.  Issue order: ... A1(t+1) | A0(t+2) B0(t+2) B1(t+2) A1(t+2) | ... in phases (t,Q1) (t,Q2) (t,Q3) (t+1,Q0) (t+1,Q1).
//    Read-after-DMA: the wait that retires a half tile sits in the load segment BEFORE the one that reads it, in every wave, with a
//    barrier between (the guide's rule "read a staged buffer one phase after the wait that retires it"; with the two groups one barrier
//    apart every reader is still at least one barrier behind every waiter).  Write-after-read: a region is restaged two or three phases
//    after its last fragment read (two: the other group's reads of that phase are retired by its lgkmcnt(0) one barrier later).
//  * Beyond the last K step the loader re-fetches the last step into regions nobody reads any more: the loop has no tail variants and
//    the in-flight count stays what vmcnt(8) assumes.
//  * Epilogue from the accumulators: B's LDS rows are dealt such that lane (q, r) holds columns 4 r .. 4 r + 3 of rows 4 q + reg:
//    one 8-byte (bf16) or 16-byte (f32) store per row, 16 lanes = 128 / 256 contiguous bytes.
#include <type_traits>

#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace gemm8 {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
union Frag {
    bf16x8 v;
    uint4 u;
    u32x4 w;
};

// 16 B out of LDS by inline asm, NOT waited for here: hipcc puts s_waitcnt vmcnt(0) in front of every LDS read it can see in a kernel
// that also uses LDS-DMA (the DMA writes LDS and may alias), which would drain the four half tiles in flight at every phase.  The
// compute segment opens with s_waitcnt lgkmcnt(0) + sched_barrier(0) (an MFMA is register-only: only the scheduling barrier keeps it
// behind the wait).
template <int OFF>
__device__ __forceinline__ void lds_rd16(Frag& f, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.w) : "v"(addr), "n"(OFF) : "memory");
}

constexpr int HALF = 128 * 64 * 2;          // one half tile: 128 rows x 64 k of bf16 = 16 KiB
constexpr int LDS_BYTES = 8 * HALF;         // [2 parities][A0 A1 B0 B1]

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds, 16, voffset, soffset, 0, 0);
}

// Behind every raw buffer store of this file.  A 16-byte store reads its data registers over several cycles after issue, and a vector
// instruction that writes one of them must keep a distance (the ISA's "VMEM store of more than 64 bits -> write of its data VGPRs" wait
// states).  hipcc inserts those for global stores but NOT for a buffer store with an SGPR soffset (LLVM's hazard recogniser exempts that
// form), and the register allocator did put the next store's address into the first data register of the store in front of it:
// `buffer_store_dwordx4 v[64:67], ...` / `v_cndmask_b32 v64, ...` back to back stored the ADDRESS in lanes 12 - 15 of a row now and then
// (float32 logits, 400 of 24.6 M elements; found by exact-integer tests on an output pre-filled with NaN: tools/_dbg_f32map.py).
#define ASR8_STORE_FENCE()                          \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("s_nop 3" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0)
#define ASR8_LOAD_END() asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory")
#define ASR8_COMPUTE_BEGIN()                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);                       \
    __builtin_amdgcn_s_setprio(1)
#define ASR8_COMPUTE_END()                                   \
    __builtin_amdgcn_s_setprio(0);                           \
    __builtin_amdgcn_sched_barrier(0);                       \
    asm volatile("s_barrier" ::: "memory")

// KT: K is a multiple of 8 but not of 64 (the logit gradient's K = V = 3000): the chunks of the last K step that lie beyond K are
// fetched from beyond the buffers' ends, i.e. as zeros -- both operands.
// CONV: operand A is the activation tensor (Ts, B, Hs, Cs) of an implicit convolution (gemm.hip: ConvDesc): row r = (t, b, h) over the
// row space (Tr, B, Hr), column k = (kh, kw, c); A[r][k] = x[t + sgn (kw - pt)][b][h + sgn (kh - ph)][c], zero outside the tensor.
// Cs % 64 == 0: a K step lies inside ONE tap, so the tap walk is scalar state -- one per A half, each advanced by its own issues, which
// come in K order -- and a lane adds a uniform offset to its row's own address and checks two ranges; a row outside the tensor or an
// empty tap (K may be padded, and the loop's re-fetches beyond K walk on into empty taps) gets an offset beyond the buffer: zeros.
struct ConvDesc8 {
    int B, Hs, Cs, Ts, KH, KW, ph, pt, sgn, Hr;
};
template <typename OutT, bool KT, bool CONV>
__global__ __launch_bounds__(512, 2) void gemm_nt_8ph_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ B, int ldb,
                                                            OutT* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                            int tiles_m, int tiles_n, unsigned a_bytes, unsigned b_bytes, ConvDesc8 cd, int stagger) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Short K, several rounds of tiles per CU: every workgroup of a round reaches its 128 KB of stores at the same moment, the chip
    // alternates between "everybody computes" and "everybody stores", and the stores of a round take as long as its K loop.  A start
    // delay that differs between the workgroups of the FIRST round (stagger x 0 .. 7 units of ~1 us by workgroup id / 8 % 8; later
    // workgroups start when a CU comes free and inherit the spread) lets one CU's stores run beside the others' K loops.
    if (stagger > 0 && blockIdx.x < 256) {
        const int n = (int)((blockIdx.x >> 3) & 7) * stagger;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
    }
    // waves 4 (M) x 2 (N): a wave owns 64 rows x 128 columns = 4 x 8 MFMA tiles; waves i and i + 4 (one SIMD) are wm and wm + 2
    const int wm = wid >> 1, wn = wid & 1;
    // tiles: XCD x (= bid % 8 under round-robin dispatch; a locality hint only) takes a contiguous range of tile ids, so the column tiles of a
    // row panel of A meet in one L2.  The map is a bijection for any number of tiles.
    int wg;
    {
        const int nwg = tiles_m * tiles_n, qq = nwg >> 3, rr = nwg & 7, x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (x < rr ? x * (qq + 1) : rr * (qq + 1) + (x - rr) * qq) + idx;
    }
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = KT ? (K + 63) >> 6 : K >> 6;

    // ---- loader: slot s = i * 512 + tid of a half tile = (LDS row rho = s / 8, position s % 8), holding chunk (s % 8) ^ (rho % 8)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
    unsigned oa[2][2], ob[2][2];            // [half][i]: byte offset of this thread's source chunk at k = 0
    int kc[2];                              // KT: first k of the slot's chunk inside a K step
    int cth[2][2];                          // CONV: (t << 8) | h of the slot's row; rows beyond M get a t far below zero
    int tw_kh[2] = {0, 0}, tw_kw[2] = {0, 0}, tw_ci[2] = {0, 0};      // CONV: tap of the next K step each A half issues
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = i * 512 + tid, rho = s >> 3, c = (s & 7) ^ (rho & 7);
        kc[i] = c * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // A half h: LDS row rho = wm' * 32 + rr  <-  row m0 + wm' * 64 + h * 32 + rr
            const int grow = m0 + (rho >> 5) * 64 + h * 32 + (rho & 31);
            if (CONV) {
                const int rc = grow < M ? grow : 0;
                const int hh = rc % cd.Hr, tb = rc / cd.Hr, bb = tb % cd.B, tt = tb / cd.B;
                cth[h][i] = (grow < M ? tt : -(1 << 20)) * 256 + hh;
                oa[h][i] = (unsigned)((tt * cd.B + bb) * cd.Hs + hh) * (unsigned)(cd.Cs * 2) + (unsigned)(c * 16);
            } else {
                oa[h][i] = (unsigned)min(grow, M - 1) * (unsigned)(lda * 2) + (unsigned)(c * 16);
            }
            // B half h: LDS row rho = wn' * 64 + j * 16 + r  <-  column n0 + wn' * 128 + 8 r + (4 h + j): a lane's eight accumulator columns
            // are 16 contiguous bytes of bf16.  Float32 output: column n0 + wn' * 128 + 64 h + 4 r + j instead -- a lane's four columns of
            // one half are 16 contiguous bytes and the 16 lanes of a row 256 contiguous bytes per store instruction (with eight columns
            // per lane a 16-byte store covered every other 16 bytes of 512: two half-dense instructions per row)
            const int gcol = sizeof(OutT) == 4 ? n0 + (rho >> 6) * 128 + 64 * h + 4 * (rho & 15) + ((rho >> 4) & 3)
                                               : n0 + (rho >> 6) * 128 + 8 * (rho & 15) + 4 * h + ((rho >> 4) & 3);
            ob[h][i] = (unsigned)min(gcol, N - 1) * (unsigned)(ldb * 2) + (unsigned)(c * 16);
        }
    }
    // KIND: 0 A0, 1 A1, 2 B0, 3 B1; region (PAR, KIND) at (PAR * 4 + KIND) * HALF
    auto issue = [&](auto kind_c, auto par_c, int kt) {
        constexpr int KIND = decltype(kind_c)::value, PAR = decltype(par_c)::value;
        char* base = smem + (PAR * 4 + KIND) * HALF + wid * 1024;
        const int so = kt * 128;
        const bool d0 = KT && kt * 64 + kc[0] >= K, d1 = KT && kt * 64 + kc[1] >= K;
        if (KIND < 2 && CONV) {
            constexpr int H = KIND & 1;
            const int dt = cd.sgn * (tw_kw[H] - cd.pt), dh = cd.sgn * (tw_kh[H] - cd.ph);
            const int delta = ((dt * cd.B * cd.Hs + dh) * cd.Cs + tw_ci[H]) * 2;
            const bool tap_ok = tw_kh[H] < cd.KH;
            tw_ci[H] += 64;
            if (tw_ci[H] >= cd.Cs) { tw_ci[H] = 0; if (++tw_kw[H] == cd.KW) { tw_kw[H] = 0; ++tw_kh[H]; } }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ti = (cth[H][i] >> 8) + dt, hi = (cth[H][i] & 255) + dh;
                const bool ok = tap_ok && (unsigned)ti < (unsigned)cd.Ts && (unsigned)hi < (unsigned)cd.Hs;
                lds_dma16(rsrc_a, base + i * 8192, ok ? oa[H][i] + (unsigned)delta : 0xfffffff0u, 0);
            }
        } else if (KIND < 2) {
            lds_dma16(rsrc_a, base, d0 ? 0xfffffff0u : oa[KIND & 1][0], so);
            lds_dma16(rsrc_a, base + 8192, d1 ? 0xfffffff0u : oa[KIND & 1][1], so);
        } else {
            lds_dma16(rsrc_b, base, d0 ? 0xfffffff0u : ob[KIND & 1][0], so);
            lds_dma16(rsrc_b, base + 8192, d1 ? 0xfffffff0u : ob[KIND & 1][1], so);
        }
    };
#define ASR8_ISSUE(KIND, PAR, KT_) issue(std::integral_constant<int, KIND>{}, std::integral_constant<int, PAR>{}, (KT_))

    // ---- reader: lane (q, r) reads row (block base) + r, chunk (4 ks + q) ^ (r % 8); the region's kind and the tile index are the
    // instruction's immediate offset (< 64 KiB), the parity is in the address register
    const int q = lane >> 4, r = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned al0 = lds0 + (unsigned)((wm * 32 + r) * 128 + ((q ^ (r & 7)) << 4)), al1 = al0 ^ 64u;
    const unsigned bl0 = lds0 + (unsigned)((wn * 64 + r) * 128 + ((q ^ (r & 7)) << 4)), bl1 = bl0 ^ 64u;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag a[2][2], b0[4][2], b1[4][2];

    auto read_a = [&](auto par_c, auto h_c) {
        constexpr int PAR = decltype(par_c)::value, H = decltype(h_c)::value;
        const unsigned p0 = al0 + PAR * 4 * HALF, p1 = al1 + PAR * 4 * HALF;
        lds_rd16<H * HALF + 0 * 2048>(a[0][0], p0); lds_rd16<H * HALF + 0 * 2048>(a[0][1], p1);
        lds_rd16<H * HALF + 1 * 2048>(a[1][0], p0); lds_rd16<H * HALF + 1 * 2048>(a[1][1], p1);
    };
    auto read_b = [&](auto par_c, auto h_c, Frag (&b)[4][2]) {
        constexpr int PAR = decltype(par_c)::value, H = decltype(h_c)::value;
        const unsigned p0 = bl0 + PAR * 4 * HALF, p1 = bl1 + PAR * 4 * HALF;
        lds_rd16<(2 + H) * HALF + 0 * 2048>(b[0][0], p0); lds_rd16<(2 + H) * HALF + 0 * 2048>(b[0][1], p1);
        lds_rd16<(2 + H) * HALF + 1 * 2048>(b[1][0], p0); lds_rd16<(2 + H) * HALF + 1 * 2048>(b[1][1], p1);
        lds_rd16<(2 + H) * HALF + 2 * 2048>(b[2][0], p0); lds_rd16<(2 + H) * HALF + 2 * 2048>(b[2][1], p1);
        lds_rd16<(2 + H) * HALF + 3 * 2048>(b[3][0], p0); lds_rd16<(2 + H) * HALF + 3 * 2048>(b[3][1], p1);
    };
    // one quadrant: row tiles I0, I0 + 1, column tiles J0 .. J0 + 3, K = 64
    auto quadrant = [&](auto i0_c, auto j0_c, const Frag (&b)[4][2]) {
        constexpr int I0 = decltype(i0_c)::value, J0 = decltype(j0_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[I0 + i][J0 + j] = ASR_MFMA_16x16x32(a[i][ks].v, b[j][ks].v, acc[I0 + i][J0 + j]);
    };
#define ASR8_C(V) std::integral_constant<int, V>{}

    // the four phases of K step t (parity PAR of its LDS regions)
    auto kstep = [&](auto par_c, int t) {
        constexpr int PAR = decltype(par_c)::value;
        const int t1 = min(t + 1, nk - 1), t2 = min(t + 2, nk - 1);
        // (A0, B0)
        read_a(ASR8_C(PAR), ASR8_C(0));
        __builtin_amdgcn_sched_barrier(0);
        read_b(ASR8_C(PAR), ASR8_C(0), b0);
        ASR8_ISSUE(3, PAR ^ 1, t1);
        ASR8_LOAD_END();
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(0), ASR8_C(0), b0);
        ASR8_COMPUTE_END();
        // (A0, B1)
        read_b(ASR8_C(PAR), ASR8_C(1), b1);
        ASR8_ISSUE(1, PAR ^ 1, t1);
        ASR8_LOAD_END();
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(0), ASR8_C(4), b1);
        ASR8_COMPUTE_END();
        // (A1, B1)
        read_a(ASR8_C(PAR), ASR8_C(1));
        ASR8_ISSUE(0, PAR, t2);
        ASR8_LOAD_END();
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(2), ASR8_C(4), b1);
        ASR8_COMPUTE_END();
        // (A1, B0)
        ASR8_ISSUE(2, PAR, t2);
        ASR8_LOAD_END();
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(2), ASR8_C(0), b0);
        ASR8_COMPUTE_END();
    };

    // prologue: K step 0 whole, A0 and B0 of step 1 -- six half tiles; the first phase issues the seventh and retires the third
    {
        const int t1 = min(1, nk - 1);
        ASR8_ISSUE(0, 0, 0);
        ASR8_ISSUE(2, 0, 0);
        ASR8_ISSUE(3, 0, 0);
        ASR8_ISSUE(1, 0, 0);
        ASR8_ISSUE(0, 1, t1);
        ASR8_ISSUE(2, 1, t1);
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    }
    const int grp = wm >> 1;                                    // waves i and i + 4 share a SIMD: the halves wm < 2 / wm >= 2
    if (grp == 1) asm volatile("s_barrier" ::: "memory");       // the second half runs one barrier behind the first
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        kstep(ASR8_C(0), t);
        kstep(ASR8_C(1), t + 1);
    }
    if (t < nk) kstep(ASR8_C(0), t);
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the re-fetches beyond K still write LDS: drained before the workgroup leaves)

    if (stagger < 0) {          // what-if (ASR_DEBUG nt8_stagger=-1: timing only, results invalid): no stores except one dword per wave
        if (lane == 0) reinterpret_cast<float*>(C)[(size_t)blockIdx.x * 8 + wid] = acc[0][0][0] + acc[3][7][3];
        return;
    }
    // ---- epilogue: acc[I][J][reg] = C[m0 + wm * 64 + I * 16 + 4 q + reg][n0 + wn * 128 + 8 r + J]: 16 bytes of bf16 (32 of f32) per lane
    // and row, the 16 lanes of a row 256 (512) contiguous bytes; 16 store instructions per lane (the store tail is issue-bound: 8-byte
    // stores, 32 per lane, took as long as the eight K steps of a K = 512 product)
    constexpr bool F32MAP = sizeof(OutT) == 4;                  // float32: columns J = 4 .. 7 of a lane lie 64 columns behind J = 0 .. 3
    constexpr int SECOND = F32MAP ? 64 : 4;
    const int col = n0 + wn * 128 + (F32MAP ? 4 : 8) * r;
    if (col >= N) return;
    float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool full = col + SECOND + 3 < N;                     // (N % 4 == 0: a lane's eight columns exist as 8 or as the first 4)
    if (bias) {
        const float4 ba = *reinterpret_cast<const float4*>(bias + col);
        bv[0] = ba.x; bv[1] = ba.y; bv[2] = ba.z; bv[3] = ba.w;
        if (full) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + col + SECOND);
            bv[4] = bb.x; bv[5] = bb.y; bv[6] = bb.z; bv[7] = bb.w;
        }
    }
    const int row0 = m0 + wm * 64 + 4 * q;
    const bool c16 = (ldc & 7) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + i * 16 + reg;
            if (row >= M) continue;
            OutT* dst = C + (size_t)row * ldc + col;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = acc[i][j][reg] + bv[j];
            if (sizeof(OutT) == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                if (full) *reinterpret_cast<float4*>(dst + SECOND) = make_float4(v[4], v[5], v[6], v[7]);
            } else if (full && c16) {
                uint4 pk;
                pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
                *reinterpret_cast<uint4*>(dst) = pk;
            } else if (full) {      // (a row pitch of 4 (mod 8) columns: rows start 8 bytes off a 16-byte boundary)
                uint2 pk, pl;
                pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                pl.x = pack_bf16x2(v[4], v[5]); pl.y = pack_bf16x2(v[6], v[7]);
                *reinterpret_cast<uint2*>(dst) = pk;
                *reinterpret_cast<uint2*>(dst + 4) = pl;
            } else {
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(dst) = pk;
            }
        }
}

// ------------------------------------------------------------------------------------------------ NT, persistent: short K, many tiles
// With K = 512 a tile's K loop is eight steps (11 us) and what surrounds it costs as much again (tools/nt8_k_sweep.py: ~45 us of a 122 us
// product): the workgroup's 128 KB of output leave after its K loop, the workgroup may only end when the last store is acknowledged, and
// its successor on the CU starts with an empty pipeline.  This form keeps ONE workgroup per CU for the whole product:
//  * the K steps of its tiles are ONE stream: the loader walks on into the next tile's operands while the last steps of the current one
//    are computed (where the one-tile kernel re-fetches the last step as a dummy);
//  * a finished tile leaves IN PIECES during the first K step of the next one, beside the partner group's MFMAs on the same SIMD (a bulk
//    epilogue between the tiles cost 3.8 us per tile in vector instructions alone, what-if in profiles/r05_gemm8_ab.txt): the load segment
//    of phase 0 converts and stores the wave's rows I = 0, 1 (quadrants 0 and 1 rewrite them in phases 0 and 1), that of phase 2 its rows
//    I = 2, 3 -- whole 16-byte pieces of a row per lane, 256 (512) contiguous bytes per 16 lanes: quadrant-wise 8-byte pieces, the two
//    halves of a 16-byte chunk written phases apart, ran SLOWER than the one-tile kernel (partial lines leave L2 before their other half
//    arrives) -- and the load segment of phase p re-initialises quadrant p;
//  * the bias is the accumulators' INITIAL value (read from the 32 KiB of LDS the operand ring leaves free, by inline asm like the
//    fragments: a vector load in the loop would make hipcc drain the ring), so a piece is a conversion and a store;
//  * `vmcnt` counts loads and stores together, in order.  The schedule's rule -- at the end of a load segment everything but the four
//    youngest half tiles is complete -- becomes vmcnt(8 + stores issued since the fourth-youngest half tile): with 8 (bf16) stores in
//    phases 0 and 2: 16, 16, 24, 16 in the K step that carries the pieces and 16, 8, 8, 8 in the one after it (f32, 16 stores: 24, 24, 40,
//    24 | 24, 8, 8, 8).  A piece's stores come BEFORE its segment's LDS-DMA instructions and are always the same number of instructions:
//    rows and columns that do not exist get an offset beyond the buffer, never a branch.  A piece's stores must be acknowledged three
//    phases after they were issued; that wait is what is left of the store tail.
// K % 64 == 0, K >= 128 (an odd K / 64 is padded with a K step of zeros: parity and mode of a K step are compile-time constants),
// N <= 8192 (bf16 output: N % 8 == 0, ldc % 8 == 0), C below 4 GiB.  Tiles: XCD x owns a contiguous range of tile ids (column tiles of a row panel meet in
// one L2); its S = gridDim / 8 workgroups take tiles beg + slot, beg + slot + S, ...
#define ASR8P_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")
// (MODE is a compile-time constant: a K step with run-time tests in its load segments lost 6 us of a 104 us product to them)
#define ASR8P_LOAD_END(H1, H2, F1, F2)                                                          \
    do {                                                                                        \
        if (MODE == 0) ASR8P_WAIT(8);                                                           \
        else if (MODE == 1) { if (ES == 4) ASR8P_WAIT(F1); else ASR8P_WAIT(H1); }               \
        else { if (ES == 4) ASR8P_WAIT(F2); else ASR8P_WAIT(H2); }                              \
    } while (0)
template <typename OutT>
__global__ __launch_bounds__(512, 2) void gemm_nt_8pp_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ B, int ldb,
                                                            OutT* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                            int tiles_m, int tiles_n, unsigned a_bytes, unsigned b_bytes, unsigned c_bytes,
                                                            int whatif) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = (int)sizeof(OutT);
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int nk = ((K >> 6) + 1) & ~1;         // an even number of K steps per tile; an odd K / 64 gets one more, fetched as zeros
    int tile, tile_end;
    const int S = (int)(gridDim.x >> 3);
    {
        const int nwg = tiles_m * tiles_n, qq = nwg >> 3, rr = nwg & 7, x = blockIdx.x & 7;
        const int beg = x < rr ? x * (qq + 1) : rr * (qq + 1) + (x - rr) * qq;
        tile_end = beg + (x < rr ? qq + 1 : qq);
        tile = beg + (int)(blockIdx.x >> 3);
    }
    if (tile >= tile_end) return;
    float* sbias = reinterpret_cast<float*>(smem + LDS_BYTES);
    for (int i = tid; i < N; i += 512) sbias[i] = bias ? bias[i] : 0.f;
    __syncthreads();

    // ---- loader (slots as in gemm_nt_8ph_kernel): offsets of the current tile and of the workgroup's next one
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (int)c_bytes, 0x00020000);
    // a slot's source offset = (its own row / column + chunk: one register per operand) + (tile, half, i: scalar), clamped to the
    // operand's last row / column per issue: no per-tile vector state, the workgroup's current and next tile are four scalars
    const int c16 = (((tid & 7) ^ ((tid >> 3) & 7)) << 4);
    const unsigned arow_off = (unsigned)(((tid >> 8) * 64 + ((tid >> 3) & 31)) * (lda * 2) + c16);         // rows + i * 128 + h * 32
    constexpr bool F32MAP = ES == 4;        // float32 output: the column map of gemm_nt_8ph_kernel's float32 form (dense 16-byte stores)
    constexpr int BH = F32MAP ? 64 : 4, BR = F32MAP ? 4 : 8;
    const unsigned bcol_off = (unsigned)((BR * ((tid >> 3) & 15) + ((tid >> 7) & 3)) * (ldb * 2) + c16);    // columns + i * 128 + BH h
    const unsigned amax = (unsigned)(M - 1) * (unsigned)(lda * 2) + (unsigned)c16, bmax = (unsigned)(N - 1) * (unsigned)(ldb * 2) + (unsigned)c16;
    int ctm = tile / tiles_n, ctn = tile - ctm * tiles_n;                 // current tile
    int xtm, xtn;                                                         // the next one (the current one again when there is none: a dummy re-fetch)
    auto next_coords = [&]() {
        const int nt = tile + S < tile_end ? tile + S : tile;
        xtm = nt / tiles_n;
        xtn = nt - xtm * tiles_n;
    };
    next_coords();
    // kt counts from the current tile's first K step: kt >= nk is K step kt - nk of the next tile (nk >= 2: never beyond that one)
    auto issue = [&](auto kind_c, auto par_c, int kt) {
        constexpr int KIND = decltype(kind_c)::value, PAR = decltype(par_c)::value, H = KIND & 1;
        char* base = smem + (PAR * 4 + KIND) * HALF + wid * 1024;
        const bool nx = kt >= nk;
        const int ks = nx ? kt - nk : kt;
        const int so = ks * 128;
        const bool pad = ks * 64 >= K;              // (uniform) the padding K step: beyond the buffer, i.e. zeros
        if (KIND < 2) {
            const unsigned s0 = (unsigned)(((nx ? xtm : ctm) * 256 + H * 32) * (lda * 2)), s1 = s0 + (unsigned)(128 * lda * 2);
            lds_dma16(rsrc_a, base, pad ? 0xfffffff0u : min(arow_off + s0, amax), so);
            lds_dma16(rsrc_a, base + 8192, pad ? 0xfffffff0u : min(arow_off + s1, amax), so);
        } else {
            const unsigned s0 = (unsigned)(((nx ? xtn : ctn) * 256 + BH * H) * (ldb * 2)), s1 = s0 + (unsigned)(128 * ldb * 2);
            lds_dma16(rsrc_b, base, pad ? 0xfffffff0u : min(bcol_off + s0, bmax), so);
            lds_dma16(rsrc_b, base + 8192, pad ? 0xfffffff0u : min(bcol_off + s1, bmax), so);
        }
    };

    const int q = lane >> 4, r = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned al0 = lds0 + (unsigned)((wm * 32 + r) * 128 + ((q ^ (r & 7)) << 4)), al1 = al0 ^ 64u;
    const unsigned bl0 = lds0 + (unsigned)((wn * 64 + r) * 128 + ((q ^ (r & 7)) << 4)), bl1 = bl0 ^ 64u;
    const unsigned sb0 = lds0 + (unsigned)LDS_BYTES;

    f32x4 acc[4][8];
    Frag a[2][2], b0[4][2], b1[4][2];
    auto read_a = [&](auto par_c, auto h_c) {
        constexpr int PAR = decltype(par_c)::value, H = decltype(h_c)::value;
        const unsigned p0 = al0 + PAR * 4 * HALF, p1 = al1 + PAR * 4 * HALF;
        lds_rd16<H * HALF + 0 * 2048>(a[0][0], p0); lds_rd16<H * HALF + 0 * 2048>(a[0][1], p1);
        lds_rd16<H * HALF + 1 * 2048>(a[1][0], p0); lds_rd16<H * HALF + 1 * 2048>(a[1][1], p1);
    };
    auto read_b = [&](auto par_c, auto h_c, Frag (&b)[4][2]) {
        constexpr int PAR = decltype(par_c)::value, H = decltype(h_c)::value;
        const unsigned p0 = bl0 + PAR * 4 * HALF, p1 = bl1 + PAR * 4 * HALF;
        lds_rd16<(2 + H) * HALF + 0 * 2048>(b[0][0], p0); lds_rd16<(2 + H) * HALF + 0 * 2048>(b[0][1], p1);
        lds_rd16<(2 + H) * HALF + 1 * 2048>(b[1][0], p0); lds_rd16<(2 + H) * HALF + 1 * 2048>(b[1][1], p1);
        lds_rd16<(2 + H) * HALF + 2 * 2048>(b[2][0], p0); lds_rd16<(2 + H) * HALF + 2 * 2048>(b[2][1], p1);
        lds_rd16<(2 + H) * HALF + 3 * 2048>(b[3][0], p0); lds_rd16<(2 + H) * HALF + 3 * 2048>(b[3][1], p1);
    };
    auto quadrant = [&](auto i0_c, auto j0_c, const Frag (&b)[4][2]) {
        constexpr int I0 = decltype(i0_c)::value, J0 = decltype(j0_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[I0 + i][J0 + j] = ASR_MFMA_16x16x32(a[i][ks].v, b[j][ks].v, acc[I0 + i][J0 + j]);
    };

    // ---- pieces.  acc[I][J][reg] = C[tm 256 + wm 64 + I 16 + 4 q + reg][tn 256 + wn 128 + 8 r + J], the bias already inside.
    // Quadrant (I0, J0) = rows I0, I0 + 1 x columns J0 .. J0 + 3: a lane's four columns are 8 (16) contiguous bytes.  The store's scalar
    // offset carries the tile and the row (the range check of a raw buffer access sees only the vector offset + immediate: a lane whose
    // row or columns do not exist gets a vector offset beyond the buffer).
    int ptm = 0, ptn = 0;                                       // the tile the pieces belong to
    int ntn = 0;                                                // column tile of the tile being computed (bias of a re-initialised quadrant)
    const unsigned vc = (unsigned)((4 * q * ldc + wn * 128 + BR * r) * ES);
    auto bias_read = [&](auto j0_c, f32x4& bv) {
        constexpr int J0 = decltype(j0_c)::value;
        const unsigned ad = sb0 + (unsigned)(min(ntn * 256 + wn * 128 + BR * r + (J0 ? BH : 0), N - 4) * 4);
        asm volatile("ds_read_b128 %0, %1" : "=v"(bv) : "v"(ad) : "memory");
    };
    // rows I0, I0 + 1 (x 4 q + reg) of the tile the pieces belong to, all eight columns of the lane: 8 (bf16: 16 bytes per row) or 16
    // (f32: two 16-byte halves) store instructions, never fewer
    auto rows_store = [&](auto i0_c) {
        constexpr int I0 = decltype(i0_c)::value;
        const int rlim = M - (ptm * 256 + wm * 64), clim = N - (ptn * 256 + wn * 128);
        const bool cok0 = BR * r < clim, cok1 = BR * r + BH < clim;
        const unsigned sbase = ((unsigned)(ptm * 256 + wm * 64) * (unsigned)ldc + (unsigned)(ptn * 256)) * (unsigned)ES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int rr_ = (I0 + i) * 16 + reg;
                const bool rok = (4 * q < rlim - rr_) & (whatif != 1);      // (scalar right side: no per-row lane constants)
                const unsigned so = sbase + (unsigned)rr_ * (unsigned)ldc * (unsigned)ES;
                const f32x4 v0 = (f32x4){acc[I0 + i][0][reg], acc[I0 + i][1][reg], acc[I0 + i][2][reg], acc[I0 + i][3][reg]};
                const f32x4 v1 = (f32x4){acc[I0 + i][4][reg], acc[I0 + i][5][reg], acc[I0 + i][6][reg], acc[I0 + i][7][reg]};
                if (ES == 4) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rsrc_c, (int)((rok & cok0) ? vc : 0xfffffff0u), (int)so, 0);
                    ASR8_STORE_FENCE();
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), rsrc_c, (int)((rok & cok1) ? vc + (unsigned)(BH * ES) : 0xfffffff0u), (int)so, 0);
                    ASR8_STORE_FENCE();
                } else {            // (N % 8 == 0 here: a lane's eight columns exist together)
                    const u32x4 pk = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc_c, (int)((rok & cok0) ? vc : 0xfffffff0u), (int)so, 0);
                    ASR8_STORE_FENCE();
                }
            }
    };
    auto piece_init = [&](auto i0_c, auto j0_c, const f32x4& bv) {
        constexpr int I0 = decltype(i0_c)::value, J0 = decltype(j0_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[I0 + i][J0 + jj] = (f32x4){bv[jj], bv[jj], bv[jj], bv[jj]};
    };
    // phase p of the K step after a tile's end: quadrant p starts again from the new tile's bias; before that the old tile's rows leave --
    // rows I = 0, 1 in phase 0 (quadrants 0 and 1 rewrite them in phases 0 and 1), rows I = 2, 3 in phase 2 (quadrants 2 and 3)
#define ASR8P_PIECE(I0, J0, STORE)                                       \
    if (MODE == 1) {                                                     \
        f32x4 bv_;                                                       \
        bias_read(ASR8_C(J0), bv_);                                      \
        if (STORE) rows_store(ASR8_C(I0));                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               \
        __builtin_amdgcn_sched_barrier(0);                               \
        piece_init(ASR8_C(I0), ASR8_C(J0), bv_);                         \
        __builtin_amdgcn_sched_barrier(0);                               \
    }
    // MODE 1: the first K step of a tile (pieces), 2: the second (the waits still count the pieces' stores), 0: any other
    auto kstep = [&](auto par_c, auto mode_c, int t) {
        constexpr int PAR = decltype(par_c)::value, MODE = decltype(mode_c)::value;
        // (A0, B0)
        read_a(ASR8_C(PAR), ASR8_C(0));
        __builtin_amdgcn_sched_barrier(0);
        read_b(ASR8_C(PAR), ASR8_C(0), b0);
        ASR8P_PIECE(0, 0, true)
        ASR8_ISSUE(3, PAR ^ 1, t + 1);
        ASR8P_LOAD_END(16, 16, 24, 24);
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(0), ASR8_C(0), b0);
        ASR8_COMPUTE_END();
        // (A0, B1)
        read_b(ASR8_C(PAR), ASR8_C(1), b1);
        ASR8P_PIECE(0, 4, false)
        ASR8_ISSUE(1, PAR ^ 1, t + 1);
        ASR8P_LOAD_END(16, 8, 24, 8);
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(0), ASR8_C(4), b1);
        ASR8_COMPUTE_END();
        // (A1, B1)
        read_a(ASR8_C(PAR), ASR8_C(1));
        ASR8P_PIECE(2, 4, true)
        ASR8_ISSUE(0, PAR, t + 2);
        ASR8P_LOAD_END(24, 8, 40, 8);
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(2), ASR8_C(4), b1);
        ASR8_COMPUTE_END();
        // (A1, B0)
        ASR8P_PIECE(2, 0, false)
        ASR8_ISSUE(2, PAR, t + 2);
        ASR8P_LOAD_END(16, 8, 24, 8);
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(2), ASR8_C(0), b0);
        ASR8_COMPUTE_END();
    };

    // The workgroup's first tile is not special: it "follows" a tile below the matrix's last row, whose pieces are stores beyond the buffer
    // (dropped by the range check, counted by vmcnt like any other), and takes its bias in its first K step like every tile.
    ntn = ctn;
    ptm = tiles_m;
    ptn = 0;
    {   // prologue: K step 0 whole, A0 and B0 of step 1
        ASR8_ISSUE(0, 0, 0);
        ASR8_ISSUE(2, 0, 0);
        ASR8_ISSUE(3, 0, 0);
        ASR8_ISSUE(1, 0, 0);
        ASR8_ISSUE(0, 1, 1);
        ASR8_ISSUE(2, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    }
    const int grp = wm >> 1;
    if (grp == 1) asm volatile("s_barrier" ::: "memory");
    // a tile's end (between two K steps; no barrier in here, the two groups pass it one barrier apart): the tile becomes the one its
    // pieces belong to, the next tile's offsets move up, the one after it is worked out.  false: that was the workgroup's last tile.
    auto tile_end_work = [&]() -> bool {
        ptm = ctm;
        ptn = ctn;
        tile += S;
        if (tile >= tile_end) return false;
        ctm = xtm;
        ctn = xtn;
        ntn = ctn;
        next_coords();
        return true;
    };
    // an EVEN number of K steps per tile: the LDS parity of a K step is then a compile-time constant as well
    for (;;) {
        kstep(ASR8_C(0), ASR8_C(1), 0);
        kstep(ASR8_C(1), ASR8_C(2), 1);
        for (int t = 2; t < nk; t += 2) {
            kstep(ASR8_C(0), ASR8_C(0), t);
            kstep(ASR8_C(1), ASR8_C(0), t + 1);
        }
        if (!tile_end_work()) break;
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
    // the last tile leaves at once; the zero fills of the tile that does not exist still write LDS: drained before the workgroup ends
    rows_store(ASR8_C(0));
    rows_store(ASR8_C(2));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#undef ASR8P_PIECE
#undef ASR8P_LOAD_END
#undef ASR8P_WAIT

// ------------------------------------------------------------------------------------------------ NT, narrow: N <= 128
// Products with few output columns (backward-data of a convolution into 64 or 128 input channels: M = T B H rows, K = taps x channels)
// waste three quarters of a 256 x 256 tile.  Tile 256 x TNW (64 or 128), waves 8 (M) x 1 (N): a wave owns 32 rows x all columns, the two
// groups of four waves (waves i and i + 4 share a SIMD) again run one barrier apart.  A K step of 64 is ONE phase at TNW = 64 (12 fragment
// reads, 16 MFMAs) and TWO at TNW = 128 (the B halves; A stays in registers).  Such a tile is bound by operand fill (40 KB per 128 MFMAs at
// TNW = 64), so the ring is as deep as LDS allows: TNW = 64: four stages of 40 KiB (all 160 KiB), a K step is issued two phases ahead and
// restaged two phases after its last read; TNW = 128: three stages of 48 KiB, half a K step (A rows 0 .. 127 + B half 0, then the rest)
// issued per phase, two K steps ahead.  `vmcnt` leaves exactly the last two phases' issues in flight (5 resp. 6 instructions per thread).
template <typename OutT, bool CONV, int TNW>
__global__ __launch_bounds__(512, 2) void gemm_nt_8pn_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ B, int ldb,
                                                            OutT* __restrict__ C, int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                            int tiles_m, unsigned a_bytes, unsigned b_bytes, ConvDesc8 cd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NJ = TNW / 16;                              // column tiles of a wave
    constexpr int A_BYTES = 256 * 128, B_BYTES = TNW * 128, STAGE = A_BYTES + B_BYTES, NSTAGE = TNW == 64 ? 4 : 3;
    constexpr int BG = TNW / 64;                              // LDS-DMA instructions per thread for a K step of B
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int wg;
    {
        const int nwg = tiles_m, qq = nwg >> 3, rr = nwg & 7, x = blockIdx.x & 7, idx = blockIdx.x >> 3;
        wg = (x < rr ? x * (qq + 1) : rr * (qq + 1) + (x - rr) * qq) + idx;      // contiguous row panels per XCD (the taps' re-reads stay in one L2)
    }
    const int m0 = wg * 256;
    const int nk = K >> 6;

    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)b_bytes, 0x00020000);
    // loader: A slot s = i * 512 + tid (i < 4): LDS row s / 8 = tile row, position s % 8 holds chunk (s % 8) ^ (row % 8);
    // B slot s = i * 512 + tid (i < BG): LDS row rho = j * 16 + r  <-  column G r + j % G + (j / G) 16 G, G = min(NJ, 8) columns per lane run
    unsigned oa[4], ob[BG];
    int cth[4];
    int tw_kh = 0, tw_kw = 0, tw_ci = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = i * 512 + tid, rho = s >> 3, c = (s & 7) ^ (rho & 7);
        const int grow = m0 + rho;
        if (CONV) {
            const int rc = grow < M ? grow : 0;
            const int hh = rc % cd.Hr, tb = rc / cd.Hr, bb = tb % cd.B, tt = tb / cd.B;
            cth[i] = (grow < M ? tt : -(1 << 20)) * 256 + hh;
            oa[i] = (unsigned)((tt * cd.B + bb) * cd.Hs + hh) * (unsigned)(cd.Cs * 2) + (unsigned)(c * 16);
        } else {
            cth[i] = 0;
            oa[i] = (unsigned)min(grow, M - 1) * (unsigned)(lda * 2) + (unsigned)(c * 16);
        }
    }
    constexpr int G = NJ < 8 ? NJ : 8;
#pragma unroll
    for (int i = 0; i < BG; ++i) {
        const int s = i * 512 + tid, rho = s >> 3, c = (s & 7) ^ (rho & 7);
        const int j = rho >> 4, r_ = rho & 15;
        const int gcol = G * r_ + (j % G) + (j / G) * (16 * G);
        ob[i] = (unsigned)min(gcol, N - 1) * (unsigned)(ldb * 2) + (unsigned)(c * 16);
    }
    // A part `part` (TNW = 128: rows 128 part .. + 127 = slots 2 part, 2 part + 1; TNW = 64: part < 0 = all four slots) of the NEXT K step in
    // K order; CONV: the tap state advances when the K step's last part has been issued
    auto issue_a = [&](int stage, int part, bool last_part, int kt) {
        char* base = smem + stage * STAGE + wid * 1024;
        int dt = 0, dh = 0, delta = 0;
        bool tap_ok = true;
        if (CONV) {
            dt = cd.sgn * (tw_kw - cd.pt); dh = cd.sgn * (tw_kh - cd.ph);
            delta = ((dt * cd.B * cd.Hs + dh) * cd.Cs + tw_ci) * 2;
            tap_ok = tw_kh < cd.KH;
            if (last_part) { tw_ci += 64; if (tw_ci >= cd.Cs) { tw_ci = 0; if (++tw_kw == cd.KW) { tw_kw = 0; ++tw_kh; } } }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (part >= 0 && (i >> 1) != part) continue;
            if (CONV) {
                const int ti = (cth[i] >> 8) + dt, hi = (cth[i] & 255) + dh;
                const bool ok = tap_ok && (unsigned)ti < (unsigned)cd.Ts && (unsigned)hi < (unsigned)cd.Hs;
                lds_dma16(rsrc_a, base + i * 8192, ok ? oa[i] + (unsigned)delta : 0xfffffff0u, 0);
            } else {
                lds_dma16(rsrc_a, base + i * 8192, oa[i], kt * 128);
            }
        }
    };
    auto issue_b = [&](int stage, int half, int kt) {          // half < 0: all of B (TNW = 64)
        char* base = smem + stage * STAGE + A_BYTES + wid * 1024;
#pragma unroll
        for (int i = 0; i < BG; ++i)
            if (half < 0 || i == half) lds_dma16(rsrc_b, base + i * 8192, ob[i], kt * 128);
    };

    const int q = lane >> 4, r = lane & 15;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned al0 = lds0 + (unsigned)((wid * 32 + r) * 128 + ((q ^ (r & 7)) << 4)), al1 = al0 ^ 64u;
    const unsigned bl0 = lds0 + (unsigned)(A_BYTES + r * 128 + ((q ^ (r & 7)) << 4)), bl1 = bl0 ^ 64u;

    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag a[2][2], b[4][2];
    auto read_a = [&](unsigned so) {
        lds_rd16<0>(a[0][0], al0 + so); lds_rd16<0>(a[0][1], al1 + so);
        lds_rd16<2048>(a[1][0], al0 + so); lds_rd16<2048>(a[1][1], al1 + so);
    };
    auto read_b = [&](unsigned so, auto half_c) {              // column tiles 4 half .. 4 half + 3
        constexpr int HO = decltype(half_c)::value * 8192;
        lds_rd16<HO + 0 * 2048>(b[0][0], bl0 + so); lds_rd16<HO + 0 * 2048>(b[0][1], bl1 + so);
        lds_rd16<HO + 1 * 2048>(b[1][0], bl0 + so); lds_rd16<HO + 1 * 2048>(b[1][1], bl1 + so);
        lds_rd16<HO + 2 * 2048>(b[2][0], bl0 + so); lds_rd16<HO + 2 * 2048>(b[2][1], bl1 + so);
        lds_rd16<HO + 3 * 2048>(b[3][0], bl0 + so); lds_rd16<HO + 3 * 2048>(b[3][1], bl1 + so);
    };
    auto mfma16 = [&](auto j0_c) {
        constexpr int J0 = decltype(j0_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][J0 + j] = ASR_MFMA_16x16x32(a[i][ks].v, b[j][ks].v, acc[i][J0 + j]);
    };
#define ASR8N_LOAD_END()                                                                                   \
    do {                                                                                                   \
        if (TNW == 64) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");                        \
        else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");                                  \
    } while (0)

    // prologue: K steps 0 and 1 (the re-fetches beyond K go on into empty taps / the clamped last step)
    issue_a(0, -1, true, 0);
    issue_b(0, -1, 0);
    issue_a(1 % NSTAGE, -1, true, min(1, nk - 1));
    issue_b(1 % NSTAGE, -1, min(1, nk - 1));
    if (TNW == 64) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    const int grp = wid >> 2;
    if (grp == 1) asm volatile("s_barrier" ::: "memory");
    int st = 0, st2 = 2 % NSTAGE;                              // stage of K step t, of K step t + 2
    for (int t = 0; t < nk; ++t) {
        const int t2 = min(t + 2, nk - 1);
        const unsigned so = (unsigned)(st * STAGE);
        if (TNW == 64) {
            read_b(so, ASR8_C(0));
            __builtin_amdgcn_sched_barrier(0);
            read_a(so);
            issue_a(st2, -1, true, t2);
            issue_b(st2, -1, t2);
            ASR8N_LOAD_END();
            ASR8_COMPUTE_BEGIN();
            mfma16(ASR8_C(0));
            ASR8_COMPUTE_END();
        } else {
            read_b(so, ASR8_C(0));
            __builtin_amdgcn_sched_barrier(0);
            read_a(so);
            issue_a(st2, 0, false, t2);
            issue_b(st2, 0, t2);
            ASR8N_LOAD_END();
            ASR8_COMPUTE_BEGIN();
            mfma16(ASR8_C(0));
            ASR8_COMPUTE_END();
            read_b(so, ASR8_C(1));
            issue_a(st2, 1, true, t2);
            issue_b(st2, 1, t2);
            ASR8N_LOAD_END();
            ASR8_COMPUTE_BEGIN();
            mfma16(ASR8_C((NJ > 4 ? 4 : 0)));
            ASR8_COMPUTE_END();
        }
        st = st == NSTAGE - 1 ? 0 : st + 1;
        st2 = st2 == NSTAGE - 1 ? 0 : st2 + 1;
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // epilogue: acc[i][j][reg] = C[m0 + wid 32 + i 16 + 4 q + reg][G r + j % G + (j / G) 16 G]
    const int row0 = m0 + wid * 32 + 4 * q;
    const bool c16 = (ldc & 7) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + i * 16 + reg;
            if (row >= M) continue;
#pragma unroll
            for (int gq = 0; gq < NJ / G; ++gq) {
                const int col = G * r + gq * (16 * G);
                if (col >= N) continue;
                OutT* dst = C + (size_t)row * ldc + col;
                float v[G];
#pragma unroll
                for (int j = 0; j < G; ++j) v[j] = acc[i][gq * G + j][reg] + ((bias && col + j < N) ? bias[col + j] : 0.f);
                if (sizeof(OutT) == 4) {
#pragma unroll
                    for (int j = 0; j < G; j += 4)
                        if (col + j < N) *reinterpret_cast<float4*>(dst + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
                } else if (G == 8 && c16 && col + 7 < N) {
                    uint4 pk;
                    pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]);
                    pk.z = pack_bf16x2(v[G == 8 ? 4 : 0], v[G == 8 ? 5 : 1]); pk.w = pack_bf16x2(v[G == 8 ? 6 : 2], v[G == 8 ? 7 : 3]);
                    *reinterpret_cast<uint4*>(dst) = pk;
                } else {
#pragma unroll
                    for (int j = 0; j < G; j += 4)
                        if (col + j < N) {
                            uint2 pk;
                            pk.x = pack_bf16x2(v[j], v[j + 1]); pk.y = pack_bf16x2(v[j + 2], v[j + 3]);
                            *reinterpret_cast<uint2*>(dst + j) = pk;
                        }
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ TN: C[M,N] += A[K,M]^T B[K,N]
// The weight gradients (dW = dy^T x: K = T B = 32000 rows, M and N a few hundred to a few thousand) on the same schedule.  Both
// operands are k-strided in memory, so the LDS images keep the memory order -- a half tile is 64 k rows x 128 columns (256 B per row) --
// and the fragments come out of LDS through ds_read_b64_tr_b16 (16 / 8 transposing reads per A / B half instead of 8 / 4 plain ones:
// the same bytes).  16-B chunk c of k row kr sits at position c ^ 2 (kr % 8): the eight k rows a 32-lane group of a transposing read
// touches then cover all 64 banks.  A lane of such a read points at k row 4 g + q (+ 16) of the 32-row K slice and at columns
// 4 p .. 4 p + 3 of the 16-column tile (g = lane / 16, q = lane / 4 % 4, p = lane % 4) -- the convention of gemm.hip's TN kernels: A and
// B fragments hold their k slots in the same (permuted) order, which is all the product needs.
// Work item = (K split, output tile of one of up to four products); a workgroup adds its 256 x 256 partial product with float atomics.
// K splits are multiples of 64 rows; the rows beyond K are beyond the operand's buffer and arrive as zeros.
struct Tn8Prob {
    const uint16_t* A;
    const uint16_t* B;
    float* C;
    int lda, ldb, ldc, M, N, K, tiles_n;
    int tile_end;                        // tiles of this and all earlier products
};
struct Tn8Group {
    Tn8Prob p[4];
    int n;
};
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
union FragT {
    bf16x8 v;
    u32x2 d[2];
};
template <int OFF>
__device__ __forceinline__ void lds_tr8(u32x2& f, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF) : "memory");
}

// CONV: operand B is the virtual im2col matrix of the activation tensor x (Ts, B, Hs, Cs): row k = (t, b, h) over the output positions
// (Tr, B, Hr), column n = (kh, kw, c) -- the weight gradient of a convolution without a column matrix in memory.  A loader slot's
// columns never change, so its tap and channel are loop invariant; its row walks on by 64 positions per issue (incremental (t, b, h)
// with carries: no division in the loop).  One product per launch (grp.n == 1; P.B = x, P.ldb unused).
template <bool CONV>
__global__ __launch_bounds__(512, 2) void gemm_tn_8ph_kernel(Tn8Group grp, int tiles, int k_per_split, ConvDesc8 cd, int whatif, int splits, int stag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
    // (split, tile) pairs in split-major order, dealt to the XCDs (workgroup id % 8 under round-robin dispatch: a locality hint) in
    // eight contiguous runs: the tiles of one K split meet in one L2 and share the rows of A and B they read
    const int item = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    const int split = item / tiles;
    int tile = item - split * tiles;
    int pi = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (i + 1 < grp.n && tile >= grp.p[i].tile_end) pi = i + 1;
    const Tn8Prob& P = grp.p[pi];
    if (pi > 0) tile -= grp.p[pi - 1].tile_end;
    const uint16_t* __restrict__ A = P.A;
    const uint16_t* __restrict__ B = P.B;
    const int lda = P.lda, ldb = P.ldb, M = P.M, N = P.N, K = P.K;
    const int tm = tile / P.tiles_n, tn = tile - tm * P.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    // K splits of unequal length (stag rows per step, a multiple of 64: split s is k_per_split + stag (2 s - splits + 1) rows long): the
    // workgroups of a launch start together, and with equal lengths all of them reach their 256 KB of float atomics together -- 63 MB at
    // the chip-wide atomic rate with nothing computing beside them (40 of 194 us for a GRU layer's weight gradients).  Staggered, the
    // atomics of the splits that end first run beside the K loops of the others.
    const int kbeg = split * k_per_split + stag * split * (split - splits);
    const int klen = k_per_split + stag * (2 * split - splits + 1);
    if (kbeg >= K) return;
    const int kend = min(K, kbeg + klen);
    const int nk = (kend - kbeg + 63) >> 6;

    // ---- loader: slot s = i * 512 + tid = (k row kr = s / 16, position s % 16) holds chunk (s % 16) ^ 2 (kr % 8); the K advance
    // goes into the VECTOR offset (the range check of a raw buffer access does not see the scalar offset): rows beyond K read zeros
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((unsigned)K * (unsigned)lda * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)B, 0, CONV ? (int)((unsigned)cd.Ts * (unsigned)cd.B * (unsigned)cd.Hs * (unsigned)cd.Cs * 2u) : (int)((unsigned)K * (unsigned)ldb * 2u), 0x00020000);
    unsigned oa[2][2], ob[2][2];
    int cv_ci[2][2], cv_dt[2][2], cv_dh[2][2], cv_t[2][2], cv_b[2][2], cv_h[2][2];      // CONV: [B half][slot]
    bool cv_ok[2][2];
    const int inc_h = CONV ? 64 % cd.Hr : 0, inc_b = CONV ? (64 / cd.Hr) % cd.B : 0, inc_t = CONV ? (64 / cd.Hr) / cd.B : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = i * 512 + tid, kr = s >> 4, c = (s & 15) ^ ((kr & 7) << 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // A half h: chunk c = 8 columns of the M-wave c / 8: m0 + (c / 8) 128 + h 64 + (c % 8) 8
            const int m = m0 + (c >> 3) * 128 + h * 64 + (c & 7) * 8;
            oa[h][i] = m < M ? (unsigned)(kbeg + kr) * (unsigned)(lda * 2) + (unsigned)(m * 2) : 0xfffffff0u;
            // B half h: chunk c = 8 columns of the N-wave c / 4: n0 + (c / 4) 64 + h 32 + (c % 4) 8
            const int n = n0 + (c >> 2) * 64 + h * 32 + (c & 3) * 8;
            if (CONV) {
                const int tap = n / cd.Cs, kh = tap / cd.KW, kw = tap - kh * cd.KW;
                cv_ci[h][i] = n - tap * cd.Cs;
                cv_dt[h][i] = kw - cd.pt;
                cv_dh[h][i] = kh - cd.ph;
                cv_ok[h][i] = n < N && kh < cd.KH;
                const int gk = kbeg + kr;
                cv_h[h][i] = gk % cd.Hr;
                const int tb = gk / cd.Hr;
                cv_b[h][i] = tb % cd.B;
                cv_t[h][i] = tb / cd.B;
                ob[h][i] = 0u;
            } else {
                ob[h][i] = n < N ? (unsigned)(kbeg + kr) * (unsigned)(ldb * 2) + (unsigned)(n * 2) : 0xfffffff0u;
            }
        }
    }
    const unsigned stepa = (unsigned)(lda * 128), stepb = (unsigned)(ldb * 128);      // 64 k rows in bytes
    auto issue = [&](auto kind_c, auto par_c, int kt) {
        constexpr int KIND = decltype(kind_c)::value, PAR = decltype(par_c)::value;
        char* base = smem + (PAR * 4 + KIND) * HALF + wid * 1024;
        if (KIND < 2) {
            const unsigned ko = (unsigned)kt * stepa;
            const unsigned v0 = oa[KIND & 1][0], v1 = oa[KIND & 1][1];
            lds_dma16(rsrc_a, base, v0 == 0xfffffff0u ? v0 : v0 + ko, 0);
            lds_dma16(rsrc_a, base + 8192, v1 == 0xfffffff0u ? v1 : v1 + ko, 0);
        } else if (CONV) {
            constexpr int H = KIND & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = cv_t[H][i], b = cv_b[H][i], hq = cv_h[H][i];
                {   // the same slot of the next K step (issues of one half come in K order; beyond the last step the walk simply goes on)
                    int hh = hq + inc_h;
                    const int c1 = hh >= cd.Hr;
                    hh -= c1 ? cd.Hr : 0;
                    int bb = b + inc_b + c1;
                    const int c2 = bb >= cd.B;
                    bb -= c2 ? cd.B : 0;
                    cv_h[H][i] = hh; cv_b[H][i] = bb; cv_t[H][i] = t + inc_t + c2;
                }
                const int ti = t + cv_dt[H][i], hi = hq + cv_dh[H][i];
                const bool ok = cv_ok[H][i] && (unsigned)ti < (unsigned)cd.Ts && (unsigned)hi < (unsigned)cd.Hs;
                const unsigned v = (unsigned)(((ti * cd.B + b) * cd.Hs + hi) * cd.Cs + cv_ci[H][i]) * 2u;
                lds_dma16(rsrc_b, base + i * 8192, ok ? v : 0xfffffff0u, 0);
            }
        } else {
            const unsigned ko = (unsigned)kt * stepb;
            const unsigned v0 = ob[KIND & 1][0], v1 = ob[KIND & 1][1];
            lds_dma16(rsrc_b, base, v0 == 0xfffffff0u ? v0 : v0 + ko, 0);
            lds_dma16(rsrc_b, base + 8192, v1 == 0xfffffff0u ? v1 : v1 + ko, 0);
        }
    };

    // ---- reader (transposing): lane (g, q, p) -> k row 4 g + q of a 16-row group, columns 4 p .. 4 p + 3 of a 16-column tile
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int kq = 4 * g + q, sw = (kq & 7) << 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    unsigned ra[4], rb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = lds0 + (unsigned)(kq * 256 + (((wm * 8 + i * 2 + (p >> 1)) ^ sw) << 4) + (p & 1) * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) rb[j] = lds0 + (unsigned)(kq * 256 + (((wn * 4 + j * 2 + (p >> 1)) ^ sw) << 4) + (p & 1) * 8);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    FragT a[4][2], b0[2][2], b1[2][2];
    // fragment (tile, ks): k rows ks 32 + {kq, 16 + kq}: immediates ks * 8192 and + 4096
    auto read_a = [&](auto par_c, auto h_c) {
        constexpr int PAR = decltype(par_c)::value, H = decltype(h_c)::value, R = (PAR * 4 + H) * HALF;
        static_assert(R + 8192 + 4096 < 65536 || PAR == 1, "immediate range");
        const unsigned po = PAR ? 4u * HALF : 0u;       // (the parity's 64 KiB go into the address register: the immediate is 16 bits)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned ad = ra[i] + po;
            lds_tr8<H * HALF + 0>(a[i][0].d[0], ad); lds_tr8<H * HALF + 4096>(a[i][0].d[1], ad);
            lds_tr8<H * HALF + 8192>(a[i][1].d[0], ad); lds_tr8<H * HALF + 12288>(a[i][1].d[1], ad);
        }
    };
    auto read_b = [&](auto par_c, auto h_c, FragT (&b)[2][2]) {
        constexpr int PAR = decltype(par_c)::value, H = decltype(h_c)::value;
        const unsigned po = PAR ? 4u * HALF : 0u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned ad = rb[j] + po;
            lds_tr8<(2 + H) * HALF + 0>(b[j][0].d[0], ad); lds_tr8<(2 + H) * HALF + 4096>(b[j][0].d[1], ad);
            lds_tr8<(2 + H) * HALF + 8192>(b[j][1].d[0], ad); lds_tr8<(2 + H) * HALF + 12288>(b[j][1].d[1], ad);
        }
    };
    auto quadrant = [&](auto i0_c, auto j0_c, const FragT (&b)[2][2]) {
        constexpr int I0 = decltype(i0_c)::value, J0 = decltype(j0_c)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[I0 + i][J0 + j] = ASR_MFMA_16x16x32(a[i][ks].v, b[j][ks].v, acc[I0 + i][J0 + j]);
    };
    auto kstep = [&](auto par_c, int t) {
        constexpr int PAR = decltype(par_c)::value;
        const int t1 = min(t + 1, nk - 1), t2 = min(t + 2, nk - 1);
        read_b(ASR8_C(PAR), ASR8_C(0), b0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(ASR8_C(PAR), ASR8_C(0));
        ASR8_ISSUE(3, PAR ^ 1, t1);
        ASR8_LOAD_END();
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(0), ASR8_C(0), b0);
        ASR8_COMPUTE_END();
        read_b(ASR8_C(PAR), ASR8_C(1), b1);
        ASR8_ISSUE(1, PAR ^ 1, t1);
        ASR8_LOAD_END();
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(0), ASR8_C(2), b1);
        ASR8_COMPUTE_END();
        read_a(ASR8_C(PAR), ASR8_C(1));
        ASR8_ISSUE(0, PAR, t2);
        ASR8_LOAD_END();
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(4), ASR8_C(2), b1);
        ASR8_COMPUTE_END();
        ASR8_ISSUE(2, PAR, t2);
        ASR8_LOAD_END();
        ASR8_COMPUTE_BEGIN();
        quadrant(ASR8_C(4), ASR8_C(0), b0);
        ASR8_COMPUTE_END();
    };
    {
        const int t1 = min(1, nk - 1);
        ASR8_ISSUE(0, 0, 0);
        ASR8_ISSUE(2, 0, 0);
        ASR8_ISSUE(3, 0, 0);
        ASR8_ISSUE(1, 0, 0);
        ASR8_ISSUE(0, 1, t1);
        ASR8_ISSUE(2, 1, t1);
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    }
    if (wm == 1) asm volatile("s_barrier" ::: "memory");
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        kstep(ASR8_C(0), t);
        kstep(ASR8_C(1), t + 1);
    }
    if (t < nk) kstep(ASR8_C(0), t);
    if (wm == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: acc[I][J][reg] += into C[m0 + wm 128 + I 16 + 4 (lane / 16) + reg][n0 + wn 64 + J 16 + lane % 16]
    float* __restrict__ C = P.C;
    const int ldc = P.ldc;
    const int col0 = n0 + wn * 64 + (lane & 15), row0 = m0 + wm * 128 + 4 * (lane >> 4);
    if (whatif == 1) {          // what-if (ASR_DEBUG tn8_whatif=1: timing only, results invalid): one atomic per wave instead of the tile's
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (lane == 0 && row0 < M && col0 < N) atomicAdd(C + (size_t)row0 * ldc + col0, sum);
        return;
    }
    if (whatif == 2) {          // the accumulator's own layout: an atomic instruction = 4 rows x 16 columns (four 64-byte pieces)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = row0 + i * 16 + reg;
                if (row >= M) continue;
                float* dst = C + (size_t)row * ldc + col0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col0 + j * 16 < N) atomicAdd(dst + j * 16, acc[i][j][reg]);
            }
        return;
    }
    // 4 x 4 transpose between the lane's 16-lane group (row group g) and the column tile j (two lane-swap instructions per pair of
    // registers): afterwards register j' holds row 4 j' + reg and lane l column l of the wave's 64 -- an atomic instruction = 256
    // contiguous bytes of one row
    const int colL = n0 + wn * 64 + lane, rowb = m0 + wm * 128;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            unsigned r0 = __float_as_uint(acc[i][0][reg]), r1 = __float_as_uint(acc[i][1][reg]);
            unsigned r2 = __float_as_uint(acc[i][2][reg]), r3 = __float_as_uint(acc[i][3][reg]);
            auto s02 = __builtin_amdgcn_permlane32_swap(r0, r2, false, false);
            auto s13 = __builtin_amdgcn_permlane32_swap(r1, r3, false, false);
            auto s01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
            auto s23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
            const unsigned v[4] = {s01[0], s01[1], s23[0], s23[1]};
#pragma unroll
            for (int jp = 0; jp < 4; ++jp) {
                const int row = rowb + i * 16 + 4 * jp + reg;
                if (row < M && colL < N) atomicAdd(C + (size_t)row * ldc + colL, __uint_as_float(v[jp]));
            }
        }
}

// K splits of a launch.  A work item's K loop takes ~1.42 us per 64 rows, its 256 KB of float atomics ~0.175 us of the chip-wide atomic
// rate, and a launch's items all add at its end: K / (64 S) x 1.42 + tiles S x 0.175 us is smallest at S = sqrt(0.127 K / tiles) -- fewer
// items than CUs where few tiles meet a short K (a dense layer's 640 x 512 gradient over 32000 rows: 26 splits = 156 items instead of 42
// = 252: 62.6 -> 52 us, tools/tn8_small.py); otherwise one item per CU (`target` = ASR_DEBUG tn8_items, 256) and at least 8 K steps each.
static inline int tn8_splits(int tiles, int kmax) {
    static const int target = debug_flag("tn8_items", 256), model = debug_flag("tn8_split_model", 1);
    int splits = target / tiles;
    if (model) {
        int s = 1;
        while ((long long)(s + 1) * (s + 1) * tiles * 1000 <= 127LL * kmax) ++s;      // floor(sqrt(0.127 kmax / tiles))
        if (s < splits) splits = s;
    }
    if (splits < 1) splits = 1;
    const int max_splits = (kmax + 511) / 512;
    return splits > max_splits ? max_splits : splits;
}

// rows per stagger step of a launch's K splits (0: equal splits).  One split's workgroups (one per tile) add tiles x 256 KB at ~1.5 TB/s
// (tiles x 0.175 us) while a K step of 64 rows takes a workgroup ~1.4 us: consecutive splits should end that far apart, i.e. differ by
// 2 stag = tiles x 8 rows, rounded to whole K steps -- as long as the shortest split keeps three quarters of the mean.
// ASR_DEBUG tn8_stag: -1 (this rule), 0 (equal splits), n > 0 (n K steps of difference between consecutive splits).
static inline int tn8_stagger(int tiles, int splits, int k_per_split) {
    static const int forced = debug_flag("tn8_stag", -1);
    if (splits < 2 || forced == 0) return 0;
    int steps = forced > 0 ? forced : (tiles * 8 + 32) / 64;            // K steps between consecutive splits (= 2 stag / 64)
    if (steps < 1) return 0;
    // stag (2 s - splits + 1) must be a multiple of 64 for every s: an even number of K steps when `splits` is even (odd multipliers)
    int stag = (splits & 1) ? steps * 32 : ((steps + 1) / 2) * 64;
    if (splits & 1) stag = ((stag + 63) / 64) * 64;                     // (odd splits: even multipliers -- whole K steps of 64 keep it simple)
    while (stag > 0 && (long long)stag * (splits - 1) * 4 > k_per_split) stag -= 64;
    return stag > 0 ? stag : 0;
}

}  // namespace gemm8
}  // namespace asr

using namespace asr;

// 1 if asr_gemm_nt_8ph serves the product (K a multiple of 8, whole groups of four columns, 16-byte aligned rows, 32-bit byte offsets)
extern "C" int asr_gemm_nt_8ph_ok(const void* A, int lda, const void* B, int ldb, const void* C, int ldc, const float* bias, int M, int N, int K,
                                  int out_bf16) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K < 64 || (K & 7) || (N & 3)) return 0;
    if (lda < K || ldb < K || ldc < N || (lda & 7) || (ldb & 7) || (ldc & 3)) return 0;
    if ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C)) & 15) return 0;
    if (bias && (((uintptr_t)bias) & 15)) return 0;
    if ((unsigned long long)M * lda * 2 >= (1ull << 31) || (unsigned long long)N * ldb * 2 >= (1ull << 31)) return 0;
    return 1;
}

extern "C" int asr_gemm_nt_8ph(void* stream_, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, int M, int N,
                               int K, int out_bf16) {
    if (!asr_gemm_nt_8ph_ok(A, lda, B, ldb, C, ldc, bias, M, N, K, out_bf16)) return ASR_ERR_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    const int tiles_m = cdiv(M, 256), tiles_n = cdiv(N, 256);
    const unsigned a_bytes = (unsigned)((unsigned long long)M * lda * 2), b_bytes = (unsigned)((unsigned long long)N * ldb * 2);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8ph_kernel<uint16_t, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8ph_kernel<float, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8ph_kernel<uint16_t, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8ph_kernel<float, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES);
        attr = true;
    }
#define ASR8_LAUNCH(T, CT, KT_)                                                                                                     \
    hipLaunchKernelGGL((gemm8::gemm_nt_8ph_kernel<T, KT_, false>), dim3(tiles_m * tiles_n), dim3(512), gemm8::LDS_BYTES, stream, (const uint16_t*)A, lda, \
                       (const uint16_t*)B, ldb, (CT*)C, ldc, bias, M, N, K, tiles_m, tiles_n, a_bytes, b_bytes, gemm8::ConvDesc8{}, stagger)
    // the persistent form for short K and more tiles than CUs (ASR_DEBUG nt_8pp=0: the one-tile-per-workgroup kernel everywhere)
    // float32 output (the logits: 384 MB) is bound by its stores in either form and the one-tile kernel is the faster one there (119.5
    // against 123 us; ASR_DEBUG nt8pp_f32=1 runs the persistent form for it: tests)
    static const int pp = debug_flag("nt_8pp", 1), pp_kmax = debug_flag("nt8pp_kmax", 1024), pp_whatif = debug_flag("nt8pp_whatif", 0),
                     pp_f32 = debug_flag("nt8pp_f32", 0);
    const unsigned long long c_total = ((unsigned long long)(M - 1) * ldc + N) * (out_bf16 ? 2 : 4);
    if (pp && (K & 63) == 0 && K >= 128 && K <= pp_kmax && tiles_m * tiles_n > 256 && N <= 8192 && c_total < 0xfffffff0ull &&
        (out_bf16 ? ((N & 7) == 0 && (ldc & 7) == 0) : pp_f32 != 0)) {
        static bool attr_pp = false;
        if (!attr_pp) {
            (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8pp_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES + 32768);
            (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8pp_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES + 32768);
            attr_pp = true;
        }
        if (out_bf16)
            hipLaunchKernelGGL((gemm8::gemm_nt_8pp_kernel<uint16_t>), dim3(256), dim3(512), gemm8::LDS_BYTES + 32768, stream, (const uint16_t*)A, lda,
                               (const uint16_t*)B, ldb, (uint16_t*)C, ldc, bias, M, N, K, tiles_m, tiles_n, a_bytes, b_bytes, (unsigned)c_total, pp_whatif);
        else
            hipLaunchKernelGGL((gemm8::gemm_nt_8pp_kernel<float>), dim3(256), dim3(512), gemm8::LDS_BYTES + 32768, stream, (const uint16_t*)A, lda,
                               (const uint16_t*)B, ldb, (float*)C, ldc, bias, M, N, K, tiles_m, tiles_n, a_bytes, b_bytes, (unsigned)c_total, pp_whatif);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    static const int stagger_env = debug_flag("nt8_stagger", 0);
    const int stagger = (stagger_env < 0 || (tiles_m * tiles_n > 256 && K <= 1024)) ? stagger_env : 0;
    const bool kt = (K & 63) != 0;
    if (out_bf16) { if (kt) ASR8_LAUNCH(uint16_t, uint16_t, true); else ASR8_LAUNCH(uint16_t, uint16_t, false); }
    else          { if (kt) ASR8_LAUNCH(float, float, true); else ASR8_LAUNCH(float, float, false); }
#undef ASR8_LAUNCH
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// 1 if asr_gemm_tn_acc_8ph serves the product: whole 16-byte chunks of eight columns (M, N, lda, ldb multiples of 8, aligned bases),
// operands below 2 GiB
extern "C" int asr_gemm_tn_8ph_ok(const void* A, int lda, const void* B, int ldb, const float* C, int ldc, int M, int N, int K) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || (M & 7) || (N & 7) || (lda & 7) || (ldb & 7)) return 0;
    if (lda < M || ldb < N || ldc < N) return 0;
    if ((((uintptr_t)A) | ((uintptr_t)B)) & 15) return 0;
    if ((unsigned long long)K * lda * 2 >= (1ull << 31) || (unsigned long long)K * ldb * 2 >= (1ull << 31)) return 0;
    return 1;
}

// n <= 4 products C_i += A_i^T B_i in one launch on the eight-wave kernel (every product must pass asr_gemm_tn_8ph_ok)
extern "C" int asr_gemm_tn_acc_group_8ph(void* stream_, int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                         float* const* C, const int* ldc, const int* M, const int* N, const int* K) {
    if (!A || !B || !C || !lda || !ldb || !ldc || !M || !N || !K || n < 1 || n > 4) return ASR_ERR_BAD_ARG;
    gemm8::Tn8Group grp{};
    int tiles = 0, kmax = 0;
    for (int i = 0; i < n; ++i) {
        if (!asr_gemm_tn_8ph_ok(A[i], lda[i], B[i], ldb[i], C[i], ldc[i], M[i], N[i], K[i])) return ASR_ERR_UNSUPPORTED;
        gemm8::Tn8Prob& q = grp.p[i];
        q.A = (const uint16_t*)A[i]; q.B = (const uint16_t*)B[i]; q.C = C[i];
        q.lda = lda[i]; q.ldb = ldb[i]; q.ldc = ldc[i]; q.M = M[i]; q.N = N[i]; q.K = K[i];
        q.tiles_n = cdiv(N[i], 256);
        tiles += cdiv(M[i], 256) * q.tiles_n;
        q.tile_end = tiles;
        kmax = K[i] > kmax ? K[i] : kmax;
    }
    grp.n = n;
    // one workgroup per CU holds a whole CU (128 KiB of LDS): K splits so that about every CU gets one item, each split a multiple of 64 rows
    int splits = gemm8::tn8_splits(tiles, kmax);
    const int k_per_split = cdiv(cdiv(kmax, splits), 64) * 64;
    splits = cdiv(kmax, k_per_split);
    const int items = tiles * splits, grid = 8 * cdiv(items, 8);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_tn_8ph_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES);
        attr = true;
    }
    static const int whatif = debug_flag("tn8_whatif", 0);
    const int stag = gemm8::tn8_stagger(tiles, splits, k_per_split);
    hipLaunchKernelGGL(gemm8::gemm_tn_8ph_kernel<false>, dim3(grid), dim3(512), gemm8::LDS_BYTES, (hipStream_t)stream_, grp, tiles, k_per_split, gemm8::ConvDesc8{}, whatif,
                       splits, stag);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// asr_conv_nt's products on the eight-wave kernel: the implicit convolution of gemm.hip (same arguments) with Cs % 64 == 0.
extern "C" int asr_conv_nt_8ph_ok(const void* x, const void* W, int ldw, const void* out, int out_bf16, const float* bias, int Ts, int B, int Hs,
                                  int Cs, int KH, int KW, int Tr, int Hr, int N) {
    if (!x || !W || !out || Ts <= 0 || B <= 0 || Hs <= 0 || Cs <= 0 || KH <= 0 || KW <= 0 || Tr <= 0 || Hr <= 0 || N <= 0) return 0;
    if ((Cs & 63) || (ldw & 63) || ldw < KH * KW * Cs || (ldw % Cs) || (N & 3) || Hs >= 256) return 0;
    if ((((uintptr_t)x) | ((uintptr_t)W) | ((uintptr_t)out)) & 15) return 0;
    if (bias && (((uintptr_t)bias) & 15)) return 0;
    if ((unsigned long long)Ts * B * Hs * Cs * 2 >= 0x7ffffff0ull || (unsigned long long)N * ldw * 2 >= (1ull << 31) || (long long)Tr * B * Hr > 0x7fffffffLL) return 0;
    return 1;
}

extern "C" int asr_conv_nt_8ph(void* stream_, const void* x, const void* W, int ldw, void* out, int out_bf16, const float* bias, int Ts, int B,
                               int Hs, int Cs, int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N) {
    if (sgn != 1 && sgn != -1) return ASR_ERR_BAD_ARG;
    if (!asr_conv_nt_8ph_ok(x, W, ldw, out, out_bf16, bias, Ts, B, Hs, Cs, KH, KW, Tr, Hr, N)) return ASR_ERR_UNSUPPORTED;
    const int M = Tr * B * Hr, K = ldw;
    const int tiles_m = cdiv(M, 256), tiles_n = cdiv(N, 256);
    const unsigned a_bytes = (unsigned)((unsigned long long)Ts * B * Hs * Cs * 2), b_bytes = (unsigned)((unsigned long long)N * ldw * 2);
    const gemm8::ConvDesc8 cd{B, Hs, Cs, Ts, KH, KW, pad_h, pad_t, sgn, Hr};
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8ph_kernel<uint16_t, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8ph_kernel<float, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES);
        attr = true;
    }
    if (out_bf16)
        hipLaunchKernelGGL((gemm8::gemm_nt_8ph_kernel<uint16_t, false, true>), dim3(tiles_m * tiles_n), dim3(512), gemm8::LDS_BYTES, (hipStream_t)stream_,
                           (const uint16_t*)x, 0, (const uint16_t*)W, ldw, (uint16_t*)out, N, bias, M, N, K, tiles_m, tiles_n, a_bytes, b_bytes, cd, 0);
    else
        hipLaunchKernelGGL((gemm8::gemm_nt_8ph_kernel<float, false, true>), dim3(tiles_m * tiles_n), dim3(512), gemm8::LDS_BYTES, (hipStream_t)stream_,
                           (const uint16_t*)x, 0, (const uint16_t*)W, ldw, (float*)out, N, bias, M, N, K, tiles_m, tiles_n, a_bytes, b_bytes, cd, 0);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// asr_conv_tn_acc on the eight-wave kernel: C[Co, KH KW Cs] += g[(t, b, h), Co]^T . im2col(x)
extern "C" int asr_conv_tn_8ph_ok(const void* g, int ldg, const void* x, const float* C, int ldc, int Co, int Ts, int B, int Hs, int Cs, int KH,
                                  int KW, int Tr, int Hr) {
    if (!g || !x || !C || Co <= 0 || Ts <= 0 || B <= 0 || Hs <= 0 || Cs <= 0 || KH <= 0 || KW <= 0 || Tr <= 0 || Hr <= 0) return 0;
    const int N = KH * KW * Cs;
    const long long K = (long long)Tr * B * Hr;
    if ((Co & 7) || (Cs & 7) || (ldg & 7) || ldg < Co || ldc < N) return 0;
    if ((((uintptr_t)g) | ((uintptr_t)x)) & 15) return 0;
    if (K > 0x7fffffffLL || (unsigned long long)K * ldg * 2 >= (1ull << 31) || (unsigned long long)Ts * B * Hs * Cs * 2 >= 0x7ffffff0ull) return 0;
    return 1;
}

extern "C" int asr_conv_tn_acc_8ph(void* stream_, const void* g, int ldg, const void* x, float* C, int ldc, int Co, int Ts, int B, int Hs,
                                   int Cs, int KH, int KW, int pad_h, int pad_t, int Tr, int Hr) {
    if (!asr_conv_tn_8ph_ok(g, ldg, x, C, ldc, Co, Ts, B, Hs, Cs, KH, KW, Tr, Hr)) return ASR_ERR_UNSUPPORTED;
    const int N = KH * KW * Cs, K = Tr * B * Hr;
    gemm8::Tn8Group grp{};
    gemm8::Tn8Prob& q = grp.p[0];
    q.A = (const uint16_t*)g; q.B = (const uint16_t*)x; q.C = C;
    q.lda = ldg; q.ldb = 0; q.ldc = ldc; q.M = Co; q.N = N; q.K = K;
    q.tiles_n = cdiv(N, 256);
    const int tiles = cdiv(Co, 256) * q.tiles_n;
    q.tile_end = tiles;
    grp.n = 1;
    int splits = gemm8::tn8_splits(tiles, K);
    const int k_per_split = cdiv(cdiv(K, splits), 64) * 64;
    splits = cdiv(K, k_per_split);
    const int items = tiles * splits, grid = 8 * cdiv(items, 8);
    const gemm8::ConvDesc8 cd{B, Hs, Cs, Ts, KH, KW, pad_h, pad_t, +1, Hr};
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_tn_8ph_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, gemm8::LDS_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL(gemm8::gemm_tn_8ph_kernel<true>, dim3(grid), dim3(512), gemm8::LDS_BYTES, (hipStream_t)stream_, grp, tiles, k_per_split, cd, 0, splits,
                       gemm8::tn8_stagger(tiles, splits, k_per_split));
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// the narrow form (N <= 128) of asr_conv_nt_8ph: same arguments
extern "C" int asr_conv_nt_8pn_ok(const void* x, const void* W, int ldw, const void* out, int out_bf16, const float* bias, int Ts, int B, int Hs,
                                  int Cs, int KH, int KW, int Tr, int Hr, int N) {
    if (N > 128) return 0;
    return asr_conv_nt_8ph_ok(x, W, ldw, out, out_bf16, bias, Ts, B, Hs, Cs, KH, KW, Tr, Hr, N);
}

extern "C" int asr_conv_nt_8pn(void* stream_, const void* x, const void* W, int ldw, void* out, int out_bf16, const float* bias, int Ts, int B,
                               int Hs, int Cs, int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N) {
    if (sgn != 1 && sgn != -1) return ASR_ERR_BAD_ARG;
    if (!asr_conv_nt_8pn_ok(x, W, ldw, out, out_bf16, bias, Ts, B, Hs, Cs, KH, KW, Tr, Hr, N)) return ASR_ERR_UNSUPPORTED;
    const int M = Tr * B * Hr, K = ldw;
    const int tiles_m = cdiv(M, 256);
    const unsigned a_bytes = (unsigned)((unsigned long long)Ts * B * Hs * Cs * 2), b_bytes = (unsigned)((unsigned long long)N * ldw * 2);
    const gemm8::ConvDesc8 cd{B, Hs, Cs, Ts, KH, KW, pad_h, pad_t, sgn, Hr};
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8pn_kernel<uint16_t, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 40960);
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8pn_kernel<float, true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 40960);
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8pn_kernel<uint16_t, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152);
        (void)hipFuncSetAttribute((const void*)gemm8::gemm_nt_8pn_kernel<float, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 49152);
        attr = true;
    }
#define ASR8N(T, CT, W_, LDS)                                                                                                        \
    hipLaunchKernelGGL((gemm8::gemm_nt_8pn_kernel<T, true, W_>), dim3(tiles_m), dim3(512), LDS, (hipStream_t)stream_, (const uint16_t*)x, 0, \
                       (const uint16_t*)W, ldw, (CT*)out, N, bias, M, N, K, tiles_m, a_bytes, b_bytes, cd)
    if (N <= 64) { if (out_bf16) ASR8N(uint16_t, uint16_t, 64, 4 * 40960); else ASR8N(float, float, 64, 4 * 40960); }
    else         { if (out_bf16) ASR8N(uint16_t, uint16_t, 128, 3 * 49152); else ASR8N(float, float, 128, 3 * 49152); }
#undef ASR8N
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
