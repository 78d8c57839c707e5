// bf16 MFMA GEMMs for the dense projections of the acoustic model (gfx950, wave64, v_mfma_f32_16x16x32_bf16).
//
//   NT:  C[M,N] = A[M,K] * B[N,K]^T (+ bias[N])      forward projections (x W^T) and backward-data (dy W, with W^T copy)
//   TN:  C[M,N] += A[K,M]^T * B[K,N]                 weight gradients (dW = dy^T x), split-K, f32 atomics
//
// Replaces the cuBLAS/cuDNN calls behind chainer.links.Linear / ConvolutionND(ksize=1) (asr/nn/convolution_1d.py:7-38),
// the SRU projection asr/nn/sru.py:340-341 (forward) / :421-429 (backward) and, through im2col, Convolution2D.
//
// Tile 128 x 128 x 64, 256 threads = 2 x 2 waves, each wave 64 x 64 = 4 x 4 MFMA tiles (64 accumulator VGPRs).
// NT: operands are k-contiguous -> LDS rows of 128 B, 16-B chunks XOR-swizzled by (row & 7), fragments by ds_read_b128.
// TN: operands are k-strided    -> LDS keeps the global [k][m] order (rows padded to 288 B), fragments by
//     ds_read_b64_tr_b16 (hardware transpose).  The k slots of a fragment are permuted (same permutation for A and B).
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace gemm {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT_LDS_BYTES = 2 * (BM + BN) * BK * 2;   // 64 KiB, double buffered

// Tile order.  Workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2), so id -> (row panel, column
// tile) is chosen such that the XCD that first touches a row panel of A also computes all of that panel's column
// tiles: panels go to XCDs in groups of 8 (panel % 8 = id % 8), the incomplete last group falls back to the plain order.
// Only the L2 hit rate depends on the dispatch behaviour, never the result (the map is a bijection either way).
// Used (tiles_n passed negative) when the whole B operand fits an L2 beside the panels in flight; otherwise plain order.
__device__ __forceinline__ void tile_of(int bid, int tiles_m, int tiles_n_signed, int& tm, int& tn) {
    const int tiles_n = tiles_n_signed < 0 ? -tiles_n_signed : tiles_n_signed;
    if (tiles_n_signed > 0) {
        tm = bid / tiles_n;
        tn = bid - tm * tiles_n;
        return;
    }
    const int full = (tiles_m >> 3) << 3;
    if (bid < full * tiles_n) {
        const int x = bid & 7, idx = bid >> 3;
        tm = (idx / tiles_n) * 8 + x;
        tn = idx - (idx / tiles_n) * tiles_n;
    } else {
        const int r = bid - full * tiles_n;
        tm = full + r / tiles_n;
        tn = r - (r / tiles_n) * tiles_n;
    }
}

union Frag {
    bf16x8 v;
    uint4 u;
    bf16x4 h[2];
};

__device__ __forceinline__ uint4 ld16(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }

template <typename OutT>
__device__ __forceinline__ void store_out(OutT* p, float v);
template <>
__device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_out<uint16_t>(uint16_t* p, float v) { *p = f32_to_bf16(v); }

// ------------------------------------------------------------------------------------------------ NT
// GLDS: operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: each lane's 16 B land lane-linear behind a
// wave-uniform base, so the XOR swizzle is applied to the SOURCE address); needs K % 64 == 0 and 16-B aligned rows.
// The register-staged form pays ~80 B/clk/CU for its ds_write_b128 pass, which with two workgroups per CU costs more
// LDS time than the MFMAs take.  Rows beyond M / N are clamped to the last row (computed, never stored).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <typename OutT, bool GLDS>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const uint16_t* __restrict__ A, int lda,
                                                      const uint16_t* __restrict__ B, int ldb, OutT* __restrict__ C,
                                                      int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                      int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tm, tn;
    tile_of(blockIdx.x, (M + BM - 1) / BM, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    char* As = smem;                          // [2][BM][BK] bf16
    char* Bs = smem + 2 * BM * BK * 2;        // [2][BN][BK] bf16

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    // 16-byte loads need K, the row pitch and the base address to be multiples of 8 elements; otherwise (odd
    // vocabulary sizes such as 119) fall back to element loads for that operand
    const bool a_vec = ((K & 7) == 0) && ((lda & 7) == 0) && ((((uintptr_t)A) & 15) == 0);
    const bool b_vec = ((K & 7) == 0) && ((ldb & 7) == 0) && ((((uintptr_t)B) & 15) == 0);
    auto load_row = [&](const uint16_t* P, int ld, bool vec, int g, int lim, int k) -> uint4 {
        if (g >= lim || k >= K) return make_uint4(0, 0, 0, 0);
        const uint16_t* src = P + (size_t)g * ld + k;
        if (vec) return ld16(src);
        union { uint4 u; uint16_t s[8]; } t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t.s[e] = (k + e < K) ? src[e] : (uint16_t)0;
        return t.u;
    };
    auto load_global = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256;
            const int row = id >> 3, c = id & 7;
            const int k = k0 + c * 8;
            ra[i] = load_row(A, lda, a_vec, m0 + row, M, k);
            rb[i] = load_row(B, ldb, b_vec, n0 + row, N, k);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + i * 256;
            const int row = id >> 3, c = id & 7;
            const int off = row * (BK * 2) + ((c ^ (row & 7)) << 4);
            *reinterpret_cast<uint4*>(As + buf * (BM * BK * 2) + off) = ra[i];
            *reinterpret_cast<uint4*>(Bs + buf * (BN * BK * 2) + off) = rb[i];
        }
    };

    const int nk = (K + BK - 1) / BK;
    auto compute_tile = [&](int buf) {
        const char* Ab = As + buf * (BM * BK * 2);
        const char* Bb = Bs + buf * (BN * BK * 2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag a[4], b[4];
            const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ar = wm * 64 + i * 16 + (lane & 15);
                a[i].u = *reinterpret_cast<const uint4*>(Ab + ar * (BK * 2) + ((chunk ^ (ar & 7)) << 4));
                const int br = wn * 64 + i * 16 + (lane & 15);
                b[i].u = *reinterpret_cast<const uint4*>(Bb + br * (BK * 2) + ((chunk ^ (br & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = ASR_MFMA_16x16x32(a[i].v, b[j].v, acc[i][j]);
        }
    };
    if (GLDS) {
        // slot s = i * 256 + wid * 64 + lane of a tile image: row s / 8, 16-B position s % 8, which holds chunk (s % 8) ^ (row & 7)
        const uint16_t* ga[4];
        const uint16_t* gb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sl = i * 256 + tid, row = sl >> 3, c = (sl & 7) ^ (row & 7);
            ga[i] = A + (size_t)min(m0 + row, M - 1) * lda + c * 8;
            gb[i] = B + (size_t)min(n0 + row, N - 1) * ldb + c * 8;
        }
        auto issue_tile = [&](int kt, int buf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int base = (i * 256 + wid * 64) * 16;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[i] + kt * BK), (lds_ptr_t)(As + buf * (BM * BK * 2) + base), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(gb[i] + kt * BK), (lds_ptr_t)(Bs + buf * (BN * BK * 2) + base), 16, 0, 0);
            }
        };
        issue_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) issue_tile(kt + 1, buf ^ 1);
            compute_tile(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
        load_global(0);
        store_lds(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_global((kt + 1) * BK);
            compute_tile(buf);
            if (kt + 1 < nk) store_lds(buf ^ 1);
            __syncthreads();
        }
    }

    // epilogue: stage 64 rows at a time through LDS (f32, row pitch 132 floats) and write whole rows
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CP = BN + 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = i * 16 + (lane >> 4) * 4 + r;
                        const int col = wn * 64 + j * 16 + (lane & 15);
                        Cs[row * CP + col] = acc[i][j][r];
                    }
        }
        __syncthreads();
        // 64 rows x 128 cols, 4 columns per thread-iteration
        for (int id = tid; id < 64 * (BN / 4); id += 256) {
            const int row = id / (BN / 4), c4 = (id - row * (BN / 4)) * 4;
            const int gm = m0 + half * 64 + row, gn = n0 + c4;
            if (gm >= M || gn >= N) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * CP + c4]);
            float vv[4] = {v.x, v.y, v.z, v.w};
            OutT* dst = C + (size_t)gm * ldc + gn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (gn + e < N) {
                    float x = vv[e];
                    if (bias) x += bias[gn + e];
                    vv[e] = x;
                }
            }
            if (gn + 3 < N && ((((uintptr_t)dst) & (sizeof(OutT) * 4 - 1)) == 0)) {
                if (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                } else {
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(vv[0]) | ((uint32_t)f32_to_bf16(vv[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(vv[2]) | ((uint32_t)f32_to_bf16(vv[3]) << 16);
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gn + e < N) store_out<OutT>(dst + e, vv[e]);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ NT, 256 x 128 tile
// At 128 x 128 x 64 every MFMA cycle needs 64 B/clk of operand fill per CU -- the whole L1-miss path -- so the K = 512
// projections of the model ran at ~450 TFLOP/s.  256 x 128 halves the fill per flop: 4 waves as 2 (M) x 2 (N), each
// 128 x 64 = 8 x 4 MFMA tiles (128 accumulator registers); BK = 32, two LDS stages of 24 KB filled by LDS-DMA and 168 VGPRs,
// so THREE workgroups share a CU and one's loads / prologue / epilogue hide behind another's MFMAs.  Measured at
// M=32000, N=3072, K=512 (f32 out): 3 stages x 2 workgroups per CU 480 TFLOP/s, 4 stages x 1 workgroup 301, 2 stages x 3
// workgroups 542 (8192^3: 957 / 537 / 1010): occupancy hides the fill latency better than pipeline depth.
// LDS image: rows of 64 B (32 k), 16-B chunk c of row r stored at position c ^ g[(r >> 2) & 3], g = {0, 2, 3, 1}: the
// ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...) then touch 16 distinct 16-B slots of the 256-B bank row.
constexpr int B2M = 256, B2N = 128, B2K = 32;

// Implicit-GEMM convolution (CONV = true): operand A is not a matrix but the activation tensor (Ts, B, Hs, Cs) bf16, and
// row r = (t, b, h) / column k = (kh, kw, c) of the virtual im2col matrix is gathered on the fly:
//   A[r][k] = x[t + sgn (kw - pt)][b][h + sgn (kh - ph)][c]      (zero outside the tensor)
// sgn = +1 with rows over the OUTPUT positions is the forward convolution (col . W^T); sgn = -1 with rows over the INPUT
// positions and x = the output gradient is the backward-data convolution.  Cs % 8 == 0 keeps every 16-B chunk inside one
// tap, so the LDS-DMA loader only changes its per-lane source address; out-of-range chunks read a page of zeros.
struct ConvDesc {
    int B, Hs, Cs, Ts, KH, KW, ph, pt, sgn, Hr;       // Hr: height of the row space
};
__device__ uint4 g_zero_page[2];
constexpr int NT2_STAGES = 2;
constexpr int NT2_LDS_BYTES = NT2_STAGES * (B2M + B2N) * B2K * 2;      // stages x 24 KiB

// WN = waves along N: 2 -> tile 256 x 128 (waves 2 x 2, 128 x 64 each); 1 -> tile 256 x 64 (waves 4 x 1, 64 x 64 each) for N <= 64
template <typename OutT, bool CONV, int WN>
__global__ __launch_bounds__(256, CONV ? 2 : 3) void gemm_nt256_kernel(const uint16_t* __restrict__ A, int lda,
                                                            const uint16_t* __restrict__ B, int ldb, OutT* __restrict__ C,
                                                            int ldc, const float* __restrict__ bias, int M, int N, int K,
                                                            int tiles_n, ConvDesc cd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tm, tn;
    tile_of(blockIdx.x, (M + B2M - 1) / B2M, tiles_n, tm, tn);
    constexpr int TN = 64 * WN, MI = 4 * WN;        // tile columns, 16-row MFMA tiles per wave along M
    const int m0 = tm * B2M, n0 = tn * TN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = WN == 2 ? wid >> 1 : wid, wn = WN == 2 ? wid & 1 : 0;
    constexpr int A_STAGE = B2M * B2K * 2, B_STAGE = TN * B2K * 2;
    char* As = smem;                           // [3][256 rows][64 B]
    char* Bs = smem + NT2_STAGES * A_STAGE;    // [3][128 rows][64 B]
    auto g4 = [](int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; };      // g = {0, 2, 3, 1}

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS-DMA sources: slot s of an image = (row s / 4, position s % 4) holds chunk (s % 4) ^ g(row); a wave-instruction
    // fills 64 consecutive slots.  A: 1024 slots = 4 per thread, B: 512 = 2 per thread.
    const uint16_t* ga[4];
    const uint16_t* gb[WN];
    int cth[4];                            // CONV: (t << 8) | h of the slot's row; rows beyond M get a t far below zero
    // CONV with Cs % 32 == 0 (every layer but the first): a 32-wide K step lies inside ONE tap, the same for every row, so
    // the tap walk (kh, kw, channel base) is scalar state advanced once per issued tile, and a lane only adds a uniform
    // offset to its own row pointer and checks two ranges.  (The general form below decomposes k per lane and per piece:
    // two integer divisions by runtime values, ~60 VALU instructions per LDS-DMA piece -- more issue time than the tile's
    // MFMAs, which is what held the implicit convolutions near 500 TFLOP/s.)
    const bool tap_uniform = CONV && (cd.Cs % B2K) == 0;
    int tw_kh = 0, tw_kw = 0, tw_ci = 0;   // tap walk of the NEXT tile to issue (tiles are issued in k order)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sl = i * 256 + tid, row = sl >> 2, c = (sl & 3) ^ g4(row);
        if (CONV) {
            const int r = m0 + row;
            const int rr = r < M ? r : 0;
            const int h = rr % cd.Hr, tb = rr / cd.Hr, b = tb % cd.B, t = r < M ? tb / cd.B : -(1 << 20);
            cth[i] = t * 256 + h;
            ga[i] = A + ((size_t)b * cd.Hs) * cd.Cs;                 // + ((ti * B) * Hs + hi) * Cs + ci per tile
            if (tap_uniform && r < M) ga[i] += ((long long)t * cd.B * cd.Hs + h) * cd.Cs + c * 8;      // the row's own (t, h), this lane's chunk
        } else {
            ga[i] = A + (size_t)min(m0 + row, M - 1) * lda + c * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int sl = i * 256 + tid, row = sl >> 2, c = (sl & 3) ^ g4(row);
        gb[i] = B + (size_t)min(n0 + row, N - 1) * ldb + c * 8;
    }
    auto issue_tile = [&](int kt, int buf) {
        // uniform part of the fast convolution form
        int dt = 0, dh = 0;
        long long delta = 0;
        bool tap_ok = false;
        if (tap_uniform) {
            dt = cd.sgn * (tw_kw - cd.pt);
            dh = cd.sgn * (tw_kh - cd.ph);
            delta = ((long long)dt * cd.B * cd.Hs + dh) * cd.Cs + tw_ci;
            tap_ok = tw_kh < cd.KH;                                  // (K may be padded with empty taps)
            tw_ci += B2K;
            if (tw_ci >= cd.Cs) { tw_ci = 0; if (++tw_kw == cd.KW) { tw_kw = 0; ++tw_kh; } }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint16_t* src;
            if (CONV && tap_uniform) {
                const int ti = (cth[i] >> 8) + dt, hi = (cth[i] & 255) + dh;
                const bool ok = tap_ok && (unsigned)ti < (unsigned)cd.Ts && (unsigned)hi < (unsigned)cd.Hs;
                src = ok ? ga[i] + delta : reinterpret_cast<const uint16_t*>(g_zero_page);
            } else if (CONV) {
                const int sl = i * 256 + tid, row = sl >> 2;
                const int k = kt * B2K + ((sl & 3) ^ g4(row)) * 8;
                const int tap = k / cd.Cs, ci = k - tap * cd.Cs;
                const int kh = tap / cd.KW, kw = tap - kh * cd.KW;
                const int ti = (cth[i] >> 8) + cd.sgn * (kw - cd.pt), hi = (cth[i] & 255) + cd.sgn * (kh - cd.ph);
                const bool ok = ti >= 0 && ti < cd.Ts && hi >= 0 && hi < cd.Hs && kh < cd.KH;      // (K may be padded with empty taps)
                src = ok ? ga[i] + ((size_t)ti * cd.B * cd.Hs + hi) * cd.Cs + ci : reinterpret_cast<const uint16_t*>(g_zero_page);
            } else {
                src = ga[i] + kt * B2K;
            }
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(As + buf * A_STAGE + (i * 256 + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WN; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(gb[i] + kt * B2K), (lds_ptr_t)(Bs + buf * B_STAGE + (i * 256 + wid * 64) * 16), 16, 0, 0);
    };
    // fragment addresses (bytes inside a stage) are loop invariant
    int aoff[MI], boff[4];
    {
        const int q = lane >> 4, r = lane & 15;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wm * (16 * MI) + i * 16 + r;
            aoff[i] = row * 64 + ((q ^ g4(row)) << 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wn * 64 + j * 16 + r;
            boff[j] = row * 64 + ((q ^ g4(row)) << 4);
        }
    }
    // NT2_STAGES stages, NT2_STAGES - 1 tiles in flight: by Little's law 48 B/clk of fill at ~2000 cycles of latency is
    // ~96 KB per CU.  Tile kt + S - 1 is issued while tile kt is computed; before the barrier each wave waits until only
    // the DMAs of the tiles behind kt + 1 are outstanding (6 per tile and wave).  Raw s_barrier: __syncthreads() would
    // drain the DMAs in flight.
    constexpr int S = NT2_STAGES;
    auto wait_tiles_in_flight = [](int tiles) {          // vmcnt needs an immediate: 4 + WN DMAs per tile and thread
        if (tiles >= 3) { if (WN == 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); }
        else if (tiles == 2) { if (WN == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
        else if (tiles == 1) { if (WN == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int nk = K / B2K;
    for (int t = 0; t < S - 1 && t < nk; ++t) issue_tile(t, t);
    wait_tiles_in_flight(min(S - 2, nk - 1));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + S - 1 < nk) issue_tile(kt + S - 1, buf == 0 ? S - 1 : buf - 1);
        const char* Ab = As + buf * A_STAGE;
        const char* Bb = Bs + buf * B_STAGE;
        Frag b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j].u = *reinterpret_cast<const uint4*>(Bb + boff[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            Frag a;
            a.u = *reinterpret_cast<const uint4*>(Ab + aoff[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = ASR_MFMA_16x16x32(a.v, b[j].v, acc[i][j]);
        }
        wait_tiles_in_flight(min(S - 2, nk - 2 - kt));          // tiles kt + 2 .. may stay in flight, tile kt + 1 has landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        buf = buf == S - 1 ? 0 : buf + 1;
    }

    // epilogue: 64 rows at a time through LDS (f32, row pitch 132 floats = 33 KB) and whole rows out
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CP = TN + 4;
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {
        if (wm == (WN == 2 ? chunk >> 1 : chunk)) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = ii * 16 + (lane >> 4) * 4 + r;
                        const int col = wn * 64 + j * 16 + (lane & 15);
                        Cs[row * CP + col] = acc[(WN == 2 ? (chunk & 1) * 4 : 0) + ii][j][r];
                    }
        }
        __syncthreads();
        for (int id = tid; id < 64 * (TN / 4); id += 256) {
            const int row = id / (TN / 4), c4 = (id - row * (TN / 4)) * 4;
            const int gm = m0 + chunk * 64 + row, gn = n0 + c4;
            if (gm >= M || gn >= N) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * CP + c4]);
            float vv[4] = {v.x, v.y, v.z, v.w};
            OutT* dst = C + (size_t)gm * ldc + gn;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (bias && gn + e < N) vv[e] += bias[gn + e];
            if (gn + 3 < N && ((((uintptr_t)dst) & (sizeof(OutT) * 4 - 1)) == 0)) {
                if (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                } else {
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(vv[0]) | ((uint32_t)f32_to_bf16(vv[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(vv[2]) | ((uint32_t)f32_to_bf16(vv[3]) << 16);
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gn + e < N) store_out<OutT>(dst + e, vv[e]);
            }
        }
        __syncthreads();
    }
}

// (a plain device function: called directly from a lambda of the kernel template, the builtin made the host pass drop the kernel's
// stub without a diagnostic)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, unsigned voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds, 16, voffset, soffset, 0, 0);
}

// ------------------------------------------------------------------------------------------------ NT, 256 x 128, persistent
// The K = 512 / 1024 projections of the model spend half of a tile's life outside its K loop: 16 K steps of fill + MFMA between a
// cold prologue (nothing to compute until the first stage lands) and an epilogue that moves 64 KB of results through LDS behind
// eight workgroup barriers -- and the three workgroups of a CU, launched together with equal work, go through those phases together
// (the same kernel reaches 1010 TFLOP/s at K = 8192 and 510-540 at K = 512).  Here a workgroup walks over its tiles (bid, bid + grid,
// ...) in ONE software pipeline: the K loop runs across tile boundaries (stage 0 of the next tile is asked for during the last K step
// of this one), and the epilogue goes from the accumulators straight to memory, without LDS (which by then belongs to the next tile)
// and without a barrier.  Stores and LDS-DMA loads share vmcnt: the wait at the end of the next tile's first K step covers both.
//
// Register epilogue with whole-line stores.  A wave owns 64 rows x all 128 columns (4 x 8 MFMA tiles).  Column index c of MFMA tile j
// stands for the ACTUAL column 8 c + j: the loader puts global row n0 + 8 (rho & 15) + (rho >> 4) of B at LDS row rho (an LDS-DMA lane
// chooses its source freely), so the fragment reads are the plain ones, and lane (q, c) ends up with columns 8 c .. 8 c + 7 of rows
// 4 q + reg: one 16-B bf16 store per row, 16 adjacent lanes = 256 contiguous bytes (f32: runs of four columns, two stores).  (What-if timings on 32000 x 3072 x 512: no stores
// 0.106 ms, stores alone 0.073 ms when every lane wrote 8 B with the rows of a lane quad 6 KB apart: a request-rate bound.)
// CONV: operand A is the activation tensor of an implicit convolution (ConvDesc; Cs % 32 == 0, so a K step lies inside one tap
// and the tap walk is scalar state of the issuing side); a row outside the tensor or an empty tap gets a voffset beyond the
// buffer's num_records -- the buffer load then writes zeros, no zero page and no select between pointers.
// NJ: column tiles per wave = tile width / 16: 8 -> 256 x 128, 4 -> 256 x 64 (N <= 64: no MFMA work on columns that do not exist).
// KT: K is a multiple of 8 but not of the 32-wide K step (the logit gradient's K = V = 3000): the chunks of the last step that lie
// beyond K are fetched from beyond the buffers' ends, i.e. as zeros -- both operands, so whatever follows a row never meets a number.
template <typename OutT, int CONV, int NJ, bool KT = false>
__global__ __launch_bounds__(256, CONV ? 2 : 3) void gemm_nt256p_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ B,
                                                                       int ldb, OutT* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                                       int M, int N, int K, int tiles_m, int tiles_n, int total,
                                                                       unsigned a_bytes, ConvDesc cd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int TNW = 16 * NJ;                                        // tile width
    constexpr int A_STAGE = B2M * B2K * 2, B_STAGE = TNW * B2K * 2;
    constexpr int BP = NJ / 4;                                          // LDS-DMA pieces of the B tile per thread
    // a lane's run of consecutive columns: 16 B of the output type (f32: two runs of four, 64 columns apart, so that the 16 lanes
    // of a row still write 256 contiguous bytes per store)
    constexpr int G = (sizeof(OutT) == 4 && NJ == 8) ? 4 : NJ;
    char* As = smem;
    char* Bs = smem + 2 * A_STAGE;
    auto g4 = [](int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; };      // g = {0, 2, 3, 1}
    const int nk = KT ? (K + B2K - 1) / B2K : K / B2K;
    // Tile walk: XCD x (= bid % 8 under round-robin dispatch; only a locality hint) owns the CONTIGUOUS tiles [x chunk, (x + 1) chunk),
    // its workgroups stride through them together: ~64-96 consecutive row panels in flight per L2.  For the implicit convolutions that
    // is what keeps the taps' re-reads of the activations (rows of neighbouring time steps = the neighbouring panels) inside one L2:
    // dealt panel by panel over the XCDs, every L2 fetched nearly the whole tensor (PMC FETCH_SIZE of the 64 -> 128 convolution of the
    // BASELINE model: 710 MB for a 53 MB input, backward-data 1333 MB for 106 MB).  Plain products: a row panel of A and all its column
    // tiles stay on one XCD as before.
    const int chunk = (total + 7) >> 3, stride = gridDim.x >> 3;
    const int limit = min(total, ((int)(blockIdx.x & 7) + 1) * chunk), first = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);

    // loader: slot sl = i * 256 + tid = (LDS row i * 64 + tid / 4, position tid % 4); the chunk it holds does not depend on i
    const int lrow = tid >> 2, lchunk = ((tid & 3) ^ g4(lrow)) * 16;
    unsigned oa[4], ob[BP];                // BYTE offsets of this lane's rows in the tile being ISSUED (uniform base + 32-bit lane offset)
    int cth[4];                            // CONV: (t << 8) | h of the row; rows beyond M get a t far below zero
    int it_tile = first, it_k = 0;
    int tw_kh = 0, tw_kw = 0, tw_ci = 0;   // CONV: tap of the next K step to issue (CONV == 2: of this lane's chunk of that step)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, KT ? N * ldb * 2 : -1, 0x00020000);
    auto set_issue = [&](int t) {
        int tm, tn;
        tm = t / tiles_n;
        tn = t - tm * tiles_n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (CONV) {
                const int rr = tm * B2M + i * 64 + lrow;
                const int rc = rr < M ? rr : 0;
                const int h = rc % cd.Hr, tb = rc / cd.Hr, b = tb % cd.B, t0 = rr < M ? tb / cd.B : -(1 << 20);
                cth[i] = t0 * 256 + h;
                oa[i] = (unsigned)(((tb / cd.B) * cd.B + b) * cd.Hs + h) * (unsigned)(cd.Cs * 2) + (CONV == 2 ? 0 : lchunk);
            } else {
                oa[i] = __umul24((unsigned)min(tm * B2M + i * 64 + lrow, M - 1), (unsigned)(lda * 2)) + lchunk;      // (24-bit factors: a 32-bit multiply-add, no 64-bit pair)
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int rho = i * 64 + lrow;
            const int jj = rho >> 4;                                     // LDS row 16 j + c holds column (j / G) 16 G + G c + j % G
            ob[i] = __umul24((unsigned)min(tn * TNW + (jj / G) * (16 * G) + G * (rho & 15) + (jj % G), N - 1), (unsigned)(ldb * 2)) + lchunk;
        }
        tw_kh = tw_kw = tw_ci = 0;
        if (CONV == 2) { const int c = lchunk >> 4; tw_kh = c / cd.KW; tw_kw = c - tw_kh * cd.KW; }
    };
    auto issue_next = [&](int buf) {
        const int koff = it_k * (B2K * 2);      // scalar offset of the buffer instruction: no per-lane address arithmetic at all
        if (CONV) {
            const int dt = cd.sgn * (tw_kw - cd.pt), dh = cd.sgn * (tw_kh - cd.ph);
            const int delta = ((dt * cd.B * cd.Hs + dh) * cd.Cs + tw_ci) * 2;
            const bool tap_ok = tw_kh < cd.KH;                        // (K may be padded with empty taps)
            if (CONV == 2) {                                          // this lane's chunk moves on by the four taps of a K step
                tw_kw += 4;
                while (tw_kw >= cd.KW) { tw_kw -= cd.KW; ++tw_kh; }
            } else {
                tw_ci += B2K;
                if (tw_ci >= cd.Cs) { tw_ci = 0; if (++tw_kw == cd.KW) { tw_kw = 0; ++tw_kh; } }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ti = (cth[i] >> 8) + dt, hi = (cth[i] & 255) + dh;
                const bool ok = tap_ok && (unsigned)ti < (unsigned)cd.Ts && (unsigned)hi < (unsigned)cd.Hs;
                lds_dma16(rsrc_a, As + buf * A_STAGE + (i * 256 + wid * 64) * 16, ok ? oa[i] + delta : 0xfffffff0u, 0);
            }
        } else {
            const bool dead = KT && it_k == nk - 1 && it_k * B2K + (lchunk >> 1) >= K;      // this lane's chunk of the last K step lies beyond K
#pragma unroll
            for (int i = 0; i < 4; ++i)
                lds_dma16(rsrc_a, As + buf * A_STAGE + (i * 256 + wid * 64) * 16, dead ? 0xfffffff0u : oa[i], koff);
        }
        {
            const bool dead = KT && it_k == nk - 1 && it_k * B2K + (lchunk >> 1) >= K;
#pragma unroll
            for (int i = 0; i < BP; ++i)
                lds_dma16(rsrc_b, Bs + buf * B_STAGE + (i * 256 + wid * 64) * 16, dead ? 0xfffffff0u : ob[i], koff);
        }
        if (++it_k == nk) {
            it_k = 0;
            it_tile += stride;
            if (it_tile < limit) set_issue(it_tile);
        }
    };
    // fragment addresses: row = (wave base) + i * 16 + r, and the swizzle key (row >> 2) & 3 does not depend on i: one register per
    // operand, the tile index is an immediate offset of the ds_read
    const int q = lane >> 4, r = lane & 15;
    const int aoff0 = (wid * 64 + r) * 64 + ((q ^ g4(r)) << 4);
    const int boff0 = r * 64 + ((q ^ g4(r)) << 4);
    if (it_tile >= limit) return;
    set_issue(it_tile);
    issue_next(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int buf = 0;
    for (int tile = first; tile < limit; tile += stride) {
        f32x4 acc[4][NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            if (it_tile < limit) issue_next(buf ^ 1);
            const char* Ab = As + buf * A_STAGE;
            const char* Bb = Bs + buf * B_STAGE;
            // B fragment j + 1 is read BEFORE the four MFMAs of fragment j are issued (the sched_group_barriers pin that order: left
            // alone, the register-starved schedule reuses one fragment register and every read waits out its full LDS latency)
            Frag a[4], b[2];
            b[0].u = *reinterpret_cast<const uint4*>(Bb + boff0);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i].u = *reinterpret_cast<const uint4*>(Ab + aoff0 + i * 1024);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j + 1 < NJ) b[(j + 1) & 1].u = *reinterpret_cast<const uint4*>(Bb + boff0 + (j + 1) * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = ASR_MFMA_16x16x32(a[i].v, b[j & 1].v, acc[i][j]);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);      // b0, a0..a3, b1
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (j + 2 < NJ) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            buf ^= 1;
        }
        // epilogue: acc[i][j][reg] = C[m0 + wid * 64 + i * 16 + 4 q + reg][n0 + (j / G) 16 G + G r + j % G]
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int col = tn * TNW + G * r;
        const int row0 = tm * B2M + wid * 64 + 4 * q;
        float bv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cj = col + (j / G) * (16 * G) + (j % G);
            bv[j] = (bias && cj < N) ? bias[cj] : 0.f;
        }
        OutT* d0 = C + (size_t)row0 * ldc + col;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (row0 + i * 16 + reg >= M) continue;
                OutT* dst = d0 + (size_t)(i * 16 + reg) * ldc;
#pragma unroll
                for (int g = 0; g < NJ / G; ++g) {
                    float v[G];
#pragma unroll
                    for (int j = 0; j < G; ++j) v[j] = acc[i][g * G + j][reg] + bv[g * G + j];
                    const int cg = col + g * (16 * G);
                    if (cg + G - 1 < N) {
                        if (sizeof(OutT) == 4) {
                            *reinterpret_cast<float4*>(dst + g * (16 * G)) = make_float4(v[0], v[1], v[2], v[3]);
                        } else if (G == 8) {
                            uint4 pk;
                            pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                            pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                            pk.z = (uint32_t)f32_to_bf16(v[G == 8 ? 4 : 0]) | ((uint32_t)f32_to_bf16(v[G == 8 ? 5 : 0]) << 16);
                            pk.w = (uint32_t)f32_to_bf16(v[G == 8 ? 6 : 0]) | ((uint32_t)f32_to_bf16(v[G == 8 ? 7 : 0]) << 16);
                            *reinterpret_cast<uint4*>(dst) = pk;
                        } else {
                            uint2 pk;
                            pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                            pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                            *reinterpret_cast<uint2*>(dst) = pk;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            if (cg + j < N) store_out<OutT>(dst + g * (16 * G) + j, v[j]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // one row at a time: keeps the conversion temporaries of 16 rows from piling up
            }
    }
}

// ------------------------------------------------------------------------------------------------ NT, 256 x 256 tile
// The 256 x 128 kernel above is bound by operand fill, not by the matrix pipes (SQ counters on the model's projections:
// waves parked on vmcnt / barriers half of their life, MFMA pipes 22 % busy): a 256 x 128 x 32 step moves 24 KB from L2
// into LDS for 128 MFMAs.  Here EIGHT waves -- 2 (M) x 4 (N), each the same 128 x 64 = 8 x 4 MFMA tiles -- share a
// 256 x 256 tile: 32 KB per 256 MFMAs, two thirds of the fill per flop (and for N <= 256 -- the convolutions of the recipes
// -- the activation operand is streamed ONCE instead of once per column tile); every wave issues 4 LDS-DMA pieces per 32
// MFMAs instead of 6.  One workgroup per CU, two waves per SIMD, STAGES-deep ring (STAGES - 1 tiles in flight).
// BKT = 32: LDS rows of 64 B, swizzle as above.  BKT = 64: rows of 128 B = whole cache lines per row and K step, 16-B chunk
// c of row r at position c ^ (r & 7) (the ds_read_b128 lane groups then cover all 64 banks).
template <int N_>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <typename OutT, bool CONV, int BKT, int STAGES, int WN_>
__global__ __launch_bounds__(128 * WN_) void gemm_nt_wide_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ B,
                                                           int ldb, OutT* __restrict__ C, int ldc, const float* __restrict__ bias,
                                                           int M, int N, int K, int tiles_n, ConvDesc cd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TM = 256, TNC = 256, THREADS = 128 * WN_, MI = 8, NJ = 16 / WN_;        // waves 2 (M) x WN_ (N), wave tile 128 x (256 / WN_)
    constexpr int CH = BKT / 8;                         // 16-B chunks per LDS row
    constexpr int ROWB = BKT * 2;                       // bytes per LDS row
    constexpr int A_SLOTS = TM * CH / THREADS, B_SLOTS = TNC * CH / THREADS, P = A_SLOTS + B_SLOTS;
    constexpr int A_STAGE = TM * ROWB, B_STAGE = TNC * ROWB;
    int tm, tn;
    if (CONV) {
        // contiguous tile ranges per XCD (bid % 8), walked in dispatch order: the taps' re-reads of neighbouring time steps stay in
        // one L2 (see gemm_nt256p_kernel); the grid is 8 x ceil(tiles / 8), the surplus workgroups leave here
        const int tn_ = tiles_n < 0 ? -tiles_n : tiles_n, total = ((M + TM - 1) / TM) * tn_, chunk = (total + 7) >> 3;
        const int x = blockIdx.x & 7, t = x * chunk + (int)(blockIdx.x >> 3);
        if (t >= min(total, (x + 1) * chunk)) return;
        tm = t / tn_;
        tn = t - tm * tn_;
    } else {
        tile_of(blockIdx.x, (M + TM - 1) / TM, tiles_n, tm, tn);
    }
    const int m0 = tm * TM, n0 = tn * TNC;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN_, wn = wid % WN_;
    char* As = smem;
    char* Bs = smem + STAGES * A_STAGE;
    auto swz = [](int row) { return BKT == 32 ? ((0x78 >> (((row >> 2) & 3) * 2)) & 3) : (row & 7); };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const uint16_t* ga[A_SLOTS];
    const uint16_t* gb[B_SLOTS];
    int cth[A_SLOTS];
    int tw_kh = 0, tw_kw = 0, tw_ci = 0;
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int sl = i * THREADS + tid, row = sl / CH, c = (sl % CH) ^ swz(row);
        if (CONV) {
            const int r = m0 + row;
            const int rr = r < M ? r : 0;
            const int h = rr % cd.Hr, tb = rr / cd.Hr, b = tb % cd.B, t = r < M ? tb / cd.B : -(1 << 20);
            cth[i] = t * 256 + h;
            ga[i] = A + ((size_t)b * cd.Hs) * cd.Cs;
            if (r < M) ga[i] += ((long long)t * cd.B * cd.Hs + h) * cd.Cs + c * 8;
        } else {
            ga[i] = A + (size_t)min(m0 + row, M - 1) * lda + c * 8;
        }
    }
#pragma unroll
    for (int i = 0; i < B_SLOTS; ++i) {
        const int sl = i * THREADS + tid, row = sl / CH, c = (sl % CH) ^ swz(row);
        gb[i] = B + (size_t)min(n0 + row, N - 1) * ldb + c * 8;
    }
    auto issue_tile = [&](int kt, int buf) {
        int dt = 0, dh = 0;
        long long delta = 0;
        bool tap_ok = false;
        if (CONV) {
            dt = cd.sgn * (tw_kw - cd.pt);
            dh = cd.sgn * (tw_kh - cd.ph);
            delta = ((long long)dt * cd.B * cd.Hs + dh) * cd.Cs + tw_ci;
            tap_ok = tw_kh < cd.KH;
            tw_ci += BKT;
            if (tw_ci >= cd.Cs) { tw_ci = 0; if (++tw_kw == cd.KW) { tw_kw = 0; ++tw_kh; } }
        }
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const uint16_t* src;
            if (CONV) {
                const int ti = (cth[i] >> 8) + dt, hi = (cth[i] & 255) + dh;
                const bool ok = tap_ok && (unsigned)ti < (unsigned)cd.Ts && (unsigned)hi < (unsigned)cd.Hs;
                src = ok ? ga[i] + delta : reinterpret_cast<const uint16_t*>(g_zero_page);
            } else {
                src = ga[i] + kt * BKT;
            }
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(As + buf * A_STAGE + (i * THREADS + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_SLOTS; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(gb[i] + kt * BKT), (lds_ptr_t)(Bs + buf * B_STAGE + (i * THREADS + wid * 64) * 16), 16, 0, 0);
    };
    // fragment byte offsets inside a stage, per 32-wide K slice
    int aoff[MI], boff[NJ];
    {
        const int q = lane >> 4, r = lane & 15;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = wm * 128 + i * 16 + r;
            aoff[i] = row * ROWB + ((q ^ swz(row)) << 4);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int row = wn * (16 * NJ) + j * 16 + r;
            boff[j] = row * ROWB + ((q ^ swz(row)) << 4);
        }
    }
    auto wait_tiles_in_flight = [](int tiles) {
        if (tiles >= 3) wait_vmcnt<3 * P>();
        else if (tiles == 2) wait_vmcnt<2 * P>();
        else if (tiles == 1) wait_vmcnt<P>();
        else wait_vmcnt<0>();
    };
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int S = STAGES;
    const int nk = K / BKT;
    for (int t = 0; t < S - 1 && t < nk; ++t) issue_tile(t, t);
    wait_tiles_in_flight(min(S - 2, nk - 1));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + S - 1 < nk) issue_tile(kt + S - 1, buf == 0 ? S - 1 : buf - 1);
        const char* Ab = As + buf * A_STAGE;
        const char* Bb = Bs + buf * B_STAGE;
#pragma unroll
        for (int ks = 0; ks < BKT / 32; ++ks) {
            // second 32-wide slice of a 128-B row: chunk index + 4, i.e. the position XORed with 4 (the swizzle only
            // touches the low bits it was given)
            const int kx = ks << 6;
            Frag b[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j].u = *reinterpret_cast<const uint4*>(Bb + (boff[j] ^ kx));
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                Frag a;
                a.u = *reinterpret_cast<const uint4*>(Ab + (aoff[i] ^ kx));
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = ASR_MFMA_16x16x32(a.v, b[j].v, acc[i][j]);
            }
        }
        wait_tiles_in_flight(min(S - 2, nk - 2 - kt));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        buf = buf == S - 1 ? 0 : buf + 1;
    }

    // epilogue: 32 rows at a time through LDS (f32, row pitch 260 floats = 33 KB) and whole rows out
    float* Cs = reinterpret_cast<float*>(smem);
    constexpr int CP = TNC + 4, ER = 32;
#pragma unroll
    for (int chunk = 0; chunk < TM / ER; ++chunk) {
        if (wm == chunk / 4) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = ii * 16 + (lane >> 4) * 4 + r;
                        const int col = wn * (16 * NJ) + j * 16 + (lane & 15);
                        Cs[row * CP + col] = acc[(chunk & 3) * 2 + ii][j][r];
                    }
        }
        __syncthreads();
        for (int id = tid; id < ER * (TNC / 4); id += THREADS) {
            const int row = id / (TNC / 4), c4 = (id - row * (TNC / 4)) * 4;
            const int gm = m0 + chunk * ER + row, gn = n0 + c4;
            if (gm >= M || gn >= N) continue;
            float4 v = *reinterpret_cast<const float4*>(&Cs[row * CP + c4]);
            float vv[4] = {v.x, v.y, v.z, v.w};
            OutT* dst = C + (size_t)gm * ldc + gn;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (bias && gn + e < N) vv[e] += bias[gn + e];
            if (gn + 3 < N && ((((uintptr_t)dst) & (sizeof(OutT) * 4 - 1)) == 0)) {
                if (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                } else {
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(vv[0]) | ((uint32_t)f32_to_bf16(vv[1]) << 16);
                    pk.y = (uint32_t)f32_to_bf16(vv[2]) | ((uint32_t)f32_to_bf16(vv[3]) << 16);
                    *reinterpret_cast<uint2*>(dst) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gn + e < N) store_out<OutT>(dst + e, vv[e]);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ TN
constexpr int TK = 32;               // k rows per LDS tile (one MFMA K step)
constexpr int TP = BM + 16;          // padded row pitch in elements (288 B): 8 consecutive rows cover all 64 banks
constexpr int TN_LDS_BYTES = 2 * 2 * TK * TP * 2;   // A and B tiles, double buffered = 36 KiB

__device__ __forceinline__ bf16x4 lds_tr16(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}

// CONV: operand B is the virtual im2col matrix of the activation tensor (see ConvDesc; sgn = +1): row = reduction index
// (t, b, h) over the output positions, column = (kh, kw, c) -- the weight gradient of a convolution without a column
// matrix in memory (1.2 GB written and read back per step for the second conv layer otherwise).
// Several products in one launch (the weight gradients a recurrence boundary releases together: dW_ih 3072 x 512 and the two
// directions' dW_hh 1536 x 512, K = 32000).  Launched one by one each needs 8 .. 16 K splits to fill the chip and a workgroup's
// fixed cost -- cold prologue, 16 K atomics -- is then a third of its life (time = 54 us per 2000 k rows + 47 us at 768
// workgroups); together they are 192 tiles, which fill the chip at 4 .. 6 splits.  Alone 4 splits are fastest (0.286 ms against
// 0.316 for the three launches, 0.349 at 6 splits); in the train step 6 are (14.73 against 14.80 ms: the next recurrence wants
// every CU and waits for the last product workgroup, so short-lived workgroups hand the chip over sooner).  n == 0: the plain call.
struct TnProb {
    const uint16_t* A;
    const uint16_t* B;
    float* C;
    int lda, ldb, ldc, M, N, K, tiles_n;
    int tile_end;                        // tiles of this and all earlier products
};
struct TnGroup {
    TnProb p[4];
    int n;
};

template <bool CONV>
__global__ __launch_bounds__(256, 4) void gemm_tn_kernel(const uint16_t* __restrict__ A, int lda,
                                                      const uint16_t* __restrict__ B, int ldb, float* __restrict__ C,
                                                      int ldc, int M, int N, int K, int tiles_n, int k_per_split,
                                                      ConvDesc cd, long long copy_stride, TnGroup grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* As = reinterpret_cast<uint16_t*>(smem);             // [2][TK][TP]
    uint16_t* Bs = As + 2 * TK * TP;                              // [2][TK][TP]
    // 1-D grid, XCD-aware: workgroup L runs on XCD L % 8 (round-robin dispatch); an XCD takes a contiguous run of the
    // split-major (split, tile) pairs -- whole K splits where the split count is a multiple of 8 -- so the rows of A and B of a
    // split are fetched into ONE L2 and shared by all the tiles there (tile-major ids put every column block on its own XCD: each
    // L2 fetched all of A).  Only a locality hint: any placement computes the same thing.
    const bool grouped = !CONV && grp.n > 0;
    const int tiles = grouped ? grp.p[grp.n - 1].tile_end : ((M + BM - 1) / BM) * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    C += (size_t)xcd * copy_stride;      // (one copy of the output per XCD when hundreds of K splits share one tile: asr_conv_tn_copies)
    // (split, tile) pairs in split-major order, dealt to the XCDs in eight contiguous runs of gridDim.x / 8
    const int item = xcd * (int)(gridDim.x >> 3) + slot;
    const int round = item / tiles;
    int bid = item - round * tiles;
    if (grouped) {                       // uniform over the workgroup
        int which = 0;
        while (which + 1 < grp.n && bid >= grp.p[which].tile_end) ++which;
        if (which > 0) bid -= grp.p[which - 1].tile_end;
        const TnProb& q = grp.p[which];
        A = q.A; B = q.B; C = q.C; lda = q.lda; ldb = q.ldb; ldc = q.ldc; M = q.M; N = q.N; K = q.K; tiles_n = q.tiles_n;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = round * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    if (kbeg >= kend) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // a tile is 32 rows x 128 elements = 32 x 16 chunks of 16 B -> 2 chunks per thread per operand
    uint4 ra[2], rb[2];
    const bool a_vec = (lda & 7) == 0, b_vec = (ldb & 7) == 0;
    auto load_one = [&](const uint16_t* P, int ld, bool vec, int gk, int gc, int lim) -> uint4 {
        if (gk >= kend) return make_uint4(0, 0, 0, 0);
        const uint16_t* src = P + (size_t)gk * ld + gc;
        if (vec && gc + 8 <= lim && ((((uintptr_t)src) & 15) == 0)) return ld16(src);
        union { uint4 u; uint16_t s[8]; } t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t.s[e] = (gc + e < lim) ? src[e] : (uint16_t)0;
        return t.u;
    };
    // CONV: a thread's column chunk (tid & 15) never changes: its tap and channel are loop invariant
    int cv_dt = 0, cv_dh = 0, cv_ci = 0;
    bool cv_ok = false;
    if (CONV) {
        const int k = n0 + (tid & 15) * 8;
        const int tap = k / cd.Cs, kh = tap / cd.KW, kw = tap - kh * cd.KW;
        cv_ci = k - tap * cd.Cs;
        cv_dt = kw - cd.pt;
        cv_dh = kh - cd.ph;
        cv_ok = k < N && kh < cd.KH;
    }
    // CONV: the (t, b, h) of this thread's two reduction rows, decomposed ONCE and then stepped by the k tile (32 rows) with
    // single carries: three integer divisions by runtime values per load were more issue time than the tile's 16 MFMAs
    int cv_t[2] = {0, 0}, cv_b[2] = {0, 0}, cv_h[2] = {0, 0};
    int inc_h = 0, inc_b = 0, inc_t = 0;
    if (CONV) {
        inc_h = TK % cd.Hr;
        const int inc_tb = TK / cd.Hr;
        inc_b = inc_tb % cd.B;
        inc_t = inc_tb / cd.B;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gk = kbeg + ((tid + i * 256) >> 4);
            cv_h[i] = gk % cd.Hr;
            const int tb = gk / cd.Hr;
            cv_b[i] = tb % cd.B;
            cv_t[i] = tb / cd.B;
        }
    }
    auto load_conv = [&](int gk, int i) -> uint4 {
        const int t = cv_t[i], b = cv_b[i], h = cv_h[i];
        {   // advance to the same row of the next k tile
            int hh = h + inc_h;
            const int c1 = hh >= cd.Hr;
            hh -= c1 ? cd.Hr : 0;
            int bb = b + inc_b + c1;
            const int c2 = bb >= cd.B;
            bb -= c2 ? cd.B : 0;
            cv_h[i] = hh; cv_b[i] = bb; cv_t[i] = t + inc_t + c2;
        }
        if (gk >= kend || !cv_ok) return make_uint4(0, 0, 0, 0);
        const int ti = t + cv_dt, hi = h + cv_dh;
        if (ti < 0 || ti >= cd.Ts || hi < 0 || hi >= cd.Hs) return make_uint4(0, 0, 0, 0);
        return ld16(B + (((size_t)ti * cd.B + b) * cd.Hs + hi) * cd.Cs + cv_ci);
    };
    auto load_global = [&](int k0) {        // called with k0 = kbeg, kbeg + TK, ... in order (the CONV row walk relies on it)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256;
            const int row = id >> 4, c = id & 15;
            ra[i] = load_one(A, lda, a_vec, k0 + row, m0 + c * 8, M);
            rb[i] = CONV ? load_conv(k0 + row, i) : load_one(B, ldb, b_vec, k0 + row, n0 + c * 8, N);
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256;
            const int row = id >> 4, c = id & 15;
            *reinterpret_cast<uint4*>(As + (buf * TK + row) * TP + c * 8) = ra[i];
            *reinterpret_cast<uint4*>(Bs + (buf * TK + row) * TP + c * 8) = rb[i];
        }
    };

    // transpose-read addressing: 16-lane group g = lane >> 4, lane 4q+p of the group points at row q, columns 4p..4p+3;
    // first read covers k rows 4g..4g+3, second read 16+4g..16+4g+3 (k slots permuted identically for A and B)
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int trow0 = 4 * g + q, trow1 = 16 + 4 * g + q, tcol = 4 * p;

    const int nk = (kend - kbeg + TK - 1) / TK;
    load_global(kbeg);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_global(kbeg + (kt + 1) * TK);
        const uint16_t* Ab = As + buf * TK * TP;
        const uint16_t* Bb = Bs + buf * TK * TP;
        Frag a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ac = wm * 64 + i * 16 + tcol;
            a[i].h[0] = lds_tr16(Ab + trow0 * TP + ac);
            a[i].h[1] = lds_tr16(Ab + trow1 * TP + ac);
            const int bc = wn * 64 + i * 16 + tcol;
            b[i].h[0] = lds_tr16(Bb + trow0 * TP + bc);
            b[i].h[1] = lds_tr16(Bb + trow1 * TP + bc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = ASR_MFMA_16x16x32(a[i].v, b[j].v, acc[i][j]);
        if (kt + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                const int gn = n0 + wn * 64 + j * 16 + (lane & 15);
                if (gm < M && gn < N) atomicAdd(C + (size_t)gm * ldc + gn, acc[i][j][r]);
            }
}

// ------------------------------------------------------------------------------------------------ TN, vector operands
// The kernel above asks for the operands of k tile kt + 1 while it works on tile kt: ONE tile in flight per workgroup, and a
// 16-MFMA tile is over long before an L2 answer is back (the what-if build without global loads ran 0.103 instead of 0.144 ms
// without atomics on 3072 x 512 x 32000); its general loader (element loads, edges) is a forest of ~16 scalar branches per tile,
// behind which the compiler waits for vmcnt(0) wherever it waits at all.  This one serves the shapes the train step has (M, N, lda,
// ldb multiples of 8, 16-B aligned bases, operands below 4 GB): buffer loads with the k offset in an SGPR -- no branch and no
// per-lane address arithmetic; rows beyond the K split and chunks beyond M / N get an offset beyond the buffer, i.e. zeros --
// and TWO tiles in flight: the registers of tile kt + 2 are asked for before tile kt is worked on, tile kt + 1 is waited for with
// vmcnt(4) (gfx9 counts loads in order: the four youngest may stay out) and goes to LDS behind the MFMAs.  Same tiling, LDS image,
// transposing reads, split-K atomics, XCD-aware order and grouped launches as gemm_tn_kernel<false>.
// CONV: operand B is the virtual im2col matrix of the activation tensor (as in gemm_tn_kernel<true>: a thread's chunk is one tap and
// channel group for the whole launch, its reduction rows (t, b, h) step by the k tile with single carries); positions outside the
// tensor and empty taps take the out-of-buffer offset.
template <bool CONV>
__global__ __launch_bounds__(256, 3) void gemm_tn_vec_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ B, int ldb,
                                                             float* __restrict__ C, int ldc, int M, int N, int K, int tiles_n,
                                                             int k_per_split, TnGroup grp, ConvDesc cd, long long copy_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* As = reinterpret_cast<uint16_t*>(smem);             // [2][TK][TP]
    uint16_t* Bs = As + 2 * TK * TP;                              // [2][TK][TP]
    const bool grouped = !CONV && grp.n > 0;
    const int tiles = grouped ? grp.p[grp.n - 1].tile_end : ((M + BM - 1) / BM) * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    C += (size_t)xcd * copy_stride;      // (one copy of the output per XCD: asr_conv_tn_copies)
    const int item = xcd * (int)(gridDim.x >> 3) + slot;
    const int round = item / tiles;
    int bid = item - round * tiles;
    if (grouped) {                       // uniform over the workgroup
        int which = 0;
        while (which + 1 < grp.n && bid >= grp.p[which].tile_end) ++which;
        if (which > 0) bid -= grp.p[which - 1].tile_end;
        const TnProb& q = grp.p[which];
        A = q.A; B = q.B; C = q.C; lda = q.lda; ldb = q.ldb; ldc = q.ldc; M = q.M; N = q.N; K = q.K; tiles_n = q.tiles_n;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = round * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    if (kbeg >= kend) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // a tile is 32 k rows x 16 chunks of 16 B per operand: chunk id = tid + 256 i -> k row id / 16, chunk id % 16 (as above)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((size_t)K * lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)B, 0, CONV ? (int)((size_t)cd.Ts * cd.B * cd.Hs * cd.Cs * 2) : (int)((size_t)K * ldb * 2), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    unsigned oa[2], ob[2];
    int krow[2];
    // CONV: this thread's chunk (tid & 15) never changes -> tap and channel are launch invariant; (t, b, h) of its two rows step by TK
    int cv_dt = 0, cv_dh = 0, cv_ci = 0, cv_t[2] = {0, 0}, cv_b[2] = {0, 0}, cv_h[2] = {0, 0}, inc_h = 0, inc_b = 0, inc_t = 0;
    bool cv_ok = false;
    if (CONV) {
        const int k = n0 + (tid & 15) * 8;
        const int tap = k / cd.Cs, kh = tap / cd.KW, kw = tap - kh * cd.KW;
        cv_ci = k - tap * cd.Cs;
        cv_dt = kw - cd.pt;
        cv_dh = kh - cd.ph;
        cv_ok = k < N && kh < cd.KH;
        inc_h = TK % cd.Hr;
        const int inc_tb = TK / cd.Hr;
        inc_b = inc_tb % cd.B;
        inc_t = inc_tb / cd.B;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + i * 256, row = id >> 4, c = id & 15;
        krow[i] = kbeg + row;
        oa[i] = m0 + c * 8 < M ? (unsigned)(kbeg + row) * (unsigned)(lda * 2) + (unsigned)(m0 + c * 8) * 2u : OOB;
        ob[i] = n0 + c * 8 < N ? (unsigned)(kbeg + row) * (unsigned)(ldb * 2) + (unsigned)(n0 + c * 8) * 2u : OOB;
        if (CONV) {
            const int gk = kbeg + row;
            cv_h[i] = gk % cd.Hr;
            const int tb = gk / cd.Hr;
            cv_b[i] = tb % cd.B;
            cv_t[i] = tb / cd.B;
        }
    }
    const int sa = TK * lda * 2, sb = TK * ldb * 2;             // bytes per k tile (SGPRs)
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    // the operands of k tile kt (beyond the split -> zeros).  CONV: call with kt = 0, 1, 2, ... in order, each once (the row walk)
    auto ask = [&](uint4 (&ra)[2], uint4 (&rb)[2], int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool live = krow[i] + kt * TK < kend;
            const u32x4_t va = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, live ? oa[i] : OOB, (int)((unsigned)kt * (unsigned)sa), 0);
            u32x4_t vb;
            if (CONV) {
                const int t = cv_t[i], b = cv_b[i], h = cv_h[i];
                {   // advance to the same row of the next k tile
                    int hh = h + inc_h;
                    const int c1 = hh >= cd.Hr;
                    hh -= c1 ? cd.Hr : 0;
                    int bb = b + inc_b + c1;
                    const int c2 = bb >= cd.B;
                    bb -= c2 ? cd.B : 0;
                    cv_h[i] = hh; cv_b[i] = bb; cv_t[i] = t + inc_t + c2;
                }
                const int ti = t + cv_dt, hi = h + cv_dh;
                const bool ok = live && cv_ok && (unsigned)ti < (unsigned)cd.Ts && (unsigned)hi < (unsigned)cd.Hs;
                const unsigned off = (unsigned)((ti * cd.B + b) * cd.Hs + hi) * (unsigned)(cd.Cs * 2) + (unsigned)cv_ci * 2u;
                vb = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, ok ? off : OOB, 0, 0);
            } else {
                vb = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, live ? ob[i] : OOB, (int)((unsigned)kt * (unsigned)sb), 0);
            }
            ra[i] = make_uint4(va[0], va[1], va[2], va[3]);
            rb[i] = make_uint4(vb[0], vb[1], vb[2], vb[3]);
        }
    };
    auto to_lds = [&](const uint4 (&ra)[2], const uint4 (&rb)[2], int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * 256, row = id >> 4, c = id & 15;
            *reinterpret_cast<uint4*>(As + (buf * TK + row) * TP + c * 8) = ra[i];
            *reinterpret_cast<uint4*>(Bs + (buf * TK + row) * TP + c * 8) = rb[i];
        }
    };
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int trow0 = 4 * g + q, trow1 = 16 + 4 * g + q, tcol = 4 * p;
    auto work = [&](int buf) {
        const uint16_t* Ab = As + buf * TK * TP;
        const uint16_t* Bb = Bs + buf * TK * TP;
        Frag a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ac = wm * 64 + i * 16 + tcol;
            a[i].h[0] = lds_tr16(Ab + trow0 * TP + ac);
            a[i].h[1] = lds_tr16(Ab + trow1 * TP + ac);
            const int bc = wn * 64 + i * 16 + tcol;
            b[i].h[0] = lds_tr16(Bb + trow0 * TP + bc);
            b[i].h[1] = lds_tr16(Bb + trow1 * TP + bc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = ASR_MFMA_16x16x32(a[i].v, b[j].v, acc[i][j]);
    };

    const int nk = (kend - kbeg + TK - 1) / TK;
    const int nk2 = (nk + 1) & ~1;                 // tiles in pairs: the body below is straight-line code (an odd count works one tile of zeros)
    uint4 ra0[2], rb0[2], ra1[2], rb1[2];          // set 0: even tiles, set 1: odd tiles
    ask(ra0, rb0, 0);
    ask(ra1, rb1, 1);
    to_lds(ra0, rb0, 0);                           // (waits for tile 0 only: vmcnt(4))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nk2; kt += 2) {
        // even tile kt (LDS buffer 0): tile kt + 1 is in set 1 (in flight or landed), tile kt + 2 is asked for into set 0
        // (the scheduling barriers keep the order asked for: left alone the compiler sinks the loads behind the MFMAs -- into the
        // fragment registers they free -- and hoists the LDS stores, with their wait, in front of them: one tile in flight again)
        ask(ra0, rb0, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        work(0);
        __builtin_amdgcn_sched_barrier(0);
        to_lds(ra1, rb1, 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // odd tile kt + 1 (LDS buffer 1): tile kt + 3 is asked for into set 1
        ask(ra1, rb1, kt + 3);
        __builtin_amdgcn_sched_barrier(0);
        work(1);
        __builtin_amdgcn_sched_barrier(0);
        to_lds(ra0, rb0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                const int gn = n0 + wn * 64 + j * 16 + (lane & 15);
                if (gm < M && gn < N) atomicAdd(C + (size_t)gm * ldc + gn, acc[i][j][r]);
            }
}

// ------------------------------------------------------------------------------------------------ TN, 256 x 128 tile
// The 128 x 128 TN kernel above stages its operands through registers (global load -> VGPR -> ds_write) and runs 16 MFMAs per
// barrier.  This one is the NT 256 x 128 kernel's structure on k-strided operands: LDS-DMA straight into a two-stage ring (no
// staging registers: 3 workgroups per CU), four waves of 128 x 64 (32 MFMAs per wave and barrier), fragments by
// ds_read_b64_tr_b16.  LDS image of a k row: its 16-B chunks in order, chunk c of row k stored at position c ^ ((k & 7) << 1) --
// the 32 lanes a transposing read services together (8 k rows x 32 B) then cover all 64 banks; LDS-DMA writes lane-linear, so
// the swizzle is applied to the SOURCE address (as in the NT kernels).  Rows beyond the K split and columns beyond M / N read a
// page of zeros.  Split-K with float atomics and the XCD-aware order of the kernel above.  CONV: B = virtual im2col matrix.
constexpr int T2M = 256, T2N = 128;
constexpr int TN2_STAGES = 2;
constexpr int TN2_LDS_BYTES = TN2_STAGES * (T2M + T2N) * TK * 2;      // 2 x 24 KiB

template <bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_tn256_kernel(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ B,
                                                            int ldb, float* __restrict__ C, int ldc, int M, int N, int K, int tiles_n,
                                                            int k_per_split, ConvDesc cd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_STAGE = TK * T2M * 2, B_STAGE = TK * T2N * 2;          // 16 KiB, 8 KiB
    char* As = smem;
    char* Bs = smem + TN2_STAGES * A_STAGE;
    const int tiles = ((M + T2M - 1) / T2M) * tiles_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int round = slot / tiles, bid = slot - round * tiles;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * T2M, n0 = tn * T2N;
    const int kbeg = (round * 8 + xcd) * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    if (kbeg >= kend) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS-DMA slots.  A image: 32 k rows x 32 chunks = 1024 slots, slot = i * 256 + tid: k row slot / 32, position slot % 32;
    // B image: 32 x 16 = 512 slots: k row slot / 16, position slot % 16.
    const uint16_t* ga[4];
    const uint16_t* gb[2];
    int ka[4], kb[2];                      // k row of the slot inside the tile
    bool oka[4], okb[2];                   // the slot's columns exist
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sl = i * 256 + tid, kr = sl >> 5, c = (sl & 31) ^ ((kr & 7) << 1);
        ka[i] = kr;
        oka[i] = m0 + c * 8 < M;           // (M % 8 == 0: a chunk is inside or outside as a whole)
        ga[i] = A + (size_t)(kbeg + kr) * lda + m0 + c * 8;
    }
    // CONV: a slot's chunk never changes -> tap and channel are loop invariant; its row (t, b, h) steps by the k tile
    int cv_dt[2] = {0, 0}, cv_dh[2] = {0, 0}, cv_ci[2] = {0, 0}, cv_t[2] = {0, 0}, cv_b[2] = {0, 0}, cv_h[2] = {0, 0};
    int inc_h = 0, inc_b = 0, inc_t = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sl = i * 256 + tid, kr = sl >> 4, c = (sl & 15) ^ ((kr & 7) << 1);
        kb[i] = kr;
        const int n = n0 + c * 8;
        okb[i] = n < N;
        if (CONV) {
            const int tap = n / cd.Cs, kh = tap / cd.KW, kw = tap - kh * cd.KW;
            cv_ci[i] = n - tap * cd.Cs;
            cv_dt[i] = kw - cd.pt;
            cv_dh[i] = kh - cd.ph;
            okb[i] = okb[i] && kh < cd.KH;
            const int gk = kbeg + kr;
            cv_h[i] = gk % cd.Hr;
            const int tb = gk / cd.Hr;
            cv_b[i] = tb % cd.B;
            cv_t[i] = tb / cd.B;
            gb[i] = B;
        } else {
            gb[i] = B + (size_t)(kbeg + kr) * ldb + n;
        }
    }
    if (CONV) {
        inc_h = TK % cd.Hr;
        const int inc_tb = TK / cd.Hr;
        inc_b = inc_tb % cd.B;
        inc_t = inc_tb / cd.B;
    }
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page);
    auto issue_tile = [&](int kt, int buf) {            // called with kt = 0, 1, 2, ... in order (the CONV row walk relies on it)
        const int k0 = kbeg + kt * TK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint16_t* src = (oka[i] && k0 + ka[i] < kend) ? ga[i] + (size_t)kt * TK * lda : zero;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(As + buf * A_STAGE + (i * 256 + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint16_t* src;
            if (CONV) {
                const int t = cv_t[i], b = cv_b[i], h = cv_h[i];
                {   // advance to the same slot of the next k tile
                    int hh = h + inc_h;
                    const int c1 = hh >= cd.Hr;
                    hh -= c1 ? cd.Hr : 0;
                    int bb = b + inc_b + c1;
                    const int c2 = bb >= cd.B;
                    bb -= c2 ? cd.B : 0;
                    cv_h[i] = hh; cv_b[i] = bb; cv_t[i] = t + inc_t + c2;
                }
                const int ti = t + cv_dt[i], hi = h + cv_dh[i];
                const bool ok = okb[i] && k0 + kb[i] < kend && (unsigned)ti < (unsigned)cd.Ts && (unsigned)hi < (unsigned)cd.Hs;
                src = ok ? B + (((size_t)ti * cd.B + b) * cd.Hs + hi) * cd.Cs + cv_ci[i] : zero;
            } else {
                src = (okb[i] && k0 + kb[i] < kend) ? gb[i] + (size_t)kt * TK * ldb : zero;
            }
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(Bs + buf * B_STAGE + (i * 256 + wid * 64) * 16), 16, 0, 0);
        }
    };
    // transposing reads: 16-lane group g = lane >> 4, lane 4 q + p of the group points at k row 4 g + q (second read: 16 + 4 g + q),
    // columns 4 p .. 4 p + 3 of the 16-column tile (k slots permuted identically for A and B: see the 128 x 128 kernel)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int k0r = 4 * g + q, k1r = 16 + 4 * g + q;
    const int sw0 = (k0r & 7) << 1;                                    // (k1r & 7 == k0r & 7: the second read is a constant 16 rows further)
    (void)k1r;
    int aoff0[8], boff0[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = wm * 16 + i * 2 + (pp >> 1);                     // 16-B chunk of columns wm*128 + i*16 + 4 pp ..
        aoff0[i] = k0r * (T2M * 2) + ((c ^ sw0) << 4) + (pp & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = wn * 8 + j * 2 + (pp >> 1);
        boff0[j] = k0r * (T2N * 2) + ((c ^ sw0) << 4) + (pp & 1) * 8;
    }
    constexpr int A16 = 16 * T2M * 2, B16 = 16 * T2N * 2;
    const int nk = (kend - kbeg + TK - 1) / TK;
    issue_tile(0, 0);
    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) issue_tile(kt + 1, buf ^ 1);
        const char* Ab = As + buf * A_STAGE;
        const char* Bb = Bs + buf * B_STAGE;
        Frag b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b[j].h[0] = lds_tr16(reinterpret_cast<const uint16_t*>(Bb + boff0[j]));
            b[j].h[1] = lds_tr16(reinterpret_cast<const uint16_t*>(Bb + boff0[j] + B16));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            Frag a;
            a.h[0] = lds_tr16(reinterpret_cast<const uint16_t*>(Ab + aoff0[i]));
            a.h[1] = lds_tr16(reinterpret_cast<const uint16_t*>(Ab + aoff0[i] + A16));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = ASR_MFMA_16x16x32(a.v, b[j].v, acc[i][j]);
        }
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + wm * 128 + i * 16 + (lane >> 4) * 4 + r;
                const int gn = n0 + wn * 64 + j * 16 + (lane & 15);
                if (gm < M && gn < N) atomicAdd(C + (size_t)gm * ldc + gn, acc[i][j][r]);
            }
}

}  // namespace gemm
}  // namespace asr

using namespace asr;
using namespace asr::gemm;

// Which NT kernel serves a call (measured, tools/time_nt.py, TFLOP/s at T=1000, B=32, 13 rows of height, k 3x5):
//                                      256x128x32 (3 WG/CU)   256x256x32, 3 stages   256x256x64, 2 stages
//   conv 128->256 fwd                         598                    617                   684
//   conv 128->512 fwd                         711                    703                   839
//   conv 256->512 fwd / bwd-data           762 / 752              781 / 754             931 / 914
//   32000 x 3072 x 512 (f32 out)              527                    428                   487
//   8192^3                                    963                    765                   953
// Whole 128-B lines per row and K step are what pays (a 64-B half line per row leaves the other half to be fetched again
// by the next K step once the tile no longer fits the L1), and only where the 256-wide tile removes a second pass over the
// activations: the convolutions with more than 128 output channels.  ASR_DEBUG nt_wide overrides (tests, experiments):
// -1 (default) = 64-wide K for those convolutions only; 0 = never; 1 / 2 = 256x256x32 / 256x256x64 wherever the shape allows.
static int nt_wide_mode() {
    static int mode = -2;
    if (mode == -2) {
        mode = debug_flag("nt_wide", -1);
        if (mode < -1 || mode > 4) mode = -1;
    }
    return mode;
}

// ASR_DEBUG nt_wide_force=1 (tests): take the wide kernel whatever the number of tiles
static bool nt_wide_force() {
    static int f = -1;
    if (f < 0) f = debug_flag("nt_wide_force", 0) ? 1 : 0;
    return f == 1;
}

template <typename OutT, bool CONV>
static int launch_nt_wide(hipStream_t stream, const uint16_t* A, int lda, const uint16_t* B, int ldb, OutT* C, int ldc, const float* bias,
                          int M, int N, int K, bool b_fits_l2, const ConvDesc& cd, int mode) {
    const int tm = cdiv(M, 256), tn = cdiv(N, 256);
    const int tn_arg = b_fits_l2 ? -tn : tn;
#define ASR_WIDE(BK, ST, WN, THREADS)                                                                                      \
    do {                                                                                                                  \
        constexpr int LDS = ST * 512 * BK * 2;                                                                            \
        static bool attr = false;                                                                                         \
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_nt_wide_kernel<OutT, CONV, BK, ST, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; } \
        hipLaunchKernelGGL((gemm_nt_wide_kernel<OutT, CONV, BK, ST, WN>), dim3(CONV ? 8 * cdiv(tm * tn, 8) : tm * tn), dim3(THREADS), LDS, stream, A, lda, B, ldb, C, ldc, bias, M, N, K, tn_arg, cd); \
    } while (0)
    if (mode == 2) ASR_WIDE(64, 2, 4, 512);
    else if (mode == 3) ASR_WIDE(32, 4, 2, 256);      // four waves of 128 x 128 (256 accumulator registers), four-deep ring
    else if (mode == 4) ASR_WIDE(64, 2, 2, 256);
    else ASR_WIDE(32, 3, 4, 512);
#undef ASR_WIDE
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// Grid of the persistent kernels: a multiple of 256 workgroups (whole CUs; a multiple of 8 keeps bid % 8 = XCD for every tile a
// workgroup walks over), chosen for the fewest rounds x the time of a tile when k workgroups share a CU (measured on 8192^3 and
// 32000 x 1024 x 3072: a tile takes ~0.87 of the three-per-CU time at two per CU; one per CU leaves the fill latency bare).
// ASR_DEBUG nt_persist_grid overrides (experiments).
static int nt_persist_grid(int total, int max_per_cu) {
    static int forced = -1;
    if (forced < 0) forced = debug_flag("nt_persist_grid", 0);
    if (forced > 0) return ((total < forced ? total : forced) + 7) & ~7;
    const float tile_time[4] = {0.f, 0.80f, 0.87f, 1.0f};
    int best = 256 * max_per_cu;
    float best_cost = 1e30f;
    for (int k = max_per_cu; k >= 2; --k) {
        const float cost = (float)cdiv(total, 256 * k) * tile_time[k];
        if (cost < best_cost - 1e-6f) { best_cost = cost; best = 256 * k; }
    }
    return total < best ? ((total + 7) & ~7) : best;       // (a multiple of 8: the kernels deal tile ranges to bid % 8)
}

extern "C" int asr_gemm_nt(void* stream_, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                           const float* bias, int M, int N, int K, int out_bf16) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return ASR_ERR_BAD_ARG;
    if (lda < K || ldb < K || ldc < N) return ASR_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<uint16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<uint16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
        attr_set = true;
    }
    const bool b_fits_l2 = (size_t)N * K * 2 <= (size_t)5 * 512 * 1024;      // 2.5 MB of the 4 MB per XCD
    const bool aligned = (lda % 8) == 0 && (ldb % 8) == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0;
    {
        // The 256 x 256 tile / eight-wave kernel of csrc/gemm8.hip wherever it qualifies (K % 64 == 0, N % 4 == 0, aligned) -- measured against
        // the kernels below (tools/time_nt8.py, us, same process, alternating): 32000 x 512 x 3072 118 -> 73, x 3072 x 512 122 -> 106, x 384 x 3072
        // 109 -> 72, x 3072 x 384 104 -> 92, x 3000 x 320 (f32) 151 -> 121, x 512 x 640 32 -> 25, 8192^3 1082 -> 753 (1017 -> 1460 TFLOP/s) -- except
        // where short K meets a ragged last column tile (32000 x 640 x 512: a third 256-wide tile for 128 columns, 34 -> 36 with one tile
        // per workgroup; its persistent form -- more tiles than CUs, K <= 1024: gemm_nt_8pp_kernel -- takes that one too, 35 -> 30: up to
        // 25 % of padding there).  With more tiles than CUs and K <= 1024 the products above run: x 3072 x 512 116 -> 91, x 3072 x 384
        // 99 -> 73 (tools/ab_nt8pp.py).
        // ASR_DEBUG nt_8ph=0: never (comparison, and the tests that pin the kernels below).
        static const int use8 = debug_flag("nt_8ph", 1), use8pp = debug_flag("nt_8pp", 1);
        const long long n_padded = (long long)cdiv(N, 256) * 256;
        const bool persistent = use8pp && (K & 63) == 0 && K >= 128 && K <= 1024 && (long long)cdiv(M, 256) * cdiv(N, 256) > 256 && N <= 8192 &&
                                out_bf16 && (N & 7) == 0 && (ldc & 7) == 0;
        if (use8 && nt_wide_mode() <= 0 && (K >= 1024 || n_padded * 10 <= (long long)N * 11 || (persistent && n_padded * 4 <= (long long)N * 5)) && M >= 256 &&
            asr_gemm_nt_8ph_ok(A, lda, B, ldb, C, ldc, bias, M, N, K, out_bf16))
            return asr_gemm_nt_8ph(stream_, A, lda, B, ldb, C, ldc, bias, M, N, K, out_bf16);
    }
    {
        const int wide = nt_wide_mode() > 0 ? nt_wide_mode() : 0;      // plain GEMMs: only on request
        if (wide && aligned && (K % ((wide == 2 || wide == 4) ? 64 : 32)) == 0 && N >= 256 && ((long long)cdiv(M, 256) * cdiv(N, 256) >= 512 || nt_wide_force())) {
            if (out_bf16) return launch_nt_wide<uint16_t, false>(stream, (const uint16_t*)A, lda, (const uint16_t*)B, ldb, (uint16_t*)C, ldc, bias, M, N, K, b_fits_l2, ConvDesc{}, wide);
            return launch_nt_wide<float, false>(stream, (const uint16_t*)A, lda, (const uint16_t*)B, ldb, (float*)C, ldc, bias, M, N, K, b_fits_l2, ConvDesc{}, wide);
        }
    }
    // persistent form of the 256 x 128 kernel: vector stores want ldc % 8 == 0 and an aligned C (bias: 16-B aligned), 32-bit byte
    // offsets want operands below 4 GB; ASR_DEBUG nt_persist=0 keeps the one-tile-per-workgroup kernel (tests, comparison)
    static int persist = -1;
    if (persist < 0) persist = debug_flag("nt_persist", 1);
    static int pmin = -1;
    if (pmin < 0) pmin = debug_flag("nt_persist_min", 256);      // (one tile per CU: 32000 x 384 x 3072 642 -> 688, x 320 x 3008 532 -> 604 TFLOP/s against the 128 x 128 kernel)
    const bool k_tail = (K % B2K) != 0;       // K = 3000 (the logit gradient): the last K step is fetched short, see the kernel's KT
    if (persist && aligned && (K % 8) == 0 && K >= B2K && (long long)cdiv(M, B2M) * cdiv(N, B2N) >= pmin && (ldc % 8) == 0 && (((uintptr_t)C) & 15) == 0 &&
        (!bias || (((uintptr_t)bias) & 15) == 0) && (unsigned long long)M * lda < (1ull << 31) && (unsigned long long)N * ldb < (1ull << 31) && M < (1 << 24) && N < (1 << 24) &&
        lda < (1 << 23) && ldb < (1 << 23)) {
        static bool attrp = false;
        if (!attrp) {
            (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<float, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<uint16_t, 0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
            attrp = true;
        }
        const int t2m = cdiv(M, B2M), t2n = cdiv(N, B2N), total = t2m * t2n;
        const int grid = nt_persist_grid(total, 3);
        if (k_tail) {
            static bool attrk = false;
            if (!attrk) {
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<float, 0, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<uint16_t, 0, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                attrk = true;
            }
            const unsigned a_bytes = (unsigned)((unsigned long long)M * lda * 2);
            if (out_bf16)
                hipLaunchKernelGGL((gemm_nt256p_kernel<uint16_t, 0, 8, true>), dim3(grid), dim3(256), NT2_LDS_BYTES, stream, (const uint16_t*)A, lda, (const uint16_t*)B, ldb,
                                   (uint16_t*)C, ldc, bias, M, N, K, t2m, t2n, total, a_bytes, ConvDesc{});
            else
                hipLaunchKernelGGL((gemm_nt256p_kernel<float, 0, 8, true>), dim3(grid), dim3(256), NT2_LDS_BYTES, stream, (const uint16_t*)A, lda, (const uint16_t*)B, ldb,
                                   (float*)C, ldc, bias, M, N, K, t2m, t2n, total, a_bytes, ConvDesc{});
        } else if (out_bf16)
            hipLaunchKernelGGL((gemm_nt256p_kernel<uint16_t, 0, 8>), dim3(grid), dim3(256), NT2_LDS_BYTES, stream, (const uint16_t*)A, lda, (const uint16_t*)B, ldb,
                               (uint16_t*)C, ldc, bias, M, N, K, t2m, t2n, total, 0xffffffffu, ConvDesc{});
        else
            hipLaunchKernelGGL((gemm_nt256p_kernel<float, 0, 8>), dim3(grid), dim3(256), NT2_LDS_BYTES, stream, (const uint16_t*)A, lda, (const uint16_t*)B, ldb,
                               (float*)C, ldc, bias, M, N, K, t2m, t2n, total, 0xffffffffu, ConvDesc{});
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    if (aligned && (K % B2K) == 0 && (long long)cdiv(M, B2M) * cdiv(N, B2N) >= 1024) {      // at least two rounds of 2 workgroups per CU
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<float, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
            (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<uint16_t, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
            attr2 = true;
        }
        const int t2m = cdiv(M, B2M), t2n = cdiv(N, B2N) * (b_fits_l2 ? -1 : 1);
        if (out_bf16)
            hipLaunchKernelGGL((gemm_nt256_kernel<uint16_t, false, 2>), dim3(t2m * cdiv(N, B2N)), dim3(256), NT2_LDS_BYTES, stream, (const uint16_t*)A, lda,
                               (const uint16_t*)B, ldb, (uint16_t*)C, ldc, bias, M, N, K, t2n, ConvDesc{});
        else
            hipLaunchKernelGGL((gemm_nt256_kernel<float, false, 2>), dim3(t2m * cdiv(N, B2N)), dim3(256), NT2_LDS_BYTES, stream, (const uint16_t*)A, lda,
                               (const uint16_t*)B, ldb, (float*)C, ldc, bias, M, N, K, t2n, ConvDesc{});
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    const bool glds = (K % BK) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0;
#define ASR_NT(T, G, CT)                                                                                                  \
    hipLaunchKernelGGL((gemm_nt_kernel<T, G>), dim3(tiles_m * tiles_n), dim3(256), NT_LDS_BYTES, stream, (const uint16_t*)A, lda, \
                       (const uint16_t*)B, ldb, (CT*)C, ldc, bias, M, N, K, b_fits_l2 ? -tiles_n : tiles_n)
    if (out_bf16) { if (glds) ASR_NT(uint16_t, true, uint16_t); else ASR_NT(uint16_t, false, uint16_t); }
    else          { if (glds) ASR_NT(float, true, float); else ASR_NT(float, false, float); }
#undef ASR_NT
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// split K so that about 3 workgroups per CU are in flight: a multiple of 8 splits (one set per XCD) where K allows, each
// split a multiple of the k tile
static int tn_splits(int tiles, int K, int& k_per_split, int target = 768, bool below8 = false) {
    int splits = cdiv(target, tiles);   // (3 workgroups per CU; 1024 was 5..17 % slower on the step's shapes, 512 hurt M = 3000)
    if (!(below8 && splits < 8)) splits = cdiv(splits, 8) * 8;      // below8 (grouped launches): 1..7 splits as they come
    const int max_splits = cdiv(K, 8 * TK);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    k_per_split = cdiv(cdiv(K, splits), TK) * TK;
    return cdiv(K, k_per_split);
}

// grid of the 128 x 128 TN kernel: the (split, tile) pairs in eight equal contiguous runs, one per XCD
static int tn_grid(int tiles, int splits) { return 8 * cdiv(tiles * splits, 8); }

// what gemm_tn_vec_kernel asks of a product (ASR_DEBUG tn_vec=0 keeps the general kernel: tests, comparison)
static bool tn_vec_ok(const void* A, int lda, const void* B, int ldb, int M, int N, int K) {
    static int on = -1;
    if (on < 0) on = debug_flag("tn_vec", 1);
    if (!on) return false;
    return (M & 7) == 0 && (N & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0 &&
           (unsigned long long)K * lda * 2 < 0xfffffff0ull && (unsigned long long)K * ldb * 2 < 0xfffffff0ull;
}

// the 256 x 128 LDS-DMA kernel wants whole 16-B chunks (M, lda, ldb multiples of 8, aligned bases) and at least one full tile
// of rows; ASR_DEBUG tn256=0 keeps the 128 x 128 kernel (tests, comparison)
// Measured (tools/time_nt.py, TFLOP/s, 256 x 128 LDS-DMA kernel at two workgroups per CU against the 128 x 128 kernel at four):
// convolution weight gradients 128->256: 668 / 638, 128->512: 783 / 745, 256->512: 823 / 737 -- but 3072 x 512 x 32000: 541 / 680,
// 1536 x 512: 387 / 555 (half the workgroups in flight and twice the atomics per workgroup behind a K split of the same depth).
// So it serves the implicit convolutions only; ASR_DEBUG tn256=2 sends the plain products there as well (tests), 0 switches it off.
static bool tn256_ok(int M, int lda, int ldb, const void* A, const void* B, bool conv) {
    static int on = -1;
    if (on < 0) on = debug_flag("tn256", 1);
    if (!on || (!conv && on != 2)) return false;
    return M >= T2M && (M & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;
}

extern "C" int asr_gemm_tn_acc(void* stream_, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M,
                               int N, int K) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return ASR_ERR_BAD_ARG;
    if (lda < M || ldb < N || ldc < N) return ASR_ERR_BAD_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    {
        // the eight-wave kernel of csrc/gemm8.hip (tools/time_tn8.py: 32000 x 3000 x 320 116 -> 102 us, 416000 x 256 x 1920 554 -> 382;
        // the small dense gradients 640 x 512 / 640 x 320 tie); ASR_DEBUG tn_8ph=0 / tn256=2: never
        static const int use8 = debug_flag("tn_8ph", 1), tn256_forced = debug_flag("tn256", 1);
        if (use8 && tn256_forced != 2 && K >= 2048 && M >= 192 && N >= 192 && asr_gemm_tn_8ph_ok(A, lda, B, ldb, C, ldc, M, N, K)) {
            const void* a1[1] = {A};
            const void* b1[1] = {B};
            float* c1[1] = {C};
            return asr_gemm_tn_acc_group_8ph(stream_, 1, a1, &lda, b1, &ldb, c1, &ldc, &M, &N, &K);
        }
    }
    if (tn256_ok(M, lda, ldb, A, B, false) && (N & 7) == 0) {
        const int t2m = cdiv(M, T2M), t2n = cdiv(N, T2N);
        int kps;
        const int sp = tn_splits(t2m * t2n, K, kps);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TN2_LDS_BYTES); attr = true; }
        hipLaunchKernelGGL(gemm_tn256_kernel<false>, dim3(8 * t2m * t2n * cdiv(sp, 8)), dim3(256), TN2_LDS_BYTES, stream,
                           (const uint16_t*)A, lda, (const uint16_t*)B, ldb, C, ldc, M, N, K, t2n, kps, ConvDesc{});
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    int k_per_split;
    const int splits = tn_splits(tiles_m * tiles_n, K, k_per_split);
    if (tn_vec_ok(A, lda, B, ldb, M, N, K))
        hipLaunchKernelGGL(gemm_tn_vec_kernel<false>, dim3(tn_grid(tiles_m * tiles_n, splits)), dim3(256), TN_LDS_BYTES, stream,
                           (const uint16_t*)A, lda, (const uint16_t*)B, ldb, C, ldc, M, N, K, tiles_n, k_per_split, TnGroup{}, ConvDesc{}, 0LL);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(tn_grid(tiles_m * tiles_n, splits)), dim3(256), TN_LDS_BYTES, stream,
                           (const uint16_t*)A, lda, (const uint16_t*)B, ldb, C, ldc, M, N, K, tiles_n, k_per_split, ConvDesc{}, 0LL, TnGroup{});
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// n <= 4 products C_i += A_i^T B_i (any shapes) in ONE launch; host arrays of n entries each.
extern "C" int asr_gemm_tn_acc_group(void* stream_, int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                     float* const* C, const int* ldc, const int* M, const int* N, const int* K) {
    if (!A || !B || !C || !lda || !ldb || !ldc || !M || !N || !K || n < 1 || n > 4) return ASR_ERR_BAD_ARG;
    {
        // the eight-wave kernel of csrc/gemm8.hip where every product qualifies (tools/time_tn8.py: the three weight gradients of a GRU
        // layer, K = 32000: 275 -> 194 us); ASR_DEBUG tn_8ph=0: never
        static const int use8 = debug_flag("tn_8ph", 1);
        bool all8 = use8 != 0;
        for (int i = 0; i < n && all8; ++i)
            all8 = A[i] && B[i] && C[i] && K[i] >= 2048 && asr_gemm_tn_8ph_ok(A[i], lda[i], B[i], ldb[i], C[i], ldc[i], M[i], N[i], K[i]);
        if (all8) return asr_gemm_tn_acc_group_8ph(stream_, n, A, lda, B, ldb, C, ldc, M, N, K);
    }
    TnGroup grp{};
    int tiles = 0, kmax = 0;
    bool vec = true;
    for (int i = 0; i < n; ++i) {
        if (!A[i] || !B[i] || !C[i] || M[i] <= 0 || N[i] <= 0 || K[i] <= 0) return ASR_ERR_BAD_ARG;
        if (lda[i] < M[i] || ldb[i] < N[i] || ldc[i] < N[i]) return ASR_ERR_BAD_ARG;
        vec = vec && tn_vec_ok(A[i], lda[i], B[i], ldb[i], M[i], N[i], K[i]);
        TnProb& q = grp.p[i];
        q.A = (const uint16_t*)A[i]; q.B = (const uint16_t*)B[i]; q.C = C[i];
        q.lda = lda[i]; q.ldb = ldb[i]; q.ldc = ldc[i]; q.M = M[i]; q.N = N[i]; q.K = K[i];
        q.tiles_n = cdiv(N[i], BN);
        tiles += cdiv(M[i], BM) * q.tiles_n;
        q.tile_end = tiles;
        kmax = K[i] > kmax ? K[i] : kmax;
    }
    grp.n = n;
    hipStream_t stream = (hipStream_t)stream_;
    int k_per_split;
    static int target = 0;
    if (!target) { target = debug_flag("tn_group_target", 1152); if (target < 1) target = 1152; }
    const int splits = tn_splits(tiles, kmax, k_per_split, target, true);
    const TnProb& q = grp.p[0];
    if (vec)
        hipLaunchKernelGGL(gemm_tn_vec_kernel<false>, dim3(tn_grid(tiles, splits)), dim3(256), TN_LDS_BYTES, stream,
                           q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, q.tiles_n, k_per_split, grp, ConvDesc{}, 0LL);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<false>, dim3(tn_grid(tiles, splits)), dim3(256), TN_LDS_BYTES, stream,
                           q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, q.tiles_n, k_per_split, ConvDesc{}, 0LL, grp);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// Few output tiles (one for the first layer: 128 x 120) mean hundreds of K splits, i.e. hundreds of workgroups on all eight XCDs adding
// into the same 64 KB: 91 of the 166 us of that call were the atomics.  With copies == 8 every XCD (workgroup id % 8) adds into its own
// copy C + xcd * Co * ldc -- the adds stay in one L2 -- and asr_conv_weight_grad_unpack_copies sums the copies.
extern "C" int asr_conv_tn_copies(int Co, int Cs, int KH, int KW) {
    const int N = KH * KW * Cs;
    if (tn256_ok(Co, 8, 8, nullptr, nullptr, true)) return 1;
    return cdiv(Co, BM) * cdiv(N, BN) <= 2 ? 8 : 1;
}

extern "C" int asr_conv_tn_acc_copies(void* stream_, const void* g, int ldg, const void* x, float* C, int ldc, int copies, int Co, int Ts,
                                      int B, int Hs, int Cs, int KH, int KW, int pad_h, int pad_t, int Tr, int Hr) {
    if (!g || !x || !C || Co <= 0 || Ts <= 0 || B <= 0 || Hs <= 0 || Cs <= 0 || KH <= 0 || KW <= 0 || Tr <= 0 || Hr <= 0)
        return ASR_ERR_BAD_ARG;
    const int N = KH * KW * Cs;
    const long long K = (long long)Tr * B * Hr;
    if (ldg < Co || ldc < N || (copies != 1 && copies != 8)) return ASR_ERR_BAD_ARG;
    if (copies == 8 && asr_conv_tn_copies(Co, Cs, KH, KW) != 8) return ASR_ERR_BAD_ARG;
    if ((Cs & 7) || K > 0x7fffffffLL || (((uintptr_t)x) & 15)) return ASR_ERR_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    {
        // the eight-wave kernel of csrc/gemm8.hip for the large gradients (at least one 256 x 256 tile's worth of outputs);
        // ASR_DEBUG tn_8ph=0 / tn256=2: never
        static const int use8 = debug_flag("tn_8ph", 1), tn256_forced = debug_flag("tn256", 1);
        if (use8 && tn256_forced != 2 && copies == 1 && Co >= 192 && N >= 192 && K >= 2048 &&      // (64 -> 128 channels, half-empty 256-row tiles: 147 -> 185 us)
            asr_conv_tn_8ph_ok(g, ldg, x, C, ldc, Co, Ts, B, Hs, Cs, KH, KW, Tr, Hr))
            return asr_conv_tn_acc_8ph(stream_, g, ldg, x, C, ldc, Co, Ts, B, Hs, Cs, KH, KW, pad_h, pad_t, Tr, Hr);
    }
    const ConvDesc cd{B, Hs, Cs, Ts, KH, KW, pad_h, pad_t, +1, Hr};
    if (tn256_ok(Co, ldg, 8, g, x, true)) {
        const int t2m = cdiv(Co, T2M), t2n = cdiv(N, T2N);
        int kps;
        const int sp = tn_splits(t2m * t2n, (int)K, kps);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_tn256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TN2_LDS_BYTES); attr = true; }
        hipLaunchKernelGGL(gemm_tn256_kernel<true>, dim3(8 * t2m * t2n * cdiv(sp, 8)), dim3(256), TN2_LDS_BYTES, stream,
                           (const uint16_t*)g, ldg, (const uint16_t*)x, 0, C, ldc, Co, N, (int)K, t2n, kps, cd);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    const int tiles_m = cdiv(Co, BM), tiles_n = cdiv(N, BN);
    int k_per_split;
    const int splits = tn_splits(tiles_m * tiles_n, (int)K, k_per_split);
    // (vector form: the virtual im2col chunks are whole 16-B pieces already -- Cs % 8 == 0 above; the gradient operand must be too)
    const unsigned long long x_bytes = (unsigned long long)Ts * B * Hs * Cs * 2;
    if (tn_vec_ok(g, ldg, x, 8, Co, N, (int)K) && x_bytes < 0xfffffff0ull) {
        hipLaunchKernelGGL(gemm_tn_vec_kernel<true>, dim3(tn_grid(tiles_m * tiles_n, splits)), dim3(256), TN_LDS_BYTES, stream,
                           (const uint16_t*)g, ldg, (const uint16_t*)x, 0, C, ldc, Co, N, (int)K, tiles_n, k_per_split, TnGroup{}, cd,
                           copies == 8 ? (long long)Co * ldc : 0LL);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    hipLaunchKernelGGL(gemm_tn_kernel<true>, dim3(tn_grid(tiles_m * tiles_n, splits)), dim3(256), TN_LDS_BYTES, stream,
                       (const uint16_t*)g, ldg, (const uint16_t*)x, 0, C, ldc, Co, N, (int)K, tiles_n, k_per_split, cd, copies == 8 ? (long long)Co * ldc : 0LL,
                       TnGroup{});
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_conv_tn_acc(void* stream_, const void* g, int ldg, const void* x, float* C, int ldc, int Co, int Ts, int B,
                               int Hs, int Cs, int KH, int KW, int pad_h, int pad_t, int Tr, int Hr) {
    return asr_conv_tn_acc_copies(stream_, g, ldg, x, C, ldc, 1, Co, Ts, B, Hs, Cs, KH, KW, pad_h, pad_t, Tr, Hr);
}

extern "C" int asr_conv_direct_ok(int Ts, int B, int Hs, int Cs, int KH, int KW, int Tr, int Hr, int N, int K, int out_bf16);
extern "C" int asr_conv_direct_nt(void* stream, const void* x, const void* W, int ldw, void* out, const float* bias, int Ts, int B, int Hs,
                                  int Cs, int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N);
extern "C" int asr_conv_nt(void* stream_, const void* x, const void* W, int ldw, void* out, int out_bf16, const float* bias,
                           int Ts, int B, int Hs, int Cs, int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N) {
    if (!x || !W || !out || Ts <= 0 || B <= 0 || Hs <= 0 || Cs <= 0 || KH <= 0 || KW <= 0 || Tr <= 0 || Hr <= 0 || N <= 0 ||
        (sgn != 1 && sgn != -1))
        return ASR_ERR_BAD_ARG;
    const long long M = (long long)Tr * B * Hr;
    const int K = ldw;                  // row pitch of W = K of the GEMM: KH*KW*Cs, or more with empty (zero) taps behind
    if (ldw < KH * KW * Cs) return ASR_ERR_BAD_ARG;
    {
        // the eight-wave kernel of csrc/gemm8.hip for the wide products -- more than 128 output columns, channels a multiple of 64 -- in front
        // of the LDS-resident kernel (tools/time_conv8.py, T = 1000, B = 32, 13 rows, 3 x 5 taps, us, LDS-resident / eight-wave:
        // 256 -> 512 channels 1769 / 1189, 128 -> 512 981 / 662, 256 -> 256 956 / 643, 512 -> 256 1724 / 1227: 0.83 - 0.95 -> 1.24 - 1.38 PFLOP/s).
        // ASR_DEBUG conv_8ph=0 (or an explicit nt_wide): never
        static const int c8 = debug_flag("conv_8ph", 1);
        if (c8 && nt_wide_mode() == -1 && N > 128 && asr_conv_nt_8ph_ok(x, W, ldw, out, out_bf16, bias, Ts, B, Hs, Cs, KH, KW, Tr, Hr, N))
            return asr_conv_nt_8ph(stream_, x, W, ldw, out, out_bf16, bias, Ts, B, Hs, Cs, KH, KW, pad_h, pad_t, sgn, Tr, Hr, N);
        // its narrow form (256 x 64 / 256 x 128 tiles) where the LDS-resident kernel needs several passes over the channels
        // (tools/time_conv8n.py: backward-data 256 -> 128 channels 527 -> 470 us, 512 -> 128 967 -> 904; with up to 128 channels the
        // LDS-resident kernel stays ahead: 64 -> 128 115 against 199, 128 -> 64 176 against 198 -- a narrow tile is bound by operand fill)
        if (c8 && nt_wide_mode() == -1 && N <= 128 && N >= 64 && Cs >= 256 &&
            asr_conv_nt_8pn_ok(x, W, ldw, out, out_bf16, bias, Ts, B, Hs, Cs, KH, KW, Tr, Hr, N))
            return asr_conv_nt_8pn(stream_, x, W, ldw, out, out_bf16, bias, Ts, B, Hs, Cs, KH, KW, pad_h, pad_t, sgn, Tr, Hr, N);
    }
    // the kernel with the activation block resident in LDS (conv_direct.hip; at most 128 channels of it at a time, more in passes):
    // T=1000, B=32, us, implicit GEMM / direct: 64 -> 64 channels 111 / 67, 128 -> 64 (a backward-data) 206 / 166, 128 -> 256 595 / 483,
    // 128 -> 512 1085 / 938, 64 -> 128 138 / 134, 256 -> 128 544 / 462, 512 -> 128 1040 / 884, 256 -> 256 998 / 902
    {
        static int direct = -1;
        if (direct < 0) direct = debug_flag("conv_direct", 1);
        if (direct && out_bf16 && asr_conv_direct_ok(Ts, B, Hs, Cs, KH, KW, Tr, Hr, N, ldw, 1) &&
            !((((uintptr_t)x) | ((uintptr_t)W) | ((uintptr_t)out)) & 15) && !(bias && (((uintptr_t)bias) & 15)))
            return asr_conv_direct_nt(stream_, x, W, ldw, out, bias, Ts, B, Hs, Cs, KH, KW, pad_h, pad_t, sgn, Tr, Hr, N);
    }
    if ((Cs & 7) || (K % B2K) || (K % Cs) || M > 0x7fffffffLL || ((((uintptr_t)x) | ((uintptr_t)W)) & 15)) return ASR_ERR_UNSUPPORTED;
    hipStream_t stream = (hipStream_t)stream_;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<float, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<uint16_t, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<float, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt256_kernel<uint16_t, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
        attr = true;
    }
    const ConvDesc cd{B, Hs, Cs, Ts, KH, KW, pad_h, pad_t, sgn, Hr};
    {
        const int wide = nt_wide_mode() == -1 ? 2 : nt_wide_mode();
        const int bkt = (wide == 2 || wide == 4) ? 64 : 32;
        if (wide && N > 128 && (Cs % bkt) == 0 && (K % bkt) == 0 && (cdiv((int)M, 256) * cdiv(N, 256) >= 256 || nt_wide_force())) {
            const bool fits = (size_t)N * K * 2 <= (size_t)5 * 512 * 1024;
            if (out_bf16) return launch_nt_wide<uint16_t, true>(stream, (const uint16_t*)x, 0, (const uint16_t*)W, K, (uint16_t*)out, N, bias, (int)M, N, K, fits, cd, wide);
            return launch_nt_wide<float, true>(stream, (const uint16_t*)x, 0, (const uint16_t*)W, K, (float*)out, N, bias, (int)M, N, K, fits, cd, wide);
        }
    }
    const bool narrow = N <= 64;                 // 256 x 64 tiles: no MFMA work on columns that do not exist
    const int t2m = cdiv((int)M, B2M), t2n = cdiv(N, narrow ? 64 : B2N);
    const bool b_fits_l2 = (size_t)N * K * 2 <= (size_t)5 * 512 * 1024;
    const int tn_arg = b_fits_l2 ? -t2n : t2n;
    // persistent form (K pipeline across tiles, register epilogue): channels a multiple of the K step, whole 16-B stores
    {
        static int persist = -1;
        if (persist < 0) persist = debug_flag("nt_persist", 1);
        const unsigned long long xbytes = (unsigned long long)Ts * B * Hs * Cs * 2;
        const int cmode = (Cs % B2K) == 0 ? 1 : (Cs == 8 ? 2 : 0);
        if (persist && cmode && (N % 8) == 0 && Hs < 256 && xbytes < 0xfffffff0ull && (((uintptr_t)out) & 15) == 0 &&
            (!bias || (((uintptr_t)bias) & 15) == 0) && (long long)t2m * t2n >= 400 && N < (1 << 24) && K < (1 << 23)) {
            static bool attrc = false;
            if (!attrc) {
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<float, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<uint16_t, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<float, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<uint16_t, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<float, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<uint16_t, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<float, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                (void)hipFuncSetAttribute((const void*)gemm_nt256p_kernel<uint16_t, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, NT2_LDS_BYTES);
                attrc = true;
            }
            const int total = t2m * t2n, grid = nt_persist_grid(total, 2);
#define ASR_CONVP(T, C_, NJ_, CT)                                                                                          \
    hipLaunchKernelGGL((gemm_nt256p_kernel<T, C_, NJ_>), dim3(grid), dim3(256), 2 * (B2M + 16 * NJ_) * B2K * 2, stream, (const uint16_t*)x, 0, \
                       (const uint16_t*)W, K, (CT*)out, N, bias, (int)M, N, K, t2m, t2n, total, (unsigned)xbytes, cd)
            if (cmode == 1) {
                if (out_bf16) { if (narrow) ASR_CONVP(uint16_t, 1, 4, uint16_t); else ASR_CONVP(uint16_t, 1, 8, uint16_t); }
                else          { if (narrow) ASR_CONVP(float, 1, 4, float); else ASR_CONVP(float, 1, 8, float); }
            } else {
                if (out_bf16) { if (narrow) ASR_CONVP(uint16_t, 2, 4, uint16_t); else ASR_CONVP(uint16_t, 2, 8, uint16_t); }
                else          { if (narrow) ASR_CONVP(float, 2, 4, float); else ASR_CONVP(float, 2, 8, float); }
            }
#undef ASR_CONVP
            ASR_LAUNCH_CHECK();
            return ASR_OK;
        }
    }
#define ASR_CONV(T, W_, CT)                                                                                               \
    hipLaunchKernelGGL((gemm_nt256_kernel<T, true, W_>), dim3(t2m * t2n), dim3(256), NT2_LDS_BYTES, stream, (const uint16_t*)x, 0, \
                       (const uint16_t*)W, K, (CT*)out, N, bias, (int)M, N, K, tn_arg, cd)
    if (out_bf16) { if (narrow) ASR_CONV(uint16_t, 1, uint16_t); else ASR_CONV(uint16_t, 2, uint16_t); }
    else          { if (narrow) ASR_CONV(float, 1, float); else ASR_CONV(float, 2, float); }
#undef ASR_CONV
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
