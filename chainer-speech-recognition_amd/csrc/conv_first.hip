// The first block of the acoustic models in one forward and one backward kernel (gfx950): Convolution2D over the (B, 3, 40, T) features
// -> Maxout(2) -> MaxPooling2D((k, 1))  (asr/nn/nn.py:235-238, :45-50, :95-103 as run/ctc/cnn/model.py:42-48 and the conv + recurrent
// template run/ctc/sru/model.py stack them).
//
// Unfused, this block is pure memory traffic around a convolution with K = 120: the convolution writes 311 MB (T=1000, B=32, 38 heights,
// 128 channels), maxout + pooling read them and keep one value in six; backward scatters the pooled gradient back into a 311 MB tensor of
// which five sixths are zeros, and the weight-gradient product reads that.  0.45 - 0.49 ms of the BASELINE configs[1] step around 37 GFLOP (15 us of MFMA time) each way.
//
// Forward: the input is (Ts, B, Hs, 8) bf16 (three real channels, asr_pack_input_pad), so ONE 16-byte chunk is one filter tap and one lane's
// share of an MFMA 16x16x32 A operand: the fragments come straight from global memory (L1 / L2 hits: every chunk is used by KH KW taps),
// the weights (128 channels x 128 k) live in 128 registers of every wave, no LDS and no barrier anywhere.  The rows of an MFMA tile are
// chosen for the epilogue: row 4 q + i = convolution row i of pooling window q, and column c of tile n = channel 8 c + n, so a lane ends up
// with the k candidates x 2 channels of FOUR maxout pairs of one window in its own accumulators -- maxout, pooling and the winner's index
// are lane-local, the store is 8 bytes per lane and 128 contiguous bytes per window (a quarter of the MFMA rows is idle at k = 3; the MFMA
// floor of this layer is 15 us).  Only the pooled output (53 MB) and one byte per output naming the winner (27 MB) are written.
//
// Backward: dW[ch][kcol] = sum over convolution rows of G[row][ch] * X[row][kcol], G = the pooled gradient at its winner and zero
// elsewhere, X = the virtual im2col row.  G is expanded from (gy, idx) on its way into LDS, X gathered as in the implicit TN kernel
// (gemm.hip), both transposed by ds_read_b64_tr_b16; each workgroup keeps a 128 x 128 accumulator over its share of the rows and writes it
// to a workspace, a second kernel sums the shares into the weight gradient's (Co, Ci, KH, KW) layout.  The empty sixteenth tap of a row
// holds a one, which makes column 8 KH KW of the product the bias gradient.
//
// Same values as asr_conv_nt + asr_maxout2_pool_fwd / asr_maxout2_pool_bwd_db + asr_conv_tn_acc (bf16 rounding of the convolution output
// before the comparisons, ties: first channel of the pair, first row of the window).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "common.hpp"

namespace asr {
namespace convf {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
union Frag {
    bf16x8 v;
    uint4 u;
    u32x4 r;
    bf16x4 h[2];
};

struct Desc {
    int Ts, B, Hs, KH, KW, ph, pt;      // input (Ts, B, Hs, 8); filter taps; padding (time: the causal / symmetric left pad)
    int Hout, Hp, k;                    // convolution output heights, pooled heights, pooling window (2 .. 4)
    int Co, Ci, taps;                   // output channels (multiple of 128), real input channels, KH KW (<= 15)
    int frames;                         // Tout * B
};

__device__ __forceinline__ uint4 ld16(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }

// Every load of these kernels is an UNCONDITIONAL buffer load: a chunk that does not exist gets an offset beyond the buffer's num_records
// and comes back as zeros, and the conditions are combined with & (not &&).  A load under a condition -- or a short-circuit condition in
// front of an address select -- becomes control flow: the loop body falls apart into basic blocks the scheduler cannot interleave, and
// with generic pointers (a select between a kernel argument and a __device__ constant) the loads turn into flat_load behind vmcnt(0).
constexpr unsigned OOB = 0xfffffff0u;

// pooled row G = frame * Hp + hp, advanced by a fixed number of rows
struct Walk {
    int tb, hp;
    __device__ __forceinline__ void init(int G, int Hp) { tb = G / Hp; hp = G - tb * Hp; }
    __device__ __forceinline__ void step(int dtb, int dhp, int Hp) {
        hp += dhp;
        const int c = hp >= Hp;
        hp -= c ? Hp : 0;
        tb += dtb + c;
    }
};

// one tap of a lane / thread: offset of its chunk from the chunk of (frame, h) and the frames for which the time step exists
struct Tap {
    int off, lo, hi, dh;
    __device__ __forceinline__ void init(int tap, const Desc& d) {
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
        const int dt = kw - d.pt;
        dh = kh - d.ph;
        off = (dt * d.B * d.Hs + dh) * 16;             // bytes
        lo = tap < d.taps ? -dt * d.B : INT_MAX;        // t + dt >= 0  <=>  frame >= -dt B   (a tap that does not exist: never)
        hi = (d.Ts - dt) * d.B;                         // t + dt < Ts  <=>  frame < (Ts - dt) B
    }
};

// ------------------------------------------------------------------------------------------------ forward
// Epilogue arithmetic is what a tile costs beside its 32 MFMAs (512 cycles): every vector instruction of a wave takes 4.  Per maxout pair
// and row: two bias adds, ONE packed conversion (the rounding of both channels), two unpacks, compare + max; then the running maximum over
// the rows with the winner's index -- ~30 instructions per pair, ~125 per tile (the first version converted and compared value by value
// with run-time row checks: 109 us for the BASELINE layer, three quarters of it vector ALU time).
template <int KP>
__global__ __launch_bounds__(256, 2) void fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ W,
                                                    const float* __restrict__ bias, uint16_t* __restrict__ y, uint8_t* __restrict__ idx,
                                                    Desc d, int tiles, int tiles_per_xcd) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int cb = blockIdx.y, Cp = d.Co >> 1;
    Frag w[8][4];
    float bs[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
#pragma unroll
        for (int s = 0; s < 4; ++s) w[n][s].u = ld16(W + (size_t)(cb * 128 + 8 * c + n) * 128 + 32 * s + 8 * g);
        bs[n] = bias ? bias[cb * 128 + 8 * c + n] : 0.f;
    }
    Tap tp[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) tp[s].init(4 * s + g, d);

    // an XCD owns a contiguous run of tiles (a tile = four pooling windows): neighbouring windows read the same input rows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wpx = (int)(gridDim.x >> 3) * 4;              // waves per XCD
    const int last = min(tiles, (xcd + 1) * tiles_per_xcd);
    int tile = xcd * tiles_per_xcd + slot * 4 + wid;
    const int qa = (lane >> 2) & 3, ia = lane & 3;          // A operand: row lane & 15 = window qa, convolution row ia of it
    Walk wa, we;                                            // ... and the epilogue's window: lane >> 4
    wa.init(tile * 4 + qa, d.Hp);
    we.init(tile * 4 + g, d.Hp);
    const int dtb = (4 * wpx) / d.Hp, dhp = (4 * wpx) - dtb * d.Hp;

    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)d.Ts * d.B * d.Hs * 16u), 0x00020000);
    auto load_a = [&](Frag (&a)[4], bool tile_ok) {
        const int h = wa.hp * KP + ia;
        const bool rok = tile_ok & (wa.tb < d.frames) & (ia < KP) & (h < d.Hout);
        const unsigned base = (unsigned)((wa.tb * d.Hs + h) * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ok = rok & (wa.tb >= tp[s].lo) & (wa.tb < tp[s].hi) & ((unsigned)(h + tp[s].dh) < (unsigned)d.Hs);
            a[s].r = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? base + (unsigned)tp[s].off : OOB, 0, 0);
        }
    };
    // one tile: the NEXT tile's chunks are asked for first and are in flight during this tile's MFMAs and epilogue
    auto process = [&](Frag (&a)[4], Frag (&an)[4]) {
        const int cur = tile;
        tile += wpx;
        wa.step(dtb, dhp, d.Hp);
        load_a(an, tile < last);

        f32x4 acc[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = ASR_MFMA_16x16x32(a[0].v, w[n][0].v, ((f32x4){0.f, 0.f, 0.f, 0.f}));
#pragma unroll
        for (int s = 1; s < 4; ++s)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = ASR_MFMA_16x16x32(a[s].v, w[n][s].v, acc[n]);

        // lane (c, g): window cur * 4 + g, channels 8 c .. 8 c + 7 = pairs 4 c .. 4 c + 3, convolution rows 0 .. 3 of the window in acc[n][0 .. 3]
        if (we.tb < d.frames) {
            const int nv = d.Hout - we.hp * KP;             // rows of this window inside the convolution output (>= 1; the last window may be short)
            float best[4];
            uint32_t wins = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                uint32_t win = 0;
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const uint32_t pk = pack_bf16x2(acc[2 * m][j] + bs[2 * m], acc[2 * m + 1][j] + bs[2 * m + 1]);
                    const float lo = bf16_to_f32((uint16_t)(pk & 0xffffu)), hi = bf16_to_f32((uint16_t)(pk >> 16));
                    const bool second = hi > lo;
                    float v = second ? hi : lo;
                    if (j == 0) {
                        best[m] = v;
                        win = second ? 1u : 0u;
                    } else {
                        v = j < nv ? v : -INFINITY;
                        const bool take = v > best[m];
                        best[m] = take ? v : best[m];
                        win = take ? (second ? 2u * j + 1u : 2u * j) : win;
                    }
                }
                wins |= win << (8 * m);
            }
            const size_t at = (size_t)(cur * 4 + g) * Cp + cb * 64 + 4 * c;
            *reinterpret_cast<uint2*>(y + at) = make_uint2(pack_bf16x2(best[0], best[1]), pack_bf16x2(best[2], best[3]));
            *reinterpret_cast<uint32_t*>(idx + at) = wins;
        }
        we.step(dtb, dhp, d.Hp);
    };

    Frag a0[4], a1[4];
    load_a(a0, tile < last);
    while (tile < last) {               // two tiles per trip: the register sets keep their names (no copies)
        process(a0, a1);
        if (!(tile < last)) break;
        process(a1, a0);
    }
}

// ------------------------------------------------------------------------------------------------ backward
constexpr int TK = 32;
constexpr int TP = 128 + 16;        // row pitch in elements (288 B): 8 consecutive rows cover all 64 banks (as the TN kernel of gemm.hip)

__device__ __forceinline__ bf16x4 lds_tr16(const uint16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)p);
}

// A k tile = 8 pooling windows = 32 virtual convolution rows (window q: rows 4 q .. 4 q + 3, rows >= k or beyond Hout are empty).
// part[p][ch][kcol]: the share of workgroup p (rows of its k tiles), kcol = tap * 8 + ci, column taps * 8 = the bias gradient.
__global__ __launch_bounds__(256, 2) void bwd_kernel(const uint16_t* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                    const uint16_t* __restrict__ x, float* __restrict__ part, Desc d, int ktiles,
                                                    int kt_per_wg) {
    __shared__ __attribute__((aligned(16))) uint16_t Gs[2 * TK * TP];
    __shared__ __attribute__((aligned(16))) uint16_t Xs[2 * TK * TP];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int cb = blockIdx.y, Cp = d.Co >> 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int p = xcd * (int)(gridDim.x >> 3) + slot;       // an XCD's workgroups own a contiguous run of rows
    const int kbeg = p * kt_per_wg, kend = min(ktiles, kbeg + kt_per_wg);
    const long long groups = (long long)d.frames * d.Hp;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // G image: thread = (window fg of the tile, 16-byte chunk fc = pairs 4 fc .. 4 fc + 3, rows 2 fh and 2 fh + 1 of the window)
    const int fg = tid >> 5, fc = (tid >> 1) & 15, fh = tid & 1;
    // X image: thread = (tap tid & 15, virtual rows tid >> 4 and 16 + (tid >> 4)): windows (tid >> 6) and 4 + (tid >> 6), row (tid >> 4) & 3 of them
    const int xtap = tid & 15, xi = (tid >> 4) & 3;
    Tap tp;
    tp.init(xtap, d);
    const bool ones = xtap == d.taps;                       // the column of ones: the bias gradient
    const uint32_t one = f32_to_bf16(1.f);
    const unsigned gy_bytes = (unsigned)(groups * Cp * 2);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)d.Ts * d.B * d.Hs * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc((void*)gy, 0, (int)gy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_i = __builtin_amdgcn_make_buffer_rsrc((void*)idx, 0, (int)(gy_bytes >> 1), 0x00020000);
    Walk wx[2];
    wx[0].init(kbeg * 8 + (tid >> 6), d.Hp);
    wx[1].init(kbeg * 8 + 4 + (tid >> 6), d.Hp);
    const int dtb = 8 / d.Hp, dhp = 8 - dtb * d.Hp;

    // Two k tiles of operands are in flight in registers while a third is worked on: the loads of tile kt + 2 are issued before the MFMAs of
    // tile kt and stored to LDS at the end of tile kt + 1 (with ONE tile ahead every iteration waited out a trip to memory: 2000 cycles for
    // 256 cycles of MFMA, 102 us for the BASELINE layer).
    struct Stage {
        uint2 rg;
        uint32_t ri;
        uint4 rx[2];
    };
    auto load_global = [&](Stage& st, int kt) {     // called with kt = kbeg, kbeg + 1, ... in order (the row walk relies on it)
        const long long G = (long long)kt * 8 + fg;
        const bool gok = (kt < kend) & (G < groups);
        const unsigned at = (unsigned)G * (unsigned)Cp + cb * 64 + 4 * fc;         // (pooled rows * Co / 2 < 2^31: fill_desc)
        const u32x2 rg = __builtin_amdgcn_raw_buffer_load_b64(rsrc_g, gok ? at * 2 : OOB, 0, 0);
        st.rg = make_uint2(rg[0], rg[1]);
        st.ri = __builtin_amdgcn_raw_buffer_load_b32(rsrc_i, gok ? at : OOB, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int h = wx[i].hp * d.k + xi;
            const bool ok = (kt < kend) & (wx[i].tb < d.frames) & (xi < d.k) & (h < d.Hout) & (wx[i].tb >= tp.lo) & (wx[i].tb < tp.hi) &
                            ((unsigned)(h + tp.dh) < (unsigned)d.Hs);
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? (unsigned)((wx[i].tb * d.Hs + h) * 16 + tp.off) : OOB, 0, 0);
            st.rx[i] = ones ? make_uint4(one, 0u, 0u, 0u) : make_uint4(v[0], v[1], v[2], v[3]);
            wx[i].step(dtb, dhp, d.Hp);
        }
    };
    auto store_lds = [&](const Stage& st, int buf) {
        const uint32_t gv[4] = {st.rg.x & 0xffffu, st.rg.x >> 16, st.rg.y & 0xffffu, st.rg.y >> 16};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t i = 2 * fh + r;                  // row of the window: the winner byte is 2 * row + (second channel of the pair)
            uint32_t o[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint32_t b = (st.ri >> (8 * jj)) & 0xffu;
                o[jj] = b == 2 * i ? gv[jj] : (b == 2 * i + 1 ? gv[jj] << 16 : 0u);
            }
            *reinterpret_cast<uint4*>(Gs + (buf * TK + fg * 4 + i) * TP + fc * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(Xs + (buf * TK + (tid >> 4) + 16 * i) * TP + xtap * 8) = st.rx[i];
    };

    // transposing reads (as gemm_tn_kernel): 16-lane group g = lane >> 4, lane 4 q + pp of the group points at row 4 g + q (second read:
    // 16 + 4 g + q), columns 4 pp .. 4 pp + 3 of a 16-column tile; the k slots are permuted identically for both operands
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int trow0 = 4 * g + q, trow1 = 16 + 4 * g + q, tcol = 4 * pp;
    auto work = [&](int buf) {
        const uint16_t* Gb = Gs + buf * TK * TP;
        const uint16_t* Xb = Xs + buf * TK * TP;
        Frag a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ac = wm * 64 + i * 16 + tcol;
            a[i].h[0] = lds_tr16(Gb + trow0 * TP + ac);
            a[i].h[1] = lds_tr16(Gb + trow1 * TP + ac);
            const int bc = wn * 64 + i * 16 + tcol;
            b[i].h[0] = lds_tr16(Xb + trow0 * TP + bc);
            b[i].h[1] = lds_tr16(Xb + trow1 * TP + bc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = ASR_MFMA_16x16x32(a[i].v, b[j].v, acc[i][j]);
    };

    if (kbeg < kend) {
        Stage s0, s1;
        load_global(s0, kbeg);
        load_global(s1, kbeg + 1);
        store_lds(s0, 0);
        __syncthreads();
        // tiles in pairs (straight-line body, the two register stages keep their names); a tile beyond kend is all zeros
        for (int kt = kbeg; kt < kend; kt += 2) {
            load_global(s0, kt + 2);
            work(0);
            store_lds(s1, 1);
            __syncthreads();
            load_global(s1, kt + 3);
            work(1);
            store_lds(s0, 0);
            __syncthreads();
        }
    }
    float* out = part + ((size_t)p * d.Co + cb * 128) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                const int gn = wn * 64 + j * 16 + (lane & 15);
                out[gm * 128 + gn] = acc[i][j][r];
            }
}

// ------------------------------------------------------------------------------------------------ backward, one frame per iteration
// The same product with a frame (t, b) as the unit of work -- the form the BASELINE layers take (at most 16 pooling windows per frame).
// What the general kernel above spends its vector instructions on is the im2col image: 512 chunks per k tile, each with its own
// (frame, height, tap) address and seven bounds checks.  Here the frame's input neighbourhood -- KW time steps x (heights + filter
// margin) chunks, 260 for the BASELINE layer against the 832 im2col chunks of its 52 rows -- goes to LDS as it is, rows and time steps
// outside the tensor as zeros, and a transposing read takes its four channels of (row, tap) straight from that block at
// (kw HR + h + kh) 16 + (4-channel half) 8: the addresses of a lane's reads do not change from frame to frame, so the loop computes none.
// The k dimension of a frame is its 4 Hp virtual rows padded to 32 or 64; G as above, one (window, 8-channel chunk) per thread.
// The ones for the bias column and the zeros for taps / windows that do not exist are two constant chunks in LDS.
constexpr int FR_MAX_CHUNKS = 512;          // chunks of the raw block: at most two per thread

__global__ __launch_bounds__(256, 2) void bwd_frame_kernel(const uint16_t* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                          const uint16_t* __restrict__ x, float* __restrict__ part, Desc d, int HR,
                                                          int nks, int frames_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [0, 16): a one in front (bias column); [16, 32): zeros; then two stages of { G: 32 nks rows x TP, raw block: KW x HR chunks }
    const int g_bytes = 32 * nks * TP * 2, x_bytes = d.KW * HR * 16, stage_bytes = g_bytes + x_bytes;
    char* const stages = smem + 32;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int cb = blockIdx.y, Cp = d.Co >> 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int p = xcd * (int)(gridDim.x >> 3) + slot;       // an XCD's workgroups own a contiguous run of frames
    const int f0 = p * frames_per_wg, f1 = min(d.frames, f0 + frames_per_wg);
    if (tid < 8) reinterpret_cast<uint32_t*>(smem)[tid] = tid == 0 ? (uint32_t)f32_to_bf16(1.f) : 0u;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const unsigned gy_bytes = (unsigned)((long long)d.frames * d.Hp * Cp * 2);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)d.Ts * d.B * d.Hs * 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc((void*)gy, 0, (int)gy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_i = __builtin_amdgcn_make_buffer_rsrc((void*)idx, 0, (int)(gy_bytes >> 1), 0x00020000);

    // raw block: chunk c = time slot c / HR (input time t + slot - pt), block row c % HR (input height row - ph)
    const int nchunks = d.KW * HR;
    int xc_off[2], xc_lo[2], xc_hi[2];
    bool xc_row[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + 256 * i;
        const int sl = c / HR, r = c - sl * HR, h = r - d.ph, dt = sl - d.pt;
        xc_row[i] = (c < nchunks) & ((unsigned)h < (unsigned)d.Hs);
        xc_off[i] = (dt * d.B * d.Hs + h) * 16;
        xc_lo[i] = -dt * d.B;                               // t + dt >= 0 <=> frame >= -dt B;  t + dt < Ts <=> frame < (Ts - dt) B
        xc_hi[i] = (d.Ts - dt) * d.B;
    }
    // G: thread = (window tid >> 4, chunk tid & 15 = pairs 4 fc .. 4 fc + 3)
    const int fw = tid >> 4, fc = tid & 15;
    const bool g_row = fw < d.Hp;
    const bool g_store = fw < 8 * nks;

    struct Stage {
        uint2 rg;
        uint32_t ri;
        u32x4 rx[2];
    };
    auto load_global = [&](Stage& st, int f) {
        const bool fok = f < f1;
        const unsigned at = ((unsigned)f * (unsigned)d.Hp + fw) * (unsigned)Cp + cb * 64 + 4 * fc;
        const bool gok = fok & g_row;
        const u32x2 rg = __builtin_amdgcn_raw_buffer_load_b64(rsrc_g, gok ? at * 2 : OOB, 0, 0);
        st.rg = make_uint2(rg[0], rg[1]);
        st.ri = __builtin_amdgcn_raw_buffer_load_b32(rsrc_i, gok ? at : OOB, 0, 0);
        const unsigned fbase = (unsigned)f * (unsigned)d.Hs * 16u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = fok & xc_row[i] & (f >= xc_lo[i]) & (f < xc_hi[i]);
            st.rx[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? fbase + (unsigned)xc_off[i] : OOB, 0, 0);
        }
    };
    auto store_lds = [&](const Stage& st, int buf) {
        char* const base = stages + buf * stage_bytes;
        if (g_store) {
            const uint32_t gv[4] = {st.rg.x & 0xffffu, st.rg.x >> 16, st.rg.y & 0xffffu, st.rg.y >> 16};
            uint32_t val[4], row[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const uint32_t b = (st.ri >> (8 * jj)) & 0xffu;             // winner: 2 * row + (second channel of the pair)
                val[jj] = (b & 1u) ? gv[jj] << 16 : gv[jj];
                row[jj] = b >> 1;
            }
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                uint4 o;
                o.x = row[0] == i ? val[0] : 0u;
                o.y = row[1] == i ? val[1] : 0u;
                o.z = row[2] == i ? val[2] : 0u;
                o.w = row[3] == i ? val[3] : 0u;
                *reinterpret_cast<uint4*>(base + ((fw * 4 + i) * TP + fc * 8) * 2) = o;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + 256 * i < nchunks) *reinterpret_cast<u32x4*>(base + g_bytes + (tid + 256 * i) * 16) = st.rx[i];
    };

    // transposing reads: lane 4 q + pp of 16-lane group g points at k row 4 g + q (second read: 16 + 4 g + q), columns 4 pp .. 4 pp + 3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int trow0 = 4 * g + q, tcol = 4 * pp;
    // X operand: k row (k step ks, half hf) = virtual row 32 ks + 16 hf + 4 g + q = window 8 ks + 4 hf + g, convolution row q of it;
    // column tile j of this wave = taps 2 (4 wn + j) and + 1, this lane's tap by pp >> 1, its channel half by pp & 1
    int xrow[2][2], xtap[4];
    bool xabs[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int w = 8 * ks + 4 * hf + g;
            xrow[ks][hf] = w < d.Hp ? (w * d.k + q) * 16 : -1;               // -1: a window that does not exist (G is zero there)
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tap = 2 * (4 * wn + j) + (pp >> 1);
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
        xabs[j] = tap >= d.taps;                                             // the column of ones, or a tap that does not exist
        xtap[j] = tap < d.taps ? 32 + g_bytes + (kw * HR + kh) * 16 + (pp & 1) * 8 : (tap == d.taps ? (pp & 1) * 8 : 16);
    }
    auto work = [&](int buf) {
        const int sb = buf * stage_bytes;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks < nks) {
                const uint16_t* Gb = reinterpret_cast<const uint16_t*>(stages + sb) + ks * 32 * TP;
                Frag a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ac = wm * 64 + i * 16 + tcol;
                    a[i].h[0] = lds_tr16(Gb + trow0 * TP + ac);
                    a[i].h[1] = lds_tr16(Gb + (trow0 + 16) * TP + ac);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int at = (xabs[i] | (xrow[ks][hf] < 0)) ? (xabs[i] ? xtap[i] : 16) : sb + xtap[i] + xrow[ks][hf];
                        b[i].h[hf] = lds_tr16(reinterpret_cast<const uint16_t*>(smem + at));
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = ASR_MFMA_16x16x32(a[i].v, b[j].v, acc[i][j]);
            }
        }
    };

    if (f0 < f1) {
        Stage s0, s1;
        load_global(s0, f0);
        load_global(s1, f0 + 1);
        store_lds(s0, 0);
        __syncthreads();
        for (int f = f0; f < f1; f += 2) {      // frames in pairs: the two register stages keep their names; a frame beyond f1 is all zeros
            load_global(s0, f + 2);
            work(0);
            store_lds(s1, 1);
            __syncthreads();
            load_global(s1, f + 3);
            work(1);
            store_lds(s0, 0);
            __syncthreads();
        }
    }
    float* out = part + ((size_t)p * d.Co + cb * 128) * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = wm * 64 + i * 16 + (lane >> 4) * 4 + r;
                const int gn = wn * 64 + j * 16 + (lane & 15);
                out[gm * 128 + gn] = acc[i][j][r];
            }
}

// gW[ch][ci][kh][kw] += sum_p part[p][ch][(kh KW + kw) 8 + ci],  gb[ch] += sum_p part[p][ch][8 KH KW]; blockIdx.y = a slice of the shares
__global__ void bwd_reduce_kernel(const float* __restrict__ part, int nparts, float* __restrict__ gW, float* __restrict__ gb, Desc d) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= d.Co * 128) return;
    const int ch = o >> 7, kc = o & 127;
    const int tap = kc >> 3, ci = kc & 7;
    const bool w_ok = tap < d.taps && ci < d.Ci, b_ok = tap == d.taps && ci == 0 && gb != nullptr;
    if (!w_ok && !b_ok) return;
    const int per = (nparts + (int)gridDim.y - 1) / (int)gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(nparts, p0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const size_t pitch = (size_t)d.Co * 128;
    int p = p0;
    for (; p + 3 < p1; p += 4) {
        s0 += part[(size_t)p * pitch + o];
        s1 += part[(size_t)(p + 1) * pitch + o];
        s2 += part[(size_t)(p + 2) * pitch + o];
        s3 += part[(size_t)(p + 3) * pitch + o];
    }
    for (; p < p1; ++p) s0 += part[(size_t)p * pitch + o];
    const float s = (s0 + s1) + (s2 + s3);
    if (w_ok) {
        const int kh = tap / d.KW, kw = tap - kh * d.KW;
        atomicAdd(gW + ((size_t)(ch * d.Ci + ci) * d.KH + kh) * d.KW + kw, s);
    } else {
        atomicAdd(gb + ch, s);
    }
}

static int parts_for(int Co, int ktiles) {
    static int forced = -1;             // ASR_DEBUG conv_mp_parts=N (experiments): workgroups of the backward kernel per 128-channel block
    if (forced < 0) forced = debug_flag("conv_mp_parts", 0);
    int n = forced > 0 ? ((forced + 7) & ~7) : 512 / (Co / 128);
    if (n < 128) n = 128;
    const int need = (ktiles + 7) & ~7;
    if (n > need) n = need;
    return n < 8 ? 8 : n;
}

static bool fill_desc(Desc& d, int Ts, int B, int Hs, int Ci, int KH, int KW, int pad_h, int pad_t, int Tout, int Hout, int Co, int k) {
    if (Ts <= 0 || B <= 0 || Hs <= 0 || Tout <= 0 || Hout <= 0 || Ci <= 0 || Ci > 8) return false;
    d.Ts = Ts; d.B = B; d.Hs = Hs; d.KH = KH; d.KW = KW; d.ph = pad_h; d.pt = pad_t;
    d.Hout = Hout; d.k = k;
    d.Hp = Hout <= k ? 1 : (Hout - k + k - 1) / k + 1;
    d.Co = Co; d.Ci = Ci; d.taps = KH * KW;
    if ((long long)Tout * B > INT_MAX / 64 / d.Hp) return false;           // pooled rows * Co / 2 and rows * 4 stay inside 31 bits
    if ((long long)Ts * B * Hs * 8 >= (1ll << 30)) return false;
    d.frames = Tout * B;
    return true;
}

}  // namespace convf
}  // namespace asr

using namespace asr;
using namespace asr::convf;

// 1 when the fused first block serves a layer: 8 padded input channels (Ci <= 8 real ones), at most 15 taps (the sixteenth chunk of a
// 128-wide im2col row carries the bias gradient's column of ones), output channels in blocks of 128, pooling window 2 .. 4
extern "C" int asr_conv_mp_ok(int Ci, int KH, int KW, int Co, int k) {
    return Ci >= 1 && Ci <= 8 && KH >= 1 && KW >= 1 && KH * KW <= 15 && Co >= 128 && Co % 128 == 0 && k >= 2 && k <= 4;
}

extern "C" int asr_conv_mp_fwd(void* stream_, const void* x8, const void* W, int ldw, const float* bias, void* y, void* idx, int Ts, int B,
                               int Hs, int KH, int KW, int pad_h, int pad_t, int Tout, int Hout, int Co, int k) {
    if (!x8 || !W || !y || !idx) return ASR_ERR_BAD_ARG;
    if (!asr_conv_mp_ok(1, KH, KW, Co, k) || ldw != 128) return ASR_ERR_UNSUPPORTED;
    if (((((uintptr_t)x8) | ((uintptr_t)W) | ((uintptr_t)y)) & 15) || (((uintptr_t)idx) & 3)) return ASR_ERR_UNSUPPORTED;
    Desc d;
    if (!fill_desc(d, Ts, B, Hs, 1, KH, KW, pad_h, pad_t, Tout, Hout, Co, k)) return ASR_ERR_UNSUPPORTED;
    const long long groups = (long long)d.frames * d.Hp;
    const int tiles = (int)((groups + 3) / 4);
    int wgs = 512;                                          // two workgroups of four waves per CU
    while (wgs > 8 && (wgs / 2) * 4 >= tiles) wgs /= 2;
    const int tiles_per_xcd = (tiles + 7) / 8;
#define ASR_CF(KP_)                                                                                                                   \
    hipLaunchKernelGGL(fwd_kernel<KP_>, dim3(wgs, Co / 128), dim3(256), 0, (hipStream_t)stream_, (const uint16_t*)x8, (const uint16_t*)W, \
                       bias, (uint16_t*)y, (uint8_t*)idx, d, tiles, tiles_per_xcd)
    if (k == 2) ASR_CF(2); else if (k == 3) ASR_CF(3); else ASR_CF(4);
#undef ASR_CF
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

// bytes of workspace asr_conv_mp_bwd needs for a layer of this size
extern "C" long long asr_conv_mp_bwd_workspace(int Tout, int B, int Hout, int Co, int k) {
    if (Tout <= 0 || B <= 0 || Hout <= 0 || Co < 128 || k < 1) return 0;
    const int Hp = Hout <= k ? 1 : (Hout - k + k - 1) / k + 1;
    const long long ktiles = ((long long)Tout * B * Hp + 7) / 8;
    return (long long)parts_for(Co, (int)(ktiles > 4096 ? 4096 : ktiles)) * Co * 128 * 4;
}

extern "C" int asr_conv_mp_bwd(void* stream_, const void* gy, const void* idx, const void* x8, void* workspace, float* gW, float* gb, int Ts,
                               int B, int Hs, int Ci, int KH, int KW, int pad_h, int pad_t, int Tout, int Hout, int Co, int k) {
    if (!gy || !idx || !x8 || !workspace || !gW) return ASR_ERR_BAD_ARG;
    if (!asr_conv_mp_ok(Ci, KH, KW, Co, k)) return ASR_ERR_UNSUPPORTED;
    if (((((uintptr_t)x8) | ((uintptr_t)workspace)) & 15) || (((uintptr_t)gy) & 7) || (((uintptr_t)idx) & 3)) return ASR_ERR_UNSUPPORTED;
    Desc d;
    if (!fill_desc(d, Ts, B, Hs, Ci, KH, KW, pad_h, pad_t, Tout, Hout, Co, k)) return ASR_ERR_UNSUPPORTED;
    const long long groups = (long long)d.frames * d.Hp;
    const int ktiles = (int)((groups + 7) / 8);
    const int nparts = parts_for(Co, ktiles > 4096 ? 4096 : ktiles);
    const int kt_per_wg = (ktiles + nparts - 1) / nparts;
    hipStream_t stream = (hipStream_t)stream_;
    // one frame per iteration where a frame has at most 16 pooling windows and its input neighbourhood at most 512 chunks
    // (ASR_DEBUG conv_mp_frame=0: the general kernel)
    // block rows: every height a (window, row, tap) can name; HR = 4 (mod 8) puts the two taps of a transposing read 64 bytes (mod 128) apart
    int HR = d.Hp * k + KH + 3 > Hs + KH - 1 ? d.Hp * k + KH + 3 : Hs + KH - 1;
    HR += (4 - HR % 8 + 8) % 8;
    static int use_frame = -1;
    if (use_frame < 0) use_frame = debug_flag("conv_mp_frame", 1);
    if (use_frame && d.Hp <= 16 && KW * HR <= FR_MAX_CHUNKS && (long long)d.frames * Hs * 16 < (1ll << 31)) {
        const int nks = d.Hp <= 8 ? 1 : 2;
        const int lds = 32 + 2 * (32 * nks * TP * 2 + KW * HR * 16);
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)bwd_frame_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); attr = true; }
        if (lds <= 64 * 1024) {
            const int frames_per_wg = (d.frames + nparts - 1) / nparts;
            hipLaunchKernelGGL(bwd_frame_kernel, dim3(nparts, Co / 128), dim3(256), lds, stream, (const uint16_t*)gy, (const uint8_t*)idx,
                               (const uint16_t*)x8, (float*)workspace, d, HR, nks, frames_per_wg);
            ASR_LAUNCH_CHECK();
            const int slices_f = nparts >= 256 ? 16 : (nparts >= 64 ? 4 : 1);
            hipLaunchKernelGGL(bwd_reduce_kernel, dim3((Co * 128 + 255) / 256, slices_f), dim3(256), 0, stream, (const float*)workspace, nparts, gW, gb, d);
            ASR_LAUNCH_CHECK();
            return ASR_OK;
        }
    }
    hipLaunchKernelGGL(bwd_kernel, dim3(nparts, Co / 128), dim3(256), 0, stream, (const uint16_t*)gy, (const uint8_t*)idx, (const uint16_t*)x8,
                       (float*)workspace, d, ktiles, kt_per_wg);
    ASR_LAUNCH_CHECK();
    const int slices = nparts >= 256 ? 16 : (nparts >= 64 ? 4 : 1);        // (16 adds per address; 1024 workgroups keep the 33 MB of shares streaming)
    hipLaunchKernelGGL(bwd_reduce_kernel, dim3((Co * 128 + 255) / 256, slices), dim3(256), 0, stream, (const float*)workspace, nparts, gW, gb, d);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
