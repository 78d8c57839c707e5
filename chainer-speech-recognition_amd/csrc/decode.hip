// Greedy CTC decoding and character error rate on the GPU (gfx950) -- SURVEY row f1.
//
// Replaces, for a whole minibatch at once,
//   xp.argmax(y_batch.data, axis=2)                       run/ctc/cnn/dev.py:106
//   the blank / repeat collapse loop                      asr/error.py:38-47 (compute_minibatch_error)
//   compute_character_error_rate (Levenshtein / len(r))   asr/error.py:7-24
//
// argmax_rows : one wave per (t, b) row of the (T, B, V) f32 logits; first index of the maximum (np.argmax tie rule).
// collapse    : one workgroup per utterance; frame t survives iff id[t] != blank and id[t] != id[t-1]
//               (asr/error.py:41-47: prev_token is the previous frame's token, reset to BLANK by a blank frame, which is
//               the same predicate); survivors are compacted in order with a block scan.  merge_repeats = 0 only drops
//               blanks: the label side of asr/error.py:33-37.
// edit_distance: one workgroup per (reference, hypothesis) pair; the DP table is swept by anti-diagonals held in LDS
//               (three diagonals of len(r) + 1 ints).  Arithmetic is exact int32: the reference's table is uint8
//               (asr/error.py:10) and is only defined for sequences up to 255 tokens, where the two agree.
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace decode {

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int T, int B, int V,
                                                          int32_t* __restrict__ ids) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;       // row = t * B + b
    if (row >= (long long)T * B) return;
    const float* p = x + row * V;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
        const float f = p[v];
        if (f > best || (f == best && v < bi)) { best = f; bi = v; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        const int t = (int)(row / B), b = (int)(row - (long long)t * B);
        ids[(size_t)b * T + t] = bi == 0x7fffffff ? 0 : bi;
    }
}

__global__ __launch_bounds__(256) void collapse_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lengths,
                                                       int T, int blank, int merge_repeats, int32_t* __restrict__ out,
                                                       int32_t* __restrict__ out_len) {
    __shared__ int wsum[4];
    __shared__ int base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t* src = ids + (size_t)b * T;
    int32_t* dst = out + (size_t)b * T;
    const int len = lengths ? min(lengths[b], T) : T;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int t0 = 0; t0 < len; t0 += 256) {
        const int t = t0 + tid;
        int id = blank, keep = 0;
        if (t < len) {
            id = src[t];
            keep = id != blank && (!merge_repeats || t == 0 || id != src[t - 1]);
        }
        // exclusive scan of keep over the 256 threads
        int incl = keep;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int n = __shfl_up(incl, off);
            if (lane >= off) incl += n;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = base;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (keep) dst[before + incl - 1] = id;
        __syncthreads();
        if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    const int total = base;
    for (int t = total + tid; t < T; t += 256) dst[t] = blank;      // padded with blank like the label arrays
    if (tid == 0) out_len[b] = total;
}

// d[i][j] over anti-diagonals k = i + j; diag arrays are indexed by i
__global__ __launch_bounds__(64) void edit_distance_kernel(const int32_t* __restrict__ ref, const int32_t* __restrict__ ref_len,
                                                           int ref_pitch, const int32_t* __restrict__ hyp,
                                                           const int32_t* __restrict__ hyp_len, int hyp_pitch,
                                                           int32_t* __restrict__ dist) {
    extern __shared__ int lds[];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int n = min(ref_len[p], ref_pitch), m = min(hyp_len[p], hyp_pitch);
    const int32_t* r = ref + (size_t)p * ref_pitch;
    const int32_t* h = hyp + (size_t)p * hyp_pitch;
    int* d0 = lds;                  // diagonal k - 2
    int* d1 = lds + (ref_pitch + 1);      // diagonal k - 1
    int* d2 = lds + 2 * (ref_pitch + 1);  // diagonal k
    if (n == 0 || m == 0) {
        if (lane == 0) dist[p] = n == 0 ? m : n;
        return;
    }
    for (int k = 0; k <= n + m; ++k) {
        const int ilo = max(0, k - m), ihi = min(n, k);
        for (int i = ilo + lane; i <= ihi; i += 64) {
            const int j = k - i;
            int v;
            if (i == 0) v = j;
            else if (j == 0) v = i;
            else if (r[i - 1] == h[j - 1]) v = d0[i - 1];
            else v = min(d0[i - 1], min(d1[i], d1[i - 1])) + 1;     // substitute, insert (d[i][j-1]), delete (d[i-1][j])
            d2[i] = v;
        }
        __syncthreads();
        int* t = d0; d0 = d1; d1 = d2; d2 = t;
    }
    if (lane == 0) dist[p] = d1[n];
}

}  // namespace decode
}  // namespace asr

using namespace asr;
using namespace asr::decode;

extern "C" int asr_argmax_rows(void* stream, const float* logits, int T, int B, int V, int32_t* ids) {
    if (!logits || !ids || T <= 0 || B <= 0 || V <= 0) return ASR_ERR_BAD_ARG;
    const long long rows = (long long)T * B;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, T, B, V, ids);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_ctc_collapse(void* stream, const int32_t* ids, const int32_t* lengths, int B, int T, int blank,
                                int merge_repeats, int32_t* out, int32_t* out_len) {
    if (!ids || !out || !out_len || B <= 0 || T <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(collapse_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ids, lengths, T, blank, merge_repeats, out, out_len);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_edit_distance(void* stream, const int32_t* ref, const int32_t* ref_len, int ref_pitch, const int32_t* hyp,
                                 const int32_t* hyp_len, int hyp_pitch, int pairs, int32_t* dist) {
    if (!ref || !ref_len || !hyp || !hyp_len || !dist || pairs <= 0 || ref_pitch <= 0 || hyp_pitch <= 0) return ASR_ERR_BAD_ARG;
    const size_t lds = (size_t)3 * (ref_pitch + 1) * sizeof(int);
    if (lds > 64 * 1024) return ASR_ERR_UNSUPPORTED;      // references up to ~5400 tokens
    hipLaunchKernelGGL(edit_distance_kernel, dim3(pairs), dim3(64), lds, (hipStream_t)stream, ref, ref_len, ref_pitch, hyp, hyp_len,
                       hyp_pitch, dist);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
