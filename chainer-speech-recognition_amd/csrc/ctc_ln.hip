// Backward of [LayerNormalization over the vocabulary of every frame] -> [CTC / Gram-CTC loss(es)] in ONE sweep  (gfx950).
//
// The train step of run/ctc/cnn/train.py:190-200 ends  y = LayerNormalization(logits);  loss = CTC(y, t)  [+ gram_ctc(y, ...),
// run/gram_ctc/cnn/train.py:163-167].  Its backward was three passes over (T*B, V) float32 tensors: ctc::grad reads y and
// writes dL/dy (768 MB at T=1000, B=32, V=3000), ln::bwd_rows_f32 reads x and dL/dy again and writes dx (960 MB).  Both own
// the same (t, b) row, and dL/dy = (softmax(y) - occupancy) * scale is a function of the row alone -- so it is never written:
// a workgroup keeps the pre-normalisation row x in registers (a thread owns fixed columns, as in ln::bwd_rows_f32),
// recomputes y = xhat * gamma + beta, scatters the occupancy of the row's lattice nodes
//        occ[v] = sum_{s : path[s] = v} exp(alpha[t][s] + beta[t][s] - total)             (asr/loss/gram_ctc.py:180-217, :289)
// into an LDS row (parity-buffered; a node clears its own entry once the row is consumed), forms
//        g = sum over losses of (exp(y - lse) - occ) * scale * gy       zero for t >= input_length   (asr/loss/gram_ctc.py:290-296)
// and applies the layer-norm backward (closed form of asr/nn/layernorm.py:50-61) with the dgamma / dbeta column sums carried in
// registers.  Traffic: x in (12 KB per row), alpha + beta of the path (4 KB), dx out in bf16 (6 KB): 0.7 GB instead of 1.7 GB.
#include "common.hpp"
#include "ctc_ws.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace ctcln {

struct Loss {               // one CTC-family loss on the normalised row (up to two: the joint Gram-CTC + CTC step)
    const int* path_label;
    const int* path_len;
    const float* lse;
    const double* alpha;
    const double* beta;
    const double* total;
    const int* x_len;
    const float* gy;        // device scalar, (B) values, or NULL (= 1)
    int gy_per_utt;
    float scale;
    int Sp;
};

// XS: also carry the plain column sums of dx (third plane of `partial`): the bias gradient of the projection in front
template <typename DT, int NV, int NL, bool XS>
__global__ __launch_bounds__(256) void bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ mean_in,
                                                  const float* __restrict__ rstd_in, DT* __restrict__ dx, float* __restrict__ partial,
                                                  int T, int B, int D, Loss l0, Loss l1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* occ = reinterpret_cast<float*>(smem);                 // [2 (row parity)][NL][D]
    __shared__ float red[2][2][4];
    const int n4 = D >> 2, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long rows = (long long)T * B;
    float4 gm[NV], bt[NV], ag[NV], ab[NV], v[NV], vn[NV], ax[XS ? NV : 1];
#pragma unroll
    for (int k = 0; k < (XS ? NV : 1); ++k) ax[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + 256 * k;
        gm[k] = i < n4 ? *reinterpret_cast<const float4*>(gamma + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[k] = i < n4 ? *reinterpret_cast<const float4*>(beta + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        ag[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        vn[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < 2 * NL * D; i += 256) occ[i] = 0.f;
    long long row = blockIdx.x;
    // alpha + beta of this thread's (at most two) lattice nodes per loss: fetched one row ahead, like the x row -- a load
    // issued and consumed inside one row would put an HBM round trip on every row of the sweep
    double pab[NL][2];
    // per-utterance constants (labels of this thread's nodes, path length, input length, log-likelihood, upstream gradient):
    // a workgroup's rows all belong to one utterance when its stride is a multiple of B -- reloaded only when b changes
    int ub = -1, u_lab[NL][2], u_S[NL], u_xl[NL], u_blank[NL];
    double u_tot[NL];
    float u_sc[NL];
    auto load_utt = [&](int b_) {
        ub = b_;
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const Loss& L = q == 0 ? l0 : l1;
            const int* pl = L.path_label + (size_t)b_ * L.Sp;
            u_S[q] = L.path_len[b_];
            u_xl[q] = L.x_len ? min(L.x_len[b_], T) : T;
            u_tot[q] = L.total[b_];
            u_sc[q] = L.scale * (L.gy ? (L.gy_per_utt ? L.gy[b_] : L.gy[0]) : 1.0f);
            u_blank[q] = pl[0];                                    // node 0 of every path is the blank
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int s = tid + 256 * j;
                u_lab[q][j] = s < L.Sp ? pl[s] : -1;
            }
        }
    };
    auto fetch_ab = [&](long long r) {
        const int t_ = (int)(r / B), b_ = (int)(r - (long long)t_ * B);
        if (b_ != ub) load_utt(b_);
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const Loss& L = q == 0 ? l0 : l1;
            const double* al = L.alpha + ((size_t)b_ * T + t_) * L.Sp;
            const double* be = L.beta + ((size_t)b_ * T + t_) * L.Sp;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int s = tid + 256 * j;
                pab[q][j] = (s < u_S[q] && t_ < u_xl[q]) ? al[s] + be[s] : -INFINITY;
            }
        }
    };
    if (row < rows) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k;
            if (i < n4) vn[k] = reinterpret_cast<const float4*>(x + row * D)[i];
        }
        fetch_ab(row);
    }
    __syncthreads();
    int par = 0;
    const float invD = 1.0f / (float)D;
    for (; row < rows; row += gridDim.x) {
        const int t = (int)(row / B);
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = vn[k];
        const float mean = mean_in[row], rstd = rstd_in[row] < INFINITY ? rstd_in[row] : 0.f;      // (zero-variance rows: ln::usable_rstd)
        // occupancy of this row's lattice nodes -> occ[par][loss][label]; every scattering thread remembers its slot
        float* oc = occ + (size_t)par * NL * D;
        // (the utterance constants in registers are those of THIS row: fetch_ab(row) loaded them; the prefetch for the next
        // row below may replace them, after their last use here)
        float lse[NL], sc[NL];
        int my_label[NL][2], blank_slot[NL];
        bool live[NL];
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const Loss& L = q == 0 ? l0 : l1;
            live[q] = t < u_xl[q];
            my_label[q][0] = my_label[q][1] = blank_slot[q] = -1;
            lse[q] = 0.f;
            sc[q] = u_sc[q];
            if (live[q]) {
                lse[q] = L.lse[row];
                const double tot = u_tot[q];
                const int blank = u_blank[q];
#pragma unroll
                for (int j = 0; j < 2; ++j) {                  // Sp <= 512 nodes: two per thread
                    const int lab = u_lab[q][j];
                    const double e = pab[q][j] - tot;          // (-inf beyond the path)
                    const float p = (tot != -INFINITY && lab >= 0 && e > -80.0) ? expf((float)e) : 0.f;     // no path at all: softmax only
                    // every other node is the blank: one LDS atomic per wave for them instead of ~120 on one address
                    const float pb = wave_sum(lab == blank ? p : 0.f);
                    if (lane == 0 && pb != 0.f) { atomicAdd(&oc[q * D + blank], pb); blank_slot[q] = blank; }
                    if (lab != blank && p != 0.f) { atomicAdd(&oc[q * D + lab], p); my_label[q][j] = lab; }
                }
            }
        }
        if (row + gridDim.x < rows) fetch_ab(row + gridDim.x);
        const long long nxt = row + gridDim.x;
        if (nxt < rows) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = tid + 256 * k;
                if (i < n4) vn[k] = reinterpret_cast<const float4*>(x + nxt * D)[i];
            }
        }
        __syncthreads();                                        // (A) the row's occupancy is complete
        float4 gy[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {      // lanes beyond the row hold zeros (gamma = 0 -> g = 0 contribution)
            const int i = tid + 256 * k;
            v[k] = make_float4((v[k].x - mean) * rstd, (v[k].y - mean) * rstd, (v[k].z - mean) * rstd, (v[k].w - mean) * rstd);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n4) {
                const float y0 = v[k].x * gm[k].x + bt[k].x, y1 = v[k].y * gm[k].y + bt[k].y;
                const float y2 = v[k].z * gm[k].z + bt[k].z, y3 = v[k].w * gm[k].w + bt[k].w;
#pragma unroll
                for (int q = 0; q < NL; ++q) {
                    if (live[q]) {
                        const float4 o = *reinterpret_cast<const float4*>(oc + q * D + i * 4);
                        g.x += (__expf(y0 - lse[q]) - o.x) * sc[q];
                        g.y += (__expf(y1 - lse[q]) - o.y) * sc[q];
                        g.z += (__expf(y2 - lse[q]) - o.z) * sc[q];
                        g.w += (__expf(y3 - lse[q]) - o.w) * sc[q];
                    }
                }
            }
            gy[k] = g;
            const float g0 = g.x * gm[k].x, g1 = g.y * gm[k].y, g2 = g.z * gm[k].z, g3 = g.w * gm[k].w;
            s1 += (g0 + g1) + (g2 + g3);
            s2 += (g0 * v[k].x + g1 * v[k].y) + (g2 * v[k].z + g3 * v[k].w);
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[par][0][wid] = s1; red[par][1][wid] = s2; }
        __syncthreads();                                        // (B) everybody has read the occupancy row
        const float m1 = ((red[par][0][0] + red[par][0][1]) + (red[par][0][2] + red[par][0][3])) * invD;
        const float m2 = ((red[par][1][0] + red[par][1][1]) + (red[par][1][2] + red[par][1][3])) * invD;
        // this parity's occupancy row is used again two rows from now (barriers A and B of the next row lie in between)
#pragma unroll
        for (int q = 0; q < NL; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (my_label[q][j] >= 0) oc[q * D + my_label[q][j]] = 0.f;
#pragma unroll
        for (int q = 0; q < NL; ++q)
            if (blank_slot[q] >= 0) oc[q * D + blank_slot[q]] = 0.f;
        par ^= 1;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k;
            if (i < n4) {
                const float d0 = (gy[k].x * gm[k].x - m1 - v[k].x * m2) * rstd, d1 = (gy[k].y * gm[k].y - m1 - v[k].y * m2) * rstd;
                const float d2 = (gy[k].z * gm[k].z - m1 - v[k].z * m2) * rstd, d3 = (gy[k].w * gm[k].w - m1 - v[k].w * m2) * rstd;
                if (dx) {
                    if (sizeof(DT) == 2) {
                        uint2 o;
                        o.x = (unsigned)f32_to_bf16(d0) | ((unsigned)f32_to_bf16(d1) << 16);
                        o.y = (unsigned)f32_to_bf16(d2) | ((unsigned)f32_to_bf16(d3) << 16);
                        reinterpret_cast<uint2*>(dx + row * D)[i] = o;
                    } else {
                        reinterpret_cast<float4*>(dx + row * D)[i] = make_float4(d0, d1, d2, d3);
                    }
                }
                ag[k].x += gy[k].x * v[k].x; ag[k].y += gy[k].y * v[k].y; ag[k].z += gy[k].z * v[k].z; ag[k].w += gy[k].w * v[k].w;
                ab[k].x += gy[k].x; ab[k].y += gy[k].y; ab[k].z += gy[k].z; ab[k].w += gy[k].w;
                if (XS) { ax[k].x += d0; ax[k].y += d1; ax[k].z += d2; ax[k].w += d3; }
            }
        }
    }
    if (partial) {
        constexpr int PL = XS ? 3 : 2;
        float* pg = partial + (size_t)blockIdx.x * PL * D;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = tid + 256 * k;
            if (i < n4) {
                reinterpret_cast<float4*>(pg)[i] = ag[k];
                reinterpret_cast<float4*>(pg + D)[i] = ab[k];
                if (XS) reinterpret_cast<float4*>(pg + 2 * D)[i] = ax[k];
            }
        }
    }
}

}  // namespace ctcln
}  // namespace asr

using namespace asr;

// rows are (t, b) pairs of a (T, B, V) tensor; a workgroup strides by its grid size, which is a multiple of B so that all its
// rows belong to one utterance (not needed for correctness: the occupancy slots are cleared per row)
// The grid is sized to what is resident at once: the kernel keeps V / 1024 float4 of seven row-shaped arrays in registers
// (V = 3000: 142 VGPRs -> three workgroups per CU; a fourth per CU would run as a second, mostly empty round: measured 0.255 ->
// 0.20 ms at T=1000, B=32; ctc::grad + ln::bwd_rows_f32 take 0.35 ms).
static int ctcln_grid(long long rows, int B, int V) {
    const int nv = (V / 4 + 255) / 256;
    const int per_cu = nv <= 2 ? 4 : (nv == 3 ? 3 : 2);
    long long g = (rows + 15) / 16;
    if (g > 256 * per_cu) g = 256 * per_cu;
    if (g >= B) g -= g % B;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" long long asr_layernorm_ctc_bwd_ws_bytes(int T, int B, int V) {
    if (T <= 0 || B <= 0 || V <= 0 || (V & 3) || V > 4096) return 0;
    return (long long)ctcln_grid((long long)T * B, B, V) * 3 * V * (long long)sizeof(float);
}

extern "C" int asr_layernorm_ctc_bwd(void* stream, const float* x, const float* gamma, const float* beta, const float* mean,
                                     const float* rstd, void* dx, int dx_bf16, float* dgamma, float* dbeta, int T, int B, int V,
                                     void* ws, long long ws_bytes, int nloss, const void* ctc_ws0, int Lmax0, int gram0,
                                     const int32_t* x_len0, const float* gy0, int gy_per_utt0, float scale0, const void* ctc_ws1,
                                     int Lmax1, int gram1, const int32_t* x_len1, const float* gy1, int gy_per_utt1, float scale1,
                                     float* dxsum) {
    if (!x || !gamma || !beta || !mean || !rstd || T <= 0 || B <= 0 || V <= 0 || nloss < 1 || nloss > 2 || !ctc_ws0 || (nloss == 2 && !ctc_ws1))
        return ASR_ERR_BAD_ARG;
    if ((V & 3) || V > 4096 || ((((uintptr_t)x) | ((uintptr_t)dx)) & 15)) return ASR_ERR_UNSUPPORTED;
    const bool params = dgamma && dbeta;
    if (dxsum && !params) return ASR_ERR_BAD_ARG;
    const long long need = asr_layernorm_ctc_bwd_ws_bytes(T, B, V);
    if (params && (!ws || ws_bytes < need)) return ASR_ERR_BAD_ARG;
    if (!dx && !params) return ASR_OK;
    auto make = [&](const void* cw, int Lmax, int gram, const int32_t* xl, const float* gy, int per, float scale) {
        ctc::Workspace w = ctc::carve(const_cast<void*>(cw), T, B, Lmax, gram);
        ctcln::Loss L;
        L.path_label = w.path_label; L.path_len = w.path_len; L.lse = w.lse; L.alpha = w.alpha; L.beta = w.beta; L.total = w.total;
        L.x_len = xl; L.gy = gy; L.gy_per_utt = per; L.scale = scale; L.Sp = ctc::path_pad(Lmax, gram);
        return L;
    };
    const ctcln::Loss l0 = make(ctc_ws0, Lmax0, gram0, x_len0, gy0, gy_per_utt0, scale0);
    const ctcln::Loss l1 = nloss == 2 ? make(ctc_ws1, Lmax1, gram1, x_len1, gy1, gy_per_utt1, scale1) : l0;
    if (l0.Sp > 512 || l1.Sp > 512) return ASR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int G = ctcln_grid((long long)T * B, B, V);
    float* partial = params ? (float*)ws : nullptr;
    const int nv = (V / 4 + 255) / 256;
    const size_t lds = sizeof(float) * 2 * (size_t)nloss * V;
#define ASR_CL_(DT, NV, NL, XS)                                                                                           \
    do {                                                                                                                  \
        if (lds > 48 * 1024)                                                                                              \
            (void)hipFuncSetAttribute((const void*)ctcln::bwd_kernel<DT, NV, NL, XS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((ctcln::bwd_kernel<DT, NV, NL, XS>), dim3(G), dim3(256), lds, s, x, gamma, beta, mean, rstd, (DT*)dx, partial, \
                           T, B, V, l0, l1);                                                                              \
    } while (0)
#define ASR_CL(DT, NV, NL)                                                                                                \
    do {                                                                                                                  \
        if (dxsum) ASR_CL_(DT, NV, NL, true); else ASR_CL_(DT, NV, NL, false);                                            \
    } while (0)
#define ASR_CLN(NV)                                                                                                       \
    do {                                                                                                                  \
        if (dx_bf16) { if (nloss == 2) ASR_CL(uint16_t, NV, 2); else ASR_CL(uint16_t, NV, 1); }                           \
        else { if (nloss == 2) ASR_CL(float, NV, 2); else ASR_CL(float, NV, 1); }                                         \
    } while (0)
    switch (nv) {
        case 1: ASR_CLN(1); break;
        case 2: ASR_CLN(2); break;
        case 3: ASR_CLN(3); break;
        default: ASR_CLN(4); break;
    }
#undef ASR_CLN
#undef ASR_CL
#undef ASR_CL_
    ASR_LAUNCH_CHECK();
    if (params) {
        const int rc = asr_layernorm_fold_partials(stream, partial, G, V, V, dgamma, dbeta, dxsum);
        if (rc != ASR_OK) return rc;
    }
    return ASR_OK;
}
