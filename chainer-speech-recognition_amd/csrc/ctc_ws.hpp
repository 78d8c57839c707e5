// Workspace of the CTC / Gram-CTC kernels (csrc/ctc.hip), shared with the fused layer-norm + CTC backward (csrc/ctc_ln.hip).
#pragma once
#include "common.hpp"

namespace asr {
namespace ctc {

struct Workspace {
    int* path_label;   // (B, Sp)  -1 = dead / outside the path
    int* path_mask;    // (B, Sp)  bit j set: edge from s - k_j into s
    int* path_len;     // (B)
    float* lse;        // (T, B)
    float* lp;         // (B, T, Sp)
    double* alpha;     // (B, T, Sp)
    double* beta;      // (B, T, Sp)
    double* total;     // (B)
    size_t bytes;
};

static inline int path_pad(int Lmax, int gram) {
    const int S = (gram ? 3 : 2) * Lmax + 1;
    return (int)align_up((size_t)S, 64);
}

static Workspace carve(void* base, int T, int B, int Lmax, int gram) {
    Workspace w;
    const size_t Sp = (size_t)path_pad(Lmax, gram);
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t n) { char* r = p ? p + off : nullptr; off += align_up(n, 256); return r; };
    w.path_label = (int*)take(sizeof(int) * B * Sp);
    w.path_mask = (int*)take(sizeof(int) * B * Sp);
    w.path_len = (int*)take(sizeof(int) * B);
    w.lse = (float*)take(sizeof(float) * (size_t)T * B);
    w.lp = (float*)take(sizeof(float) * (size_t)B * T * Sp);
    w.alpha = (double*)take(sizeof(double) * (size_t)B * T * Sp);
    w.beta = (double*)take(sizeof(double) * (size_t)B * T * Sp);
    w.total = (double*)take(sizeof(double) * B);
    w.bytes = off;
    return w;
}

}  // namespace ctc
}  // namespace asr
