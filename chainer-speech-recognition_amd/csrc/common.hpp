// Shared device/host helpers for libasr_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define ASR_OK 0
#define ASR_ERR_BAD_ARG (-1)
#define ASR_ERR_WORKSPACE (-2)
#define ASR_ERR_UNSUPPORTED (-3)
#define ASR_ERR_LAUNCH (-4)

#define ASR_LAUNCH_CHECK()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return ASR_ERR_LAUNCH; \
    } while (0)

namespace asr {

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide reductions through a small LDS scratch (>= 32 floats). All threads get the result.
// blockDim.x must be a multiple of 64. Ends with a barrier so `scratch` can be reused at once.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_max(v);
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    __syncthreads();
    return r;
}

// ---- the 16-bit activation format
// Every activation, MFMA operand and compute copy of a weight is a 16-bit value that the kernels touch only through the three
// helpers below (packing two of them into a dword is format agnostic).  The default build stores bfloat16 -- hence the helpers'
// names and the `bf16` in the names of the C ABI's arguments.  -DASR_ACT_F16 builds the SAME kernels with IEEE half storage and the
// f16 MFMA (v_mfma_f32_16x16x32_f16, the bf16 form's rate): libasr_hip_f16.so, BASELINE configs[4]'s "fp16 MFMA" (the reference allows
// float16 convolutions, asr/nn/convolution_2d.py:17-19).  Half has 11 significant bits instead of 8 and an exponent range of
// 6e-8 .. 65504: the train step then needs loss scaling (asr/optimizers.py).  The recurrent kernels (gru.hip, sru.hip) and the feature
// kernels manipulate bfloat16 bits directly in places and refuse to run in the half build (ASR_ERR_UNSUPPORTED): the convolutional
// recipes -- configs[4] -- do not use them.
#ifdef ASR_ACT_F16
#define ASR_ACT_IS_F16 1
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {           // round to nearest even, overflow to infinity, NaN stays NaN
    const _Float16 v = (_Float16)f;
    uint16_t h;
    __builtin_memcpy(&h, &v, 2);
    return h;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    const f16x2_t r = __builtin_convertvector(v, f16x2_t);
    return *reinterpret_cast<const uint32_t*>(&r);
}
// D = A (16 x 32) . B (32 x 16) + C on eight 16-bit operands per lane (passed as the 8 x short the kernels hold them in)
#define ASR_MFMA_16x16x32(a, b, c)                                                                                  \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) _Float16, (a)),   \
                                           __builtin_bit_cast(__attribute__((ext_vector_type(8))) _Float16, (b)), (c), 0, 0, 0)
#else
#define ASR_ACT_IS_F16 0
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even f32 -> bf16 through the hardware conversion (keeps NaN a NaN)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<uint16_t*>(&b);
}

// two floats -> packed bf16 pair (lo in bits 0..15), round to nearest even: ONE v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const uint32_t*>(&r);
}
#define ASR_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#endif

// ONE debug table for every what-if switch of the library: ASR_DEBUG="key=value,key=value" (integers).  Kernel selections that differ
// from the defaults exist for measurements and for the tests that pin the non-default kernels; nothing in a normal run sets them.
// Keys (default): nt_8ph (1: the 256 x 256 / eight-wave NT kernel of gemm8.hip wherever it qualifies), nt_wide (-1: convolutions only), nt_wide_force (0), nt_persist (1), nt_persist_min (256), nt_persist_grid (0),
// tn_8ph (1: the eight-wave TN kernel of gemm8.hip wherever it qualifies), tn_vec (1), tn256 (1), tn_group_target (1152), fwd_ring (1), gru_poll_delay (-1: per-shape table), gru_fwd_rows (8; 4: two half-slab workgroups per CU),
// conv_8ph (1: the eight-wave implicit convolution of gemm8.hip for more than 128 output columns), conv_direct (1), conv_mp_parts (0: by size), conv_mp_frame (1), fbank_fast (1; 0: the one-workgroup-per-frame feature kernels) -- and, read by the Python layer
// (asr/_lib.py: debug_flag), tn_group (1), gru_gates_f16 (1), side_join (0), side_priority (0), conv_mp (1).  DESIGN.md section 13.4 documents them.
static inline int debug_flag(const char* key, int dflt) {
    const char* e = getenv("ASR_DEBUG");
    if (!e) return dflt;
    const size_t n = strlen(key);
    for (const char* p = e; *p;) {
        while (*p == ',' || *p == ' ') ++p;
        if (strncmp(p, key, n) == 0 && p[n] == '=') return atoi(p + n + 1);
        while (*p && *p != ',') ++p;
    }
    return dflt;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace asr
