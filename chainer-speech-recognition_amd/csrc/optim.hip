// Fused optimiser step over ONE flat f32 parameter / gradient buffer (gfx950, HBM-bound: 16 B read + 12 B written per
// parameter).  Replaces the Chainer optimiser pipeline of the train step (run/ctc/cnn/train.py:142-147,200;
// asr/optimizers.py:43-52):   GradientClipping(threshold) -> WeightDecay(rate) -> Adam(alpha, beta1, beta2, eps)
//   rate = threshold / ||g||_2 over ALL parameters, applied when < 1;  g += decay * p;
//   m += (1-b1)(g-m); v += (1-b2)(g^2-v); p -= alpha*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)      (Chainer v2 Adam)
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace optim {

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long long n, float* __restrict__ out) {
    __shared__ float scratch[32];
    float s = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = g[i];
        s += v * v;
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float lr_t,
                                                   float beta1, float beta2, float eps, float decay, float clip,
                                                   float grad_scale, const float* __restrict__ sqnorm) {
    float rate = grad_scale;
    if (sqnorm) {
        // the reference's loop drops a step whose loss is NaN (run/ctc/cnn/train.py:193-197) after reading the loss on the
        // host; here the step is dropped on the device when the gradient norm is not finite (a NaN loss makes it so), so
        // the train loop never has to synchronise for it
        if (!isfinite(sqnorm[0])) return;
        if (clip > 0.f) {
            const float norm = sqrtf(sqnorm[0]) * fabsf(grad_scale);
            const float r = clip / norm;
            if (r < 1.f) rate *= r;
        }
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] * rate + decay * pi;
        float mi = m[i], vi = v[i];
        mi += (1.f - beta1) * (gi - mi);
        vi += (1.f - beta2) * (gi * gi - vi);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - lr_t * mi / (sqrtf(vi) + eps);
    }
}

// kind 0: SGD  p -= lr g;  1: MomentumSGD  v = mu v - lr g, p += v;  2: NesterovAG  v = mu v - lr g, p += mu^2 v - (1+mu) lr g
// (chainer.optimizers.SGD / MomentumSGD / NesterovAG update rules, selected by asr/optimizers.py:43-52)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
                                                  long long n, int kind, float lr, float mu, float decay, float clip,
                                                  float grad_scale, const float* __restrict__ sqnorm) {
    float rate = grad_scale;
    if (sqnorm) {
        // the reference's loop drops a step whose loss is NaN (run/ctc/cnn/train.py:193-197) after reading the loss on the
        // host; here the step is dropped on the device when the gradient norm is not finite (a NaN loss makes it so), so
        // the train loop never has to synchronise for it
        if (!isfinite(sqnorm[0])) return;
        if (clip > 0.f) {
            const float norm = sqrtf(sqnorm[0]) * fabsf(grad_scale);
            const float r = clip / norm;
            if (r < 1.f) rate *= r;
        }
    }
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] * rate + decay * pi;
        if (kind == 0) {
            p[i] = pi - lr * gi;
        } else {
            const float vi = mu * v[i] - lr * gi;
            v[i] = vi;
            p[i] = kind == 1 ? pi + vi : pi + mu * mu * vi - (1.f + mu) * lr * gi;
        }
    }
}

__global__ void fill_kernel(float* __restrict__ p, long long n, float value) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = value;
}

}  // namespace optim
}  // namespace asr

using namespace asr;
using namespace asr::optim;

static inline int grid_for(long long n) {
    long long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    return g < 1 ? 1 : (int)g;
}

extern "C" int asr_fill_f32(void* stream, float* p, long long n, float value) {
    if (!p || n <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, n, value);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sqnorm_acc(void* stream, const float* g, long long n, float* out) {
    if (!g || !out || n <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, n, out);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_clip_decay_adam(void* stream, float* p, const float* g, float* m, float* v, long long n, float alpha,
                                   float beta1, float beta2, float eps, float weight_decay, float clip_threshold,
                                   float grad_scale, const float* sqnorm, int step) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return ASR_ERR_BAD_ARG;
    const double fix1 = 1.0 - pow((double)beta1, (double)step), fix2 = 1.0 - pow((double)beta2, (double)step);
    const float lr_t = (float)(alpha * sqrt(fix2) / fix1);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr_t, beta1,
                       beta2, eps, weight_decay, clip_threshold, grad_scale, sqnorm);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_clip_decay_sgd(void* stream, float* p, const float* g, float* v, long long n, int kind, float lr,
                                  float momentum, float weight_decay, float clip_threshold, float grad_scale,
                                  const float* sqnorm) {
    if (!p || !g || n <= 0 || kind < 0 || kind > 2 || (kind > 0 && !v)) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, v, n, kind, lr, momentum,
                       weight_decay, clip_threshold, grad_scale, sqnorm);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
