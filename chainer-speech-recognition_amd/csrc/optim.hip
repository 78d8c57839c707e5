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

// One thread decides what the update kernels of this step do, so that neither the host nor 500 000 threads have to:
//   ctl[0] = 1 when the step is dropped (non-finite gradient norm -- the reference's NaN check, run/ctc/cnn/train.py:193-197 --
//            or a raised abort word of a persistent GRU launch whose outputs are therefore garbage), else 0
//   ctl[1] = gradient factor = grad_scale * min(1, clip / (||g|| * |grad_scale|))        (GradientClipping on the reduced gradient)
//   ctl[2] = Adam's alpha_t = alpha * sqrt(1 - beta2^t) / (1 - beta1^t), t = number of APPLIED steps including this one
//            (a dropped step does not advance t: the reference `continue`s before optimizer.update)
//   ctl[3] = t
//   ctl[4] = the squared gradient norm (sum of the per-workgroup partial sums in a FIXED order: every data-parallel rank
//            derives bit-identical clipping factors from bit-identical reduced gradients -- float atomics would not)
//   ctl[5] = 1 when this step was dropped because a recurrence gave up (here or on a data-parallel peer), ctl[6] = how many steps
//            ever were (the caller zeroes ctl once; the host compares with the count it last reported)
__global__ __launch_bounds__(256) void sqnorm_partials_kernel(const float* __restrict__ g, long long n, float* __restrict__ partials) {
    __shared__ float scratch[32];
    float s = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = g[i];
        s += v * v;
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// every abort word of the process (any number of control buffers: one per (device, stream) that launched a recurrence) ORed into
// ONE word, and -- data parallelism -- a NaN planted in the reserved element of the local gradient buffer BEFORE the last slice is
// summed over the ranks, so that every rank's step_control sees a non-finite reserved element and drops the same step
__global__ void gather_abort_kernel(const long long* __restrict__ word_ptrs, int n, int* __restrict__ any_word, float* __restrict__ poison) {
    int any = 0;
    for (int i = 0; i < n; ++i) {
        const int* w = reinterpret_cast<const int*>(word_ptrs[i]);
        if (w && w[0] != 0) any = 1;
    }
    if (any_word) any_word[0] = any;
    if (any && poison) poison[0] = __int_as_float(0x7fc00000);
}

__global__ __launch_bounds__(256) void step_control_kernel(const float* __restrict__ partials, int npartials,
                                                           const int* __restrict__ abort0, const int* __restrict__ abort1,
                                                           float clip, float grad_scale, float alpha, float beta1, float beta2,
                                                           int* __restrict__ applied, float* __restrict__ ctl,
                                                           const float* __restrict__ reserved, float* __restrict__ ls) {
    __shared__ float scratch[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < npartials; i += blockDim.x) s += partials[i];
    const float sq = block_sum(s, scratch);
    if (threadIdx.x != 0) return;
    ctl[4] = sq;
    const float* sqnorm = &sq;
    bool drop = !isfinite(sqnorm[0]);
    bool gave_up = false;               // a recurrence of this rank (abort words) or of a peer (the reduced reserved element)
    if (abort0 && abort0[0] != 0) gave_up = true;
    if (abort1 && abort1[0] != 0) gave_up = true;
    if (reserved && !isfinite(reserved[0])) gave_up = true;
    if (gave_up) drop = true;
    // loss scaling (the IEEE-half build, configs[4]): the backward pass was seeded with ls[0], so the gradient factor carries 1 / ls[0];
    // a non-finite norm that no recurrence explains is an overflow of a half activation gradient: the step is dropped (as above) and the
    // scale halved; ls[2] applied steps in a row double it.  All of it here, on the device, identically on every data-parallel rank
    // (the decision is taken on the REDUCED gradient) -- ls = {scale, applied steps since the last change, growth interval, overflows}
    if (ls) {
        grad_scale /= ls[0];
        if (drop && !gave_up) {
            // static scale (growth interval 0, chainer's loss_scaling(scale=s)): the step is dropped and counted, the scale stays
            if (ls[2] > 0.f) ls[0] = fmaxf(ls[0] * 0.5f, 1.f / 16777216.f);   // below 1 when the unscaled gradients already leave the half range
            ls[1] = 0.f;
            ls[3] += 1.f;
        } else if (!drop) {
            ls[1] += 1.f;
            if (ls[2] > 0.f && ls[1] >= ls[2]) {
                ls[0] = fminf(ls[0] * 2.f, 16777216.f);
                ls[1] = 0.f;
            }
        }
    }
    float rate = grad_scale;
    if (!drop && clip > 0.f) {
        const float norm = sqrtf(sqnorm[0]) * fabsf(grad_scale);
        const float r = clip / norm;
        if (r < 1.f) rate *= r;
    }
    int t = applied[0];
    if (!drop) applied[0] = ++t;
    const double tt = t < 1 ? 1.0 : (double)t;
    const double fix1 = 1.0 - pow((double)beta1, tt), fix2 = 1.0 - pow((double)beta2, tt);
    ctl[0] = drop ? 1.f : 0.f;
    ctl[1] = rate;
    ctl[2] = (float)((double)alpha * sqrt(fix2) / fix1);
    ctl[3] = (float)t;
    ctl[5] = gave_up ? 1.f : 0.f;
    if (gave_up) ctl[6] += 1.f;         // steps dropped because a recurrence gave up, ever: a host that looks late still sees them
}

__global__ __launch_bounds__(256) void adam_ctl_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n, float beta1, float beta2, float eps,
                                                       float decay, const float* __restrict__ ctl) {
    if (ctl[0] != 0.f) return;
    const float rate = ctl[1], lr_t = ctl[2];
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

__global__ __launch_bounds__(256) void sgd_ctl_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
                                                      long long n, int kind, float lr, float mu, float decay,
                                                      const float* __restrict__ ctl) {
    if (ctl[0] != 0.f) return;
    const float rate = ctl[1];
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

extern "C" int asr_sqnorm_partials_count(long long n) { return grid_for(n); }

extern "C" int asr_gather_abort(void* stream, const long long* word_ptrs, int n, int* any_word, float* poison) {
    if (n < 0 || (n > 0 && !word_ptrs) || (!any_word && !poison)) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(gather_abort_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, word_ptrs, n, any_word, poison);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_step_control_scaled(void* stream, const float* g, long long n, float* partials, const int* abort0, const int* abort1,
                                       float clip_threshold, float grad_scale, float alpha, float beta1, float beta2,
                                       int* applied_steps, float* ctl, int reserved_index, float* loss_scale) {
    if (!g || n <= 0 || !partials || !applied_steps || !ctl || reserved_index >= n) return ASR_ERR_BAD_ARG;
    const int blocks = grid_for(n);
    hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, partials);
    hipLaunchKernelGGL(step_control_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, blocks, abort0,
                       abort1, clip_threshold, grad_scale, alpha, beta1, beta2, applied_steps, ctl,
                       reserved_index >= 0 ? g + reserved_index : (const float*)nullptr, loss_scale);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_step_control(void* stream, const float* g, long long n, float* partials, const int* abort0, const int* abort1,
                                float clip_threshold, float grad_scale, float alpha, float beta1, float beta2,
                                int* applied_steps, float* ctl, int reserved_index) {
    return asr_step_control_scaled(stream, g, n, partials, abort0, abort1, clip_threshold, grad_scale, alpha, beta1, beta2,
                                   applied_steps, ctl, reserved_index, nullptr);
}

extern "C" int asr_adam_ctl(void* stream, float* p, const float* g, float* m, float* v, long long n, float beta1, float beta2,
                            float eps, float weight_decay, const float* ctl) {
    if (!p || !g || !m || !v || !ctl || n <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(adam_ctl_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2, eps,
                       weight_decay, ctl);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_sgd_ctl(void* stream, float* p, const float* g, float* v, long long n, int kind, float lr, float momentum,
                           float weight_decay, const float* ctl) {
    if (!p || !g || !ctl || n <= 0 || kind < 0 || kind > 2 || (kind > 0 && !v)) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(sgd_ctl_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, v, n, kind, lr, momentum,
                       weight_decay, ctl);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
