/* libasr_hip -- C ABI of the MI355X (gfx950) CTC acoustic-model training path.
 *
 * Flat C boundary: raw device pointers, sizes, a hipStream_t passed as void*.  No C++ or torch types.
 * The caller owns every buffer (outputs and workspace are allocated before the call, as the
 * reference's Function objects do: asr/nn/sru.py:348-349,388-393); functions enqueue kernels on the
 * given stream and never synchronise, allocate or free.  Return value: 0 = ok, < 0 = error
 * (-1 bad argument, -2 workspace too small, -3 unsupported shape, -4 launch failure); no exception
 * crosses the boundary.  All tensors are dense row-major unless a pitch argument says otherwise.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference root).
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int asr_version(void);

/* ---------------------------------------------------------------------------------------- CTC family
 * Replaces chainer.functions.connectionist_temporal_classification (call sites run/ctc/cnn/train.py:162,191,
 * run/ctc/sru/train.py:161,191) when label_bigram == NULL, and asr/loss/gram_ctc.py:219-315 (GramCTC /
 * gram_ctc) otherwise.
 *   xs             (T, B, V) f32 pre-softmax activations (the reference passes T arrays (B, V))
 *   label_unigram  (B, Lmax) int32, padded; label_bigram (B, Lmax) int32 with -1 = absent, or NULL
 *   x_len, l_len   (B) int32 or NULL (= T / Lmax)       blank: blank symbol id
 *   loss_per_utt   (B) f32  = -log p(labels | x)         loss_mean: scalar f32 or NULL
 *   workspace      asr_ctc_workspace_bytes(...) bytes; it carries alpha/beta from forward to backward
 * backward: grad (T, B, V) f32 = (softmax - occupancy) * scale * gy, zero rows for t >= x_len
 *   gy: device pointer to one f32 (gy_per_utt = 0) or (B) f32 (gy_per_utt = 1), or NULL (= 1)
 *   scale = 1/B reproduces reduce='mean' (asr/loss/gram_ctc.py:291-292), scale = 1 reduce='no'.
 */
size_t asr_ctc_workspace_bytes(int T, int B, int V, int Lmax, int gram);
int asr_ctc_forward(void* stream, const float* xs, const int32_t* label_unigram, const int32_t* label_bigram,
                    const int32_t* x_len, const int32_t* l_len, int T, int B, int V, int Lmax, int blank,
                    float* loss_per_utt, float* loss_mean, void* workspace, size_t workspace_bytes);
int asr_ctc_backward(void* stream, const float* xs, const int32_t* x_len, int T, int B, int V, int Lmax, int gram,
                     const float* gy, int gy_per_utt, float scale, float* grad, const void* workspace,
                     size_t workspace_bytes);
/* forward + backward with gy = 1 in one call (the reference derives both from the same alpha+beta table,
 * asr/loss/gram_ctc.py:276,288-290) */
int asr_ctc_loss_grad(void* stream, const float* xs, const int32_t* label_unigram, const int32_t* label_bigram,
                      const int32_t* x_len, const int32_t* l_len, int T, int B, int V, int Lmax, int blank, float scale,
                      float* loss_per_utt, float* loss_mean, float* grad, void* workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif
