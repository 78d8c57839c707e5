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
/* the 16-bit format of every activation, MFMA operand and weight compute copy of THIS build: 0 = bfloat16 (libasr_hip.so, the default),
 * 1 = IEEE half (libasr_hip_f16.so: the same kernels built with -DASR_ACT_F16 for BASELINE configs[4]'s "fp16 MFMA"; the reference
 * allows float16 convolutions, asr/nn/convolution_2d.py:17-19).  Wherever this header says `bf16` it means that format.  The half build
 * refuses the recurrent entry points (asr_gru_*, asr_sru_*: ASR_ERR_UNSUPPORTED); a train step in half needs loss scaling on the caller's
 * side (the gradient factor of asr_step_control). */
int asr_act_dtype(void);
/* a one-wave kernel that idles for `microseconds` (<= 100000) on `stream`: a timed gap (experiments) */
int asr_stream_delay(void* stream, int microseconds);
/* diagnostic: `blocks` workgroups holding `lds_bytes` of LDS each idle for `microseconds` (<= 200000) on `stream` -- makes CUs
 * temporarily unavailable to launches on other streams (tests of the persistent GRU kernels under partial residency) */
int asr_occupy_cus(void* stream, int microseconds, int lds_bytes, int blocks);
/* diagnostic: the stand-in for a resident collective -- `blocks` workgroups (256 threads, `lds_bytes` of LDS each) that stream
 * buf[0, bytes / 2) into buf[bytes / 2, bytes) (read, add, write; buf 16-B aligned, float32) until `microseconds` (<= 200000) have
 * passed.  The reference has no multi-GPU path ("NCCL call sites: zero"); this measures, on one GPU, what an
 * RCCL ring kernel resident beside a persistent recurrence would cost it. */
int asr_stream_traffic(void* stream, int microseconds, int lds_bytes, int blocks, void* buf, long long bytes);

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
/* The same with the log-sum-exp of every (t, b) row of xs handed in (row_lse[t * B + b], e.g. from asr_layernorm_fwd_lse; NULL: as
 * asr_ctc_forward): the rows pass then only gathers the label entries instead of reading the logits twice. */
int asr_ctc_forward_lse(void* stream, const float* xs, const int32_t* label_unigram, const int32_t* label_bigram,
                        const int32_t* x_len, const int32_t* l_len, int T, int B, int V, int Lmax, int blank, float* loss_per_utt,
                        float* loss_mean, void* workspace, size_t workspace_bytes, const float* row_lse);
int asr_ctc_backward(void* stream, const float* xs, const int32_t* x_len, int T, int B, int V, int Lmax, int gram,
                     const float* gy, int gy_per_utt, float scale, float* grad, const void* workspace,
                     size_t workspace_bytes);
/* forward + backward with gy = 1 in one call (the reference derives both from the same alpha+beta table,
 * asr/loss/gram_ctc.py:276,288-290) */
int asr_ctc_loss_grad(void* stream, const float* xs, const int32_t* label_unigram, const int32_t* label_bigram,
                      const int32_t* x_len, const int32_t* l_len, int T, int B, int V, int Lmax, int blank, float scale,
                      float* loss_per_utt, float* loss_mean, float* grad, void* workspace, size_t workspace_bytes);

/* ---------------------------------------------------------------------------------------- log-mel features
 * Replace fft.get_specgram / compute_logmel / compute_deltas (asr/fft.py:52-66, 6-19, 90-99), the per-utterance loop of
 * Processor.extract_batch_features (asr/data/processing.py:67-111) and the Loader's normalisation
 * (asr/data/loaders/base.py:22-24), batched over utterances.
 *   asr_specgram  signals (B, sig_pitch) int16 or f32, lengths (B); frame f of utterance b covers samples
 *                 [f*frame_step, f*frame_step+frame_len) zero padded, pre-emphasised, times window (frame_len);
 *                 nframes (B) frames are produced per utterance.  Writes the power spectrum
 *                 (B, Fmax, nfft/2+1) and/or log(pspec . fbank^T) (B, Fmax, nfilt); nfft a power of two <= 1024.
 *                 nfft = 512 (the reference's frame, asr/data/processing.py:54) with frame_step % 4 == 0, sig_pitch % 4 == 0 and an
 *                 aligned buffer: one frame per wave, 256-point complex radix-4 transform in registers; else one workgroup per frame.
 *   asr_logmel    log-mel of a caller-supplied power spectrum (F, nbins) with fbank (nfilt, nbins)
 *   asr_deltas    (B, Fmax, nfilt) log-mel -> the minibatch x (B, 3, nfilt, Tmax) f32: static / delta / delta-delta,
 *                 T_b = nframes[b] - 2 frames, zero beyond; mean/std (3, nfilt) optional
 */
int asr_specgram(void* stream, const void* signals, int sig_is_f32, const int32_t* lengths, long long sig_pitch, int B,
                 int frame_len, int frame_step, int nfft, float preemph, const float* window, const int32_t* nframes,
                 int Fmax, float* pspec_out, const float* fbank, int nfilt, float* logmel_out);
/*   asr_mel_bands       the sparse form of a mel matrix fbank (nfilt, nbins) (asr/fft.py:68-82 builds triangles: 454 of 10280 entries are
 *                       non-zero at 40 x 257): per filter the first non-zero bin, the band length padded to a multiple of 8, the offset of its
 *                       taps, then the taps -- written ONCE per matrix into a caller-owned table of asr_mel_bands_bytes() bytes (16-byte aligned)
 *   asr_specgram_bands  asr_specgram with that table of ITS fbank (nbins = nfft / 2 + 1): the mel stage touches the bands only; a null table,
 *                       or one whose matrix did not fit it (more than 64 filters / 1024 padded taps), falls back to the dense rows */
size_t asr_mel_bands_bytes(void);
int asr_mel_bands(void* stream, const float* fbank, int nfilt, int nbins, void* table, size_t table_bytes);
int asr_specgram_bands(void* stream, const void* signals, int sig_is_f32, const int32_t* lengths, long long sig_pitch, int B,
                       int frame_len, int frame_step, int nfft, float preemph, const float* window, const int32_t* nframes,
                       int Fmax, float* pspec_out, const float* fbank, int nfilt, float* logmel_out, const void* bands);
int asr_logmel(void* stream, const float* pspec, const float* fbank, long long F, int nbins, int nfilt, float* out);
int asr_deltas(void* stream, const float* logmel, const int32_t* nframes, int B, int Fmax, int nfilt, int Tmax,
               const float* mean, const float* stdv, float* out);

/* optional steps of Processor.extract_batch_features and the Loader's running statistics
 *   asr_cmn_pspec            cepstral mean normalisation in the log-power domain (asr/data/processing.py:86-89), in place
 *                            on the (B, Fmax, nbins) power spectrum, frames < nframes[b]
 *   asr_add_white_noise      signal[b] += trunc(gain[b] * n), n ~ N(0, 1) (asr/data/processing.py:74-78; counter-based
 *                            generator: the distribution matches, NumPy's stream does not), f32 signals in place
 *   asr_augment_specgram     speed / vocal-tract perturbation by nearest-index resampling (asr/fft.py:21-50):
 *                            out[b][f][k] = in[b][int(f speed_b)][min(int(k ratio_b), nbins-1)], f < nframes_out[b]
 *                            (= int(nframes_in[b] / speed_b), computed by the caller), zero beyond
 *   asr_running_stats_update asr/data/loaders/base.py:64-80 for every utterance of x (B, CM, T) f32 in turn (frames
 *                            < lengths[b]); mean / nvar (CM) float64 state, total_before = frames seen so far; also
 *                            writes mean32 and the unbiased std32 = sqrt(nvar / (total - 1)) (:39-41)
 *   asr_normalize_bcmt       x <- (x - mean[cm]) / std[cm] over the whole padded array (:24), in place
 */
int asr_cmn_pspec(void* stream, float* pspec, const int32_t* nframes, int B, int Fmax, int nbins);
int asr_add_white_noise(void* stream, float* signals, const int32_t* lengths, long long pitch, int B, const float* gain,
                        unsigned long long seed);
int asr_augment_specgram(void* stream, const float* pspec_in, const int32_t* nframes_out, const double* speed,
                         const double* ratio, int B, int Fmax_in, int Fmax_out, int nbins, float* pspec_out);
int asr_running_stats_update(void* stream, const float* x, const int32_t* lengths, int B, int CM, int T,
                             long long total_before, double* mean, double* nvar, float* mean32, float* std32);
int asr_normalize_bcmt(void* stream, float* x, const float* mean, const float* stdv, int B, int CM, int T);

/* ---------------------------------------------------------------------------------------- greedy decode + CER
 * The evaluation loop of run/ctc/cnn/dev.py:100-108 and asr/error.py:7-68 for a whole minibatch.
 *   asr_argmax_rows    ids[b][t] = argmax_v logits[t][b][v] (first maximum, as np.argmax), logits (T, B, V) f32
 *   asr_ctc_collapse   per utterance: drop blanks and repeats (asr/error.py:38-47) of ids (B, T), frames < lengths[b]
 *                      (NULL: all T, as the reference), compacted into out (B, T) padded with blank, out_len (B);
 *                      merge_repeats = 0 drops blanks only (the label side, asr/error.py:33-37)
 *   asr_edit_distance  Levenshtein distance of `pairs` (reference, hypothesis) id rows (asr/error.py:7-24, without the
 *                      division by len(r)); dist = len(h) when len(r) == 0.  int32 arithmetic (the reference's uint8
 *                      table is defined up to 255 tokens)
 */
int asr_argmax_rows(void* stream, const float* logits, int T, int B, int V, int32_t* ids);
int asr_ctc_collapse(void* stream, const int32_t* ids, const int32_t* lengths, int B, int T, int blank, int merge_repeats,
                     int32_t* out, int32_t* out_len);
int asr_edit_distance(void* stream, const int32_t* ref, const int32_t* ref_len, int ref_pitch, const int32_t* hyp,
                      const int32_t* hyp_len, int hyp_pitch, int pairs, int32_t* dist);

/* ---------------------------------------------------------------------------------------- dense projections
 * bf16 MFMA GEMMs (f32 accumulate).  Replace the BLAS/cuDNN calls behind chainer.links.Linear, the 1x1
 * ConvolutionND of asr/nn/convolution_1d.py:7-38, the SRU projection asr/nn/sru.py:340-341,421-429 and -- through
 * asr_im2col / asr_col2im -- chainer.links.Convolution2D (asr/nn/nn.py:235-238).
 *   asr_gemm_nt      C[M,N] = A[M,K] * B[N,K]^T (+ bias[N]);  A, B bf16, k-contiguous (fast path: lda/ldb/K multiples
 *                    of 8 and 16-byte aligned bases; anything else takes element loads); C f32 (out_bf16 = 0) or bf16 (1)
 *   asr_gemm_tn_acc  C[M,N] += A[K,M]^T * B[K,N];  A, B bf16 row-major; C f32, accumulated with atomics (split-K)
 *   asr_gemm_tn_acc_group  n <= 4 such products (any shapes) in ONE launch; every argument a host array of n entries:
 *                    the weight gradients a recurrence releases together (dW_ih and the per-direction dW_hh of
 *                    chainer.links.NStepBiGRU, asr/nn/nn.py:3 -- cuDNN's RNN backward-weights forms them in one call too).
 *                    Two products may add into the same output.
 */
int asr_gemm_nt(void* stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias,
                int M, int N, int K, int out_bf16);
/* asr_gemm_nt_8ph: the same product (no element-load fall-back: K % 8 == 0, N % 4 == 0, lda / ldb multiples of 8, ldc of 4, 16-byte aligned
 * bases, operands below 2 GiB -- asr_gemm_nt_8ph_ok says whether a call qualifies) on the 256 x 256 tile kernel with eight waves in
 * two staggered groups (csrc/gemm8.hip); asr_gemm_nt routes the model's products to it.  With more than 256 tiles and K % 64 == 0,
 * 128 <= K <= 1024 (N <= 8192; bf16 output: N and ldc multiples of 8) it runs as ONE persistent workgroup per CU that walks its tiles as one
 * stream of K steps (gemm_nt_8pp_kernel): the forward projections of a recurrent layer (chainer.links.NStepBiGRU's W x, asr/nn/nn.py:3). */
int asr_gemm_nt_8ph_ok(const void* A, int lda, const void* B, int ldb, const void* C, int ldc, const float* bias, int M, int N, int K,
                       int out_bf16);
int asr_gemm_nt_8ph(void* stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias, int M, int N, int K,
                    int out_bf16);
int asr_gemm_tn_acc(void* stream, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K);
int asr_gemm_tn_acc_group(void* stream, int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                          float* const* C, const int* ldc, const int* M, const int* N, const int* K);
/* asr_gemm_tn_acc_group_8ph: the same products on the 256 x 256 tile / eight-wave kernel of csrc/gemm8.hip (transposing LDS reads);
 * every product must pass asr_gemm_tn_8ph_ok (M, N, lda, ldb multiples of 8, 16-byte aligned bases, operands below 2 GiB).
 * asr_gemm_tn_acc / asr_gemm_tn_acc_group route qualifying calls to it. */
int asr_gemm_tn_8ph_ok(const void* A, int lda, const void* B, int ldb, const float* C, int ldc, int M, int N, int K);
int asr_gemm_tn_acc_group_8ph(void* stream, int n, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                              float* const* C, const int* ldc, const int* M, const int* N, const int* K);

/* ---------------------------------------------------------------------------------------- layout / activations
 * Internal activations are (T, B, H, C) bf16 (time-major, channel-last); see DESIGN.md "Data layout in HBM".
 *   asr_cast_bf16     f32 (rows, cols) -> bf16, optionally transposed to (cols, rows)    [weight copies]
 *   asr_cast_bf16_many  the same for a table of matrices in one launch: jobs_dev = njobs x 6 long long in device
 *                     memory {src, dst, rows, cols, transpose, first_tile}; a tile is 64x64 elements of src, tiles are
 *                     numbered job after job, total_tiles = their sum  [all weight copies after an optimiser step]
 *   asr_permute4      dense dst (d0,d1,d2,d3) <- strided src (element strides s0..s3), f32/bf16 either side
 *   asr_im2col        col[(t,b,ho)][(kh,kw,ci)] (row pitch Kp, zero padded) from x with element strides
 *                     (sT,sB,sH,sC); reads x[t+kw-pad_t, b, ho+kh-pad_h, ci] for t in [0, Tout).  pad_t = KW-1 with
 *                     Tout = T + KW-1 is the reference's conv (pad both sides), Tout = T its causal crop
 *                     x[..., :-pad] (run/ctc/cnn/model.py:43-44)
 *   asr_col2im        adjoint of asr_im2col for a (T,B,Hin,Cin) bf16 input gradient
 *   asr_maxout2_*     nn.Maxout(2): asr/nn/nn.py:45-50 (pairs of adjacent channels; ties -> first)
 *   asr_maxpool_h_*   nn.MaxPooling2D(ksize=(k,1)): asr/nn/nn.py:95-103, stride k, cover_all=True
 *   asr_add_bf16      residual add (asr/nn/nn.py:322-328)
 *   asr_colsum_acc    out[c] += sum_r x[r][c]   (bias gradients)
 */
int asr_cast_bf16(void* stream, const float* src, void* dst, int rows, int cols, int transpose);
int asr_cast_bf16_many(void* stream, const long long* jobs_dev, int njobs, long long total_tiles);
int asr_bf16_to_f32(void* stream, const void* src, float* dst, long long n);
int asr_permute4(void* stream, const void* src, int src_bf16, void* dst, int dst_bf16, int d0, int d1, int d2, int d3,
                 long long s0, long long s1, long long s2, long long s3);
int asr_im2col(void* stream, const void* x, int x_bf16, long long sT, long long sB, long long sH, long long sC, int T,
               int B, int Hin, int Cin, int KH, int KW, int pad_h, int pad_t, int Tout, int Kp, void* col);
int asr_col2im(void* stream, const void* dcol, int T, int B, int Hin, int Cin, int KH, int KW, int pad_h, int pad_t,
               int Tout, int Kp, void* dx);
/* Convolution2D weights (Co, Ci, kh, kw) f32 <-> the GEMM's (Co, Kp) matrix with k = (kh, kw, ci) */
int asr_conv_weight_pack(void* stream, const float* W, void* dst_bf16, int Co, int Ci, int KH, int KW, int Kp,
                         int transpose);
/* implicit-GEMM convolution (no column matrix): x (Ts, B, Hs, Cs) bf16 with Cs % 8 == 0; W rows of pitch ldw >= KH*KW*Cs,
 * ldw % 32 == 0 and ldw % Cs == 0 (columns behind KH*KW*Cs are empty taps and must hold zeros);
 * out[(t, b, h)][n] = sum_{kh,kw,c} x[t + sgn (kw - pad_t)][b][h + sgn (kh - pad_h)][c] * W[n][(kh*KW + kw)*Cs + c]  (+ bias)
 * over rows t < Tr, h < Hr (zero outside x).  sgn = +1, rows = output positions, W = asr_conv_weight_pack(.., 0): the
 * forward convolution of nn.Convolution2D (asr/nn/nn.py:235-238); sgn = -1, x = output gradient, rows = input positions,
 * W = asr_conv_weight_pack_bwd: its backward-data.  out (Tr*B*Hr, N) bf16 or f32.
 * asr_conv_weight_pack_bwd: dst[ci][(kh*KW + kw)*Co + co] = W[co][ci][kh][kw] as bf16. */
int asr_conv_nt(void* stream, const void* x, const void* W, int ldw, void* out, int out_bf16, const float* bias, int Ts, int B,
                int Hs, int Cs, int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N);
/* asr_conv_nt_8ph: the same implicit convolution on the eight-wave kernel of csrc/gemm8.hip (Cs and the row pitch of W multiples of 64,
 * N % 4 == 0, Hs < 256, 16-byte aligned operands: asr_conv_nt_8ph_ok); asr_conv_nt routes the products with more than 128 output
 * columns that the LDS-resident kernel does not take to it. */
int asr_conv_nt_8ph_ok(const void* x, const void* W, int ldw, const void* out, int out_bf16, const float* bias, int Ts, int B, int Hs, int Cs,
                       int KH, int KW, int Tr, int Hr, int N);
int asr_conv_nt_8ph(void* stream, const void* x, const void* W, int ldw, void* out, int out_bf16, const float* bias, int Ts, int B, int Hs,
                    int Cs, int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N);
/* asr_conv_nt_8pn: the narrow form (N <= 128 output columns: 256 x 64 / 256 x 128 tiles) of the same kernel family */
int asr_conv_nt_8pn_ok(const void* x, const void* W, int ldw, const void* out, int out_bf16, const float* bias, int Ts, int B, int Hs, int Cs,
                       int KH, int KW, int Tr, int Hr, int N);
int asr_conv_nt_8pn(void* stream, const void* x, const void* W, int ldw, void* out, int out_bf16, const float* bias, int Ts, int B, int Hs,
                    int Cs, int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N);
/* The same product with the activation block of a tile resident in LDS (csrc/conv_direct.hip): one workgroup = one utterance x 256 / Hr
 * time steps x all Hr heights x 64 or 128 output columns; the (Tt + KW - 1) x (Hr + KH - 1) x Cs activations it can touch are loaded
 * once instead of once per tap.  bf16 output, Cs in {32, 64, 128, 256}, ldw % 32 == 0; asr_conv_direct_ok says whether a shape is
 * served (asr_conv_nt asks and dispatches by itself; ASR_DEBUG conv_direct=0 keeps the implicit-GEMM kernels). */
int asr_conv_direct_ok(int Ts, int B, int Hs, int Cs, int KH, int KW, int Tr, int Hr, int N, int K, int out_bf16);
int asr_conv_direct_nt(void* stream, const void* x, const void* W, int ldw, void* out, const float* bias, int Ts, int B, int Hs, int Cs,
                       int KH, int KW, int pad_h, int pad_t, int sgn, int Tr, int Hr, int N);
/* The first block of the models -- Convolution2D over the 8-channel padded features, Maxout(2), MaxPooling2D((k, 1)) (asr/nn/nn.py:235-238,
 * :45-50, :95-103 as run/ctc/cnn/model.py:42-48 stacks them) -- as ONE forward and ONE backward pass (csrc/conv_first.hip): the convolution
 * output (311 MB at T=1000, B=32 for BASELINE configs[1]) and its gradient never exist.
 * asr_conv_mp_ok: 1 <= Ci <= 8 real input channels, KH KW <= 15 taps, Co % 128 == 0, 2 <= k <= 4.
 * asr_conv_mp_fwd: x8 (Ts, B, Hs, 8) bf16 (asr_pack_input_pad); W (Co, 128) bf16, k = (kh KW + kw) 8 + ci, zeros behind 8 KH KW
 *   (asr_conv_weight_pack with Kp = 128); bias (Co) f32 or NULL; y (Tout, B, Hp, Co / 2) bf16 with Hp = cover_all pooled heights of Hout;
 *   idx (same shape, bytes): 2 * (row of the window) + (second channel of the pair) of the winner.  Values and tie rules of asr_conv_nt
 *   (bf16 out) followed by asr_maxout2_pool_fwd.
 * asr_conv_mp_bwd: gy (Tout, B, Hp, Co / 2) bf16, idx and x8 of the forward call; gW (Co, Ci, KH, KW) f32 += the weight gradient,
 *   gb (Co) f32 += the bias gradient (NULL: none); workspace: asr_conv_mp_bwd_workspace(Tout, B, Hout, Co, k) bytes, 16-byte aligned. */
int asr_conv_mp_ok(int Ci, int KH, int KW, int Co, int k);
int asr_conv_mp_fwd(void* stream, const void* x8, const void* W, int ldw, const float* bias, void* y, void* idx, int Ts, int B, int Hs,
                    int KH, int KW, int pad_h, int pad_t, int Tout, int Hout, int Co, int k);
long long asr_conv_mp_bwd_workspace(int Tout, int B, int Hout, int Co, int k);
int asr_conv_mp_bwd(void* stream, const void* gy, const void* idx, const void* x8, void* workspace, float* gW, float* gb, int Ts, int B,
                    int Hs, int Ci, int KH, int KW, int pad_h, int pad_t, int Tout, int Hout, int Co, int k);
/* any strided (T, B, H, C) f32 / bf16 tensor -> dense (T, B, H, Cpad) bf16, channels C..Cpad-1 zero: brings the loader's
 * (B, 3, 40, T) float32 minibatch into the layout of asr_conv_nt (first layer: C = 3 -> Cpad = 8) */
int asr_pack_input_pad(void* stream, const void* x, int x_bf16, long long sT, long long sB, long long sH, long long sC, int T,
                       int B, int H, int C, int Cpad, void* out_bf16);
int asr_conv_weight_pack_bwd(void* stream, const float* W, void* dst, int Co, int Ci, int KH, int KW);
/* weight gradient of the same convolution without a column matrix: g (Tr*B*Hr, Co) bf16 rows of pitch ldg = output gradient,
 * C[co][(kh*KW + kw)*Cs + c] += sum_{t,b,h} g[(t, b, h)][co] * x[t + kw - pad_t][b][h + kh - pad_h][c]   (f32, split-K atomics);
 * Cs % 8 == 0.  asr_conv_weight_grad_unpack adds such a (Co, Kp) scratch into the (Co, Ci, KH, KW) gradient; Cs = channel
 * pitch of the scratch rows (>= Ci; 0 = Ci). */
int asr_conv_tn_acc(void* stream, const void* g, int ldg, const void* x, float* C, int ldc, int Co, int Ts, int B, int Hs,
                    int Cs, int KH, int KW, int pad_h, int pad_t, int Tr, int Hr);
/* asr_conv_tn_acc_8ph: the same weight gradient on the eight-wave kernel of csrc/gemm8.hip (Co, Cs, ldg multiples of 8, 16-byte aligned
 * operands: asr_conv_tn_8ph_ok); asr_conv_tn_acc routes qualifying calls with at least 256 x 256 outputs to it. */
int asr_conv_tn_8ph_ok(const void* g, int ldg, const void* x, const float* C, int ldc, int Co, int Ts, int B, int Hs, int Cs, int KH, int KW,
                       int Tr, int Hr);
int asr_conv_tn_acc_8ph(void* stream, const void* g, int ldg, const void* x, float* C, int ldc, int Co, int Ts, int B, int Hs, int Cs, int KH,
                        int KW, int pad_h, int pad_t, int Tr, int Hr);
int asr_conv_weight_grad_unpack(void* stream, const float* scratch, float* gW, int Co, int Ci, int KH, int KW, int Kp, int Cs);
/* Few output tiles -> hundreds of K splits adding into the same few KB (first layer: 91 of 166 us were the atomics): with
 * asr_conv_tn_copies(...) == 8 the caller hands a zeroed scratch of 8 x (Co, ldc) floats, every XCD adds into its own copy, and
 * asr_conv_weight_grad_unpack_copies sums them into the gradient. */
int asr_conv_tn_copies(int Co, int Cs, int KH, int KW);
int asr_conv_tn_acc_copies(void* stream, const void* g, int ldg, const void* x, float* C, int ldc, int copies, int Co, int Ts, int B,
                           int Hs, int Cs, int KH, int KW, int pad_h, int pad_t, int Tr, int Hr);
int asr_conv_weight_grad_unpack_copies(void* stream, const float* scratch, int copies, float* gW, int Co, int Ci, int KH, int KW, int Kp,
                                       int Cs);
int asr_maxout2_fwd(void* stream, const void* x, void* y, long long n_out);
int asr_maxout2_bwd(void* stream, const void* x, const void* dy, void* dx, long long n_out);
/* Maxout(2) + MaxPooling2D((k, 1)) in one pass: x (R, Hin, 2C) bf16 -> y (R, ceil(Hin / k), C); C % 8 == 0 (else
 * ASR_ERR_UNSUPPORTED: use the two calls).  Same results and tie rules as asr_maxout2_* followed by asr_maxpool_h_*. */
int asr_maxout2_pool_fwd(void* stream, const void* x, void* y, long long R, int Hin, int C, int k);
int asr_maxout2_pool_bwd(void* stream, const void* x, const void* dy, void* dx, long long R, int Hin, int C, int k);
/* The same with the bias gradient of the convolution in front: db (2 C floats, accumulated; may be NULL) += column sums of dx, formed
 * from dy and the winners while both are in registers (replaces the chainer Convolution2D backward's gy.sum over (N, H, W),
 * asr/nn/convolution_2d.py:76-90 through chainer.functions.convolution_2d).  asr_maxout2_pool_bwd_db_ok(C): 1 if db may be given. */
int asr_maxout2_pool_bwd_db(void* stream, const void* x, const void* dy, void* dx, float* db, long long R, int Hin, int C, int k);
int asr_maxout2_pool_bwd_db_ok(int C);
int asr_maxpool_h_fwd(void* stream, const void* x, void* y, long long R, int Hin, int C, int k);
int asr_maxpool_h_bwd(void* stream, const void* x, const void* dy, void* dx, long long R, int Hin, int C, int k);
int asr_add_bf16(void* stream, const void* a, const void* b, void* y, long long n);
/* activations of asr/nn/nn.py:11-73 on bf16: kind 0 relu, 1 clipped_relu(alpha=z), 2 leaky_relu(alpha=slope),
 * 3 elu(alpha), 4 sigmoid, 5 tanh, 6 hard_sigmoid, 7 softplus(alpha=beta); GLU asr/nn/nn.py:267-281 on rows [A|B];
 * dropout (asr/nn/nn.py:211-218) with a counter-based mask that backward regenerates from the same seed. */
int asr_activation_fwd(void* stream, const void* x, void* y, long long n, int kind, float alpha);
int asr_activation_bwd(void* stream, const void* x, const void* dy, void* dx, long long n, int kind, float alpha);
int asr_glu_fwd(void* stream, const void* x, void* y, long long rows, int C);
int asr_glu_bwd(void* stream, const void* x, const void* dy, void* dx, long long rows, int C);
int asr_dropout(void* stream, const void* x, void* y, long long n, float ratio, unsigned int seed);
int asr_colsum_acc(void* stream, const void* x, int x_bf16, long long rows, int cols, int ld, float* out);

/* ---------------------------------------------------------------------------------------- layer normalisation
 * nn.LayerNormalization (asr/nn/nn.py:240-265) = NormalizeLayer (asr/nn/layernorm.py:29-64) + scale/bias on axis 1.
 * rows = T*B, D = H*C contiguous per row, channel = index % C.  No epsilon (the reference ignores it).
 */
int asr_layernorm_fwd(void* stream, const void* x, int x_bf16, void* y, int y_bf16, const float* gamma,
                      const float* beta, float* mean, float* rstd, long long rows, int D, int C);
/* float32 rows normalised over their whole width (C == D, D % 4 == 0, D <= 4096: the logits of a frame over the vocabulary,
 * asr/nn/layernorm.py:33-48 as used by run/ctc/model.py on the output layer): one wave per row, the row in registers, and --
 * lse != NULL -- the log-sum-exp of every output row for asr_ctc_forward_lse.  asr_layernorm_fwd_lse_ok(D, C): 1 if it applies. */
int asr_layernorm_fwd_lse(void* stream, const float* x, float* y, const float* gamma, const float* beta, float* mean, float* rstd,
                          float* lse, long long rows, int D);
int asr_layernorm_fwd_lse_ok(int D, int C);
int asr_layernorm_bwd(void* stream, const void* x, int x_bf16, const void* dy, int dy_bf16, const float* gamma,
                      const float* mean, const float* rstd, void* dx, int dx_bf16, float* dgamma, float* dbeta,
                      long long rows, int D, int C);
/* one-sweep backward of the float32 logits normalisation (x and dy float32, D % 4 == 0, C % 4 == 0, D <= 4096; other
 * shapes: ASR_ERR_UNSUPPORTED, use asr_layernorm_bwd): dx (bf16 or f32, may be NULL) and dgamma += / dbeta += (both or
 * neither) in one pass over x and dy; ws: asr_layernorm_bwd_rows_ws_bytes(rows, D) bytes of scratch for the per-workgroup
 * column sums (0 = shape not supported).  Same formulas as asr_layernorm_bwd (asr/nn/layernorm.py:50-61). */
long long asr_layernorm_bwd_rows_ws_bytes(long long rows, int D);
int asr_layernorm_bwd_rows(void* stream, const float* x, const float* dy, const float* gamma, const float* mean,
                           const float* rstd, void* dx, int dx_bf16, float* dgamma, float* dbeta, long long rows, int D,
                           int C, void* ws, long long ws_bytes);

/* weight normalisation of asr/nn/convolution_2d.py: W = g V / (||V|| + 1e-9) per output channel (:21-25,62-64), its
 * gradient (:92-93, accumulated into gV / gg), and the data-dependent initialisation g = 1/std_t, b = -mean_t/std_t
 * from the first batch's g=1 output (:152-167,177-187). */
int asr_weightnorm_fwd(void* stream, const float* V, const float* g, float* W, float* norm, int Co, int K);
int asr_weightnorm_bwd(void* stream, const float* gW, const float* V, const float* g, const float* norm, float* gV,
                       float* gg, int Co, int K);
int asr_channel_stats(void* stream, const float* x, long long rows, int C, float* mean, float* stdv);
int asr_channel_affine(void* stream, const float* x, const float* scale, const float* shift, void* y_bf16, long long n,
                       int C);
int asr_weightnorm_init(void* stream, const float* mean, const float* stdv, float* g, float* b, int C);

/* dgamma / dbeta += column sums of partial (G, 2, D) written by a one-sweep backward (channel = index % C); with `extra`
 * the buffer is (G, 3, D) and extra[D] += the column sums of the third plane */
int asr_layernorm_fold_partials(void* stream, const float* partial, int G, int D, int C, float* dgamma, float* dbeta, float* extra);
/* Backward of [LayerNormalization over the vocabulary of each frame] -> [one or two CTC-family losses] in one sweep: the
 * gradient with respect to the normalised logits, (softmax - occupancy) * scale * gy summed over the losses
 * (asr/loss/gram_ctc.py:284-297), is formed in registers from the pre-normalisation rows x (T*B, V) f32, the row
 * statistics of asr_layernorm_fwd and the workspaces asr_ctc_forward left behind (ctc_ws: same T, B, Lmax, gram as that
 * call; x_len / gy / gy_per_utt / scale as for asr_ctc_backward), and goes straight into the layer-norm backward
 * (asr/nn/layernorm.py:50-61): dx (f32 or bf16, may be NULL) and dgamma / dbeta ACCUMULATED (both or neither).  V % 4 == 0,
 * V <= 4096, 3*Lmax+1 <= 512.  ws: asr_layernorm_ctc_bwd_ws_bytes(T, B, V) bytes.  dxsum (V floats, or NULL; needs dgamma / dbeta):
 * ACCUMULATES the column sums of dx -- the bias gradient of the projection that produced x. */
long long asr_layernorm_ctc_bwd_ws_bytes(int T, int B, int V);
int asr_layernorm_ctc_bwd(void* stream, const float* x, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, void* dx, int dx_bf16, float* dgamma, float* dbeta, int T, int B, int V, void* ws,
                          long long ws_bytes, int nloss, const void* ctc_ws0, int Lmax0, int gram0, const int32_t* x_len0,
                          const float* gy0, int gy_per_utt0, float scale0, const void* ctc_ws1, int Lmax1, int gram1,
                          const int32_t* x_len1, const float* gy1, int gy_per_utt1, float scale1, float* dxsum);

/* ---------------------------------------------------------------------------------------- batch normalisation
 * chainer.links.BatchNormalization reaches the reference API through `from chainer.links import *` (asr/nn/nn.py:3);
 * Chainer's rule: normalise with the biased batch variance, eps = 2e-5, running averages with decay 0.9 and the
 * unbiased variance.  x, gy, y, dx: (R, C) bf16, channel last (R = every other axis).  ws2C: 2*C doubles of scratch.
 *   asr_batchnorm_stats  mean, rstd = 1/sqrt(var + eps) (C) f32; avg_mean / avg_var (both or neither) updated in place
 *   asr_batchnorm_fwd    y = gamma (x - mean) rstd + beta   (also the inference form, with rstd from the running variance)
 *   asr_batchnorm_bwd    dx (may be NULL) and dgamma / dbeta ACCUMULATED (may be NULL)
 */
int asr_batchnorm_stats(void* stream, const void* x_bf16, long long R, int C, float eps, float decay, double* ws2C,
                        float* mean, float* rstd, float* avg_mean, float* avg_var);
/* out[i] = 1 / sqrt(var[i] + eps): rstd of the inference form from the running variance */
int asr_rsqrt_eps(void* stream, const float* var, float eps, float* out, int n);
int asr_batchnorm_fwd(void* stream, const void* x_bf16, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, long long R, int C, void* y_bf16);
int asr_batchnorm_bwd(void* stream, const void* x_bf16, const void* gy_bf16, const float* mean, const float* rstd,
                      const float* gamma, long long R, int C, double* ws2C, void* dx_bf16, float* dgamma_acc,
                      float* dbeta_acc);

/* ---------------------------------------------------------------------------------------- (Bi)GRU recurrence
 * nn.GRU / nn.NStepBiGRU reach the reference API through `from chainer.links import *` (asr/nn/nn.py:3); gate
 * convention = cuDNN / torch.nn.GRU (r, z, n).  Layouts in csrc/gru.hip.  gi comes from asr_gemm_nt.
 * db_ih / db_hh (ndir, 3H) f32 or NULL: bias gradients, ACCUMULATED (sum over all rows of dgi / dgh).
 * sync_ws: asr_gru_sync_bytes(B, H, ndir) bytes of device memory (control words + the in-launch exchange
 * buffer, zeroed by the call) enabling the persistent
 * one-launch-per-layer form.  mode: 0 = automatic: persistent; a batch of more than 32 utterances (the reference trains with 128 per
 * bucket, run/ctc/cnn/train.py:72-73) runs as consecutive slabs of <= 32 rows through the default kernel pair where that pair serves
 * the shape (H % 128 == 0, offsets below 2 GiB) -- utterances are independent, the results are those of one launch, the time is
 * ceil(B / 32) recurrences -- and otherwise with one launch per time step,
 * 1 = one launch per time step, 2 = persistent with the placement-free hand-off only (sc1 write-through + agent-scope
 * counter; the wide kernels: 32 units x 4-row recurrences), 4 = persistent, XCD-local hand-off where the workgroups of a recurrence
 * find themselves on one XCD (decided inside the launch, falls back to the mode-2 protocol otherwise) signalled through a
 * line of per-producer flags, 8 = the same with the payload as its own signal (what mode 0 selects: consumers load until no
 * sentinel word is left: no flags, no store drain, one barrier per step), 7 = mode 4 with a forged split placement (test hook for
 * that fall-back).  Backward only: where H % 128 == 0 modes 0 / 8 run the partial-sum exchange kernel (a workgroup publishes the H
 * partial sums of dh of its 32 units, 4 KB per step in bf16, instead of every workgroup fetching the 3H gate gradients; dgh_bf16 is
 * then written by a storer wave and needs no sentinel fill); 9 = ask for that kernel, 10 = the same with a forged split placement
 * (its write-through stores).  Other values: ASR_ERR_BAD_ARG (3, 5 and 6 named kernel forms that no shape selected; removed).
 * Modes >= 2 return ASR_ERR_UNSUPPORTED instead of falling back to mode 1.
 * After a synchronisation ((int*)sync_ws)[1023] != 0 reports a timed-out in-launch wait (results invalid).  That word
 * is sticky: the calls zero every other control word but never this one, so a caller that reuses one sync_ws (zeroed
 * once) can look at it whenever convenient; once set, later launches give up immediately.
 */
size_t asr_gru_sync_bytes(int B, int H, int ndir);
/* gi: (T*B, ndir*3H) input projections, f32 (gi_bf16 = 0) or bf16 (gi_bf16 = 1: half the bytes written by the projection
 * GEMM and streamed here; accepted by the default persistent kernel only -- ask asr_gru_fwd_accepts_bf16_gi first, the call
 * returns -3 otherwise) */
int asr_gru_fwd_accepts_bf16_gi(int T, int B, int H, int ndir, int mode);
/* x_len (B) int32 or NULL: per-utterance frame counts -- chainer.links.NStepBiGRU (asr/nn/nn.py:3) runs every sequence over
 * its own length; with the padded (T, B) block of asr/data/processing.py:113-126 that means: row b is live for t < x_len[b], the
 * state is frozen beyond it (the reverse direction, which meets the padding first, therefore reaches t = x_len[b] - 1 with the zero
 * state it would start from), y is zero there and no gradient flows through those steps.  Implemented off the latency chain:
 * asr_gru_fwd overwrites the update-gate columns of gi on the dead rows (gi is scratch of the caller's projection GEMM: it is
 * MODIFIED when x_len is given) so that z == 1.0f exactly, asr_gru_bwd drops dy on the dead rows into dy_ws ((T*B, H) bf16,
 * required with x_len) -- every recurrence kernel form serves ragged batches unchanged. */
/* gates: (T*B, ndir, 4, H) = r | z | n | q saved for the backward pass, float32 -- or IEEE half (gates_f16 = 1: 2^-11 relative
 * rounding, half the bytes the forward storer writes and the backward loader reads beside the per-step hand-off) where the default
 * kernel pair serves the shape: ask asr_gru_gates_f16_ok (the calls return -3 otherwise); both calls must agree.  The half form is
 * BLOCKED by workgroup: (T*B, ndir, H/16, 4, 16) -- a row's r | z | n | q of 16 units are one 128-B line -- and private to the pair. */
int asr_gru_gates_f16_ok(int T, int B, int H, int ndir, int mode);
int asr_gru_fwd(void* stream, void* gi, int gi_bf16, const void* whh_bf16, const float* bhh, float* hseq, void* hseq_bf16,
                void* gates, void* y_bf16, int T, int B, int H, int ndir, void* sync_ws, int mode, const int* x_len, int gates_f16);
int asr_gru_bwd(void* stream, const void* dy_bf16, const void* gates, const float* hseq, const void* whhT_bf16,
                void* dgi_bf16, void* dgh_bf16, float* carry_ws, float* db_ih, float* db_hh, int T, int B, int H,
                int ndir, void* sync_ws, int mode, const int* x_len, void* dy_ws, int gates_f16);
/* The same layer with a GIVEN initial state (chainer.links.NStepGRU / NStepBiGRU: `hy, ys = rnn(hx, xs)`, exported by asr/nn/nn.py:3;
 * the reference's SRU model carries a state across calls, run/ctc/sru/model.py:105-122, no GRU recipe does).  hx (ndir, B, H) float32 or
 * NULL (= zeros); gi float32; gates float32 in the plain [T*B][ndir][4][H] layout.  They run on the one-launch-per-time-step kernels (the
 * persistent kernels start from a zero state).  The final state hy is rows of hseq: direction 0 at t = T - 1, direction 1 at t = 0 (with
 * x_len: the frozen state of a shorter utterance).  asr_gru_bwd_state: dhy (ndir, B, H) float32 or NULL = gradient arriving at hy; dhx
 * (ndir, B, H) float32 or NULL receives the gradient of hx; the caller adds dgh_{first step}^T . hx to the W_hh gradient (the
 * products over t >= 1 are those of asr_gru_bwd's caller). */
int asr_gru_fwd_state(void* stream, const float* gi, const void* whh_bf16, const float* bhh, const float* hx, float* hseq,
                      void* hseq_bf16, float* gates, void* y_bf16, int T, int B, int H, int ndir, const int* x_len);
int asr_gru_bwd_state(void* stream, const void* dy_bf16, const float* gates, const float* hseq, const float* hx, const float* dhy,
                      const void* whhT_bf16, void* dgi_bf16, void* dgh_bf16, float* carry_ws, float* db_ih, float* db_hh, float* dhx,
                      int T, int B, int H, int ndir, const int* x_len, void* dy_ws);

/* ---------------------------------------------------------------------------------------- SRU scan
 * Replaces the CUDA kernels `forward` / `backward` of asr/nn/sru.py:17-73,75-191 (SRUFunction.forward_gpu :327-367,
 * backward_gpu :372-433).  x, H, gH, gxh: (T, B, D) bf16;  U, gU: (T*B, 3D) [z | f | r] (U f32, gU bf16);
 * C (T, B, D) f32; bias (2D) [b_f | b_r]; c0, cT, gcT, gc0, mask: (B, D) f32 (c0 / mask / gH / gcT may be NULL).
 * asr_sru_combine: out = (a + b) * mask  (b NULL: out = a * mask) -- highway + projection gradient, input masking.
 * ws: asr_sru_ws_bytes(T, B, D) bytes of scratch (16-B aligned) for the chunked scans -- the cell recurrence is linear in c (and its
 * backward in gc), so time is cut into chunks that run in parallel, joined through per-chunk (product, sum) summaries (csrc/sru.hip);
 * NULL / too small / odd D / short T: the one-thread-per-column scans of the reference's shape serve.
 */
size_t asr_sru_ws_bytes(int T, int B, int D);
int asr_sru_fwd(void* stream, const void* x_bf16, const float* U, const float* bias, const float* c0, const float* mask,
                void* H_bf16, float* C, float* cT, int T, int B, int D, int use_tanh, void* ws, size_t ws_bytes);
int asr_sru_bwd(void* stream, const void* x_bf16, const float* U, const float* bias, const float* C, const float* c0,
                const float* mask, const void* gH_bf16, const float* gcT, void* gU_bf16, void* gxh_bf16, float* gbias,
                float* gc0, int T, int B, int D, int use_tanh, void* ws, size_t ws_bytes);
int asr_sru_combine(void* stream, const void* a_bf16, const void* b_bf16, const float* mask, void* out_bf16, long long n,
                    int BD);

/* ---------------------------------------------------------------------------------------- optimiser step
 * GradientClipping -> WeightDecay -> Adam over one flat buffer (run/ctc/cnn/train.py:142-147,200).
 * grad_scale multiplies every gradient first (1/world_size after a sum all-reduce).
 */
int asr_fill_f32(void* stream, float* p, long long n, float value);
int asr_sqnorm_acc(void* stream, const float* g, long long n, float* out);
int asr_clip_decay_adam(void* stream, float* p, const float* g, float* m, float* v, long long n, float alpha,
                        float beta1, float beta2, float eps, float weight_decay, float clip_threshold,
                        float grad_scale, const float* sqnorm, int step);
/* kind 0 SGD, 1 MomentumSGD, 2 NesterovAG (asr/optimizers.py:43-52), same clipping / decay front end */
int asr_clip_decay_sgd(void* stream, float* p, const float* g, float* v, long long n, int kind, float lr, float momentum,
                       float weight_decay, float clip_threshold, float grad_scale, const float* sqnorm);
/* The same step with its decisions taken on the device by one thread (no host synchronisation, no host-side step count):
 * asr_step_control takes the squared norm of the flat gradient g (n floats; per-workgroup partial sums into `partials`,
 * asr_sqnorm_partials_count(n) floats, then summed in a fixed order -- bit-identical on every data-parallel rank, which
 * float atomics are not), reads up to two abort words of persistent GRU launches (sync_ws int 1023 of asr_gru_fwd /
 * asr_gru_bwd; NULL = none) and writes ctl[8] = {drop, gradient factor, Adam's alpha_t, t, squared norm, ...}: the step is
 * dropped when the norm is not finite (the reference's NaN check, run/ctc/cnn/train.py:193-197) or an abort word is raised
 * (the recurrence's outputs are garbage); *applied_steps (device int, zero at start) counts the steps that were NOT dropped
 * and is the t of Adam's bias correction (the reference `continue`s before optimizer.update).  asr_adam_ctl / asr_sgd_ctl
 * apply the step ctl describes.  ctl[5] = 1 when the drop was caused by a recurrence that gave up -- here (abort words) or on
 * another data-parallel rank: reserved_index >= 0 names an element of g that belongs to no parameter; asr_gather_abort ORs any
 * number of abort words (device array of n device addresses) into *any_word and, when one is raised, plants a NaN in *poison
 * (that element of the LOCAL gradient buffer, before it is summed over the ranks), so every rank drops the same step and every
 * rank can tell why; ctl[6] (zeroed once by the caller) counts such steps, so a host that reads ctl late misses none. */
int asr_sqnorm_partials_count(long long n);
int asr_gather_abort(void* stream, const long long* word_ptrs, int n, int* any_word, float* poison);
int asr_step_control(void* stream, const float* g, long long n, float* partials, const int* abort0, const int* abort1,
                     float clip_threshold, float grad_scale, float alpha, float beta1, float beta2, int* applied_steps,
                     float* ctl, int reserved_index);
/* asr_step_control with loss scaling (chainer.Optimizer.loss_scaling(interval, scale); the IEEE-half build of BASELINE configs[4]):
 * loss_scale = device float[4] {scale S, applied steps since S last changed, growth interval (0 = S is static), overflows so far}.
 * The caller seeds the backward pass with S (the gradient handed to asr_ctc_backward's gy is the device float loss_scale[0] itself,
 * read when that kernel runs); this call divides the gradient factor by S, and when the norm is not finite without a recurrence
 * having given up -- an activation gradient overflowed the half range -- drops the step and halves S (down to 2^-24: a model whose
 * unscaled activation gradients leave the half range needs S < 1); `interval` applied steps in a row double it (up to 2^24).  No host synchronisation; identical on every data-parallel rank.
 * loss_scale == NULL: asr_step_control. */
int asr_step_control_scaled(void* stream, const float* g, long long n, float* partials, const int* abort0, const int* abort1,
                            float clip_threshold, float grad_scale, float alpha, float beta1, float beta2, int* applied_steps,
                            float* ctl, int reserved_index, float* loss_scale);
int asr_adam_ctl(void* stream, float* p, const float* g, float* m, float* v, long long n, float beta1, float beta2, float eps,
                 float weight_decay, const float* ctl);
int asr_sgd_ctl(void* stream, float* p, const float* g, float* v, long long n, int kind, float lr, float momentum,
                float weight_decay, const float* ctl);

/* ---------------------------------------------------------------------------------------- the rest of asr.nn's function layers
 * asr/nn/nn.py:18-23 CReLU, :42-43 LogSoftmax, :58-63 Softmax (axis 1 = the channels = the contiguous axis of the physical layout),
 * :77-93 AveragePooling2D / ND (ksize (k, 1), stride = ksize, pad 0; Chainer's average pooling has no cover_all: Hout = (Hin - k) / k
 * + 1), :123-133 Unpooling2D (ksize (k, 1), stride = ksize; Hout = k (Hin - 1) + 1 with cover_all, k Hin without), :220-231
 * GaussianNoise (x + N(0, std^2), counter-based).  No recipe of the reference uses them; rows = every axis but the channels;
 * x, y, dy, dx bf16. */
int asr_crelu_fwd(void* stream, const void* x, void* y, long long rows, int C);
int asr_crelu_bwd(void* stream, const void* x, const void* dy, void* dx, long long rows, int C);
int asr_softmax_fwd(void* stream, const void* x, void* y, long long rows, int C, int log_form);
int asr_softmax_bwd(void* stream, const void* y, const void* dy, void* dx, long long rows, int C, int log_form);
int asr_avgpool_h_fwd(void* stream, const void* x, void* y, long long R, int Hin, int C, int k);
int asr_avgpool_h_bwd(void* stream, const void* dy, void* dx, long long R, int Hin, int C, int k);
int asr_unpool_h_fwd(void* stream, const void* x, void* y, long long R, int Hin, int Hout, int C, int k);
int asr_unpool_h_bwd(void* stream, const void* dy, void* dx, long long R, int Hin, int Hout, int C, int k);
int asr_gaussian_noise(void* stream, const void* x, void* y, long long n, float stdv, unsigned int seed);
/* asr/nn/nn.py:135-146 UpSampling2D = chainer.functions.upsampling_2d: the inverse of a max pooling given its argmax positions
 * (`indexes` of a MaxPooling2D function object; ksize (k, 1), stride = ksize: the row inside the window).  asr_maxpool_h_indexes forms
 * them as uint8 (R, Hout, C) with Hout = the cover_all output height; upsample forward: y (R, Hout', C) with x[r][h][c] at row
 * h k + idx and zeros elsewhere; backward: dx[r][h][c] = dy[r][h k + idx][c]. */
int asr_maxpool_h_indexes(void* stream, const void* x, void* idx_u8, long long R, int Hin, int C, int k);
int asr_upsample_h_fwd(void* stream, const void* x, const void* idx_u8, void* y, long long R, int Hin, int Hout, int C, int k);
int asr_upsample_h_bwd(void* stream, const void* dy, const void* idx_u8, void* dx, long long R, int Hin, int Hout, int C, int k);
/* asr/nn/nn.py:115-121 SpatialPyramidPooling2D = chainer.functions.spatial_pyramid_pooling_2d with max pooling: level l < pyramid_height
 * pools the (H, T) plane of every (utterance, channel) over 2^l x 2^l bins (window ceil(size / 2^l), stride = window, -inf padding of
 * (2^l k - size + 1) / 2).  x: the physical (T, B, H, C) bf16 tensor; y: (B, asr_spp_bins(pyramid_height), C) bf16, bins level after
 * level, [by][bx] inside a level; pos (same shape, int32, may be NULL): flat h T + t position of each maximum (first one in (h, t)
 * order) for asr_spp_bwd, which ADDS dy into dx32 (T, B, H, C) float32 (zeroed by the caller: levels overlap). */
int asr_spp_bins(int pyramid_height);
int asr_spp_fwd(void* stream, const void* x, void* y, int* pos, int T, int B, int H, int C, int pyramid_height);
int asr_spp_bwd(void* stream, const void* dy, const int* pos, float* dx32, int T, int B, int H, int C, int pyramid_height);

#ifdef __cplusplus
}
#endif
#endif
