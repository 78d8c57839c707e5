// Log-mel filterbank features on the GPU, batched over utterances (gfx950).
//
// Replaces the per-utterance float64 NumPy loop of Processor.extract_batch_features (asr/data/processing.py:67-111):
//   fft.get_specgram   asr/fft.py:52-56  -> python_speech_features.sigproc.preemphasis / framesig / powspec
//                       (absent third party; published algorithm: y[0]=x[0], y[n]=x[n]-c*x[n-1]; frames of frame_len
//                        every frame_step, numframes = 1 + ceil((N - frame_len)/frame_step), zero padded, times window;
//                        |rfft_nfft|^2 / nfft)
//   fft.compute_logmel asr/fft.py:58-66  feat = pspec . fbank^T, exact zeros -> DBL eps, log
//   fft.compute_deltas asr/fft.py:6-19,90-99   delta[t] = (x[t+1] - x[t-1]) / 2 with edge padding, delta of delta, last 2 frames dropped
//   Loader normalisation asr/data/loaders/base.py:22-24   (x - mean) / std per (channel, mel)
//
// specgram, nfft = 512 (the reference's 0.032 s x 16 kHz: asr/data/processing.py:54) -- specgram512_kernel: persistent workgroups of four
// waves, ONE FRAME PER WAVE.  The real 512-point transform is a 256-point complex one (z[n] = x[2n] + i x[2n+1]) plus a split pass; the 256
// points live four per lane in registers, four radix-4 stages, three exchanges through a padded LDS plane private to the wave (no workgroup
// barrier: a wave's LDS operations execute in order).  Twiddles and window are PER-LANE CONSTANTS (a lane meets the same butterfly of every
// frame): computed once per wave with sincospi, no table, no sin/cos in the loop.  Four consecutive frames share one staged, pre-emphasised
// span of samples (double-buffered: one workgroup barrier per four frames).  The mel stage runs on the filters' non-zero bands only (band
// start / length found once per workgroup from the dense matrix the caller passes: 454 of 10280 entries at 40 x 257).
// Other sizes: specgram_kernel, one workgroup per frame, radix-2 in LDS.
// deltas:   a workgroup per (utterance, 64 frames): the 68 log-mel rows it needs come in as ONE contiguous block, staged in LDS; a wave
// writes 64 consecutive frames of one (channel, mel) row of the reference's (B, 3, nmel, Tmax) float32 minibatch (zero padded beyond each
// utterance's length, asr/data/processing.py:124).
#include "common.hpp"
#include "../../include/asr_hip.h"

namespace asr {
namespace fbank {

constexpr int kMaxFft = 1024;
constexpr float kPi = 3.14159265358979323846f;

__device__ __forceinline__ int bitrev(int x, int bits) { return (int)(__brev((unsigned)x) >> (32 - bits)); }

template <typename SigT>
__global__ __launch_bounds__(256) void specgram_kernel(const SigT* __restrict__ signals, const int* __restrict__ lengths,
                                                       long long sig_pitch, int frame_len, int frame_step, int nfft,
                                                       int logn, float preemph, const float* __restrict__ window,
                                                       const int* __restrict__ nframes, int Fmax,
                                                       float* __restrict__ pspec_out,        // (B, Fmax, nfft/2+1) or null
                                                       const float* __restrict__ fbank, int nfilt,
                                                       float* __restrict__ logmel_out) {     // (B, Fmax, nfilt) or null
    __shared__ float re[kMaxFft], im[kMaxFft];
    __shared__ float ps[kMaxFft / 2 + 1];
    const int b = blockIdx.y, f = blockIdx.x;
    if (f >= nframes[b]) return;
    const SigT* sig = signals + (size_t)b * sig_pitch;
    const int N = lengths[b];
    const int start = f * frame_step;
    for (int i = threadIdx.x; i < nfft; i += blockDim.x) {
        float v = 0.f;
        if (i < frame_len) {
            const int n = start + i;
            if (n < N) {
                const float x = (float)sig[n];
                v = n == 0 ? x : x - preemph * (float)sig[n - 1];
                v *= window[i];
            }
        }
        const int r = bitrev(i, logn);
        re[r] = v;
        im[r] = 0.f;
    }
    __syncthreads();
    for (int s = 1; s <= logn; ++s) {
        const int half = 1 << (s - 1);
        for (int k = threadIdx.x; k < (nfft >> 1); k += blockDim.x) {
            const int grp = k / half, pos = k - grp * half;
            const int i0 = grp * (half << 1) + pos, i1 = i0 + half;
            float sn, cs;
            sincosf(-kPi * (float)pos / (float)half, &sn, &cs);
            const float tr = re[i1] * cs - im[i1] * sn, ti = re[i1] * sn + im[i1] * cs;
            const float ur = re[i0], ui = im[i0];
            re[i0] = ur + tr; im[i0] = ui + ti;
            re[i1] = ur - tr; im[i1] = ui - ti;
        }
        __syncthreads();
    }
    const int nbins = (nfft >> 1) + 1;
    const size_t frame = (size_t)b * Fmax + f;
    for (int k = threadIdx.x; k < nbins; k += blockDim.x) {
        const float p = (re[k] * re[k] + im[k] * im[k]) / (float)nfft;
        ps[k] = p;
        if (pspec_out) pspec_out[frame * nbins + k] = p;
    }
    if (logmel_out) {
        __syncthreads();
        for (int m = threadIdx.x; m < nfilt; m += blockDim.x) {
            const float* w = fbank + (size_t)m * nbins;
            float acc = 0.f;
            for (int k = 0; k < nbins; ++k) acc += ps[k] * w[k];
            if (acc == 0.f) acc = 2.220446049250313e-16f;          // np.finfo(float).eps (asr/fft.py:64)
            logmel_out[frame * nfilt + m] = logf(acc);
        }
    }
}


// ------------------------------------------------------------------------------------------------ nfft = 512: one frame per wave
namespace f512 {
constexpr int kPts = 256;                           // complex points of the packed real transform
constexpr int kPlane = kPts + (kPts >> 4) * 4;      // phys(p) = p + 4 (p >> 4): every exchange below is free of LDS bank conflicts
constexpr int kFramesPerGroup = 4;                  // = waves of the workgroup
constexpr int kSpanMax = 1024;                      // samples four frames span: 3 frame_step + 512, four per thread
constexpr int kBandFilt = 64, kMaxTaps = 1024;      // asr_mel_bands' table: 3 x 64 ints + 1024 taps (else: the dense rows from global memory)
constexpr int kMaxFilt = kBandFilt;

__device__ __forceinline__ int phys(int p) { return p + ((p >> 4) << 2); }

// a wave's LDS operations execute in issue order: between a wave's own writes and its reads of other lanes' words only the compiler
// has to be held back
__device__ __forceinline__ void wave_exchange_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// forward radix-4 decimation-in-frequency butterfly on x[p] = x(j + p L/4): y[q] = sum_p x[p] (-i)^(p q), then y[q] *= w[q] (q = 1..3)
__device__ __forceinline__ void bfly4(float (&xr)[4], float (&xi)[4]) {
    const float ar = xr[0] + xr[2], ai = xi[0] + xi[2], cr = xr[0] - xr[2], ci = xi[0] - xi[2];
    const float br = xr[1] + xr[3], bi = xi[1] + xi[3], er = xr[1] - xr[3], ei = xi[1] - xi[3];
    xr[0] = ar + br; xi[0] = ai + bi;
    xr[2] = ar - br; xi[2] = ai - bi;
    xr[1] = cr + ei; xi[1] = ci - er;               // c - i e
    xr[3] = cr - ei; xi[3] = ci + er;               // c + i e
}
__device__ __forceinline__ void twiddle(float (&xr)[4], float (&xi)[4], const float (&wr)[3], const float (&wi)[3]) {
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const float r = xr[q] * wr[q - 1] - xi[q] * wi[q - 1], i = xr[q] * wi[q - 1] + xi[q] * wr[q - 1];
        xr[q] = r; xi[q] = i;
    }
}
// w[q - 1] = exp(-2 pi i j q / L)
__device__ __forceinline__ void make_twiddles(int j, int L, float (&wr)[3], float (&wi)[3]) {
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        float sn, cs;
        sincospif(2.0f * (float)(j * q) / (float)L, &sn, &cs);
        wr[q - 1] = cs; wi[q - 1] = -sn;
    }
}


// The mel matrix is triangles on a few bins each (454 of 10280 entries non-zero at 40 x 257): asr_mel_bands writes, once per matrix,
//   int start[64], len8[64], off[64]; float taps[1024]
// -- per filter its first non-zero bin, the band's length rounded up to a multiple of 8, the offset of its taps; the taps of a band in
// ascending bin order, zeros behind its end.  len8[0] = -1: the matrix does not fit (more than 64 filters or 1024 padded taps); the
// transform then reads the dense rows.  One workgroup; two passes over the dense matrix.
__global__ __launch_bounds__(256) void mel_bands_kernel(const float* __restrict__ fbank, int nfilt, int nbins, int* __restrict__ table) {
    __shared__ int lo[kBandFilt], hi[kBandFilt], off[kBandFilt], ok;
    const int tid = threadIdx.x, lane = tid & 63;
    float* taps = reinterpret_cast<float*>(table + 3 * kBandFilt);
    for (int i = tid; i < kMaxTaps; i += 256) taps[i] = 0.f;
    if (tid < kBandFilt) { lo[tid] = nbins; hi[tid] = -1; }
    __syncthreads();
    const bool fits = nfilt <= kBandFilt;
    if (fits)
        for (int e = tid; e < nfilt * nbins; e += 256)
            if (fbank[e] != 0.f) {
                const int m = e / nbins, k = e - m * nbins;
                atomicMin(&lo[m], k);
                atomicMax(&hi[m], k);
            }
    __syncthreads();
    if (tid < 64) {
        const int h = (fits && lane < nfilt) ? hi[lane] : -1;
        const int len = h >= 0 ? ((h - lo[lane] + 1 + 7) & ~7) : 0;
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        const int total = __shfl(incl, 63, 64);
        const bool good = fits && total <= kMaxTaps;
        off[lane] = incl - len;
        if (h < 0) lo[lane] = 0;
        table[lane] = lo[lane];
        table[kBandFilt + lane] = good ? len : (lane == 0 ? -1 : 0);
        table[2 * kBandFilt + lane] = incl - len;
        if (lane == 0) ok = good ? 1 : 0;
    }
    __syncthreads();
    if (ok)
        for (int e = tid; e < nfilt * nbins; e += 256) {
            const float v = fbank[e];
            if (v != 0.f) {
                const int m = e / nbins, k = e - m * nbins;
                taps[off[m] + k - lo[m]] = v;
            }
        }
}

template <typename SigT>
__global__ __launch_bounds__(256) void specgram512_kernel(const SigT* __restrict__ signals, const int* __restrict__ lengths,
                                                          long long sig_pitch, int frame_len, int frame_step, float preemph,
                                                          const float* __restrict__ window, const int* __restrict__ nframes,
                                                          int Fmax, int B, int groups_per_utt, float* __restrict__ pspec_out,
                                                          const float* __restrict__ fbank, int nfilt, float* __restrict__ logmel_out,
                                                          const int* __restrict__ bands) {
    constexpr int nbins = 257;
    constexpr int kPsRow = 272;         // 257 bins + zeros: the mel loop reads whole groups of 8 taps
    __shared__ __attribute__((aligned(16))) float stage[2][kSpanMax];
    __shared__ __attribute__((aligned(16))) float2 plane[kFramesPerGroup][kPlane];      // (re, im) of point p at phys(p)
    __shared__ __attribute__((aligned(16))) float ps[kFramesPerGroup][kPsRow];
    __shared__ __attribute__((aligned(16))) float taps[kMaxTaps];
    __shared__ int fstart[kMaxFilt], flen8[kMaxFilt], foff[kMaxFilt];
    __shared__ int sparse_ok;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* pl = plane[wave];
    float* psw = ps[wave];

    // ---- once per workgroup: the mel filters' non-zero bands (asr_mel_bands' table: start, padded length, offset per filter and the
    // taps, each band padded with zeros to a multiple of 8) -- one 16-byte load per thread
    if (logmel_out) {
        const bool have = bands != nullptr && nfilt <= kBandFilt;
        if (have) {
            if (tid < kBandFilt) { fstart[tid] = bands[tid]; flen8[tid] = bands[kBandFilt + tid]; foff[tid] = bands[2 * kBandFilt + tid]; }
            reinterpret_cast<float4*>(taps)[tid] = reinterpret_cast<const float4*>(bands + 3 * kBandFilt)[tid];
        }
        for (int i = nbins + lane; i < kPsRow; i += 64) psw[i] = 0.f;
        __syncthreads();
        if (tid == 0) sparse_ok = (have && flen8[0] >= 0) ? 1 : 0;      // (a table that did not fit says so in its first length)
        __syncthreads();
    }

    // ---- once per wave: the lane's constants
    float w0r[3], w0i[3], w1r[3], w1i[3], w2r[3], w2i[3];
    make_twiddles(lane, 256, w0r, w0i);
    make_twiddles(lane & 15, 64, w1r, w1i);
    make_twiddles(lane & 3, 16, w2r, w2i);
    float win[8];                       // window at samples 2 n, 2 n + 1 for the lane's points n = lane + 64 p
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i0 = 2 * (lane + 64 * p);
        win[2 * p] = i0 < frame_len ? window[i0] : 0.f;
        win[2 * p + 1] = i0 + 1 < frame_len ? window[i0 + 1] : 0.f;
    }
    float sr[3], si[3];                 // exp(-2 pi i k / 512) for the split pass: k = lane, lane + 64, 128
    {
        const int ks[3] = {lane, lane + 64, 128};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            float sn, cs;
            sincospif((float)ks[u] / 256.0f, &sn, &cs);
            sr[u] = cs; si[u] = -sn;
        }
    }
    const int span = 3 * frame_step + 512;
    const int groups = B * groups_per_utt;

    // the samples of a group's span travel one group ahead of the transform: asked for before the wave starts on its frame, they are
    // in registers when the next round stores them (raw: x[n] and x[n - 1]; the pre-emphasis is formed at the store)
    // (the window loads above retire HERE: left pending, the compiler can only cover them inside the loop with vmcnt(0), which would also
    // wait for the samples fetched one group ahead)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    // thread tid owns samples 4 tid .. 4 tid + 3 of the span (span = 3 frame_step + 512 <= 1024 here): ONE aligned load of four samples + the
    // sample in front of them, branch-free (a piece outside the signal loads the utterance's first samples and is dropped at the store)
    typedef SigT sig4_t __attribute__((ext_vector_type(4)));
    sig4_t cur;
    SigT prv;
    auto fetch = [&](int g) {
        const int b = g / groups_per_utt, f0 = (g - b * groups_per_utt) * kFramesPerGroup;
        const SigT* sig = signals + (size_t)b * sig_pitch;
        const int N = f0 < nframes[b] ? lengths[b] : 0;
        const int n = f0 * frame_step + 4 * tid;
        const bool ok = 4 * tid < span && n < N;        // (the buffer behind a signal is the row's padding: sig_pitch >= N rounded up to 4)
        cur = *reinterpret_cast<const sig4_t*>(sig + (ok ? n : 0));
        prv = sig[ok && n > 0 ? n - 1 : 0];
    };
    int g = blockIdx.x;
    if (g < groups) fetch(g);
    int buf = 0;
    for (; g < groups; g += gridDim.x, buf ^= 1) {
        const int b = g / groups_per_utt, f0 = (g - b * groups_per_utt) * kFramesPerGroup;
        const int F = nframes[b];
        const int N = f0 < F ? lengths[b] : 0, n = f0 * frame_step + 4 * tid;
        float* st = stage[buf];
        // pre-emphasised samples of the four frames' span; the padding behind the signal is zero (framesig pads AFTER pre-emphasis)
        if (4 * tid < span) {
            const float x0 = (float)cur.x, x1 = (float)cur.y, x2 = (float)cur.z, x3 = (float)cur.w;
            float4 v;
            v.x = n < N ? x0 - (n > 0 ? preemph * (float)prv : 0.f) : 0.f;
            v.y = n + 1 < N ? x1 - preemph * x0 : 0.f;
            v.z = n + 2 < N ? x2 - preemph * x1 : 0.f;
            v.w = n + 3 < N ? x3 - preemph * x2 : 0.f;
            *reinterpret_cast<float4*>(st + 4 * tid) = v;
        }
        __syncthreads();                // (the other buffer is still being read by slower waves: two buffers, one barrier)
        if (g + (int)gridDim.x < groups) fetch(g + gridDim.x);
        const int f = f0 + wave;
        if (f >= F) continue;           // (per wave; no workgroup barrier below)
        const float* fr = st + wave * frame_step;
        float xr[4], xi[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float2 v = *reinterpret_cast<const float2*>(fr + 2 * (lane + 64 * p));
            xr[p] = v.x * win[2 * p];
            xi[p] = v.y * win[2 * p + 1];
        }
        // stage 0: L = 256, butterfly j = lane on points lane + 64 p
        bfly4(xr, xi);
        twiddle(xr, xi, w0r, w0i);
#pragma unroll
        for (int q = 0; q < 4; ++q) pl[phys(lane + 64 * q)] = make_float2(xr[q], xi[q]);
        wave_exchange_fence();
        // stage 1: L = 64, lane (j, m) = (lane & 15, lane >> 4) on points 64 m + j + 16 p
        const int j1 = lane & 15, m1 = lane >> 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) { const float2 v = pl[phys(64 * m1 + j1 + 16 * p)]; xr[p] = v.x; xi[p] = v.y; }
        wave_exchange_fence();
        bfly4(xr, xi);
        twiddle(xr, xi, w1r, w1i);
#pragma unroll
        for (int q = 0; q < 4; ++q) pl[phys(64 * m1 + j1 + 16 * q)] = make_float2(xr[q], xi[q]);
        wave_exchange_fence();
        // stage 2: L = 16, lane (j, b) = (lane & 3, lane >> 2) on points 16 b + j + 4 p
        const int j2 = lane & 3, b2 = lane >> 2;
#pragma unroll
        for (int p = 0; p < 4; ++p) { const float2 v = pl[phys(16 * b2 + j2 + 4 * p)]; xr[p] = v.x; xi[p] = v.y; }
        wave_exchange_fence();
        bfly4(xr, xi);
        twiddle(xr, xi, w2r, w2i);
#pragma unroll
        for (int q = 0; q < 4; ++q) pl[phys(16 * b2 + j2 + 4 * q)] = make_float2(xr[q], xi[q]);
        wave_exchange_fence();
        // stage 3: L = 4, points 4 lane + p (32 contiguous bytes), no twiddle
        {
            const float4 a = *reinterpret_cast<const float4*>(pl + phys(4 * lane));
            const float4 c = *reinterpret_cast<const float4*>(pl + phys(4 * lane) + 2);
            xr[0] = a.x; xi[0] = a.y; xr[1] = a.z; xi[1] = a.w;
            xr[2] = c.x; xi[2] = c.y; xr[3] = c.z; xi[3] = c.w;
        }
        wave_exchange_fence();
        bfly4(xr, xi);
        // position 4 lane + q = (d3 d2 d1 d0) base 4 holds Z[k], k = (d0 d1 d2 d3): into natural order
        {
            const int kb = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) pl[phys(kb + 64 * q)] = make_float2(xr[q], xi[q]);
        }
        wave_exchange_fence();
        // split pass: X[k] = E + W O, X[256 - k] = conj(E - W O), E = (Z[k] + conj Z[256 - k]) / 2, O = -i (Z[k] - conj Z[256 - k]) / 2,
        // W = exp(-2 pi i k / 512); power = |X|^2 / 512.  Pairs k = lane, lane + 64 and (every lane, lane 0 stores) k = 128.
        const size_t frame = (size_t)b * Fmax + f;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int k = u == 0 ? lane : (u == 1 ? lane + 64 : 128), kk = (256 - k) & 255;
            const float2 za = pl[phys(k)], zb = pl[phys(kk)];
            const float er = za.x + zb.x, ei = za.y - zb.y;             // 2 E
            const float dr = za.x - zb.x, di = za.y + zb.y;             // Z[k] - conj Z[256 - k];  2 O = -i (dr + i di) = (di, -dr)
            const float tr = di * sr[u] + dr * si[u], ti = di * si[u] - dr * sr[u];     // 2 W O
            const float ar = er + tr, ai = ei + ti, mr = er - tr, mi = ei - ti;
            const float pk = (ar * ar + ai * ai) * (1.0f / 2048.0f), pm = (mr * mr + mi * mi) * (1.0f / 2048.0f);
            if (u < 2 || lane == 0) {
                psw[k] = pk;
                if (u < 2) psw[256 - k] = pm;
                if (pspec_out) {
                    pspec_out[frame * nbins + k] = pk;
                    if (u < 2) pspec_out[frame * nbins + 256 - k] = pm;
                }
            }
        }
        if (logmel_out) {
            wave_exchange_fence();
            for (int m = lane; m < nfilt; m += 64) {
                float acc = 0.f;
                if (sparse_ok) {        // the band's taps in ascending order, eight at a time (zero taps behind the band's end)
                    const float* w = taps + foff[m];
                    const float* p = psw + fstart[m];
                    const int n8 = flen8[m];
                    for (int i = 0; i < n8; i += 8) {
                        const float4 wa = *reinterpret_cast<const float4*>(w + i), wb = *reinterpret_cast<const float4*>(w + i + 4);
                        float pv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) pv[u] = p[i + u];
                        acc += pv[0] * wa.x; acc += pv[1] * wa.y; acc += pv[2] * wa.z; acc += pv[3] * wa.w;
                        acc += pv[4] * wb.x; acc += pv[5] * wb.y; acc += pv[6] * wb.z; acc += pv[7] * wb.w;
                    }
                } else {
                    const float* w = fbank + (size_t)m * nbins;
                    for (int k = 0; k < nbins; ++k) acc += psw[k] * w[k];
                }
                if (acc == 0.f) acc = 2.220446049250313e-16f;          // np.finfo(float).eps (asr/fft.py:64)
                logmel_out[frame * nfilt + m] = logf(acc);
            }
            wave_exchange_fence();      // (the next frame's split pass writes ps again)
        }
    }
}
}  // namespace f512

// log(pspec . fbank^T) for a caller-supplied power spectrum (F, nbins) -> (F, nfilt)
__global__ void logmel_kernel(const float* __restrict__ pspec, const float* __restrict__ fbank, long long F, int nbins,
                              int nfilt, float* __restrict__ out) {
    const long long n = F * nfilt;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / nfilt;
        const int m = (int)(i - f * nfilt);
        const float* p = pspec + f * nbins;
        const float* w = fbank + (size_t)m * nbins;
        float acc = 0.f;
        for (int k = 0; k < nbins; ++k) acc += p[k] * w[k];
        if (acc == 0.f) acc = 2.220446049250313e-16f;
        out[i] = logf(acc);
    }
}

// x[b][c][m][t] for c = static, delta, delta-delta; t < nframes[b] - 2, zero beyond; optional (x - mean) / std
__global__ void deltas_kernel(const float* __restrict__ logmel, const int* __restrict__ nframes, int Fmax, int nfilt,
                              int Tmax, const float* __restrict__ mean, const float* __restrict__ stdv,
                              float* __restrict__ out, int B) {
    const long long n = (long long)B * nfilt * Tmax;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % Tmax);
        const int m = (int)((i / Tmax) % nfilt);
        const int b = (int)(i / ((long long)Tmax * nfilt));
        const int F = nframes[b];
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (t < F - 2) {
            const float* lm = logmel + (size_t)b * Fmax * nfilt + m;
            auto at = [&](int q) -> float { q = q < 0 ? 0 : (q > F - 1 ? F - 1 : q); return lm[(size_t)q * nfilt]; };
            auto dl = [&](int q) -> float { q = q < 0 ? 0 : (q > F - 1 ? F - 1 : q); return (at(q + 1) - at(q - 1)) * 0.5f; };
            v0 = at(t);
            v1 = dl(t);
            v2 = (dl(t + 1) - dl(t - 1)) * 0.5f;
        }
        if (mean) {     // the reference normalises the zero padding as well (asr/data/loaders/base.py:24)
            v0 = (v0 - mean[m]) / stdv[m];
            v1 = (v1 - mean[nfilt + m]) / stdv[nfilt + m];
            v2 = (v2 - mean[2 * nfilt + m]) / stdv[2 * nfilt + m];
        }
        float* o = out + ((size_t)b * 3 * nfilt + m) * Tmax + t;
        o[0] = v0;
        o[(size_t)nfilt * Tmax] = v1;
        o[(size_t)2 * nfilt * Tmax] = v2;
    }
}


// the same through an LDS tile: workgroup = (utterance, 32 output frames); rows t0 - 2 .. t0 + 33 of the log-mel matrix are one contiguous
// block of global memory (clamped rows at the utterance's ends); a half wave then owns one mel row at a time and its lanes 32 consecutive
// frames: whole 128-byte lines of the (B, 3, nmel, Tmax) output.  Row pitch nfilt | 1 (odd): conflict-free column reads.  1024 workgroups
// at the BASELINE size (four per CU): the kernel is two dependent memory round trips, so it wants many workgroups in flight.
constexpr int kDeltaTile = 32, kDeltaMaxFilt = 64;
__global__ __launch_bounds__(256) void deltas_tile_kernel(const float* __restrict__ logmel, const int* __restrict__ nframes, int Fmax, int nfilt,
                                                          int Tmax, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                          float* __restrict__ out) {
    __shared__ float tile[(kDeltaTile + 4) * (kDeltaMaxFilt + 1)];
    __shared__ float smean[3 * kDeltaMaxFilt], sstd[3 * kDeltaMaxFilt];      // (read per mel row below: from LDS, not a memory round trip per row)
    const int b = blockIdx.y, t0 = blockIdx.x * kDeltaTile;
    if (mean && (int)threadIdx.x < 3 * nfilt) { smean[threadIdx.x] = mean[threadIdx.x]; sstd[threadIdx.x] = stdv[threadIdx.x]; }
    const int F = nframes[b];
    const int pitch = nfilt | 1;
    const int rows = kDeltaTile + 4;
    if (t0 < F - 2) {
        const float* lm = logmel + (size_t)b * Fmax * nfilt;
        for (int i = threadIdx.x; i < rows * nfilt; i += 256) {
            const int r = i / nfilt, c = i - r * nfilt;
            int q = t0 - 2 + r;
            q = q < 0 ? 0 : (q > F - 1 ? F - 1 : q);
            tile[r * pitch + c] = lm[(size_t)q * nfilt + c];
        }
    }
    __syncthreads();
    const int tl = threadIdx.x & 31, t = t0 + tl;
    if (t >= Tmax) return;
    const bool live = t < F - 2;
    // rows of frames t - 2 .. t + 2, clamped the way the reference pads (edge frames repeated): at(q) = logmel[clamp(q)],
    // dl(q) = (at(clamp(q) + 1) - at(clamp(q) - 1)) / 2 with clamp(q) in [0, F - 1]
    auto cl = [&](int q) -> int { return q < 0 ? 0 : (q > F - 1 ? F - 1 : q); };
    const int c0 = cl(t - 1), c1 = cl(t), c2 = cl(t + 1);
    const int r_t = (c1 - t0 + 2) * pitch;
    const int r_d0a = (cl(c0 + 1) - t0 + 2) * pitch, r_d0b = (cl(c0 - 1) - t0 + 2) * pitch;       // dl(t - 1)
    const int r_d1a = (cl(c1 + 1) - t0 + 2) * pitch, r_d1b = (cl(c1 - 1) - t0 + 2) * pitch;       // dl(t)
    const int r_d2a = (cl(c2 + 1) - t0 + 2) * pitch, r_d2b = (cl(c2 - 1) - t0 + 2) * pitch;       // dl(t + 1)
    for (int m = threadIdx.x >> 5; m < nfilt; m += 8) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (live) {
            v0 = tile[r_t + m];
            v1 = (tile[r_d1a + m] - tile[r_d1b + m]) * 0.5f;
            v2 = ((tile[r_d2a + m] - tile[r_d2b + m]) * 0.5f - (tile[r_d0a + m] - tile[r_d0b + m]) * 0.5f) * 0.5f;
        }
        if (mean) {     // the reference normalises the zero padding as well (asr/data/loaders/base.py:24)
            v0 = (v0 - smean[m]) / sstd[m];
            v1 = (v1 - smean[nfilt + m]) / sstd[nfilt + m];
            v2 = (v2 - smean[2 * nfilt + m]) / sstd[2 * nfilt + m];
        }
        float* o = out + ((size_t)b * 3 * nfilt + m) * Tmax + t;
        o[0] = v0;
        o[(size_t)nfilt * Tmax] = v1;
        o[(size_t)2 * nfilt * Tmax] = v2;
    }
}

// cepstral mean normalisation in the log-power domain (asr/data/processing.py:86-89):
// pspec[f][k] <- exp(log pspec[f][k] - mean_f log pspec[f][k]); one thread per (utterance, bin), frames in sequence
__global__ void cmn_pspec_kernel(float* __restrict__ pspec, const int* __restrict__ nframes, int B, int Fmax, int nbins) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nbins) return;
    const int b = i / nbins, k = i - b * nbins;
    const int F = nframes[b];
    float* p = pspec + (size_t)b * Fmax * nbins + k;
    double acc = 0.0;
    for (int f = 0; f < F; ++f) acc += (double)logf(p[(size_t)f * nbins]);
    const float mean = F > 0 ? (float)(acc / F) : 0.f;
    for (int f = 0; f < F; ++f) p[(size_t)f * nbins] = expf(logf(p[(size_t)f * nbins]) - mean);
}

// speed / vocal-tract-length perturbation by nearest-index resampling (asr/fft.py:21-50):
//   out[f][k] = in[int(f * speed)][min(int(k * ratio), nbins - 1)],  f < nframes_out[b] = int(nframes_in[b] / speed)
// index arithmetic in float64 like NumPy's (np.arange(n) * speed).astype(int); speed = ratio = 1 is the identity
__global__ void augment_specgram_kernel(const float* __restrict__ in, const int* __restrict__ nframes_out,
                                        const double* __restrict__ speed, const double* __restrict__ ratio, int B, int Fin,
                                        int Fout, int nbins, float* __restrict__ out) {
    const long long n = (long long)B * Fout * nbins;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % nbins);
        const int f = (int)((i / nbins) % Fout);
        const int b = (int)(i / ((long long)nbins * Fout));
        float v = 0.f;
        if (f < nframes_out[b]) {
            int fs = (int)((double)f * speed[b]);
            if (fs > Fin - 1) fs = Fin - 1;
            int ks = (int)((double)k * ratio[b]);
            if (ks > nbins - 1) ks = nbins - 1;
            v = in[((size_t)b * Fin + fs) * nbins + ks];
        }
        out[i] = v;
    }
}

// white-noise augmentation (asr/data/processing.py:74-78): signal += trunc(gain_b * n), n ~ N(0, 1) from a counter-based
// generator (Box-Muller over two 32-bit hashes of (seed, utterance, sample)); the reference draws from NumPy's global
// stream, so only the distribution can match, not the samples
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__global__ void add_white_noise_kernel(float* __restrict__ signals, const int* __restrict__ lengths, long long pitch, int B,
                                       const float* __restrict__ gain, unsigned long long seed) {
    const int b = blockIdx.y;
    const int N = lengths[b];
    const float g = gain[b];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const unsigned c = (unsigned)n * 2u;
        const unsigned k = hash32((unsigned)seed ^ (unsigned)(seed >> 32) ^ ((unsigned)b * 0x9e3779b9u));
        const unsigned u1 = hash32(c ^ k), u2 = hash32((c + 1u) ^ k);
        const float f1 = ((float)(u1 >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
        const float f2 = (float)(u2 >> 8) * (1.0f / 16777216.0f);
        const float z = sqrtf(-2.0f * logf(f1)) * cosf(6.283185307179586f * f2);
        signals[(size_t)b * pitch + n] += truncf(g * z);                        // noise.astype(np.int16) truncates toward zero
    }
}

// running per-(channel, mel) statistics over every frame seen (asr/data/loaders/base.py:64-80), updated utterance by
// utterance with the reference's recursion, in float64; one workgroup per (channel, mel) pair.
//   new_mean = old_mean + (sum - n old_mean) / (total + n)
//   new_nvar = old_nvar + sqsum - sum (new_mean + old_mean) + n new_mean old_mean
// also writes mean and the unbiased standard deviation sqrt(nvar / (total - 1)) (:39-41) as float32
__global__ __launch_bounds__(256) void running_stats_kernel(const float* __restrict__ x, const int* __restrict__ lengths, int B,
                                                            int CM, int T, long long total_before, double* __restrict__ mean,
                                                            double* __restrict__ nvar, float* __restrict__ mean32,
                                                            float* __restrict__ std32) {
    __shared__ double ssum[4], ssq[4];
    const int cm = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double mu = mean[cm], nv = nvar[cm];
    long long total = total_before;
    for (int b = 0; b < B; ++b) {
        const int n = min(lengths[b], T);
        if (n <= 0) continue;
        const float* p = x + ((size_t)b * CM + cm) * T;
        double s = 0.0, q = 0.0;
        for (int t = tid; t < n; t += 256) {
            const double v = (double)p[t];
            s += v; q += v * v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
        __syncthreads();
        if (lane == 0) { ssum[wave] = s; ssq[wave] = q; }
        __syncthreads();
        s = ssum[0] + ssum[1] + ssum[2] + ssum[3];
        q = ssq[0] + ssq[1] + ssq[2] + ssq[3];
        const double new_mu = mu + (s - (double)n * mu) / (double)(total + n);
        nv = nv + q - s * (new_mu + mu) + (double)n * new_mu * mu;
        mu = new_mu;
        total += n;
    }
    if (tid == 0) {
        mean[cm] = mu;
        nvar[cm] = nv;
        mean32[cm] = (float)mu;
        std32[cm] = total > 1 ? (float)sqrt(nv / (double)(total - 1)) : 1.0f;
    }
}

// x[b][cm][t] <- (x - mean[cm]) / std[cm] over the whole padded array (asr/data/loaders/base.py:24)
__global__ void normalize_bcmt_kernel(float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ stdv,
                                      long long n, int CM, int T) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int cm = (int)((i / T) % CM);
        x[i] = (x[i] - mean[cm]) / stdv[cm];
    }
}

}  // namespace fbank
}  // namespace asr

using namespace asr;
using namespace asr::fbank;

static int specgram_launch(void* stream, const void* signals, int sig_is_f32, const int32_t* lengths, long long sig_pitch,
                           int B, int frame_len, int frame_step, int nfft, float preemph, const float* window,
                           const int32_t* nframes, int Fmax, float* pspec_out, const float* fbank, int nfilt,
                           float* logmel_out, const void* bands) {
    if (!signals || !lengths || !window || !nframes || B <= 0 || Fmax <= 0 || frame_step <= 0) return ASR_ERR_BAD_ARG;
    if (!pspec_out && !logmel_out) return ASR_ERR_BAD_ARG;
    if (logmel_out && (!fbank || nfilt <= 0)) return ASR_ERR_BAD_ARG;
    int logn = 0;
    while ((1 << logn) < nfft) ++logn;
    if ((1 << logn) != nfft || nfft > kMaxFft || nfft < 64 || frame_len > nfft || frame_len <= 0) return ASR_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    static const int fast = debug_flag("fbank_fast", 1);
    // one frame per wave (nfft 512): four samples per load need 4-sample alignment of every group's first sample (frame_step % 4 == 0 does
    // it: a group starts at frame 4 i), of the rows (sig_pitch % 4 == 0) and of the buffer; the span of four frames must fit 1024 samples
    const size_t esz = sig_is_f32 ? 4 : 2;
    if (fast && nfft == 512 && (frame_step & 3) == 0 && 3 * frame_step + 512 <= f512::kSpanMax && (sig_pitch & 3) == 0 &&
        (((uintptr_t)signals) & (4 * esz - 1)) == 0 && (((uintptr_t)bands) & 15) == 0) {
        // persistent workgroups of four frames (4 workgroups per CU keep every SIMD at 4 waves)
        const int gpu = (Fmax + f512::kFramesPerGroup - 1) / f512::kFramesPerGroup;
        const long long groups = (long long)B * gpu;
        const dim3 grid((unsigned)(groups < 1024 ? groups : 1024)), block(256);
        if (sig_is_f32)
            hipLaunchKernelGGL(f512::specgram512_kernel<float>, grid, block, 0, s, (const float*)signals, lengths, sig_pitch, frame_len,
                               frame_step, preemph, window, nframes, Fmax, B, gpu, pspec_out, fbank, nfilt, logmel_out, (const int*)bands);
        else
            hipLaunchKernelGGL(f512::specgram512_kernel<short>, grid, block, 0, s, (const short*)signals, lengths, sig_pitch, frame_len,
                               frame_step, preemph, window, nframes, Fmax, B, gpu, pspec_out, fbank, nfilt, logmel_out, (const int*)bands);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    const dim3 grid(Fmax, B), block(256);
    if (sig_is_f32)
        hipLaunchKernelGGL(specgram_kernel<float>, grid, block, 0, s, (const float*)signals, lengths, sig_pitch, frame_len,
                           frame_step, nfft, logn, preemph, window, nframes, Fmax, pspec_out, fbank, nfilt, logmel_out);
    else
        hipLaunchKernelGGL(specgram_kernel<short>, grid, block, 0, s, (const short*)signals, lengths, sig_pitch, frame_len,
                           frame_step, nfft, logn, preemph, window, nframes, Fmax, pspec_out, fbank, nfilt, logmel_out);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_specgram(void* stream, const void* signals, int sig_is_f32, const int32_t* lengths, long long sig_pitch,
                            int B, int frame_len, int frame_step, int nfft, float preemph, const float* window,
                            const int32_t* nframes, int Fmax, float* pspec_out, const float* fbank, int nfilt,
                            float* logmel_out) {
    return specgram_launch(stream, signals, sig_is_f32, lengths, sig_pitch, B, frame_len, frame_step, nfft, preemph, window, nframes, Fmax,
                           pspec_out, fbank, nfilt, logmel_out, nullptr);
}

extern "C" int asr_specgram_bands(void* stream, const void* signals, int sig_is_f32, const int32_t* lengths, long long sig_pitch,
                                  int B, int frame_len, int frame_step, int nfft, float preemph, const float* window,
                                  const int32_t* nframes, int Fmax, float* pspec_out, const float* fbank, int nfilt,
                                  float* logmel_out, const void* bands) {
    return specgram_launch(stream, signals, sig_is_f32, lengths, sig_pitch, B, frame_len, frame_step, nfft, preemph, window, nframes, Fmax,
                           pspec_out, fbank, nfilt, logmel_out, bands);
}

extern "C" size_t asr_mel_bands_bytes(void) { return (size_t)(3 * f512::kBandFilt + f512::kMaxTaps) * 4; }

extern "C" int asr_mel_bands(void* stream, const float* fbank, int nfilt, int nbins, void* table, size_t table_bytes) {
    if (!fbank || !table || nfilt <= 0 || nbins <= 0) return ASR_ERR_BAD_ARG;
    if (table_bytes < asr_mel_bands_bytes() || (((uintptr_t)table) & 15) != 0) return ASR_ERR_WORKSPACE;
    hipLaunchKernelGGL(f512::mel_bands_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fbank, nfilt, nbins, (int*)table);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_logmel(void* stream, const float* pspec, const float* fbank, long long F, int nbins, int nfilt,
                          float* out) {
    if (!pspec || !fbank || !out || F <= 0 || nbins <= 0 || nfilt <= 0) return ASR_ERR_BAD_ARG;
    long long g = (F * nfilt + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pspec, fbank, F, nbins, nfilt, out);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_deltas(void* stream, const float* logmel, const int32_t* nframes, int B, int Fmax, int nfilt, int Tmax,
                          const float* mean, const float* stdv, float* out) {
    if (!logmel || !nframes || !out || B <= 0 || Fmax <= 0 || nfilt <= 0 || Tmax <= 0) return ASR_ERR_BAD_ARG;
    if ((mean == nullptr) != (stdv == nullptr)) return ASR_ERR_BAD_ARG;
    static const int fast = debug_flag("fbank_fast", 1);
    if (fast && nfilt <= kDeltaMaxFilt) {
        hipLaunchKernelGGL(deltas_tile_kernel, dim3((Tmax + kDeltaTile - 1) / kDeltaTile, B), dim3(256), 0, (hipStream_t)stream, logmel, nframes,
                           Fmax, nfilt, Tmax, mean, stdv, out);
        ASR_LAUNCH_CHECK();
        return ASR_OK;
    }
    long long g = ((long long)B * nfilt * Tmax + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(deltas_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, logmel, nframes, Fmax, nfilt, Tmax,
                       mean, stdv, out, B);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_cmn_pspec(void* stream, float* pspec, const int32_t* nframes, int B, int Fmax, int nbins) {
    if (!pspec || !nframes || B <= 0 || Fmax <= 0 || nbins <= 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(cmn_pspec_kernel, dim3((B * nbins + 255) / 256), dim3(256), 0, (hipStream_t)stream, pspec, nframes, B, Fmax, nbins);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_add_white_noise(void* stream, float* signals, const int32_t* lengths, long long pitch, int B,
                                   const float* gain, unsigned long long seed) {
    if (!signals || !lengths || !gain || B <= 0 || pitch <= 0) return ASR_ERR_BAD_ARG;
    long long g = (pitch + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(add_white_noise_kernel, dim3((unsigned)g, B), dim3(256), 0, (hipStream_t)stream, signals, lengths, pitch, B, gain, seed);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_running_stats_update(void* stream, const float* x, const int32_t* lengths, int B, int CM, int T,
                                        long long total_before, double* mean, double* nvar, float* mean32, float* std32) {
    if (!x || !lengths || !mean || !nvar || !mean32 || !std32 || B <= 0 || CM <= 0 || T <= 0 || total_before < 0) return ASR_ERR_BAD_ARG;
    hipLaunchKernelGGL(running_stats_kernel, dim3(CM), dim3(256), 0, (hipStream_t)stream, x, lengths, B, CM, T, total_before, mean, nvar,
                       mean32, std32);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_normalize_bcmt(void* stream, float* x, const float* mean, const float* stdv, int B, int CM, int T) {
    if (!x || !mean || !stdv || B <= 0 || CM <= 0 || T <= 0) return ASR_ERR_BAD_ARG;
    const long long n = (long long)B * CM * T;
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(normalize_bcmt_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, mean, stdv, n, CM, T);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}

extern "C" int asr_augment_specgram(void* stream, const float* pspec_in, const int32_t* nframes_out, const double* speed,
                                    const double* ratio, int B, int Fmax_in, int Fmax_out, int nbins, float* pspec_out) {
    if (!pspec_in || !nframes_out || !speed || !ratio || !pspec_out || B <= 0 || Fmax_in <= 0 || Fmax_out <= 0 || nbins <= 0)
        return ASR_ERR_BAD_ARG;
    const long long n = (long long)B * Fmax_out * nbins;
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(augment_specgram_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, pspec_in, nframes_out, speed, ratio,
                       B, Fmax_in, Fmax_out, nbins, pspec_out);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
}
