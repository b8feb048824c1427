// MFCC front-end: 16 kHz audio -> 39-dimensional features (13 per-frame-normalised cepstra,
// 13 deltas, 13 delta-deltas), reproducing the librosa chain the reference calls
// (melspectrogram n_fft=320 hop=160 40 Slaney mels -> power_to_db(ref=max) -> DCT-II ->
// Savitzky-Golay deltas of width 9 -> per-frame z-normalisation of the statics).
//
// Kernel 1 (frame index): utterance id of every frame of the ragged batch.
// Kernel 2 (mel power):  one 256-thread workgroup per 16 frames.  The 320-point real FFT is
//   a 160-point complex FFT of (even, odd) sample pairs, factored 10 x 16: 16 lanes per frame
//   each do a 10-point DFT in registers, the data is transposed through LDS, 10 lanes per
//   frame each do a 16-point DFT, and the real spectrum is untangled from Z[k], Z[160-k].
//   Window product and FFT are float64 (librosa multiplies by a float64 window and rounds
//   the transform to complex64); magnitude, power and everything after are float32.
//   The 40 mel energies go to a [frame][40] scratch array and the utterance maximum is kept
//   with atomicMax on the (non-negative) float bit pattern.
// Kernel 3 (finish): one workgroup per 64 frames (+8 frames halo each side): dB with the
//   utterance maximum as reference and the -80 dB floor, DCT, deltas (edge frames repeat the
//   polynomial fit of the first/last full window), normalisation, coalesced store.
#include "lsr_internal.h"

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cplx cmul_negi(cplx a) { return {a.im, -a.re}; }   // a * (-i)
__device__ __forceinline__ cplx cmul_posi(cplx a) { return {-a.im, a.re}; }   // a * (+i)

// forward 5-point DFT
__device__ __forceinline__ void dft5(const cplx x0, const cplx x1, const cplx x2, const cplx x3, const cplx x4,
                                     cplx* X) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;   // cos(2pi/5), cos(4pi/5)
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;    // sin(2pi/5), sin(4pi/5)
    const cplx t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    X[0] = {x0.re + t1.re + t2.re, x0.im + t1.im + t2.im};
    const cplx m1 = {x0.re + c1 * t1.re + c2 * t2.re, x0.im + c1 * t1.im + c2 * t2.im};
    const cplx m2 = {x0.re + c2 * t1.re + c1 * t2.re, x0.im + c2 * t1.im + c1 * t2.im};
    const cplx n1 = {s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im};
    const cplx n2 = {s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im};
    X[1] = cadd(m1, cmul_negi(n1));
    X[4] = cadd(m1, cmul_posi(n1));
    X[2] = cadd(m2, cmul_negi(n2));
    X[3] = cadd(m2, cmul_posi(n2));
}

// forward 4-point DFT
__device__ __forceinline__ void dft4(const cplx a, const cplx b, const cplx c, const cplx d, cplx* X) {
    const cplx s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
    X[0] = cadd(s0, s2);
    X[2] = csub(s0, s2);
    X[1] = cadd(s1, cmul_negi(s3));
    X[3] = cadd(s1, cmul_posi(s3));
}

__global__ void lsr_frame_index_kernel(const int64_t* __restrict__ frame_off, int32_t* __restrict__ frame_utt) {
    const int b = blockIdx.x;
    const int64_t lo = frame_off[b], hi = frame_off[b + 1];
    for (int64_t f = lo + threadIdx.x; f < hi; f += blockDim.x) frame_utt[f] = b;
}

#define MP_FRAMES 16
#define MP_YLD 17            // complex elements per (frame, k1) row: 16 + 1 pad
#define MP_ZLD 161
#define MP_PLD 164

__global__ __launch_bounds__(256) void lsr_mel_power_kernel(
    const float* __restrict__ audio, const int64_t* __restrict__ sample_off, const int64_t* __restrict__ frame_off,
    const int32_t* __restrict__ frame_utt, int64_t total_frames, const double* __restrict__ window,
    const double* __restrict__ tw160, const double* __restrict__ tw320, const float* __restrict__ melw,
    const int* __restrict__ mel_lo, const int* __restrict__ mel_hi, float* __restrict__ mel,
    unsigned int* __restrict__ uttmax) {
    __shared__ __attribute__((aligned(16))) cplx buf[MP_FRAMES * 10 * MP_YLD];   // Y, then reused as Z [16][161]
    __shared__ float power[MP_FRAMES][MP_PLD];
    __shared__ unsigned int fmax_bits[MP_FRAMES];
    __shared__ int utt_of[MP_FRAMES];

    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * MP_FRAMES;
    if (tid < MP_FRAMES) {
        fmax_bits[tid] = 0u;
        utt_of[tid] = (g0 + tid < total_frames) ? frame_utt[g0 + tid] : -1;
    }

    // ---- stage A: 10-point DFTs over n1 for fixed (frame, n2) --------------------------------
    {
        const int f = tid >> 4, n2 = tid & 15;
        const int64_t g = g0 + f;
        cplx z[10];
        if (g < total_frames) {
            const int b = frame_utt[g];
            const int64_t s0 = sample_off[b];
            const int64_t ns = sample_off[b + 1] - s0;
            const int64_t first = (int64_t)LSR_HOP * (g - frame_off[b]) - LSR_N_FFT / 2;   // centre padding
            const float* sig = audio + s0;
#pragma unroll
            for (int n1 = 0; n1 < 10; ++n1) {
                const int j = 2 * (16 * n1 + n2);
                const int64_t i0 = first + j, i1 = i0 + 1;
                const double x0 = (i0 >= 0 && i0 < ns) ? (double)sig[i0] : 0.0;
                const double x1 = (i1 >= 0 && i1 < ns) ? (double)sig[i1] : 0.0;
                z[n1] = {window[j] * x0, window[j + 1] * x1};
            }
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 10; ++n1) z[n1] = {0.0, 0.0};
        }
        cplx E[5], O[5];
        dft5(z[0], z[2], z[4], z[6], z[8], E);
        dft5(z[1], z[3], z[5], z[7], z[9], O);
        // W10^k, k = 0..4
        const double wc[5] = {1.0, 0.80901699437494742410, 0.30901699437494742410, -0.30901699437494742410,
                              -0.80901699437494742410};
        const double wsn[5] = {0.0, -0.58778525229247312917, -0.95105651629515357212, -0.95105651629515357212,
                               -0.58778525229247312917};
        const double* tw = tw160 + n2 * 20;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const cplx wo = cmul(O[k], cplx{wc[k], wsn[k]});
            const cplx lo = cadd(E[k], wo), hi = csub(E[k], wo);
            buf[(f * 10 + k) * MP_YLD + n2] = cmul(lo, cplx{tw[2 * k], tw[2 * k + 1]});
            buf[(f * 10 + k + 5) * MP_YLD + n2] = cmul(hi, cplx{tw[2 * (k + 5)], tw[2 * (k + 5) + 1]});
        }
    }
    __syncthreads();

    // ---- stage B: 16-point DFTs over n2 for fixed (frame, k1) ---------------------------------
    {
        cplx y[16], X[16];
        const bool live = tid < MP_FRAMES * 10;
        if (live) {
#pragma unroll
            for (int i = 0; i < 16; ++i) y[i] = buf[tid * MP_YLD + i];
        }
        __syncthreads();                     // every Y row is in registers: the buffer becomes Z
        if (live) {
            const double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178;   // cos, sin(pi/8)
            const double R = 0.70710678118654752440;
            // W16^m = (cos(pi m/8), -sin(pi m/8)) for m = 0,1,2,3,4,6,9
            const cplx w1 = {C1, -S1}, w2 = {R, -R}, w3 = {S1, -C1}, w4 = {0.0, -1.0}, w6 = {-R, -R}, w9 = {-C1, S1};
            cplx u[4][4];                    // u[n2][k1] = DFT4 over n1 of y[4 n1 + n2]
#pragma unroll
            for (int n2 = 0; n2 < 4; ++n2) dft4(y[n2], y[4 + n2], y[8 + n2], y[12 + n2], u[n2]);
            u[1][1] = cmul(u[1][1], w1);
            u[1][2] = cmul(u[1][2], w2);
            u[1][3] = cmul(u[1][3], w3);
            u[2][1] = cmul(u[2][1], w2);
            u[2][2] = cmul(u[2][2], w4);
            u[2][3] = cmul(u[2][3], w6);
            u[3][1] = cmul(u[3][1], w3);
            u[3][2] = cmul(u[3][2], w6);
            u[3][3] = cmul(u[3][3], w9);
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) {
                cplx o[4];
                dft4(u[0][k1], u[1][k1], u[2][k1], u[3][k1], o);
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) X[k1 + 4 * k2] = o[k2];
            }
            const int f = tid / 10, k1 = tid % 10;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) buf[f * MP_ZLD + k1 + 10 * k2] = X[k2];
        }
    }
    __syncthreads();

    // ---- stage C: untangle the real spectrum, magnitude and power in float32 -------------------
    for (int i = tid; i < MP_FRAMES * LSR_N_BINS; i += 256) {
        const int f = i / LSR_N_BINS, k = i % LSR_N_BINS;
        const cplx zk = buf[f * MP_ZLD + (k == 160 ? 0 : k)];
        const cplx zm = buf[f * MP_ZLD + (k == 0 ? 0 : 160 - k)];
        const cplx e = {0.5 * (zk.re + zm.re), 0.5 * (zk.im - zm.im)};
        const cplx d = {0.5 * (zk.re - zm.re), 0.5 * (zk.im + zm.im)};
        const cplx x = cadd(e, cmul(cmul_negi(d), cplx{tw320[2 * k], tw320[2 * k + 1]}));
        const float re = (float)x.re, im = (float)x.im;
        const float mag = (float)sqrt((double)re * (double)re + (double)im * (double)im);
        power[f][k] = mag * mag;
    }
    __syncthreads();

    // ---- stage D: mel filterbank (float64 accumulate, rounded to float32) + utterance maximum ----
    for (int i = tid; i < MP_FRAMES * LSR_N_MELS; i += 256) {
        const int f = i / LSR_N_MELS, band = i % LSR_N_MELS;
        if (g0 + f < total_frames) {
            const float* w = melw + band * LSR_N_BINS;
            double acc = 0.0;
            for (int k = mel_lo[band]; k < mel_hi[band]; ++k) acc = fma((double)w[k], (double)power[f][k], acc);
            const float m = (float)acc;
            mel[(g0 + f) * LSR_N_MELS + band] = m;
            atomicMax(&fmax_bits[f], __float_as_uint(m));
        }
    }
    __syncthreads();
    if (tid < MP_FRAMES && utt_of[tid] >= 0) atomicMax(&uttmax[utt_of[tid]], fmax_bits[tid]);
}

#define FN_FRAMES 64
#define FN_HALO 8
#define FN_STAGED (FN_FRAMES + 2 * FN_HALO)

__global__ __launch_bounds__(256) void lsr_mfcc_finish_kernel(
    const float* __restrict__ mel, const unsigned int* __restrict__ uttmax, const int64_t* __restrict__ frame_off,
    const int32_t* __restrict__ frame_utt, int64_t total_frames, const double* __restrict__ dct,
    float* __restrict__ feats) {
    __shared__ float logmel[FN_STAGED][LSR_N_MELS + 1];
    __shared__ float ceps[FN_STAGED][LSR_N_MFCC];
    __shared__ float outb[FN_FRAMES][LSR_DIM];
    __shared__ double dctm[LSR_N_MFCC * LSR_N_MELS];

    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * FN_FRAMES;
    for (int i = tid; i < LSR_N_MFCC * LSR_N_MELS; i += 256) dctm[i] = dct[i];

    // power_to_db(ref=max, amin=1e-10, top_db=80): the utterance maximum maps to exactly 0 dB,
    // so the floor is the constant -80
    for (int i = tid; i < FN_STAGED * LSR_N_MELS; i += 256) {
        const int fr = i / LSR_N_MELS, band = i % LSR_N_MELS;
        const int64_t g = g0 - FN_HALO + fr;
        float v = 0.f;
        if (g >= 0 && g < total_frames) {
            const float amin = 1e-10f;
            const float ref = __uint_as_float(uttmax[frame_utt[g]]);
            const float m = mel[g * LSR_N_MELS + band];
            const float a = 10.0f * (float)log10((double)fmaxf(amin, m));
            const float r = 10.0f * (float)log10((double)fmaxf(amin, ref));
            v = fmaxf(a - r, -80.0f);
        }
        logmel[fr][band] = v;
    }
    __syncthreads();

    for (int i = tid; i < FN_STAGED * LSR_N_MFCC; i += 256) {
        const int fr = i / LSR_N_MFCC, k = i % LSR_N_MFCC;
        double acc = 0.0;
#pragma unroll 8
        for (int n = 0; n < LSR_N_MELS; ++n) acc = fma(dctm[k * LSR_N_MELS + n], (double)logmel[fr][n], acc);
        ceps[fr][k] = (float)acc;
    }
    __syncthreads();

    // Savitzky-Golay derivative filters of width 9 (polynomial order = derivative order); the first
    // and last four frames of an utterance take the value of the nearest full window.
    for (int i = tid; i < FN_FRAMES * LSR_N_MFCC; i += 256) {
        const int fo = i / LSR_N_MFCC, k = i % LSR_N_MFCC;
        const int64_t g = g0 + fo;
        if (g < total_frames) {
            const int b = frame_utt[g];
            const int64_t lo = frame_off[b];
            const int t = (int)(g - lo), T = (int)(frame_off[b + 1] - lo);
            int tc = t < 4 ? 4 : t;
            if (tc > T - 5) tc = T - 5;
            const int c = fo + FN_HALO + (tc - t);
            const double w2[9] = {28.0 / 462, 7.0 / 462, -8.0 / 462, -17.0 / 462, -20.0 / 462,
                                  -17.0 / 462, -8.0 / 462, 7.0 / 462, 28.0 / 462};
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int j = -4; j <= 4; ++j) {
                const double v = (double)ceps[c + j][k];
                d1 = fma((double)j / 60.0, v, d1);
                d2 = fma(w2[j + 4], v, d2);
            }
            outb[fo][LSR_N_MFCC + k] = (float)d1;
            outb[fo][2 * LSR_N_MFCC + k] = (float)d2;
        }
    }
    // per-frame normalisation across the 13 statics, float32 and in NumPy's order:
    // mean = (c0 + c1 + ... + c12) / 13, std = sqrt(sum((c - mean)^2) / 13), (c - mean) / (std + 1e-8)
    if (tid < FN_FRAMES && g0 + tid < total_frames) {
        const float* c = ceps[tid + FN_HALO];
        float sum = c[0];
#pragma unroll
        for (int k = 1; k < LSR_N_MFCC; ++k) sum = sum + c[k];
        const float mean = sum / 13.0f;
        float dev[LSR_N_MFCC];
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < LSR_N_MFCC; ++k) {
            dev[k] = c[k] - mean;
            const float p = dev[k] * dev[k];
            sq = (k == 0) ? p : sq + p;
        }
        const float sd = sqrtf(sq / 13.0f) + 1e-8f;
#pragma unroll
        for (int k = 0; k < LSR_N_MFCC; ++k) outb[tid][k] = dev[k] / sd;
    }
    __syncthreads();
    const int64_t nvalid = (total_frames - g0 < FN_FRAMES) ? total_frames - g0 : FN_FRAMES;
    const float* flat = &outb[0][0];
    for (int i = tid; i < nvalid * LSR_DIM; i += 256) feats[g0 * LSR_DIM + i] = flat[i];
}

int lsr_launch_mfcc(lsr_ctx* ctx, const float* audio_d, const int64_t* sample_off_d, const int64_t* frame_off_d,
                    int n_utts, int64_t total_frames, float* feats_d) {
    int rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_frame_utt, (size_t)total_frames * sizeof(int32_t), "frame index"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_mel, (size_t)total_frames * LSR_N_MELS * sizeof(float), "mel energies"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_uttmax, (size_t)n_utts * sizeof(unsigned int), "utterance maxima"))) return rc;
    int32_t* frame_utt = (int32_t*)ctx->ws_frame_utt.ptr;
    float* mel = (float*)ctx->ws_mel.ptr;
    unsigned int* uttmax = (unsigned int*)ctx->ws_uttmax.ptr;
    LSR_HIP_CHECK(ctx, hipMemsetAsync(uttmax, 0, (size_t)n_utts * sizeof(unsigned int), ctx->stream));

    lsr_prof_begin(ctx, LSR_K_FRAME_INDEX);
    hipLaunchKernelGGL(lsr_frame_index_kernel, dim3(n_utts), dim3(256), 0, ctx->stream, frame_off_d, frame_utt);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_frame_index_kernel");

    const unsigned blocks_mp = (unsigned)((total_frames + MP_FRAMES - 1) / MP_FRAMES);
    lsr_prof_begin(ctx, LSR_K_MEL_POWER);
    hipLaunchKernelGGL(lsr_mel_power_kernel, dim3(blocks_mp), dim3(256), 0, ctx->stream, audio_d, sample_off_d,
                       frame_off_d, frame_utt, total_frames, ctx->d_window, ctx->d_tw160, ctx->d_tw320, ctx->d_melw,
                       ctx->d_mel_lo, ctx->d_mel_hi, mel, uttmax);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_mel_power_kernel");

    const unsigned blocks_fn = (unsigned)((total_frames + FN_FRAMES - 1) / FN_FRAMES);
    lsr_prof_begin(ctx, LSR_K_MFCC_FINISH);
    hipLaunchKernelGGL(lsr_mfcc_finish_kernel, dim3(blocks_fn), dim3(256), 0, ctx->stream, mel, uttmax, frame_off_d,
                       frame_utt, total_frames, ctx->d_dct, feats_d);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_mfcc_finish_kernel");
    return LSR_OK;
}
