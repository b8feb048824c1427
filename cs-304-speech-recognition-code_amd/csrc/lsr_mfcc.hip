// MFCC front-end: 16 kHz audio -> 39-dimensional features (13 per-frame-normalised cepstra,
// 13 deltas, 13 delta-deltas), reproducing the librosa chain the reference calls
// (melspectrogram n_fft=320 hop=160 40 Slaney mels -> power_to_db(ref=max) -> DCT-II ->
// Savitzky-Golay deltas of width 9 -> per-frame z-normalisation of the statics).
//
// Kernel 1 (frame index): utterance id of every frame of the ragged batch.
// Kernel 2 (mel power): persistent wavefronts (four independent ones per workgroup, sharing the window and
//   twiddle tables in LDS; three per SIMD), four frames per wavefront pass.  The 320-point real
//   FFT is factored 20 x 16 on the real samples n = 16 n1 + n2: 16 lanes per frame each do a 20-point DFT of
//   real input in registers (three radix steps 5, 2, 2; only the outputs k1 = 0..10 are needed, the others are
//   their conjugates), the twiddled results cross LDS once, and 11 lanes per frame each do a 16-point complex
//   DFT whose sixteen outputs are, as they stand or conjugated, sixteen of the bins 0..160 - there is no
//   separate pass that untangles a half-size complex transform.  Window product and FFT are float64 (librosa
//   multiplies by a float64 window and rounds the transform to complex64); magnitude, power and everything
//   after are float32.  The 40 mel energies go to a [frame][40] scratch array and the utterance maximum is
//   kept with atomicMax on the (non-negative) float bit pattern.
// Kernel 3 (finish): one workgroup per 128 frames (+8 frames halo each side): dB with the
//   utterance maximum as reference and the -80 dB floor, DCT (float64 on the matrix unit), deltas
//   (edge frames repeat the polynomial fit of the first/last full window), normalisation,
//   coalesced store.
#include "lsr_internal.h"

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)};
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
    const cplx m1 = {fma(c2, t2.re, fma(c1, t1.re, x0.re)), fma(c2, t2.im, fma(c1, t1.im, x0.im))};
    const cplx m2 = {fma(c1, t2.re, fma(c2, t1.re, x0.re)), fma(c1, t2.im, fma(c2, t1.im, x0.im))};
    const cplx n1 = {fma(s2, t4.re, s1 * t3.re), fma(s2, t4.im, s1 * t3.im)};
    const cplx n2 = {fma(-s1, t4.re, s2 * t3.re), fma(-s1, t4.im, s2 * t3.im)};
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

// Per-frame record for the transform kernel, so that it reaches a frame's samples with one load
// instead of a chain of three: x/y = position in the audio buffer of sample 0 of the centred
// window (may lie before the clip), z = utterance, w = [lo, hi) range of window samples that fall
// inside the clip (lo | hi << 16); the rest of the window is the zero padding.
__global__ void lsr_frame_index_kernel(const int64_t* __restrict__ frame_off, const int64_t* __restrict__ sample_off,
                                       int64_t total_frames, int32_t* __restrict__ frame_utt,
                                       int4* __restrict__ frame_info, unsigned int* __restrict__ uttmax) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) uttmax[b] = 0u;               // the utterance maximum starts at zero (one block per utterance)
    const int64_t lo = frame_off[b];
    int64_t hi = frame_off[b + 1];
    if (hi > total_frames) hi = total_frames;           // the tables hold total_frames entries whatever the offsets say
    if (lo < 0) return;
    // frames before the first and after the last utterance's range (offsets that do not span the batch) get an empty
    // record of utterance 0: nothing downstream reads through a stale entry
    if (b == 0)
        for (int64_t f = threadIdx.x; f < lo && f < total_frames; f += blockDim.x) {
            frame_utt[f] = 0;
            frame_info[f] = make_int4(0, 0, 0, 0);
        }
    if (b == (int)gridDim.x - 1)
        for (int64_t f = (hi < 0 ? 0 : hi) + threadIdx.x; f < total_frames; f += blockDim.x) {
            frame_utt[f] = 0;
            frame_info[f] = make_int4(0, 0, 0, 0);
        }
    const int64_t s0 = sample_off[b], ns = sample_off[b + 1] - s0;
    for (int64_t f = lo + threadIdx.x; f < hi; f += blockDim.x) {
        frame_utt[f] = b;
        const int64_t first = (int64_t)LSR_HOP * (f - lo) - LSR_N_FFT / 2;           // centre padding
        const int64_t start = s0 + first;
        int64_t vlo = first < 0 ? -first : 0, vhi = ns - first;
        if (vlo > LSR_N_FFT) vlo = LSR_N_FFT;
        if (vhi > LSR_N_FFT) vhi = LSR_N_FFT;
        if (vhi < vlo) vhi = vlo;
        frame_info[f] = make_int4((int)(uint32_t)start, (int)(start >> 32), b, (int)(vlo | (vhi << 16)));
    }
}

#define MP_FRAMES 4          // frames per wavefront pass
#define MP_K1 11             // outputs kept of a 20-point real DFT (k1 = 0..10)
#define MP_YLD 17            // complex elements per (frame, k1) row: 16 + 1 pad
#define MP_PLD 164
#define MP_SLOT_BINS 6       // bins per mel slot (61 slots cover the 260 non-zero weights)
#define MP_WAVES 4           // wavefronts per workgroup: independent workers that share the stage tables in LDS

// The load is unconditional (clamped index): a load under a branch is waited for where the branch
// joins, which would stall the pipeline in the iteration that issues it.
__device__ __forceinline__ int4 mp_load_info(const int4* __restrict__ frame_info, int64_t total_frames, int64_t g) {
    int4 r = frame_info[g < total_frames ? g : total_frames - 1];
    if (g >= total_frames) r.w = 0;                                        // empty range: all padding
    return r;
}

// samples 16 n1 + n2 (n1 = 0..19) of the frame described by `info` for one lane; zero outside the clip
template <typename SAMPLE>
__device__ __forceinline__ void mp_load_samples(const SAMPLE* __restrict__ audio, const int4 info, int n2,
                                                float (&xs)[20]) {
    const int lo = info.w & 0xffff, hi = (int)((unsigned)info.w >> 16);
    const int64_t start = (int64_t)(((uint64_t)(uint32_t)info.y << 32) | (uint32_t)info.x);
    const SAMPLE* sig = audio + start + n2;
    if (lo == 0 && hi == LSR_N_FFT) {
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) xs[n1] = (float)sig[16 * n1];
    } else {
        // sample 16 n1 + n2 lies in [lo, hi) exactly when n1 lies in [first, last): two per-lane bounds instead of
        // twenty per-lane sample indices
        const int first = (lo - n2 + 15) >> 4, last = (hi - n2 + 15) >> 4;
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) xs[n1] = (n1 >= first && n1 < last) ? (float)sig[16 * n1] : 0.f;
    }
}

// |X|^2 of the spectrum rounded to complex64.  NumPy squares the float32 magnitude
// (hypotf); re*re + im*im in float32 agrees with that to 2 ulp, far inside the tolerance of
// the dB / DCT chain that follows, and needs no square root.
__device__ __forceinline__ float mp_power(double re64, double im64) {
    const float re = (float)re64, im = (float)im64;
    return fmaf(re, re, im * im);
}

// 5-point DFT of real samples: X0 (real), X1, X2; X3 = conj X2, X4 = conj X1
__device__ __forceinline__ void rdft5(double x0, double x1, double x2, double x3, double x4, double& X0, cplx& X1,
                                      cplx& X2) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;   // cos(2pi/5), cos(4pi/5)
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;    // sin(2pi/5), sin(4pi/5)
    const double t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    X0 = x0 + t1 + t2;
    X1 = {fma(c2, t2, fma(c1, t1, x0)), fma(-s2, t4, -s1 * t3)};
    X2 = {fma(c1, t2, fma(c2, t1, x0)), fma(s1, t4, -s2 * t3)};
}

// 10-point DFT of real samples from the 5-point DFTs of its even (A) and odd (B) samples: E[0..5], E0 and E5 real
struct rspec10 {
    double e0, e5;
    cplx e1, e2, e3, e4;
};
__device__ __forceinline__ rspec10 rdft10(double a0, cplx a1, cplx a2, double b0, cplx b1, cplx b2) {
    rspec10 r;
    r.e0 = a0 + b0;
    r.e5 = a0 - b0;
    const cplx t1 = cmul(b1, cplx{0.80901699437494742410, -0.58778525229247312917});      // W10^1
    const cplx t2 = cmul(b2, cplx{0.30901699437494742410, -0.95105651629515357212});      // W10^2
    r.e1 = cadd(a1, t1);
    r.e4 = {a1.re - t1.re, t1.im - a1.im};                // conj(a1 - t1)
    r.e2 = cadd(a2, t2);
    r.e3 = {a2.re - t2.re, t2.im - a2.im};
    return r;
}

// value of the next higher lane (lane 63 gets 0): DPP wave shift left by one
__device__ __forceinline__ float mp_lane_above(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

// one pass's results: the lane that owns a band writes its energy of each frame (the 40 owners cover a whole row);
// the lanes with utt >= 0 fold their maximum into that utterance's maximum (non-negative floats order like their bit
// patterns)
__device__ __forceinline__ void mp_store_pass(float* __restrict__ mel, unsigned int* __restrict__ uttmax,
                                              int64_t total_frames, int64_t g0, int lane, int band,
                                              const float (&m)[MP_FRAMES], float frame_max, int utt) {
    if (band >= 0) {
#pragma unroll
        for (int fd = 0; fd < MP_FRAMES; ++fd)
            if (g0 + fd < total_frames) mel[(g0 + fd) * LSR_N_MELS + band] = m[fd];
    }
    if (utt >= 0) atomicMax(&uttmax[utt], __float_as_uint(frame_max));      // lanes that report a frame (or the group)
}

// Ordering point between two stages that exchange data through LDS.  A wavefront only touches its own tile and the
// LDS executes one wavefront's instructions in issue order, so a stage's reads may queue right behind the previous
// stage's writes: all that is needed is that the compiler keeps the order.  (A workgroup barrier would first wait
// for every outstanding write to complete, and the reads would only be issued after that.)
__device__ __forceinline__ void mp_stage_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename SAMPLE>
__global__ __launch_bounds__(64 * MP_WAVES, 3) void lsr_mel_power_kernel(
    const SAMPLE* __restrict__ audio, const int4* __restrict__ frame_info, int64_t total_frames, int64_t n_groups,
    const double* __restrict__ window, const double* __restrict__ tw_stage, const float* __restrict__ slot_w,
    const int* __restrict__ slot_info, const int* __restrict__ band_slots, float* __restrict__ mel,
    unsigned int* __restrict__ uttmax) {
    // Y[frame][k1][n2], twiddled, one tile per wavefront.  The power table of a pass lies over the start of the same
    // tile: it is written after the second stage has read its whole input into registers and read before the next
    // pass's first stage writes the tile again, and one wavefront's LDS operations execute in order.
    __shared__ __attribute__((aligned(16))) cplx buf_all[MP_WAVES][MP_FRAMES * MP_K1 * MP_YLD];
    __shared__ __attribute__((aligned(16))) cplx win_s[160];     // window pairs (w[16 (2j) + n2], w[16 (2j + 1) + n2])
    __shared__ __attribute__((aligned(16))) cplx tw_s[10 * 16];  // W320^(n2 k1) as [k1 - 1][n2], k1 = 1..10

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f = lane >> 4, n2 = lane & 15;
    cplx* buf = buf_all[wave];
    float(*power)[MP_PLD] = (float(*)[MP_PLD])buf_all[wave];
    static_assert(sizeof(float) * MP_FRAMES * MP_PLD <= sizeof(cplx) * MP_FRAMES * MP_K1 * MP_YLD, "power table fits the tile");
    const int64_t worker = (int64_t)blockIdx.x * MP_WAVES + wave, n_workers = (int64_t)gridDim.x * MP_WAVES;

    // tables of the first stage live in LDS (read as 16-byte pairs, one bank row per 16 lanes)
    for (int i = threadIdx.x; i < 160; i += 64 * MP_WAVES) {
        const int j = i >> 4, m = i & 15;
        win_s[i] = {window[16 * (2 * j) + m], window[16 * (2 * j + 1) + m]};
        tw_s[i] = {tw_stage[2 * (16 * (j + 1) + m)], tw_stage[2 * (16 * (j + 1) + m) + 1]};
    }
    // The slots of a band are consecutive lanes: the lane of a band's first slot adds the band's 1-3 slot sums
    // (fetched from the lanes above it) and owns the band's energy.
    int my_band = -1, my_count = 0;
    for (int b = 0; b < LSR_N_MELS; ++b)
        if (band_slots[2 * b] == lane) {
            my_band = b;
            my_count = band_slots[2 * b + 1];
        }
    // mel filterbank slot of this lane: up to MP_SLOT_BINS consecutive bins of one band (weights in registers)
    float sw[MP_SLOT_BINS];
#pragma unroll
    for (int j = 0; j < MP_SLOT_BINS; ++j) sw[j] = slot_w[lane * MP_SLOT_BINS + j];
    const int slot_start = slot_info[lane];
    // second stage: lane = (frame, k1) for the first 44 lanes
    const int fb = lane / MP_K1, kb = lane - fb * MP_K1;
    const bool live_b = lane < MP_FRAMES * MP_K1;
    // software pipeline over the grid-stride loop: samples one group ahead, frame records two ahead,
    // so no load is waited for in the iteration that issues it
    float xs[20];
    int4 info_cur = mp_load_info(frame_info, total_frames, worker * MP_FRAMES + f);
    int4 info_nxt = mp_load_info(frame_info, total_frames, (worker + n_workers) * MP_FRAMES + f);
    mp_load_samples(audio, info_cur, n2, xs);
    __syncthreads();

    // Results of a pass (a band's energies of the four frames, the frame maxima) are written during the
    // first stage of the NEXT pass, just before that pass issues its prefetch: the memory counter is
    // in-order and counts stores, so stores issued at the end of a pass would be waited for at the top
    // of the next one together with the prefetched samples.
    float out_m[MP_FRAMES] = {0.f, 0.f, 0.f, 0.f}, out_max = 0.f;
    int out_utt = 0;
    int64_t out_g0 = -1;

    for (int64_t grp = worker; grp < n_groups; grp += n_workers) {
        const int64_t g0 = grp * MP_FRAMES;
        const int utt_cur = info_cur.z;

        // ---- stage A: 20-point DFT of the REAL windowed samples 16 n1 + n2 for fixed (frame, n2); the outputs
        // k1 = 0..10 (the rest are their conjugates) times W320^(n2 k1) go to the transform buffer -------------
        {
            // window pairs and stage twiddles: all twenty LDS reads are in flight before the first use
            cplx wv[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) wv[j] = win_s[16 * j + n2];
            double z[20];
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                z[2 * j] = wv[j].re * (double)xs[2 * j];
                z[2 * j + 1] = wv[j].im * (double)xs[2 * j + 1];
            }
            // The next group's samples travel while this group is transformed.  The loads must be issued
            // after the last read of the current samples: the memory counter is in-order, so a read of
            // `xs` placed after them would wait for the new loads as well.  The empty asm statements pin
            // that order (the products are complete, then the loads start).
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) asm volatile("" : "+v"(z[n1]));
            asm volatile("" ::: "memory");
            // samples 4 c + r, r = 0..3: four 5-point DFTs, two 10-point combinations (even / odd samples), one 20-point
            double a0, b0;
            cplx a1, a2, b1, b2;
            rdft5(z[0], z[4], z[8], z[12], z[16], a0, a1, a2);
            rdft5(z[2], z[6], z[10], z[14], z[18], b0, b1, b2);
            const rspec10 E = rdft10(a0, a1, a2, b0, b1, b2);
            rdft5(z[1], z[5], z[9], z[13], z[17], a0, a1, a2);
            rdft5(z[3], z[7], z[11], z[15], z[19], b0, b1, b2);
            const rspec10 O = rdft10(a0, a1, a2, b0, b1, b2);
            // The stage twiddles W320^(n2 k1) are fetched two at a time, one pair ahead of their use: with three
            // wavefronts per SIMD the registers that all ten would occupy next to E and O are not there.  The empty
            // asm statements keep the compiler from hoisting the reads.
            asm volatile("" ::: "memory");
            const cplx* tw_lane = tw_s + n2;                       // tw_lane[16 (k1 - 1)]
            cplx* col = buf + (f * MP_K1) * MP_YLD + n2;
            cplx ta = tw_lane[16 * 0], tb = tw_lane[16 * 8];       // k1 = 1 and 9
            // Y[k] = E[k] + W20^k O[k], Y[10 - k] = conj(E[k] - W20^k O[k]); Y0, Y10 real, Y5 = E5 - i O5
            col[0] = {E.e0 + O.e0, 0.0};
            const cplx ek[4] = {E.e1, E.e2, E.e3, E.e4}, ok[4] = {O.e1, O.e2, O.e3, O.e4};
            const double wc[4] = {0.95105651629515357212, 0.80901699437494742410, 0.58778525229247312917,
                                  0.30901699437494742410};                                      // cos(k pi / 10)
            const double ws[4] = {-0.30901699437494742410, -0.58778525229247312917, -0.80901699437494742410,
                                  -0.95105651629515357212};                                     // -sin(k pi / 10)
#pragma unroll
            for (int k = 1; k <= 4; ++k) {
                // next pair: k1 = k + 1 and 9 - k; after the last step k1 = 5 and 10
                const cplx na = tw_lane[16 * k], nb = tw_lane[16 * (k < 4 ? 8 - k : 9)];
                asm volatile("" ::: "memory");
                const cplx t = cmul(ok[k - 1], cplx{wc[k - 1], ws[k - 1]});
                col[k * MP_YLD] = cmul(cadd(ek[k - 1], t), ta);
                col[(10 - k) * MP_YLD] = cmul(cplx{ek[k - 1].re - t.re, t.im - ek[k - 1].im}, tb);
                ta = na;
                tb = nb;
            }
            col[5 * MP_YLD] = cmul(cplx{E.e5, -O.e5}, ta);
            {
                const double y10 = E.e0 - O.e0;
                col[10 * MP_YLD] = {y10 * tb.re, y10 * tb.im};
            }
        }
        mp_stage_fence();

        // ---- stage B: 16-point DFTs over n2 for fixed (frame, k1): X[k1 + 20 q], q = 0..15.  Bins up to 160 are
        // wanted: q <= 7 directly (and q = 8 for k1 = 0), the other outputs are the conjugates of the bins
        // 320 - 20 q - k1 (for k1 = 0 and k1 = 10 those repeat bins the lane already has).  The power goes straight
        // to the power table: no second pass over the spectrum -----------------------------------------------
        if (live_b) {
            // the sixteen inputs are read in the order of use (the first 4-point DFT needs elements 0, 4, 8, 12), two
            // groups of four ahead of the arithmetic: reading all sixteen first would cost 64 registers at the point
            // where the results start to pile up.  The empty asm statements keep the compiler from hoisting the reads.
            cplx y[16];
            auto read_group = [&](int m) {
#pragma unroll
                for (int j = 0; j < 4; ++j) y[4 * j + m] = buf[lane * MP_YLD + 4 * j + m];
            };
            read_group(0);
            read_group(1);
            asm volatile("" ::: "memory");
            const double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178;   // cos, sin(pi/8)
            const double R = 0.70710678118654752440;
            // W16^m = (cos(pi m/8), -sin(pi m/8)) for m = 1,2,3,4,6,9
            const cplx w1 = {C1, -S1}, w2 = {R, -R}, w3 = {S1, -C1}, w6 = {-R, -R}, w9 = {-C1, S1};
            cplx u[4][4];                // u[n2][k1] = DFT4 over n1 of y[4 n1 + n2]
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m + 2 < 4) {
                    read_group(m + 2);
                    asm volatile("" ::: "memory");
                }
                dft4(y[m], y[4 + m], y[8 + m], y[12 + m], u[m]);
            }
            u[1][1] = cmul(u[1][1], w1);
            u[1][2] = cmul(u[1][2], w2);
            u[1][3] = cmul(u[1][3], w3);
            u[2][1] = cmul(u[2][1], w2);
            u[2][2] = cmul_negi(u[2][2]);          // W16^4 = -i
            u[2][3] = cmul(u[2][3], w6);
            u[3][1] = cmul(u[3][1], w3);
            u[3][2] = cmul(u[3][2], w6);
            u[3][3] = cmul(u[3][3], w9);
            float* prow = power[fb];
            float* mirror = prow + (20 - kb);        // mirror[20 (15 - q)] = prow[320 - 20 q - kb]: constant offsets >= 0
            const bool inner = kb != 0 && kb != MP_K1 - 1;
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) {
                cplx o[4];
                dft4(u[0][k1], u[1][k1], u[2][k1], u[3][k1], o);
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const int q = k1 + 4 * k2;
                    const float pw = mp_power(o[k2].re, o[k2].im);
                    if (q <= 7)
                        prow[kb + 20 * q] = pw;
                    else if (q == 8 ? kb != MP_K1 - 1 : inner)
                        mirror[20 * (15 - q)] = pw;
                }
            }
        }
        mp_stage_fence();

        // The previous pass's results are stored and the next group's samples requested here, after the second
        // stage: its sixteen complex values per lane are the kernel's register peak, and the twenty sample
        // registers are not part of it this way.  (Stores first: the memory counter is in order.)
        if (out_g0 >= 0) mp_store_pass(mel, uttmax, total_frames, out_g0, lane, my_band, out_m, out_max, out_utt);
        mp_load_samples(audio, info_nxt, n2, xs);
        info_cur = info_nxt;
        info_nxt = mp_load_info(frame_info, total_frames, (grp + 2 * n_workers) * MP_FRAMES + f);

        // ---- stage D: mel filterbank.  Lane = slot (a run of <= 6 bins of one band, weights in registers, float32
        // FMA chain).  A band's slots are consecutive lanes: its first lane adds the 1-3 slot sums in slot order
        // (neighbours through DPP wave shifts, no LDS) and keeps the band's energy; the frame maximum is a wave
        // reduction over those lanes (the energies are non-negative) ------------------------------------------------
        {
            float pv[MP_FRAMES][MP_SLOT_BINS];                 // all reads first, then the FMA chains
#pragma unroll
            for (int fd = 0; fd < MP_FRAMES; ++fd)
#pragma unroll
                for (int j = 0; j < MP_SLOT_BINS; ++j) pv[fd][j] = power[fd][slot_start + j];
            const bool owner = my_band >= 0;
#pragma unroll
            for (int fd = 0; fd < MP_FRAMES; ++fd) {
                float acc = sw[0] * pv[fd][0];
#pragma unroll
                for (int j = 1; j < MP_SLOT_BINS; ++j) acc = fmaf(sw[j], pv[fd][j], acc);
                const float up1 = mp_lane_above(acc), up2 = mp_lane_above(up1);      // slot sums of lanes + 1, + 2
                float m = acc;
                if (my_count > 1) m += up1;
                if (my_count > 2) m += up2;
                out_m[fd] = m;
            }
            // utterance of each frame of the group (held by the first lane of each 16-lane row); in nearly every
            // group the four frames belong to one utterance and a single reduction and atomic serve all four
            const int u0 = __builtin_amdgcn_readlane(utt_cur, 0), u1 = __builtin_amdgcn_readlane(utt_cur, 16);
            const int u2 = __builtin_amdgcn_readlane(utt_cur, 32), u3 = __builtin_amdgcn_readlane(utt_cur, 48);
            if (u0 == u3 && u1 == u0 && u2 == u0 && g0 + MP_FRAMES <= total_frames) {
                const float mm = fmaxf(fmaxf(out_m[0], out_m[1]), fmaxf(out_m[2], out_m[3]));
                out_max = lsr_wave_max_f32(owner ? mm : 0.f);
                out_utt = lane == 0 ? u0 : -1;                 // lane 0 reports for the group
            } else {
#pragma unroll
                for (int fd = 0; fd < MP_FRAMES; ++fd) {
                    const float top = lsr_wave_max_f32(owner && g0 + fd < total_frames ? out_m[fd] : 0.f);
                    if (lane == fd) out_max = top;             // lane fd keeps the maximum of frame fd
                }
                const int u = lane == 0 ? u0 : (lane == 1 ? u1 : (lane == 2 ? u2 : u3));
                out_utt = lane < MP_FRAMES && g0 + lane < total_frames ? u : -1;
            }
            out_g0 = g0;
        }
        mp_stage_fence();                                      // the transform buffer and the power table are reused
    }
    if (out_g0 >= 0) mp_store_pass(mel, uttmax, total_frames, out_g0, lane, my_band, out_m, out_max, out_utt);
}

#define FN_FRAMES 128
#define FN_HALO 8
#define FN_STAGED (FN_FRAMES + 2 * FN_HALO)      // 144 = 9 tiles of 16 frames
#define FN_MEL_STRIDE 44
#define FN_CEPS_LD 14

typedef double fn_v4d __attribute__((ext_vector_type(4)));

// 10 log10(x) for x >= 1e-10 as log2(x) * 10 log10(2): one hardware logarithm and one multiply, within 2 ulp of
// the float32 result NumPy computes (the tolerance of the features is four orders of magnitude wider)
__device__ __forceinline__ float fn_db(float x) { return log2f(x) * 3.01029995663981195f; }

__global__ __launch_bounds__(256) void lsr_mfcc_finish_kernel(
    const float* __restrict__ mel, const unsigned int* __restrict__ uttmax, const int64_t* __restrict__ frame_off,
    const int32_t* __restrict__ frame_utt, int64_t total_frames, const double* __restrict__ dct,
    float* __restrict__ feats) {
    // rows of 44 floats: 16-byte aligned for 128-bit accesses.  The log-mel tile is dead once the cepstra exist;
    // the output staging area reuses its memory.
    __shared__ __attribute__((aligned(16))) float logmel[FN_STAGED][FN_MEL_STRIDE];
    __shared__ float ceps[FN_STAGED][FN_CEPS_LD];
    __shared__ float refdb[FN_STAGED];
    __shared__ int pos_t[FN_FRAMES], len_t[FN_FRAMES];     // frame's index in its utterance, utterance length
    float(*outb)[LSR_DIM] = (float(*)[LSR_DIM]) & logmel[0][0];            // [FN_FRAMES][39] <= the log-mel tile
    static_assert(sizeof(float) * FN_FRAMES * LSR_DIM <= sizeof(float) * FN_STAGED * FN_MEL_STRIDE, "staging fits");

    const int tid = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * FN_FRAMES;

    // Every global load of the block is issued before the first use, so the block pays one memory
    // latency: the staged mel rows are one contiguous run of FN_STAGED * 40 floats (clamped at the
    // ends of the batch), fetched as 128-bit loads.
    constexpr int N_VEC = FN_STAGED * LSR_N_MELS / 4, VEC_PER_THREAD = (N_VEC + 255) / 256;
    float4 mv[VEC_PER_THREAD];
#pragma unroll
    for (int u = 0; u < VEC_PER_THREAD; ++u) {
        const int e = 4 * (tid + 256 * u);
        int64_t g = g0 - FN_HALO + e / LSR_N_MELS;
        g = g < 0 ? 0 : (g >= total_frames ? total_frames - 1 : g);
        mv[u] = *(const float4*)(mel + g * LSR_N_MELS + e % LSR_N_MELS);
    }
    // B operands of the DCT product below: lane (k-group kg = lane >> 4, column n = lane & 15) holds
    // dct[n][4 ks + kg] for the ten k-steps (zero for the three unused columns)
    const int lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, col = lane & 15;
    double bop[LSR_N_MELS / 4];
#pragma unroll
    for (int ks = 0; ks < LSR_N_MELS / 4; ++ks) bop[ks] = col < LSR_N_MFCC ? dct[col * LSR_N_MELS + 4 * ks + kg] : 0.0;

    // power_to_db(ref=max, amin=1e-10, top_db=80): the utterance maximum maps to exactly 0 dB,
    // so the floor is the constant -80
    if (tid < FN_STAGED) {
        const int64_t g = g0 - FN_HALO + tid;
        float r = 0.f;
        if (g >= 0 && g < total_frames) {
            const int b = frame_utt[g];
            r = fn_db(fmaxf(1e-10f, __uint_as_float(uttmax[b])));
            const int fo = tid - FN_HALO;
            if (fo >= 0 && fo < FN_FRAMES) {
                const int64_t lo = frame_off[b];
                pos_t[fo] = (int)(g - lo);
                len_t[fo] = (int)(frame_off[b + 1] - lo);
            }
        }
        refdb[tid] = r;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < VEC_PER_THREAD; ++u) {
        const int e = 4 * (tid + 256 * u);
        if (e < FN_STAGED * LSR_N_MELS) {
            const int fr = e / LSR_N_MELS, band = e % LSR_N_MELS;
            const int64_t g = g0 - FN_HALO + fr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g >= 0 && g < total_frames) {
                const float r = refdb[fr];
                v.x = fmaxf(fn_db(fmaxf(1e-10f, mv[u].x)) - r, -80.0f);
                v.y = fmaxf(fn_db(fmaxf(1e-10f, mv[u].y)) - r, -80.0f);
                v.z = fmaxf(fn_db(fmaxf(1e-10f, mv[u].z)) - r, -80.0f);
                v.w = fmaxf(fn_db(fmaxf(1e-10f, mv[u].w)) - r, -80.0f);
            }
            *(float4*)&logmel[fr][band] = v;
        }
    }
    __syncthreads();

    // DCT-II on the float64 matrix unit: cepstra[16 frames][13] = logmel[16 frames][40] . dct^T, ten
    // v_mfma_f64_16x16x4_f64 per tile of 16 frames (A: lane (kg, m) supplies logmel[frame m][4 ks + kg]; the result
    // register r of lane (kg, n) is coefficient n of frame 4 r + kg).  The wavefronts take the nine tiles in turn.
    for (int tile = wave; tile < FN_STAGED / 16; tile += 4) {
        const float* arow = &logmel[tile * 16 + col][kg];
        double aop[LSR_N_MELS / 4];
#pragma unroll
        for (int ks = 0; ks < LSR_N_MELS / 4; ++ks) aop[ks] = (double)arow[4 * ks];
        fn_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < LSR_N_MELS / 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[ks], bop[ks], acc, 0, 0, 0);
        if (col < LSR_N_MFCC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ceps[tile * 16 + 4 * r + kg][col] = (float)acc[r];
        }
    }
    __syncthreads();

    // Savitzky-Golay derivative filters of width 9 (polynomial order = derivative order); the first
    // and last four frames of an utterance take the value of the nearest full window.
    for (int i = tid; i < FN_FRAMES * LSR_N_MFCC; i += 256) {
        const int fo = i / LSR_N_MFCC, k = i % LSR_N_MFCC;
        const int64_t g = g0 + fo;
        if (g < total_frames) {
            const int t = pos_t[fo], T = len_t[fo];
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

int lsr_launch_mfcc(lsr_ctx* ctx, const void* audio_d, int audio_is_i16, const int64_t* sample_off_d,
                    const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d) {
    int rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_frame_utt, (size_t)total_frames * sizeof(int32_t), "frame index"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_mel, (size_t)total_frames * LSR_N_MELS * sizeof(float), "mel energies"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_uttmax, (size_t)n_utts * sizeof(unsigned int), "utterance maxima"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_frame_info, (size_t)total_frames * sizeof(int4), "frame records"))) return rc;
    lsr_mfcc_ws ws;
    ws.frame_info = ctx->ws_frame_info.ptr;
    ws.frame_utt = (int32_t*)ctx->ws_frame_utt.ptr;
    ws.mel = (float*)ctx->ws_mel.ptr;
    ws.uttmax = (unsigned int*)ctx->ws_uttmax.ptr;
    return lsr_launch_mfcc_ws(ctx, audio_d, audio_is_i16, sample_off_d, frame_off_d, n_utts, total_frames, feats_d, ws);
}

int lsr_launch_mfcc_ws(lsr_ctx* ctx, const void* audio_d, int audio_is_i16, const int64_t* sample_off_d,
                       const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d,
                       const lsr_mfcc_ws& ws) {
    int4* frame_info = (int4*)ws.frame_info;
    int32_t* frame_utt = ws.frame_utt;
    float* mel = ws.mel;
    unsigned int* uttmax = ws.uttmax;
    lsr_prof_begin(ctx, LSR_K_FRAME_INDEX);
    hipLaunchKernelGGL(lsr_frame_index_kernel, dim3(n_utts), dim3(256), 0, ctx->stream, frame_off_d, sample_off_d,
                       total_frames, frame_utt, frame_info, uttmax);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_frame_index_kernel");

    // persistent wavefronts: three workgroups of four per CU (three per SIMD: <= 168 VGPRs, and 11.7 KiB of LDS per
    // wavefront plus 5 KiB of tables per workgroup); each wavefront walks over groups of 4 frames with a grid stride
    const int64_t n_groups = (total_frames + MP_FRAMES - 1) / MP_FRAMES;
    const int64_t wgs_needed = (n_groups + MP_WAVES - 1) / MP_WAVES;
    const unsigned blocks_mp = (unsigned)(wgs_needed < 3 * (int64_t)ctx->n_cus ? wgs_needed : 3 * (int64_t)ctx->n_cus);
    lsr_prof_begin(ctx, LSR_K_MEL_POWER);
    if (audio_is_i16)
        hipLaunchKernelGGL(lsr_mel_power_kernel<int16_t>, dim3(blocks_mp), dim3(64 * MP_WAVES), 0, ctx->stream,
                           (const int16_t*)audio_d, frame_info, total_frames, n_groups,
                           ctx->d_window, ctx->d_tw_stage, ctx->d_slot_w, ctx->d_slot_info, ctx->d_band_slots, mel, uttmax);
    else
        hipLaunchKernelGGL(lsr_mel_power_kernel<float>, dim3(blocks_mp), dim3(64 * MP_WAVES), 0, ctx->stream,
                           (const float*)audio_d, frame_info, total_frames, n_groups,
                           ctx->d_window, ctx->d_tw_stage, ctx->d_slot_w, ctx->d_slot_info, ctx->d_band_slots, mel, uttmax);
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
