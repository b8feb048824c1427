// Energy-based silence stripping of a ragged batch of clips (the reference's SignalSeparation,
// src/loe_speech_recognition/signal_separation.py:88-164).
//
// Pass 1 (HBM-bound, every sample read once): the float32 mean of |x| and the maximum of |x| of
// every scan frame.  The mean must round exactly as NumPy's float32 mean does, because it is
// compared against thresholds: NumPy adds a contiguous float32 block of up to 128 elements with
// eight strided partial sums r[j] += a[8 i + j], folds them as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
// adds the n % 8 trailing elements one by one, and splits longer blocks in two halves (the first
// rounded down to a multiple of 8) whose sums it adds.  Eight lanes hold the eight partial sums of
// a block, sixteen lanes own a frame of up to 256 samples (two blocks), and the fold is an
// xor-butterfly (float addition commutes, so every lane ends with the same bits).
//
// Pass 2 (one thread per clip): the clip maximum, the two float32 thresholds and the sequential
// two-threshold state machine over the clip's frames.
#include "lsr_internal.h"

#define SEP_MAX_FRAME 256

template <typename SAMPLE>
__device__ __forceinline__ float sep_abs(const SAMPLE* p, int64_t i) { return fabsf((float)p[i]); }

template <typename SAMPLE>
__global__ __launch_bounds__(256) void lsr_sep_energy_kernel(const SAMPLE* __restrict__ audio,
                                                             const int64_t* __restrict__ sample_off,
                                                             const int64_t* __restrict__ scan_off,
                                                             const int32_t* __restrict__ frame_utt,
                                                             int64_t total_scan_frames, int frame_size,
                                                             float* __restrict__ energy, float* __restrict__ fmax) {
    const int lane16 = threadIdx.x & 15;
    const int leaf = lane16 >> 3, j = lane16 & 7;
    int64_t k = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = k < total_scan_frames;
    if (!live) k = total_scan_frames - 1;                 // keep every lane in the shuffles below
    const int b = frame_utt[k];
    const int64_t start = sample_off[b] + (k - scan_off[b]) * frame_size;
    int64_t rest = sample_off[b + 1] - start;
    const int len = (int)(rest < frame_size ? rest : frame_size);
    // the one or two blocks NumPy's pairwise sum cuts the frame into
    int n_a = len, n_b = 0;
    if (len > 128) {
        n_a = len / 2;
        n_a -= n_a % 8;
        n_b = len - n_a;
    }
    const SAMPLE* p = audio + start + (leaf ? n_a : 0);
    const int n = leaf ? n_b : n_a;
    const int n8 = n - n % 8;
    float r = 0.0f, mx = 0.0f, seq = 0.0f;
    if (n >= 8) {
        r = sep_abs(p, j);
        mx = r;
        for (int i = 8; i < n8; i += 8) {
            const float v = sep_abs(p, i + j);
            r += v;
            mx = fmaxf(mx, v);
        }
    } else {
        for (int i = 0; i < n; ++i) {                     // short block: plain left-to-right sum
            const float v = sep_abs(p, i);
            seq += v;
            mx = fmaxf(mx, v);
        }
    }
    r += __shfl_xor(r, 1);
    r += __shfl_xor(r, 2);
    r += __shfl_xor(r, 4);
    if (n >= 8) {
        for (int i = n8; i < n; ++i) {                    // trailing n % 8 elements
            const float v = sep_abs(p, i);
            r += v;
            mx = fmaxf(mx, v);
        }
    } else {
        r = seq;
    }
    const float other = __shfl_xor(r, 8);                 // the other block's sum
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    mx = fmaxf(mx, __shfl_xor(mx, 4));
    mx = fmaxf(mx, __shfl_xor(mx, 8));
    if (live && lane16 == 0) {
        const float total = (len > 128) ? r + other : r;
        energy[k] = total / (float)len;                   // 0/0 = NaN for an empty trailing frame, as NumPy
        fmax[k] = mx;
    }
}

__global__ __launch_bounds__(64) void lsr_sep_scan_kernel(const float* __restrict__ energy,
                                                          const float* __restrict__ fmax,
                                                          const int64_t* __restrict__ scan_off, int n_utts,
                                                          double high, double low, int max_silence_frames,
                                                          int min_kept_frames, int32_t* __restrict__ first_out,
                                                          int32_t* __restrict__ end_out, int32_t* __restrict__ status_out,
                                                          uint8_t* __restrict__ noise, float* __restrict__ max_volume) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_utts) return;
    const int64_t lo = scan_off[b], hi = scan_off[b + 1];
    float mx = 0.0f;
    for (int64_t k = lo; k < hi; ++k) mx = fmaxf(mx, fmax[k]);
    // threshold * max_volume is an np.float64 (the maximum is converted with .astype(float)), so NumPy promotes
    // the float32 energy and compares in float64; NaN (the empty trailing frame) compares false either way
    const double thr_high = high * (double)mx;
    const double thr_low = low * (double)mx;
    bool between = false, ever = false;
    int counter = 0, first = -1, end = -1;
    for (int64_t k = lo; k < hi; ++k) {
        const double e = (double)energy[k];
        bool tripped = false;
        if (between) {
            if (e > thr_low) {
                counter = 0;
            } else {
                between = false;
                tripped = ++counter >= max_silence_frames;
            }
        } else if (e > thr_high) {
            between = true;
            if (!ever) first = (int)(k - lo);
            ever = true;
            counter = 0;
        } else {
            noise[k] = 1;
            if (ever) tripped = ++counter >= max_silence_frames;
        }
        if (tripped) {
            end = (int)(k - lo);
            break;
        }
    }
    first_out[b] = first;
    end_out[b] = end;
    status_out[b] = end < 0 ? 1 : (end - first < min_kept_frames ? 2 : 0);
    if (max_volume) max_volume[b] = mx;
}

template <typename WORD>
__global__ __launch_bounds__(256) void lsr_gather_kernel(const WORD* __restrict__ src,
                                                         const int64_t* __restrict__ src_start,
                                                         const int64_t* __restrict__ dst_off, WORD* __restrict__ dst) {
    const int s = blockIdx.x;
    const int64_t from = src_start[s], to = dst_off[s], n = dst_off[s + 1] - to;
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.y * 256)
        dst[to + i] = src[from + i];
}

extern "C" int lsr_separation_scan(lsr_ctx* ctx, const void* audio_d, int audio_is_i16, const int64_t* sample_off_d,
                                   const int64_t* scan_off_d, int n_utts, int64_t total_scan_frames, int frame_size,
                                   double high, double low, int max_silence_frames, int min_kept_frames,
                                   int32_t* first_d, int32_t* end_d, int32_t* status_d, uint8_t* noise_d,
                                   float* energy_d, float* max_volume_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!audio_d || !sample_off_d || !scan_off_d || !first_d || !end_d || !status_d || !noise_d)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_separation_scan: null pointer");
    if (n_utts < 1 || total_scan_frames < n_utts || max_silence_frames < 0 || min_kept_frames < 0)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_separation_scan: bad sizes");
    if (frame_size < 1 || frame_size > SEP_MAX_FRAME)
        return lsr_fail(ctx, LSR_ERR_LIMIT, "lsr_separation_scan: frame_size %d outside 1..%d", frame_size,
                        SEP_MAX_FRAME);
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = lsr_reserve(ctx, &ctx->ws_frame_utt, (size_t)total_scan_frames * sizeof(int32_t), "frame index");
    if (rc) return rc;
    rc = lsr_reserve(ctx, &ctx->ws_sep, (size_t)total_scan_frames * 2 * sizeof(float), "frame energies");
    if (rc) return rc;
    int32_t* frame_utt = (int32_t*)ctx->ws_frame_utt.ptr;
    float* fmax = (float*)ctx->ws_sep.ptr;
    float* energy = energy_d ? energy_d : fmax + total_scan_frames;
    rc = lsr_launch_utt_index(ctx, scan_off_d, n_utts, total_scan_frames, frame_utt);
    if (rc) return rc;
    LSR_HIP_CHECK(ctx, hipMemsetAsync(noise_d, 0, (size_t)total_scan_frames, ctx->stream));
    const dim3 grid((unsigned)((total_scan_frames + 15) / 16));
    lsr_prof_begin(ctx, LSR_K_SEPARATION);
    if (audio_is_i16)
        hipLaunchKernelGGL(lsr_sep_energy_kernel<int16_t>, grid, dim3(256), 0, ctx->stream, (const int16_t*)audio_d,
                           sample_off_d, scan_off_d, frame_utt, total_scan_frames, frame_size, energy, fmax);
    else
        hipLaunchKernelGGL(lsr_sep_energy_kernel<float>, grid, dim3(256), 0, ctx->stream, (const float*)audio_d,
                           sample_off_d, scan_off_d, frame_utt, total_scan_frames, frame_size, energy, fmax);
    LSR_LAUNCH_CHECK(ctx, "lsr_sep_energy_kernel");
    hipLaunchKernelGGL(lsr_sep_scan_kernel, dim3((unsigned)((n_utts + 63) / 64)), dim3(64), 0, ctx->stream, energy, fmax,
                       scan_off_d, n_utts, high, low, max_silence_frames, min_kept_frames, first_d, end_d, status_d,
                       noise_d, max_volume_d);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_sep_scan_kernel");
    return LSR_OK;
}

extern "C" int lsr_gather_segments(lsr_ctx* ctx, const void* src_d, int elem_bytes, const int64_t* src_start_d,
                                   const int64_t* dst_off_d, int n_segments, int64_t max_segment, void* dst_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!src_d || !src_start_d || !dst_off_d || !dst_d)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_gather_segments: null pointer");
    if (n_segments < 1 || max_segment < 0 || (elem_bytes != 2 && elem_bytes != 4))
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_gather_segments: bad sizes");
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int64_t chunks = (max_segment + 4095) / 4096;         // 16 elements per thread
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    const dim3 grid((unsigned)n_segments, (unsigned)chunks);
    if (elem_bytes == 2)
        hipLaunchKernelGGL(lsr_gather_kernel<uint16_t>, grid, dim3(256), 0, ctx->stream, (const uint16_t*)src_d,
                           src_start_d, dst_off_d, (uint16_t*)dst_d);
    else
        hipLaunchKernelGGL(lsr_gather_kernel<uint32_t>, grid, dim3(256), 0, ctx->stream, (const uint32_t*)src_d,
                           src_start_d, dst_off_d, (uint32_t*)dst_d);
    LSR_LAUNCH_CHECK(ctx, "lsr_gather_kernel");
    return LSR_OK;
}
