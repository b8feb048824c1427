// Sufficient statistics of a hard alignment: for every slot (model state) the frame count,
// the sum of the frames and the sum of their outer products, all in float64.
//
// Frames arrive in utterance order, so a slot persists over runs of consecutive frames.  Each
// workgroup owns a contiguous range of frames; its 256 threads each own up to seven entries of
// the 40 x 40 outer product of the augmented frame [x, 1] (which holds sum_xx, sum_x and the
// count at once), accumulate them in registers while the slot stays the same and add them to
// the workgroup's private slab in global memory when it changes.  A second kernel adds the
// slabs in workgroup order, so the result does not depend on scheduling.
#include "lsr_internal.h"

#define ST_AUG 40
#define ST_ENTRIES (ST_AUG * ST_AUG)
#define ST_PER_THREAD 7
#define ST_CHUNK 64
#define ST_MAX_BLOCKS 256

__global__ __launch_bounds__(256) void lsr_stats_accumulate_kernel(const float* __restrict__ feats,
                                                                   const int32_t* __restrict__ slot,
                                                                   int64_t total_frames, int64_t frames_per_block,
                                                                   int n_slots, double* __restrict__ slabs) {
    __shared__ float xs[ST_CHUNK][ST_AUG];
    __shared__ int sl[ST_CHUNK];
    const int tid = threadIdx.x;
    double* slab = slabs + (size_t)blockIdx.x * n_slots * ST_ENTRIES;
    int ei[ST_PER_THREAD], ej[ST_PER_THREAD];
    double acc[ST_PER_THREAD];
#pragma unroll
    for (int r = 0; r < ST_PER_THREAD; ++r) {
        const int e = tid + 256 * r;
        ei[r] = e < ST_ENTRIES ? e / ST_AUG : 0;
        ej[r] = e < ST_ENTRIES ? e % ST_AUG : 0;
        acc[r] = 0.0;
    }
    const int64_t lo = (int64_t)blockIdx.x * frames_per_block;
    int64_t hi = lo + frames_per_block;
    if (hi > total_frames) hi = total_frames;
    int cur = -1;
    for (int64_t c0 = lo; c0 < hi; c0 += ST_CHUNK) {
        const int nf = (int)((hi - c0 < ST_CHUNK) ? hi - c0 : ST_CHUNK);
        __syncthreads();
        for (int i = tid; i < nf * LSR_DIM; i += 256) xs[i / LSR_DIM][i % LSR_DIM] = feats[c0 * LSR_DIM + i];
        if (tid < nf) {
            xs[tid][LSR_DIM] = 1.0f;
            const int s = slot[c0 + tid];
            sl[tid] = (s >= 0 && s < n_slots) ? s : -1;
        }
        __syncthreads();
        for (int f = 0; f < nf; ++f) {
            const int s = sl[f];
            if (s != cur) {
                if (cur >= 0) {
#pragma unroll
                    for (int r = 0; r < ST_PER_THREAD; ++r) {
                        const int e = tid + 256 * r;
                        if (e < ST_ENTRIES) slab[(size_t)cur * ST_ENTRIES + e] += acc[r];
                        acc[r] = 0.0;
                    }
                }
                cur = s;
            }
            if (s >= 0) {
#pragma unroll
                for (int r = 0; r < ST_PER_THREAD; ++r)
                    acc[r] = fma((double)xs[f][ei[r]], (double)xs[f][ej[r]], acc[r]);
            }
        }
    }
    if (cur >= 0) {
#pragma unroll
        for (int r = 0; r < ST_PER_THREAD; ++r) {
            const int e = tid + 256 * r;
            if (e < ST_ENTRIES) slab[(size_t)cur * ST_ENTRIES + e] += acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void lsr_stats_reduce_kernel(const double* __restrict__ slabs, int n_blocks,
                                                               int n_slots, double* __restrict__ count,
                                                               double* __restrict__ sum_x,
                                                               double* __restrict__ sum_xx) {
    const int s = blockIdx.x;
    for (int e = threadIdx.x; e < ST_ENTRIES; e += 256) {
        double t = 0.0;
        for (int b = 0; b < n_blocks; ++b) t += slabs[((size_t)b * n_slots + s) * ST_ENTRIES + e];
        const int i = e / ST_AUG, j = e % ST_AUG;
        if (i < LSR_DIM && j < LSR_DIM)
            sum_xx[((size_t)s * LSR_DIM + i) * LSR_DIM + j] = t;
        else if (i == LSR_DIM && j < LSR_DIM)
            sum_x[(size_t)s * LSR_DIM + j] = t;
        else if (i == LSR_DIM && j == LSR_DIM)
            count[s] = t;
    }
}

int lsr_launch_stats(lsr_ctx* ctx, const float* feats_d, const int32_t* slot_d, int64_t total_frames, int n_slots,
                     double* count_d, double* sum_x_d, double* sum_xx_d) {
    int64_t blocks = (total_frames + 1023) / 1024;
    if (blocks > ST_MAX_BLOCKS) blocks = ST_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    int64_t per_block = (total_frames + blocks - 1) / blocks;
    per_block = (per_block + ST_CHUNK - 1) / ST_CHUNK * ST_CHUNK;
    const size_t slab_bytes = (size_t)blocks * n_slots * ST_ENTRIES * sizeof(double);
    int rc = lsr_reserve(ctx, &ctx->ws_slab, slab_bytes, "statistics slabs");
    if (rc) return rc;
    double* slabs = (double*)ctx->ws_slab.ptr;
    LSR_HIP_CHECK(ctx, hipMemsetAsync(slabs, 0, slab_bytes, ctx->stream));
    lsr_prof_begin(ctx, LSR_K_STATS);
    hipLaunchKernelGGL(lsr_stats_accumulate_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, feats_d, slot_d,
                       total_frames, per_block, n_slots, slabs);
    LSR_LAUNCH_CHECK(ctx, "lsr_stats_accumulate_kernel");
    hipLaunchKernelGGL(lsr_stats_reduce_kernel, dim3((unsigned)n_slots), dim3(256), 0, ctx->stream, slabs, (int)blocks,
                       n_slots, count_d, sum_x_d, sum_xx_d);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_stats_reduce_kernel");
    return LSR_OK;
}

// ---- path -> statistics slots and transition counts (integer work, one thread per frame) -------------
__global__ void lsr_utt_index_kernel(const int64_t* __restrict__ frame_off, int32_t* __restrict__ frame_utt) {
    const int b = blockIdx.x;
    const int64_t lo = frame_off[b], hi = frame_off[b + 1];
    for (int64_t f = lo + threadIdx.x; f < hi; f += blockDim.x) frame_utt[f] = b;
}

int lsr_launch_utt_index(lsr_ctx* ctx, const int64_t* off_d, int n_utts, int32_t* out_d) {
    hipLaunchKernelGGL(lsr_utt_index_kernel, dim3(n_utts), dim3(256), 0, ctx->stream, off_d, out_d);
    LSR_LAUNCH_CHECK(ctx, "lsr_utt_index_kernel");
    return LSR_OK;
}

__global__ __launch_bounds__(256) void lsr_remux_kernel(const int8_t* __restrict__ paths,
                                                        const int64_t* __restrict__ frame_off,
                                                        const int64_t* __restrict__ chain_off,
                                                        const int32_t* __restrict__ chain_state,
                                                        const int32_t* __restrict__ chain_word,
                                                        const int32_t* __restrict__ frame_utt, int64_t total_frames,
                                                        int n_states, int drop_last_run, int32_t* __restrict__ slots,
                                                        unsigned long long* __restrict__ trans) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= total_frames) return;
    const int b = frame_utt[f];
    const int64_t c0 = chain_off[b], n_chain = chain_off[b + 1] - c0;
    const int64_t last = frame_off[b + 1] - 1;
    const int p = paths[f];
    if (p < 0 || p >= n_chain) {        // only a one-frame utterance has no back-pointer (-1)
        slots[f] = -1;
        return;
    }
    const int pl = paths[last];
    const int w_last = (pl >= 0 && pl < n_chain) ? chain_word[c0 + pl] : -1;
    const int g = chain_state[c0 + p], w = chain_word[c0 + p];
    const bool keep = !drop_last_run || w != w_last;
    slots[f] = (keep && g >= 0 && g < n_states) ? g : -1;
    if (keep && f < last) {
        const int p1 = paths[f + 1];
        if (p1 >= 0 && p1 < n_chain) {
            const int g1 = chain_state[c0 + p1], w1 = chain_word[c0 + p1];
            const bool keep1 = !drop_last_run || w1 != w_last;
            if (keep1 && w1 == w && g >= 0 && g < n_states && g1 >= 0 && g1 < n_states)
                atomicAdd(&trans[(size_t)g * n_states + g1], 1ull);
        }
    }
}

extern "C" int lsr_remux_slots(lsr_ctx* ctx, const int8_t* paths_d, const int64_t* frame_off_d,
                               const int64_t* chain_off_d, const int32_t* chain_state_d, const int32_t* chain_word_d,
                               int n_utts, int64_t total_frames, int n_states, int drop_last_run, int32_t* slots_d,
                               int64_t* trans_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!paths_d || !frame_off_d || !chain_off_d || !chain_state_d || !chain_word_d || !slots_d || !trans_d)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_remux_slots: null pointer");
    if (n_utts < 1 || total_frames < n_utts || n_states < 1 || n_states > 4096)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_remux_slots: bad sizes");
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = lsr_reserve(ctx, &ctx->ws_frame_utt, (size_t)total_frames * sizeof(int32_t), "frame index");
    if (rc) return rc;
    int32_t* frame_utt = (int32_t*)ctx->ws_frame_utt.ptr;
    LSR_HIP_CHECK(ctx, hipMemsetAsync(trans_d, 0, (size_t)n_states * n_states * sizeof(int64_t), ctx->stream));
    hipLaunchKernelGGL(lsr_utt_index_kernel, dim3(n_utts), dim3(256), 0, ctx->stream, frame_off_d, frame_utt);
    LSR_LAUNCH_CHECK(ctx, "lsr_utt_index_kernel");
    hipLaunchKernelGGL(lsr_remux_kernel, dim3((unsigned)((total_frames + 255) / 256)), dim3(256), 0, ctx->stream, paths_d,
                       frame_off_d, chain_off_d, chain_state_d, chain_word_d, frame_utt, total_frames, n_states,
                       drop_last_run, slots_d, (unsigned long long*)trans_d);
    LSR_LAUNCH_CHECK(ctx, "lsr_remux_kernel");
    return LSR_OK;
}
