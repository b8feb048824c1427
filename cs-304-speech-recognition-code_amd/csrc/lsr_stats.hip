// Sufficient statistics of a hard alignment: for every slot (model state) the frame count,
// the sum of the frames and the sum of their outer products, all in float64.
//
// The sums of one slot are the Gram matrix X^T X of its frames augmented with a constant 1
// ([x, 1]: 40 columns, holding sum_xx, sum_x and the count at once), which is matrix-unit work:
//   1. a stable counting sort of the frame indices by slot (per-wavefront histograms of 1024-frame
//      units, a scan over units, a scatter that ranks equal slots inside a wavefront by lane),
//   2. one wavefront per (slot, part of the slot's sorted list): v_mfma_f64_16x16x4_f64 with four
//      frames per k-step; lane (g, c) loads features c, 16 + c, 32 + c of frame 4 ks + g, and the same
//      register is the A operand (X^T) of its row tile and the B operand (X) of its column tile, so
//      a k-step is 3 gathers and 6 MFMAs (upper triangle of the 3 x 3 tiles),
//   3. a reduction that adds the parts of a slot in part order and unpacks the tiles.
// Every step has a fixed order of additions, so the result does not depend on scheduling.
#include "lsr_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define SS_UNIT 1024         // frames per wavefront in the sort kernels
#define SS_TILES 6           // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
#define SS_AUG 40            // 39 features + the constant

__global__ __launch_bounds__(64) void lsr_slot_count_kernel(const int32_t* __restrict__ slot, int64_t total_frames,
                                                            int n_slots, int32_t* __restrict__ counts) {
    extern __shared__ int cnt[];
    const int lane = threadIdx.x;
    for (int i = lane; i < n_slots; i += 64) cnt[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SS_UNIT;
    for (int it = 0; it < SS_UNIT / 64; ++it) {
        const int64_t f = base + it * 64 + lane;
        const int s = f < total_frames ? slot[f] : -1;
        if (s >= 0 && s < n_slots) atomicAdd(&cnt[s], 1);
    }
    __syncthreads();
    for (int i = lane; i < n_slots; i += 64) counts[(size_t)i * gridDim.x + blockIdx.x] = cnt[i];      // [slot][unit]
}

// exclusive scan of `n` ints in place with one 256-thread workgroup (contiguous chunk per thread);
// returns the total in every thread
__device__ int ss_block_scan(int32_t* __restrict__ v, int n) {
    __shared__ int part[256];
    const int t = threadIdx.x;
    const int chunk = (n + 255) / 256, lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += v[i];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                  // inclusive Hillis-Steele scan of the chunk sums
        const int add = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int run = part[t] - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = v[i];
        v[i] = run;
        run += c;
    }
    const int total = part[255];
    __syncthreads();
    return total;
}

// counts[slot][unit] -> position, relative to the slot's segment, of the unit's first frame of
// that slot in the sorted list (one workgroup per slot); slot_n[slot] = length of the segment
__global__ __launch_bounds__(256) void lsr_slot_scan_kernel(int32_t* __restrict__ counts, int n_units,
                                                            int32_t* __restrict__ slot_n) {
    const int total = ss_block_scan(counts + (size_t)blockIdx.x * n_units, n_units);
    if (threadIdx.x == 0) slot_n[blockIdx.x] = total;
}

// slot_base[slot] = start of the slot's segment
__global__ __launch_bounds__(256) void lsr_slot_base_kernel(const int32_t* __restrict__ slot_n, int n_slots,
                                                            int32_t* __restrict__ slot_base) {
    for (int i = threadIdx.x; i < n_slots; i += 256) slot_base[i] = slot_n[i];
    __syncthreads();
    ss_block_scan(slot_base, n_slots);
}

__global__ __launch_bounds__(64) void lsr_slot_scatter_kernel(const int32_t* __restrict__ slot, int64_t total_frames,
                                                              int n_slots, const int32_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ slot_base,
                                                              int32_t* __restrict__ order) {
    extern __shared__ int cnt[];
    const int lane = threadIdx.x;
    for (int i = lane; i < n_slots; i += 64) cnt[i] = slot_base[i] + offsets[(size_t)i * gridDim.x + blockIdx.x];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SS_UNIT;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int it = 0; it < SS_UNIT / 64; ++it) {
        const int64_t f = base + it * 64 + lane;
        int s = f < total_frames ? slot[f] : -1;
        if (s >= n_slots) s = -1;
        bool todo = s >= 0;
        // frames of equal slot keep their order: rank by lane inside the wavefront
        while (true) {
            const unsigned long long pending = __ballot(todo);
            if (pending == 0ull) break;
            const int lead = __shfl(s, __ffsll((long long)pending) - 1);
            const unsigned long long same = __ballot(todo && s == lead);
            const int start = cnt[lead];
            __syncthreads();                                     // every lane has read the counter
            if (todo && s == lead) {
                order[start + __popcll(same & below)] = (int32_t)f;
                todo = false;
            }
            if (lane == 0) cnt[lead] = start + __popcll(same);
            __syncthreads();
        }
    }
}

__device__ __forceinline__ void ss_load(const float* __restrict__ feats, const int32_t* __restrict__ order, int pos,
                                        int end, int c, double (&x)[3]) {
    x[0] = x[1] = x[2] = 0.0;
    if (pos < end) {
        const float* row = feats + (size_t)order[pos] * LSR_DIM;
        x[0] = (double)row[c];
        x[1] = (double)row[16 + c];
        x[2] = c < 7 ? (double)row[32 + c] : (c == 7 ? 1.0 : 0.0);       // column 39 is the constant
    }
}

__global__ __launch_bounds__(64) void lsr_gram_kernel(const float* __restrict__ feats, const int32_t* __restrict__ order,
                                                      const int32_t* __restrict__ slot_base,
                                                      const int32_t* __restrict__ slot_n, int n_parts,
                                                      v4d* __restrict__ partial) {
    const int s = blockIdx.x / n_parts, part = blockIdx.x % n_parts;
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int n = slot_n[s], base = slot_base[s];
    const int n_ks = (n + 3) / 4;
    const int ks0 = (int)((int64_t)part * n_ks / n_parts), ks1 = (int)((int64_t)(part + 1) * n_ks / n_parts);
    const int end = base + n;
    v4d acc[SS_TILES];
#pragma unroll
    for (int t = 0; t < SS_TILES; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int ks = ks0; ks < ks1; ks += 4) {
        double x[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = ks + u;
            ss_load(feats, order, k < ks1 ? base + 4 * k + g : end, end, c, x[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][0], x[u][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][0], x[u][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][0], x[u][2], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][1], x[u][1], acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][1], x[u][2], acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u][2], x[u][2], acc[5], 0, 0, 0);
        }
    }
    v4d* out = partial + ((size_t)blockIdx.x * SS_TILES) * 64 + lane;
#pragma unroll
    for (int t = 0; t < SS_TILES; ++t) out[t * 64] = acc[t];
}

// entry (i, j), i <= j, of the augmented Gram matrix lives in tile (i / 16, j / 16) at lane
// (i % 4) * 16 + j % 16, register (i % 16) / 4
__global__ __launch_bounds__(256) void lsr_gram_reduce_kernel(const double* __restrict__ partial, int n_parts,
                                                              double* __restrict__ count, double* __restrict__ sum_x,
                                                              double* __restrict__ sum_xx) {
    const int s = blockIdx.x;
    for (int e = threadIdx.x; e < SS_AUG * SS_AUG; e += 256) {
        int i = e / SS_AUG, j = e % SS_AUG;
        const int row = i, colm = j;
        if (i > j) {
            const int t = i;
            i = j;
            j = t;
        }
        const int ti = i / 16, tj = j / 16;
        const int tile = ti == 0 ? tj : (ti == 1 ? 2 + tj : 5);
        const int lane = (i % 4) * 16 + j % 16, reg = (i % 16) / 4;
        double t = 0.0;
        for (int p = 0; p < n_parts; ++p)
            t += partial[((((size_t)s * n_parts + p) * SS_TILES + tile) * 64 + lane) * 4 + reg];
        if (row < LSR_DIM && colm < LSR_DIM)
            sum_xx[((size_t)s * LSR_DIM + row) * LSR_DIM + colm] = t;
        else if (row == LSR_DIM && colm < LSR_DIM)
            sum_x[(size_t)s * LSR_DIM + colm] = t;
        else if (row == LSR_DIM && colm == LSR_DIM)
            count[s] = t;
    }
}

int lsr_launch_stats(lsr_ctx* ctx, const float* feats_d, const int32_t* slot_d, int64_t total_frames, int n_slots,
                     double* count_d, double* sum_x_d, double* sum_xx_d) {
    const int n_units = (int)((total_frames + SS_UNIT - 1) / SS_UNIT);
    int n_parts = 4096 / n_slots;
    if (n_parts < 1) n_parts = 1;
    if (n_parts > 256) n_parts = 256;
    // sort workspace: counts/offsets [slots][units], slot_base, slot_n, order [frames]
    const size_t n_counts = (size_t)n_units * n_slots;
    const size_t sort_ints = n_counts + 2 * (size_t)n_slots + (size_t)total_frames;
    int rc = lsr_reserve(ctx, &ctx->ws_sort, sort_ints * sizeof(int32_t), "slot sort");
    if (rc) return rc;
    const size_t partial_bytes = (size_t)n_slots * n_parts * SS_TILES * 64 * sizeof(v4d);
    if ((rc = lsr_reserve(ctx, &ctx->ws_slab, partial_bytes, "statistics partial sums"))) return rc;
    int32_t* counts = (int32_t*)ctx->ws_sort.ptr;
    int32_t* slot_base = counts + n_counts;
    int32_t* slot_n = slot_base + n_slots;
    int32_t* order = slot_n + n_slots;
    v4d* partial = (v4d*)ctx->ws_slab.ptr;
    const size_t lds = (size_t)n_slots * sizeof(int);
    lsr_prof_begin(ctx, LSR_K_STATS);
    hipLaunchKernelGGL(lsr_slot_count_kernel, dim3(n_units), dim3(64), lds, ctx->stream, slot_d, total_frames, n_slots,
                       counts);
    LSR_LAUNCH_CHECK(ctx, "lsr_slot_count_kernel");
    hipLaunchKernelGGL(lsr_slot_scan_kernel, dim3(n_slots), dim3(256), 0, ctx->stream, counts, n_units, slot_n);
    LSR_LAUNCH_CHECK(ctx, "lsr_slot_scan_kernel");
    hipLaunchKernelGGL(lsr_slot_base_kernel, dim3(1), dim3(256), 0, ctx->stream, slot_n, n_slots, slot_base);
    LSR_LAUNCH_CHECK(ctx, "lsr_slot_base_kernel");
    hipLaunchKernelGGL(lsr_slot_scatter_kernel, dim3(n_units), dim3(64), lds, ctx->stream, slot_d, total_frames, n_slots,
                       counts, slot_base, order);
    LSR_LAUNCH_CHECK(ctx, "lsr_slot_scatter_kernel");
    hipLaunchKernelGGL(lsr_gram_kernel, dim3((unsigned)(n_slots * n_parts)), dim3(64), 0, ctx->stream, feats_d, order,
                       slot_base, slot_n, n_parts, partial);
    LSR_LAUNCH_CHECK(ctx, "lsr_gram_kernel");
    hipLaunchKernelGGL(lsr_gram_reduce_kernel, dim3((unsigned)n_slots), dim3(256), 0, ctx->stream,
                       (const double*)partial, n_parts, count_d, sum_x_d, sum_xx_d);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_gram_reduce_kernel");
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
