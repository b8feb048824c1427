// Sufficient statistics of a hard alignment: for every slot (model state) the frame count,
// the sum of the frames and the sum of their outer products, all in float64.
//
// The sums of one slot are the Gram matrix X^T X of its frames augmented with a constant 1
// ([x, 1]: 40 columns, holding sum_xx, sum_x and the count at once), which is matrix-unit work:
//   1. a stable counting sort of the frame indices by slot (per-wavefront histograms of 1024-frame
//      units, a scan over units, a scatter that ranks equal slots inside a wavefront by lane),
//   2. one wavefront per part (up to 512 consecutive frames of a slot's sorted list, so that a slot with many
//      frames - silence - gets proportionally many wavefronts): v_mfma_f64_16x16x4_f64 with four
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
#define SS_CHUNK 512         // frames per wavefront of the Gram kernel (a multiple of 16)

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

// slot_base[slot] = start of the slot's segment; part_base[slot] = first part (chunk of SS_CHUNK frames) of the
// slot in the list of all parts, part_base[n_slots] = number of parts
__global__ __launch_bounds__(256) void lsr_slot_base_kernel(const int32_t* __restrict__ slot_n, int n_slots,
                                                            int32_t* __restrict__ slot_base,
                                                            int32_t* __restrict__ part_base) {
    for (int i = threadIdx.x; i < n_slots; i += 256) {
        slot_base[i] = slot_n[i];
        part_base[i] = (slot_n[i] + SS_CHUNK - 1) / SS_CHUNK;
    }
    __syncthreads();
    ss_block_scan(slot_base, n_slots);
    const int total = ss_block_scan(part_base, n_slots);
    if (threadIdx.x == 0) part_base[n_slots] = total;
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

// frame index at position pos of the sorted list (-1 past the end of the part)
__device__ __forceinline__ int ss_index(const int32_t* __restrict__ order, int pos, int end) {
    return pos < end ? order[pos] : -1;
}

// the lane's three values of frame `frame` (zeros for -1)
__device__ __forceinline__ void ss_load(const float* __restrict__ feats, int frame, int c, double (&x)[3]) {
    x[0] = x[1] = x[2] = 0.0;
    if (frame >= 0) {
        const float* row = feats + (size_t)frame * LSR_DIM;
        x[0] = (double)row[c];
        x[1] = (double)row[16 + c];
        x[2] = c < 7 ? (double)row[32 + c] : (c == 7 ? 1.0 : 0.0);       // column 39 is the constant
    }
}

__global__ __launch_bounds__(64) void lsr_gram_kernel(const float* __restrict__ feats, const int32_t* __restrict__ order,
                                                      const int32_t* __restrict__ slot_base,
                                                      const int32_t* __restrict__ slot_n,
                                                      const int32_t* __restrict__ part_base, int n_slots,
                                                      v4d* __restrict__ partial) {
    const int b = blockIdx.x;
    if (b >= part_base[n_slots]) return;                 // the grid is sized for the worst case
    // slot of this part: the last slot whose first part is <= b (slots without frames have no parts)
    int lo = 0, hi = n_slots;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (part_base[mid] <= b) lo = mid; else hi = mid;
    }
    const int s = lo, part = b - part_base[s];
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    const int first = slot_base[s] + part * SS_CHUNK;
    const int seg_end = slot_base[s] + slot_n[s];
    const int end = first + SS_CHUNK < seg_end ? first + SS_CHUNK : seg_end;
    v4d acc[SS_TILES];
#pragma unroll
    for (int t = 0; t < SS_TILES; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    // sixteen frames per iteration (four k-steps of four frames).  A frame is reached through two dependent loads (its
    // index in the sorted list, then its row), so the rows of the next sixteen frames and the indices of the sixteen
    // after those are in flight while the current ones are multiplied
    double x[4][3], nx[4][3];
    int ni[4], nni[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        ss_load(feats, ss_index(order, first + 4 * u + g, end), c, x[u]);
        ni[u] = ss_index(order, first + 16 + 4 * u + g, end);
    }
    for (int pos = first; pos < end; pos += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            nni[u] = ss_index(order, pos + 32 + 4 * u + g, end);
            ss_load(feats, ni[u], c, nx[u]);
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
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 3; ++k) x[u][k] = nx[u][k];
            ni[u] = nni[u];
        }
    }
    v4d* out = partial + ((size_t)b * SS_TILES) * 64 + lane;
#pragma unroll
    for (int t = 0; t < SS_TILES; ++t) out[t * 64] = acc[t];
}

// entry (i, j), i <= j, of the augmented Gram matrix lives in tile (i / 16, j / 16) at lane
// (i % 4) * 16 + j % 16, register (i % 16) / 4
// One thread per entry of a slot's augmented Gram matrix (blockIdx.y selects the 256 entries): the slot's parts are
// added in list order - a fixed order of additions - with eight loads in flight.
__global__ __launch_bounds__(256) void lsr_gram_reduce_kernel(const double* __restrict__ partial,
                                                              const int32_t* __restrict__ part_base,
                                                              double* __restrict__ count, double* __restrict__ sum_x,
                                                              double* __restrict__ sum_xx) {
    const int s = blockIdx.x;
    const int p0 = part_base[s], p1 = part_base[s + 1];
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= SS_AUG * SS_AUG) return;
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
    const double* src = partial + ((size_t)tile * 64 + lane) * 4 + reg;
    const size_t stride = (size_t)SS_TILES * 64 * 4;              // doubles per part
    double t = 0.0;
    int p = p0;
    for (; p + 8 <= p1; p += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(p + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; p < p1; ++p) t += src[(size_t)p * stride];
    if (row < LSR_DIM && colm < LSR_DIM)
        sum_xx[((size_t)s * LSR_DIM + row) * LSR_DIM + colm] = t;
    else if (row == LSR_DIM && colm < LSR_DIM)
        sum_x[(size_t)s * LSR_DIM + colm] = t;
    else if (row == LSR_DIM && colm == LSR_DIM)
        count[s] = t;
}

int lsr_launch_stats(lsr_ctx* ctx, const float* feats_d, const int32_t* slot_d, int64_t total_frames, int n_slots,
                     double* count_d, double* sum_x_d, double* sum_xx_d) {
    const int n_units = (int)((total_frames + SS_UNIT - 1) / SS_UNIT);
    // every slot with frames has at most one part that is not full
    const size_t max_parts = (size_t)(total_frames / SS_CHUNK) + (size_t)n_slots;
    // sort workspace: counts/offsets [slots][units], slot_base, slot_n, part_base [slots + 1], order [frames]
    const size_t n_counts = (size_t)n_units * n_slots;
    const size_t sort_ints = n_counts + 3 * (size_t)n_slots + 1 + (size_t)total_frames;
    int rc = lsr_reserve(ctx, &ctx->ws_sort, sort_ints * sizeof(int32_t), "slot sort");
    if (rc) return rc;
    const size_t partial_bytes = max_parts * SS_TILES * 64 * sizeof(v4d);
    if ((rc = lsr_reserve(ctx, &ctx->ws_slab, partial_bytes, "statistics partial sums"))) return rc;
    int32_t* counts = (int32_t*)ctx->ws_sort.ptr;
    int32_t* slot_base = counts + n_counts;
    int32_t* slot_n = slot_base + n_slots;
    int32_t* part_base = slot_n + n_slots;
    int32_t* order = part_base + n_slots + 1;
    v4d* partial = (v4d*)ctx->ws_slab.ptr;
    const size_t lds = (size_t)n_slots * sizeof(int);
    lsr_prof_begin(ctx, LSR_K_STATS);
    hipLaunchKernelGGL(lsr_slot_count_kernel, dim3(n_units), dim3(64), lds, ctx->stream, slot_d, total_frames, n_slots,
                       counts);
    LSR_LAUNCH_CHECK(ctx, "lsr_slot_count_kernel");
    hipLaunchKernelGGL(lsr_slot_scan_kernel, dim3(n_slots), dim3(256), 0, ctx->stream, counts, n_units, slot_n);
    LSR_LAUNCH_CHECK(ctx, "lsr_slot_scan_kernel");
    hipLaunchKernelGGL(lsr_slot_base_kernel, dim3(1), dim3(256), 0, ctx->stream, slot_n, n_slots, slot_base, part_base);
    LSR_LAUNCH_CHECK(ctx, "lsr_slot_base_kernel");
    hipLaunchKernelGGL(lsr_slot_scatter_kernel, dim3(n_units), dim3(64), lds, ctx->stream, slot_d, total_frames, n_slots,
                       counts, slot_base, order);
    LSR_LAUNCH_CHECK(ctx, "lsr_slot_scatter_kernel");
    hipLaunchKernelGGL(lsr_gram_kernel, dim3((unsigned)max_parts), dim3(64), 0, ctx->stream, feats_d, order, slot_base,
                       slot_n, part_base, n_slots, partial);
    LSR_LAUNCH_CHECK(ctx, "lsr_gram_kernel");
    hipLaunchKernelGGL(lsr_gram_reduce_kernel, dim3((unsigned)n_slots, (SS_AUG * SS_AUG + 255) / 256), dim3(256), 0, ctx->stream,
                       (const double*)partial, part_base, count_d, sum_x_d, sum_xx_d);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_gram_reduce_kernel");
    return LSR_OK;
}

// ---- path -> statistics slots and transition counts (integer work, one thread per frame) -------------
// The offsets come from the caller's device buffer: ranges are clamped to the table, so that inconsistent offsets
// give wrong results in the caller's own outputs rather than writes outside the workspace.
__global__ void lsr_utt_index_kernel(const int64_t* __restrict__ frame_off, int64_t total, int32_t* __restrict__ frame_utt) {
    const int b = blockIdx.x;
    int64_t lo = frame_off[b], hi = frame_off[b + 1];
    if (lo < 0) lo = 0;
    if (hi > total) hi = total;
    for (int64_t f = lo + threadIdx.x; f < hi; f += blockDim.x) frame_utt[f] = b;
}

int lsr_launch_utt_index(lsr_ctx* ctx, const int64_t* off_d, int n_utts, int64_t total, int32_t* out_d) {
    hipLaunchKernelGGL(lsr_utt_index_kernel, dim3(n_utts), dim3(256), 0, ctx->stream, off_d, total, out_d);
    LSR_LAUNCH_CHECK(ctx, "lsr_utt_index_kernel");
    return LSR_OK;
}

#define RM_FRAMES 4096       // frames per workgroup of the re-mux kernel
#define RM_LDS_STATES 128    // largest model whose transition table (n x n counters) is kept in LDS

// Transition counts concentrate on a few hundred (state, state) pairs, so one global atomic per frame would
// serialise on those addresses: a workgroup counts its 4 096 frames in an LDS table and adds the non-zero
// entries to the global table once (integer sums: the order does not matter).  Models too large for the LDS
// table count in global memory directly.
__global__ __launch_bounds__(256) void lsr_remux_kernel(const int8_t* __restrict__ paths,
                                                        const int64_t* __restrict__ frame_off,
                                                        const int64_t* __restrict__ chain_off,
                                                        const int32_t* __restrict__ chain_state,
                                                        const int32_t* __restrict__ chain_word,
                                                        const int32_t* __restrict__ frame_utt, int n_utts,
                                                        int64_t total_frames, int n_states, int drop_last_run, int use_lds,
                                                        int32_t* __restrict__ slots,
                                                        unsigned long long* __restrict__ trans) {
    extern __shared__ unsigned int local_trans[];
    const int n_entries = n_states * n_states;
    if (use_lds) {
        for (int e = threadIdx.x; e < n_entries; e += 256) local_trans[e] = 0u;
        __syncthreads();
    }
    const int64_t f0 = (int64_t)blockIdx.x * RM_FRAMES;
    for (int it = 0; it < RM_FRAMES / 256; ++it) {
        const int64_t f = f0 + it * 256 + threadIdx.x;
        if (f >= total_frames) break;
        const int b = frame_utt[f];
        if (b < 0 || b >= n_utts) {         // a frame outside every utterance's range
            slots[f] = -1;
            continue;
        }
        const int64_t c0 = chain_off[b], n_chain = chain_off[b + 1] - c0;
        const int64_t last = frame_off[b + 1] - 1;
        const int p = paths[f];
        if (p < 0 || p >= n_chain || last >= total_frames) {        // only a one-frame utterance has no back-pointer (-1)
            slots[f] = -1;
            continue;
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
                if (keep1 && w1 == w && g >= 0 && g < n_states && g1 >= 0 && g1 < n_states) {
                    if (use_lds)
                        atomicAdd(&local_trans[g * n_states + g1], 1u);
                    else
                        atomicAdd(&trans[(size_t)g * n_states + g1], 1ull);
                }
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int e = threadIdx.x; e < n_entries; e += 256) {
            const unsigned int c = local_trans[e];
            if (c) atomicAdd(&trans[e], (unsigned long long)c);
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
    // frames that no utterance covers (inconsistent offsets) keep -1 and are skipped by the re-mux kernel
    LSR_HIP_CHECK(ctx, hipMemsetAsync(frame_utt, 0xFF, (size_t)total_frames * sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(lsr_utt_index_kernel, dim3(n_utts), dim3(256), 0, ctx->stream, frame_off_d, total_frames, frame_utt);
    LSR_LAUNCH_CHECK(ctx, "lsr_utt_index_kernel");
    const int use_lds = n_states <= RM_LDS_STATES;
    const size_t lds = use_lds ? (size_t)n_states * n_states * sizeof(unsigned int) : 0;
    hipLaunchKernelGGL(lsr_remux_kernel, dim3((unsigned)((total_frames + RM_FRAMES - 1) / RM_FRAMES)), dim3(256), lds,
                       ctx->stream, paths_d, frame_off_d, chain_off_d, chain_state_d, chain_word_d, frame_utt, n_utts,
                       total_frames, n_states, drop_last_run, use_lds, slots_d, (unsigned long long*)trans_d);
    LSR_LAUNCH_CHECK(ctx, "lsr_remux_kernel");
    return LSR_OK;
}
