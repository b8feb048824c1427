// Gaussian emission scores for every (frame, state): float64 whitening on the MFMA f64 unit.
//
//   ll[f][s] = float32( -0.5 * (log_norm[s] + sum_j ( sum_k (x[f][k] - mean[s][k]) * U[s][k][j] )^2 ) )
//
// evaluated as sum_j ( sum_k x~[f][k] U~[s][k][j] )^2 with x~ = [x, 1] and U~ = [U; -(mean U)] (the
// offset c = mean U is computed on the host in extended precision): the matrix unit does all the
// arithmetic of the whitening, float64 vector adds would compete with the MFMAs for the same pipe.
//
// The work is the list of (64-frame tile, state) pairs in tile-major order.  The grid is persistent
// (three workgroups of four wavefronts per compute unit) and every wavefront takes an equal
// contiguous share of that list, so all wavefronts finish together whatever the batch size; a
// share is a few runs of at most 32 consecutive states of one tile.  For a run the wavefront holds the
// tile's 64 frames (4 column groups of 16) as B operands in registers.
//
// The matrix instruction is v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products, here the
// same 4 rows x 4 features block of U~ against four groups of 4 frames.  It has the MAC rate of the
// 16 x 16 x 4 form, and its 4 x 4 granularity fits a triangular U: with U zero above the diagonal
// (TRI), row block jb only needs feature blocks kb >= jb, 55 of 100 blocks.  Lane l supplies
// A[k = l >> 4][i = l & 3] (the same for its four frame groups), B[k = l >> 4][frame l & 15] and
// receives row i = l >> 4 of frame l & 15; A values stream through a small ring of registers from
// the table built by lsr_model_upload (16 distinct values per block: the loads of a wavefront
// coalesce into one or two 128-byte lines).  The accumulators of a row block are squared and
// summed in-lane; after the last block the four 16-lane groups (rows) are folded so that lane l
// holds the score of frame l, which goes to the wavefront's LDS tile and, at the end of the run,
// to memory row by row.
//
// Every other vector instruction takes issue slots from the float64 MFMAs (they share the
// double-precision pipe), so the loop keeps them to the 40 squares, the fold and the conversion.
//
// Mixtures: with n_mix > 1 the n_states Gaussians are n_states / n_mix states of n_mix components
// each; consecutive components are folded with a sequential float64 log-add-exp of
// log w + log N (the order NumPy's np.logaddexp loop uses) before the float32 rounding.
#include "lsr_internal.h"

#define EM_WAVES 4
#define EM_RUN 32             // most states of one tile a wavefront scores before it writes them out

typedef double v2d __attribute__((ext_vector_type(2)));

// Exchange the upper 32 lanes of a with the lower 32 lanes of b (v_permlane32_swap), and the odd
// 16-lane rows of a with the even rows of b (v_permlane16_swap): after the exchange a + b adds
// lane pairs 32 (16) apart, the first operand's sums landing in the lower (even) half.
__device__ __forceinline__ void em_swap32(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void em_swap16(double& a, double& b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}

template <bool TRI, bool MIX>
__global__ __launch_bounds__(64 * EM_WAVES, MIX ? 2 : 3) void lsr_emission_kernel(
    const float* __restrict__ feats, int64_t total_frames, const double* __restrict__ wfrag,
    const double* __restrict__ lognorm, const double* __restrict__ logw,
    int n_states, int n_mix, float* __restrict__ ll, int ll_stride) {
    // [frame][(state - first state of the run) ^ frame]: the XOR keeps the per-row reads on distinct LDS
    // banks without padding; 8 KiB per wavefront, so three workgroups fit a compute unit
    __shared__ float tile[EM_WAVES][64][EM_RUN];
    // readfirstlane: the wavefront index (and everything derived from it) lives in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int n_cols = n_states / n_mix;                       // states = columns of the score matrix
    const int64_t n_tiles = (total_frames + 63) / 64;
    const int64_t n_pairs = n_tiles * n_cols;
    const int64_t n_waves = (int64_t)gridDim.x * EM_WAVES, me = (int64_t)blockIdx.x * EM_WAVES + wave;
    int64_t p = me * n_pairs / n_waves;
    const int64_t p_end = (me + 1) * n_pairs / n_waves;

    int64_t loaded = -1;                                       // tile whose frames are in x[][]
    double x[4][LSR_KSTEPS];
    while (p < p_end) {
        // one run: columns c0..c1-1 of tile t, at most EM_RUN of them
        const int64_t t = p / n_cols;
        const int c0 = (int)(p - t * n_cols);
        int c1 = (int)((int64_t)n_cols - c0 < p_end - p ? n_cols : c0 + (p_end - p));
        if (c1 > c0 + EM_RUN) c1 = c0 + EM_RUN;
        const int64_t f0 = t * 64;
        p += c1 - c0;

        // B-operand source: x~[frame 16q + col][k = 4ks + grp]; frames beyond the batch are zero
        if (t != loaded) {
            loaded = t;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t f = f0 + 16 * q + col;
                const bool live = f < total_frames;
                const float* row = feats + (live ? f : 0) * LSR_DIM;
#pragma unroll
                for (int ks = 0; ks < LSR_KSTEPS; ++ks) {
                    const int k = 4 * ks + grp;
                    x[q][ks] = k < LSR_DIM ? (live ? (double)row[k] : 0.0) : 1.0;      // feature 39 is the constant
                }
            }
        }

        // A values are a stream: the blocks of a state in the order of their use (see the loop below),
        // SLOTS places per state, states back to back.  RING of them are in flight: the register of a
        // block is refilled with the block RING places further down the stream as soon as its four
        // MFMAs are issued, which is ~900 cycles before that block is needed.
        constexpr int SLOTS = TRI ? LSR_TRI_SLOTS : LSR_JBLOCKS * LSR_KSTEPS;
        constexpr int RING = TRI ? 14 : 10;                                    // divides SLOTS, even
        const int s_begin = c0 * n_mix, s_end = c1 * n_mix;
        const int a_idx = (lane >> 4) * 4 + (lane & 3);
        // two consecutive places of the stream are interleaved in memory: one 16-byte load fills two registers
        v2d w[RING / 2];
        {
            const v2d* src = (const v2d*)wfrag + (size_t)s_begin * (SLOTS * 8) + a_idx;
#pragma unroll
            for (int v = 0; v < RING / 2; ++v) w[v] = src[v * 16];
        }

        double run = 0.0;          // log-sum of the components of the current state seen so far
        int mix = 0;
        for (int s = s_begin; s < s_end; ++s) {
            const v2d* ahead = (const v2d*)wfrag + ((size_t)s * SLOTS + RING) * 8 + a_idx;      // wave-uniform base
            double ss[4] = {0.0, 0.0, 0.0, 0.0};
            // Row blocks are taken two at a time (jb, jb + 1) with their feature blocks interleaved: eight
            // independent accumulator chains keep the matrix pipe fed (a chain can take a new MFMA only
            // when its previous one has finished).  Place in the stream: pair base + 2 t (+ 1 for the second).
#pragma unroll
            for (int jp = 0; jp < LSR_JBLOCKS; jp += 2) {
                const int ka = TRI ? jp : 0, kb0 = TRI ? jp + 1 : 0;             // first feature block of each row block
                const int na = LSR_KSTEPS - ka, nb = LSR_KSTEPS - kb0;
                // blocks before row block jp: sum_{j < jp} (10 - j) for a triangular matrix
                const int v_base = TRI ? jp * LSR_KSTEPS - jp * (jp - 1) / 2 : jp * LSR_KSTEPS;
                double acc[8];
#pragma unroll
                for (int t = 0; t < LSR_KSTEPS; ++t) {
                    if (t < na) {
                        const int v = v_base + 2 * t;
                        const double wv = w[(v % RING) / 2][v % 2];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(wv, x[q][ka + t], t == 0 ? 0.0 : acc[q], 0, 0, 0);
                        if (v % 2 == 1) w[(v % RING) / 2] = ahead[(v / 2) * 16];      // both places of the pair are used
                    }
                    if (t < nb) {
                        const int v = v_base + 2 * t + 1;
                        const double wv = w[(v % RING) / 2][v % 2];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[4 + q] = __builtin_amdgcn_mfma_f64_4x4x4f64(wv, x[q][kb0 + t], t == 0 ? 0.0 : acc[4 + q], 0, 0, 0);
                        if (v % 2 == 1) w[(v % RING) / 2] = ahead[(v / 2) * 16];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) ss[q] = fma(acc[q], acc[q], ss[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) ss[q] = fma(acc[4 + q], acc[4 + q], ss[q]);
            }
            // the padding places of the stream carry no block, but their turn to refill a register remains
#pragma unroll
            for (int v = TRI ? 55 : SLOTS; v < SLOTS; ++v)
                if (v % 2 == 1) w[(v % RING) / 2] = ahead[(v / 2) * 16];
            const double ln = lognorm[s];
            // fold the four lane groups (rows of the C tiles) so that group g keeps frame tile g
            em_swap32(ss[0], ss[2]);
            em_swap32(ss[1], ss[3]);
            double v0 = ss[0] + ss[2];            // lanes 0..31: frame tile 0, lanes 32..63: frame tile 2
            double v1 = ss[1] + ss[3];            // frame tiles 1 and 3
            em_swap16(v0, v1);
            const double maha = v0 + v1;
            const double lp = -0.5 * (ln + maha);
            if (!MIX) {
                tile[wave][lane][((s - s_begin) ^ lane) & (EM_RUN - 1)] = (float)lp;
            } else {
                const double v = logw[s] + lp;
                if (mix == 0) {
                    run = v;                                   // logaddexp(-inf, v)
                } else if (run == v) {
                    run = run + 0.693147180559945309417232121458176568;
                } else {
                    const double d = run - v;                  // NumPy: x + log1p(exp(-d)) for d > 0, else y + log1p(exp(d))
                    run = d > 0 ? run + log1p(exp(-d)) : (d <= 0 ? v + log1p(exp(d)) : d);
                }
                if (++mix == n_mix) {
                    const int c = s / n_mix;
                    tile[wave][lane][((c - c0) ^ lane) & (EM_RUN - 1)] = (float)run;
                    mix = 0;
                }
            }
        }
        // write-out of columns c0..c1-1 of the wavefront's own tile, one row per instruction (lane = column
        // c0 + lane); the run that ends a tile also fills the padding columns up to the row stride with -inf.  Only
        // this wavefront touches tile[wave].
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < c1 - c0) {
#pragma unroll 8
            for (int r = 0; r < 64; ++r)
                ll[(f0 + r) * ll_stride + c0 + lane] = tile[wave][r][(lane ^ r) & (EM_RUN - 1)];
        }
        if (c1 == n_cols && n_cols + lane < ll_stride) {
#pragma unroll 8
            for (int r = 0; r < 64; ++r) ll[(f0 + r) * ll_stride + n_cols + lane] = -INFINITY;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

int lsr_launch_emission(lsr_ctx* ctx, const float* feats_d, int64_t total_frames, float* ll_d) {
    // persistent grid: as many workgroups as fit a compute unit (three of 32 KiB of LDS and <= 168 VGPRs;
    // two for mixtures), fewer when the batch has fewer (tile, state) pairs than that many wavefronts
    const bool mix = ctx->n_mix > 1;
    const int per_cu = mix ? 2 : 3;
    const int64_t pairs = (total_frames + 63) / 64 * (ctx->n_states / ctx->n_mix);
    int64_t blocks = (pairs + EM_WAVES - 1) / EM_WAVES;
    if (blocks > per_cu * (int64_t)ctx->n_cus) blocks = per_cu * (int64_t)ctx->n_cus;
    void (*kern)(const float*, int64_t, const double*, const double*, const double*, int, int, float*, int) =
        ctx->whiten_triangular ? (mix ? lsr_emission_kernel<true, true> : lsr_emission_kernel<true, false>)
                               : (mix ? lsr_emission_kernel<false, true> : lsr_emission_kernel<false, false>);
    lsr_prof_begin(ctx, LSR_K_EMISSION);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * EM_WAVES), 0, ctx->stream, feats_d, total_frames,
                       ctx->d_wfrag, ctx->d_lognorm, ctx->d_logw, ctx->n_states, ctx->n_mix, ll_d, lsr_ctx_ll_stride(ctx));
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_emission_kernel");
    return LSR_OK;
}
