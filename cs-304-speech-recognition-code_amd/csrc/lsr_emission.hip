// Gaussian emission scores for every (frame, state): float64 whitening on the MFMA f64 unit.
//
//   ll[f][s] = float32( -0.5 * (log_norm[s] + sum_j ( sum_k (x[f][k] - mean[s][k]) * U[s][k][j] )^2 ) )
//
// evaluated as sum_j ( sum_k x[f][k] U[s][k][j] - c[s][j] )^2 with c = mean U precomputed on the
// host: the accumulators start at -c, so the matrix unit does all the arithmetic of the
// whitening (float64 vector adds would compete with the float64 MFMAs for the same pipe).
//
// The work is the list of (64-frame tile, state) pairs in tile-major order.  The grid is persistent
// (two workgroups of four wavefronts per compute unit) and every wavefront takes an equal
// contiguous share of that list, so all wavefronts finish together whatever the batch size; a
// share is a few runs of consecutive states of one tile.  For a run the wavefront holds the
// tile's 64 frames (4 column tiles of 16) as B operands in registers.  A state is three row tiles
// of 16 whitened components; a row tile is k-steps ks_begin[a]..9 of v_mfma_f64_16x16x4_f64 for
// each of the 4 frame tiles, with A = U_s^T fragments (coalesced 512-byte loads from the
// pre-swizzled table built by lsr_model_upload, fetched one row tile ahead of their use).  The
// accumulators of a row tile are squared and summed in-lane; after the third tile the four
// 16-lane groups (rows of the C tiles) are folded so that lane l holds the score of frame l,
// which goes to the wavefront's LDS tile and, at the end of the run, to memory row by row.
//
// Every other vector instruction takes issue slots from the float64 MFMAs (they share the
// double-precision pipe), so the loop keeps them to the 40 squares, the fold and the conversion.
//
// Mixtures: with n_mix > 1 the n_states Gaussians are n_states / n_mix states of n_mix components
// each; consecutive components are folded with a sequential float64 log-add-exp of
// log w + log N (the order NumPy's np.logaddexp loop uses) before the float32 rounding.
//
// ks_begin lets a lower-triangular whitening matrix (zero above the diagonal) skip the
// all-zero k-steps of the second and third row tile; a dense matrix uses 0, 0, 0.
#include "lsr_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define EM_WAVES 4

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

template <int KB1, int KB2>
__global__ __launch_bounds__(64 * EM_WAVES, 2) void lsr_emission_kernel(
    const float* __restrict__ feats, int64_t total_frames, const double* __restrict__ wfrag,
    const v4d* __restrict__ cfrag, const double* __restrict__ lognorm, const double* __restrict__ logw,
    int n_states, int n_mix, float* __restrict__ ll) {
    // [frame][state ^ frame]: the XOR keeps both the per-frame writes and the per-row reads
    // on distinct LDS banks without padding (4 x 16 KiB = the 64 KiB static limit)
    __shared__ float tile[EM_WAVES][64][64];
    // readfirstlane: the wavefront index (and everything derived from it) lives in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int n_cols = n_states / n_mix;                       // states = columns of the score matrix
    const int64_t n_tiles = (total_frames + 63) / 64;
    const int64_t n_pairs = n_tiles * n_cols;
    const int64_t n_waves = (int64_t)gridDim.x * EM_WAVES, me = (int64_t)blockIdx.x * EM_WAVES + wave;
    int64_t p = me * n_pairs / n_waves;
    const int64_t p_end = (me + 1) * n_pairs / n_waves;

    while (p < p_end) {
        // one run: columns c0..c1-1 of tile t
        const int64_t t = p / n_cols;
        const int c0 = (int)(p - t * n_cols);
        const int c1 = (int)((int64_t)n_cols - c0 < p_end - p ? n_cols : c0 + (p_end - p));
        const int64_t f0 = t * 64;
        p += c1 - c0;

        // B-operand source: x[frame 16q + col][k = 4ks + grp], zero beyond the batch / beyond k = 38
        double x[4][LSR_KSTEPS];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t f = f0 + 16 * q + col;
            const bool live = f < total_frames;
            const float* row = feats + (live ? f : 0) * LSR_DIM;
#pragma unroll
            for (int ks = 0; ks < LSR_KSTEPS; ++ks) {
                const int k = 4 * ks + grp;
                x[q][ks] = (live && k < LSR_DIM) ? (double)row[k] : 0.0;
            }
        }

        // [state][row tile][k-step][lane] and [state][row tile][lane] (-c for the lane's 4 rows of the
        // C tile); both tables end with one zero tile, so the look-ahead below needs no bounds test
        const int s_begin = c0 * n_mix, s_end = c1 * n_mix;
        double wc[LSR_KSTEPS], wn[LSR_KSTEPS];
        {
            const double* src = wfrag + (size_t)s_begin * (LSR_JTILES * LSR_KSTEPS * 64);
#pragma unroll
            for (int ks = 0; ks < LSR_KSTEPS; ++ks) wc[ks] = src[ks * 64 + lane];
        }
        v4d cc = cfrag[(size_t)s_begin * (LSR_JTILES * 64) + lane], cn;

        double run = 0.0;          // log-sum of the components of the current state seen so far
        int mix = 0;
        for (int s = s_begin; s < s_end; ++s) {
            double ss[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int a = 0; a < LSR_JTILES; ++a) {
                const int kb = a == 0 ? 0 : (a == 1 ? KB1 : KB2);
                // fetch the fragments of the next row tile (next state after the third tile)
                {
                    const int nxt = s * LSR_JTILES + a + 1;                      // wave-uniform
                    const double* src = wfrag + (size_t)nxt * (LSR_KSTEPS * 64);
                    const int nkb = a == 0 ? KB1 : (a == 1 ? KB2 : 0);
#pragma unroll
                    for (int ks = 0; ks < LSR_KSTEPS; ++ks)
                        if (ks >= nkb) wn[ks] = src[ks * 64 + lane];
                    cn = cfrag[(size_t)nxt * 64 + lane];
                }
                v4d acc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = cc;
#pragma unroll
                for (int ks = 0; ks < LSR_KSTEPS; ++ks) {
                    if (ks >= kb) {
                        const double w = wc[ks];
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, x[0][ks], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, x[1][ks], acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, x[2][ks], acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, x[3][ks], acc[3], 0, 0, 0);
                    }
                }
                // rows 40..47 (registers 2 and 3 of the third tile) are padding: always zero
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < (a == 2 ? 2 : 4); ++r) ss[q] = fma(acc[q][r], acc[q][r], ss[q]);
#pragma unroll
                for (int ks = 0; ks < LSR_KSTEPS; ++ks) wc[ks] = wn[ks];
                cc = cn;
            }
            const double ln = lognorm[s];
            // fold the four lane groups (rows of the C tiles) so that group g keeps frame tile g
            em_swap32(ss[0], ss[2]);
            em_swap32(ss[1], ss[3]);
            double v0 = ss[0] + ss[2];            // lanes 0..31: frame tile 0, lanes 32..63: frame tile 2
            double v1 = ss[1] + ss[3];            // frame tiles 1 and 3
            em_swap16(v0, v1);
            const double maha = v0 + v1;
            const double lp = -0.5 * (ln + maha);
            if (n_mix == 1) {
                tile[wave][lane][(s ^ lane) & 63] = (float)lp;
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
                    tile[wave][lane][(c ^ lane) & 63] = (float)run;
                    mix = 0;
                }
            }
        }
        // write-out of columns c0..c1-1 (and the -inf padding columns after the last state) of the
        // wavefront's own tile, one row per instruction; only this wavefront touches tile[wave]
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool mine = lane >= c0 && lane < (c1 == n_cols ? 64 : c1);
        for (int r = 0; r < 64; ++r) {
            const float v = lane < n_cols ? tile[wave][r][(lane ^ r) & 63] : -INFINITY;
            if (mine) ll[(f0 + r) * LSR_LL_STRIDE + lane] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

int lsr_launch_emission(lsr_ctx* ctx, const float* feats_d, int64_t total_frames, float* ll_d) {
    // persistent grid: two resident workgroups per compute unit (64 KiB of LDS each), fewer when the
    // batch has fewer (tile, state) pairs than that many wavefronts
    const int64_t pairs = (total_frames + 63) / 64 * (ctx->n_states / ctx->n_mix);
    int64_t blocks = (pairs + EM_WAVES - 1) / EM_WAVES;
    if (blocks > 2 * (int64_t)ctx->n_cus) blocks = 2 * (int64_t)ctx->n_cus;
    lsr_prof_begin(ctx, LSR_K_EMISSION);
    if (ctx->whiten_triangular)
        hipLaunchKernelGGL((lsr_emission_kernel<4, 8>), dim3((unsigned)blocks), dim3(64 * EM_WAVES), 0, ctx->stream,
                           feats_d, total_frames, ctx->d_wfrag, (const v4d*)ctx->d_cfrag, ctx->d_lognorm, ctx->d_logw, ctx->n_states, ctx->n_mix, ll_d);
    else
        hipLaunchKernelGGL((lsr_emission_kernel<0, 0>), dim3((unsigned)blocks), dim3(64 * EM_WAVES), 0, ctx->stream,
                           feats_d, total_frames, ctx->d_wfrag, (const v4d*)ctx->d_cfrag, ctx->d_lognorm, ctx->d_logw, ctx->n_states, ctx->n_mix, ll_d);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_emission_kernel");
    return LSR_OK;
}
