// Gaussian emission scores for every (frame, state): float64 whitening on the MFMA f64 unit.
//
//   ll[f][s] = float32( -0.5 * (log_norm[s] + sum_j ( sum_k (x[f][k] - mean[s][k]) * U[s][k][j] )^2 ) )
//
// One wavefront owns 64 consecutive frames (4 column tiles of 16 frames) and walks over all
// states.  Per state it issues 10 k-steps x 3 row tiles x 4 frame tiles of
// v_mfma_f64_16x16x4_f64 with A = U_s^T fragments (read once per wave, coalesced, from the
// pre-swizzled table built by lsr_model_upload) and B = (x - mean_s) formed in registers.
// The 12 accumulators of a state are squared and summed in-lane, then reduced across the
// four 16-lane groups so that lane l ends up with the score of frame l.
#include "lsr_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define EM_WAVES 4

__global__ __launch_bounds__(64 * EM_WAVES) void lsr_emission_kernel(
    const float* __restrict__ feats, int64_t total_frames, const double* __restrict__ wfrag,
    const double* __restrict__ mean, const double* __restrict__ lognorm, int n_states,
    float* __restrict__ ll) {
    // [frame][state ^ frame]: the XOR keeps both the per-frame writes and the per-row reads
    // on distinct LDS banks without padding (4 x 16 KiB = the 64 KiB static limit)
    __shared__ float tile[EM_WAVES][64][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int64_t f0 = ((int64_t)blockIdx.x * EM_WAVES + wave) * 64;

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

    const bool hi32 = (lane & 32) != 0, hi16 = (lane & 16) != 0;
    for (int s = 0; s < n_states; ++s) {
        v4d acc[LSR_JTILES][4];
#pragma unroll
        for (int a = 0; a < LSR_JTILES; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[a][q] = (v4d){0.0, 0.0, 0.0, 0.0};
        const double* wf = wfrag + (size_t)s * (LSR_KSTEPS * LSR_JTILES * 64) + lane;
        const double* mu = mean + (size_t)s * LSR_KPAD + grp;
#pragma unroll
        for (int ks = 0; ks < LSR_KSTEPS; ++ks) {
            const double m = mu[4 * ks];
            const double d0 = x[0][ks] - m, d1 = x[1][ks] - m, d2 = x[2][ks] - m, d3 = x[3][ks] - m;
#pragma unroll
            for (int a = 0; a < LSR_JTILES; ++a) {
                const double w = wf[(ks * LSR_JTILES + a) * 64];
                acc[a][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, d0, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, d1, acc[a][1], 0, 0, 0);
                acc[a][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, d2, acc[a][2], 0, 0, 0);
                acc[a][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, d3, acc[a][3], 0, 0, 0);
            }
        }
        double ss[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double t = 0.0;
#pragma unroll
            for (int a = 0; a < LSR_JTILES; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) t = fma(acc[a][q][r], acc[a][q][r], t);
            ss[q] = t;
        }
        // fold the four lane groups (rows of the C tiles) so that group g keeps frame tile g
        const double keep_a = hi32 ? ss[2] : ss[0], send_a = hi32 ? ss[0] : ss[2];
        const double keep_b = hi32 ? ss[3] : ss[1], send_b = hi32 ? ss[1] : ss[3];
        const double v0 = keep_a + __shfl_xor(send_a, 32);
        const double v1 = keep_b + __shfl_xor(send_b, 32);
        const double keep = hi16 ? v1 : v0, send = hi16 ? v0 : v1;
        const double maha = keep + __shfl_xor(send, 16);
        tile[wave][lane][(s ^ lane) & 63] = (float)(-0.5 * (lognorm[s] + maha));
    }
    __syncthreads();
    // coalesced write-out: each wave stores its own 64 x 64 tile, one 256-byte row per instruction
    for (int r = 0; r < 64; ++r) {
        const int64_t f = f0 + r;
        const float v = lane < n_states ? tile[wave][r][(lane ^ r) & 63] : -INFINITY;
        if (f < (total_frames + 63) / 64 * 64) ll[f * LSR_LL_STRIDE + lane] = v;
    }
}

int lsr_launch_emission(lsr_ctx* ctx, const float* feats_d, int64_t total_frames, float* ll_d) {
    const int64_t per_block = 64 * EM_WAVES;
    const int64_t blocks = (total_frames + per_block - 1) / per_block;
    lsr_prof_begin(ctx, LSR_K_EMISSION);
    hipLaunchKernelGGL(lsr_emission_kernel, dim3((unsigned)blocks), dim3(64 * EM_WAVES), 0, ctx->stream, feats_d,
                       total_frames, ctx->d_wfrag, ctx->d_mean, ctx->d_lognorm, ctx->n_states, ll_d);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_emission_kernel");
    return LSR_OK;
}
