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

// Ring of A operands in flight per wavefront (places of the operand stream, see below): divides the places of a
// state and is even.
template <bool TRI>
struct em_ring {
    static constexpr int SLOTS = TRI ? LSR_TRI_SLOTS : LSR_JBLOCKS * LSR_KSTEPS;
    static constexpr int RING = TRI ? 14 : 10;
};

// Squared whitened components of the wavefront's 64 frames for ONE Gaussian, summed per lane into ss[4] (lane
// group g = rows 4 jb + g of every row block, frame column q): the MFMA loop.
//
// A values are a stream: the blocks of a Gaussian in the order of their use, SLOTS places per Gaussian.  RING of
// them are in flight: the register of a block is refilled with the block RING places further down the stream as
// soon as its four MFMAs are issued, which is ~900 cycles before that block is needed.  `cur` is the stream of
// this Gaussian (lane's element of place 0), `nxt` that of the Gaussian scored next: with SPLIT = false the
// Gaussians are consecutive in memory and the stream simply runs on (nxt is unused).
template <bool TRI, bool SPLIT>
__device__ __forceinline__ void em_gaussian(v2d (&w)[em_ring<TRI>::RING / 2], const double (&x)[4][LSR_KSTEPS],
                                            const v2d* __restrict__ cur, const v2d* __restrict__ nxt, double (&ss)[4]) {
    constexpr int SLOTS = em_ring<TRI>::SLOTS, RING = em_ring<TRI>::RING;
    // place v (odd: both places of its register pair are used) has been consumed: fetch the pair RING places on
    auto refill = [&](int v) {
        const int pn = v / 2 + RING / 2;                     // register pair = two consecutive places
        if (!SPLIT || pn < SLOTS / 2) w[(v % RING) / 2] = cur[pn * 16];
        else w[(v % RING) / 2] = nxt[(pn - SLOTS / 2) * 16];
    };
    ss[0] = ss[1] = ss[2] = ss[3] = 0.0;
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
                if (v % 2 == 1) refill(v);
            }
            if (t < nb) {
                const int v = v_base + 2 * t + 1;
                const double wv = w[(v % RING) / 2][v % 2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[4 + q] = __builtin_amdgcn_mfma_f64_4x4x4f64(wv, x[q][kb0 + t], t == 0 ? 0.0 : acc[4 + q], 0, 0, 0);
                if (v % 2 == 1) refill(v);
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
        if (v % 2 == 1) refill(v);
}

// log density of the Gaussian for the frame of each lane from the per-lane sums of em_gaussian: folds the four
// lane groups (rows of the C tiles) so that lane l ends up with frame l of the tile
__device__ __forceinline__ double em_fold(double (&ss)[4], double log_norm) {
    em_swap32(ss[0], ss[2]);
    em_swap32(ss[1], ss[3]);
    double v0 = ss[0] + ss[2];            // lanes 0..31: frame tile 0, lanes 32..63: frame tile 2
    double v1 = ss[1] + ss[3];            // frame tiles 1 and 3
    em_swap16(v0, v1);
    const double maha = v0 + v1;
    return -0.5 * (log_norm + maha);
}

// B-operand source of a tile: x~[frame 16q + col][k = 4ks + grp]; frames beyond the batch are zero.
// The 64 feature rows of a tile are one contiguous span of 64 * 39 floats: the wavefront copies it to its LDS
// area with ten coalesced 16-byte loads per lane (the rows start on 4-byte boundaries only, which global loads
// accept) and every lane then picks its forty values from there - a lane reading its own values straight from
// memory touches sixteen different rows per load instruction.  `stage` is the wavefront's LDS area
// (EM_LDS_FLOATS floats), free between two runs.  The last tile of the batch, whose span would run past the
// features, reads memory directly.
#define EM_LDS_FLOATS 2560          // >= 64 * 39 staged features, >= 64 * EM_RUN scores
typedef float em_f4 __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void em_load_frames(double (&x)[4][LSR_KSTEPS], float* __restrict__ stage,
                                               const float* __restrict__ feats, int64_t f0, int64_t total_frames,
                                               int lane, int col, int grp) {
    if (f0 + 64 <= total_frames) {
        const float* src = feats + f0 * LSR_DIM;
        // groups of four loads off one wave-uniform base: the lane's offset is one register, the rest immediates
        // (the lane index goes through an opaque move so that the addresses are formed here, once per tile, instead
        // of being kept in registers across the matrix loop, where there are none to spare)
        int l = lane;
        asm volatile("" : "+v"(l));
        const float* lane_src = src + l * 4;
        float* lane_dst = stage + l * 4;
#pragma unroll
        for (int g = 0; g < 10; g += 4) {
            em_f4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (g + i < 9 || (g + i == 9 && l < (64 * LSR_DIM - 9 * 256) / 4)) v[i] = *(const em_f4*)(lane_src + (g + i) * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (g + i < 9 || (g + i == 9 && l < (64 * LSR_DIM - 9 * 256) / 4))
                    *(float4*)(lane_dst + (g + i) * 256) = float4{v[i][0], v[i][1], v[i][2], v[i][3]};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* row = stage + (16 * q + col) * LSR_DIM;
#pragma unroll
            for (int ks = 0; ks < LSR_KSTEPS; ++ks) {
                const int k = 4 * ks + grp;
                x[q][ks] = k < LSR_DIM ? (double)row[k] : 1.0;                  // feature 39 is the constant
            }
        }
        // the area is written again (scores) only after every lane has read its values
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t f = f0 + 16 * q + col;
        const bool live = f < total_frames;
        const float* row = feats + (live ? f : 0) * LSR_DIM;
#pragma unroll
        for (int ks = 0; ks < LSR_KSTEPS; ++ks) {
            const int k = 4 * ks + grp;
            x[q][ks] = k < LSR_DIM ? (live ? (double)row[k] : 0.0) : 1.0;
        }
    }
}

template <bool TRI, bool MIX>
__global__ __launch_bounds__(64 * EM_WAVES, MIX ? 2 : 3) void lsr_emission_kernel(
    const float* __restrict__ feats, int64_t total_frames, const double* __restrict__ wfrag,
    const double* __restrict__ lognorm, const double* __restrict__ logw,
    int n_states, int n_mix, float* __restrict__ ll, int ll_stride) {
    // per wavefront: scores [frame][(state - first state of the run) ^ frame] - the XOR keeps the per-row reads on
    // distinct LDS banks without padding - and, between runs, the staged features of the next tile; 10 KiB per
    // wavefront, so three workgroups fit a compute unit
    __shared__ __attribute__((aligned(16))) float lds[EM_WAVES][EM_LDS_FLOATS];
    // readfirstlane: the wavefront index (and everything derived from it) lives in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float (*tile)[EM_RUN] = (float (*)[EM_RUN])lds[wave];
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

        if (t != loaded) {
            loaded = t;
            em_load_frames(x, lds[wave], feats, f0, total_frames, lane, col, grp);
        }

        constexpr int SLOTS = em_ring<TRI>::SLOTS, RING = em_ring<TRI>::RING;
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
            const v2d* cur = (const v2d*)wfrag + (size_t)s * (SLOTS * 8) + a_idx;      // wave-uniform base
            double ss[4];
            em_gaussian<TRI, false>(w, x, cur, cur, ss);
            const double lp = em_fold(ss, lognorm[s]);
            if (!MIX) {
                tile[lane][((s - s_begin) ^ lane) & (EM_RUN - 1)] = (float)lp;
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
                    tile[lane][((c - c0) ^ lane) & (EM_RUN - 1)] = (float)run;
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
                ll[(f0 + r) * ll_stride + c0 + lane] = tile[r][(lane ^ r) & (EM_RUN - 1)];
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

// ---- scores of selected states only ------------------------------------------------------------------------
// Forced alignment reads, for the frames of an utterance, only the columns of the states in that utterance's
// chain (hidden_markov_model.py:638-664: the chain S d S d ... S holds the models of the utterance's own words).
// Here the 64-frame tiles are anchored at the utterance starts (tile k of utterance b = its frames 64 k ..
// 64 k + 63), so a tile belongs to one utterance and needs exactly the states of that utterance's chain:
// utt_mask[b] (two 64-bit words).  The work list is every (utterance, tile, selected state) in that order -
// utterance b owns pairs pair_first[b] .. pair_first[b + 1] - 1 - cut into equal contiguous shares like the
// full kernel's.  Columns outside the mask are left as they are.

// a wave-uniform 64-bit value moved to scalar registers
__device__ __forceinline__ unsigned long long em_uniform(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

template <bool TRI>
__global__ __launch_bounds__(64 * EM_WAVES, 3) void lsr_emission_masked_kernel(
    const float* __restrict__ feats, int64_t total_frames, const double* __restrict__ wfrag,
    const double* __restrict__ lognorm, float* __restrict__ ll, int ll_stride,
    const int64_t* __restrict__ frame_off, int n_utts, const unsigned long long* __restrict__ utt_mask,
    const int64_t* __restrict__ pair_first) {
    __shared__ __attribute__((aligned(16))) float lds[EM_WAVES][EM_LDS_FLOATS];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float (*tile)[EM_RUN] = (float (*)[EM_RUN])lds[wave];
    const int col = lane & 15, grp = lane >> 4;
    const int64_t n_pairs = (int64_t)em_uniform((unsigned long long)pair_first[n_utts]);
    const int64_t n_waves = (int64_t)gridDim.x * EM_WAVES, me = (int64_t)blockIdx.x * EM_WAVES + wave;
    int64_t p = (int64_t)em_uniform((unsigned long long)(me * n_pairs / n_waves));
    const int64_t p_end = (int64_t)em_uniform((unsigned long long)((me + 1) * n_pairs / n_waves));
    if (p >= p_end) return;

    // the utterance that owns pair p: the last one whose first pair is <= p (utterances without pairs in between
    // share that value and are passed over)
    int b = 0;
    {
        int hi = n_utts;
        while (hi - b > 1) {
            const int mid = (b + hi) >> 1;
            if ((int64_t)em_uniform((unsigned long long)pair_first[mid]) <= p) b = mid; else hi = mid;
        }
    }
    unsigned long long u0 = em_uniform(utt_mask[2 * b]), u1 = em_uniform(utt_mask[2 * b + 1]);     // states of utterance b
    int64_t f_utt = (int64_t)em_uniform((unsigned long long)frame_off[b]);
    int T = (int)((int64_t)em_uniform((unsigned long long)frame_off[b + 1]) - f_utt);
    int k;                                                   // tile within the utterance
    unsigned long long m0 = u0, m1 = u1;                     // states of that tile not yet scored
    {
        const int per_tile = __popcll(u0) + __popcll(u1);
        const int64_t local = p - (int64_t)em_uniform((unsigned long long)pair_first[b]);
        k = (int)(local / per_tile);
        for (int skip = (int)(local - (int64_t)k * per_tile); skip > 0; --skip) {
            if (m0) m0 &= m0 - 1; else m1 &= m1 - 1;
        }
    }

    double x[4][LSR_KSTEPS];
    while (p < p_end) {
        if ((m0 | m1) == 0) {                 // tile finished (p < p_end <= n_pairs: some later tile has states left)
            if (64 * (++k) >= T) {            // utterance finished
                ++b;
                u0 = em_uniform(utt_mask[2 * b]);
                u1 = em_uniform(utt_mask[2 * b + 1]);
                f_utt = (int64_t)em_uniform((unsigned long long)frame_off[b]);
                T = (int)((int64_t)em_uniform((unsigned long long)frame_off[b + 1]) - f_utt);
                k = T > 0 ? 0 : -1;           // an utterance without frames has no tile: the test above moves on
            }
            m0 = k >= 0 ? u0 : 0;
            m1 = k >= 0 ? u1 : 0;
            continue;
        }
        int cnt = __popcll(m0) + __popcll(m1);
        if (cnt > EM_RUN) cnt = EM_RUN;
        if ((int64_t)cnt > p_end - p) cnt = (int)(p_end - p);
        p += cnt;
        const int64_t f0 = f_utt + 64 * (int64_t)k;
        const int n_rows = T - 64 * k < 64 ? T - 64 * k : 64;          // rows of the tile that belong to the utterance
        // every run loads its tile's frames: a tile with more than EM_RUN selected states (two runs) is the rare case
        // here, and an unconditional load keeps the frame registers from being carried across the branch
        em_load_frames(x, lds[wave], feats, f0, total_frames, lane, col, grp);

        constexpr int SLOTS = em_ring<TRI>::SLOTS, RING = em_ring<TRI>::RING;
        const int a_idx = (lane >> 4) * 4 + (lane & 3);
        auto pop = [&]() -> int {
            int s;
            if (m0) {
                s = __builtin_ctzll(m0);
                m0 &= m0 - 1;
            } else {
                s = 64 + __builtin_ctzll(m1);
                m1 &= m1 - 1;
            }
            return s;
        };
        int s = pop();
        v2d w[RING / 2];
        {
            const v2d* src = (const v2d*)wfrag + (size_t)s * (SLOTS * 8) + a_idx;
#pragma unroll
            for (int v = 0; v < RING / 2; ++v) w[v] = src[v * 16];
        }
        int my_col = 0;                                        // column this lane writes out: the lane-th state of the run
#pragma nounroll
        for (int i = 0; i < cnt; ++i) {
            const int s_next = i + 1 < cnt ? pop() : s;        // after the last state the ring refills with places it never uses
            const v2d* cur = (const v2d*)wfrag + (size_t)s * (SLOTS * 8) + a_idx;
            const v2d* nxt = (const v2d*)wfrag + (size_t)s_next * (SLOTS * 8) + a_idx;
            double ss[4];
            em_gaussian<TRI, true>(w, x, cur, nxt, ss);
            const double lp = em_fold(ss, lognorm[s]);
            tile[lane][(i ^ lane) & (EM_RUN - 1)] = (float)lp;
            if (lane == i) my_col = s;
            s = s_next;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < cnt) {
            float* dst = ll + f0 * ll_stride + my_col;
#pragma unroll 8
            for (int r = 0; r < 64; ++r)
                if (r < n_rows) dst[(int64_t)r * ll_stride] = tile[r][(lane ^ r) & (EM_RUN - 1)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// states visited by the chain of each utterance and the number of (tile, state) pairs it owns: one wavefront
// per utterance
__global__ __launch_bounds__(64) void lsr_chain_utt_mask_kernel(const int64_t* __restrict__ chain_off,
                                                                const int32_t* __restrict__ chain_state, int n_cols,
                                                                const int64_t* __restrict__ frame_off,
                                                                unsigned long long* __restrict__ utt_mask,
                                                                int32_t* __restrict__ utt_pairs) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t c0 = chain_off[b], c1 = chain_off[b + 1];
    unsigned long long m0 = 0, m1 = 0;
    for (int64_t i = c0 + lane; i < c1; i += 64) {
        const int s = chain_state[i];
        if (s >= 0 && s < n_cols) {
            if (s < 64) m0 |= 1ull << s; else m1 |= 1ull << (s - 64);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        m0 |= __shfl_xor(m0, o);
        m1 |= __shfl_xor(m1, o);
    }
    if (lane == 0) {
        utt_mask[2 * b] = m0;
        utt_mask[2 * b + 1] = m1;
        const int64_t T = frame_off[b + 1] - frame_off[b];
        utt_pairs[b] = T > 0 ? (int32_t)((T + 63) / 64) * (__popcll(m0) + __popcll(m1)) : 0;
    }
}

// exclusive running sum of the utterances' pair counts (one workgroup: every thread a contiguous piece, the
// pieces' totals scanned in LDS); out has n + 1 entries
__global__ __launch_bounds__(1024) void lsr_chain_scan_kernel(const int32_t* __restrict__ count, int64_t n,
                                                              int64_t* __restrict__ out) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t a = tid * per < n ? tid * per : n, b = a + per < n ? a + per : n;
    int64_t sum = 0;
    for (int64_t i = a; i < b; ++i) sum += count[i];
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = tid ? part[tid - 1] : 0;
    for (int64_t i = a; i < b; ++i) {
        out[i] = run;
        run += count[i];
    }
    if (tid == 1023) out[n] = part[1023];
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

int lsr_launch_emission_chains(lsr_ctx* ctx, const float* feats_d, int64_t total_frames, float* ll_d,
                               const int64_t* frame_off_d, int n_utts, const int64_t* chain_off_d,
                               const int32_t* chain_state_d) {
    if (ctx->n_mix > 1) return lsr_launch_emission(ctx, feats_d, total_frames, ll_d);      // mixtures: all states
    const int n_cols = ctx->n_states;
    // [utterance masks 16 B][first pair of each utterance 8 B x (utterances + 1)][pairs of each utterance 4 B]
    const size_t need = (size_t)n_utts * 16 + (size_t)(n_utts + 1) * 8 + (size_t)n_utts * 4;
    int rc = lsr_reserve(ctx, &ctx->ws_mask, need, "state masks of the alignment chains");
    if (rc) return rc;
    unsigned long long* utt_mask = (unsigned long long*)ctx->ws_mask.ptr;
    int64_t* pair_first = (int64_t*)(utt_mask + (size_t)n_utts * 2);
    int32_t* utt_pairs = (int32_t*)(pair_first + n_utts + 1);
    lsr_prof_begin(ctx, LSR_K_EMISSION);
    hipLaunchKernelGGL(lsr_chain_utt_mask_kernel, dim3((unsigned)n_utts), dim3(64), 0, ctx->stream, chain_off_d,
                       chain_state_d, n_cols, frame_off_d, utt_mask, utt_pairs);
    hipLaunchKernelGGL(lsr_chain_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, utt_pairs, (int64_t)n_utts, pair_first);
    // the number of pairs is only known on the device: the full persistent grid (at most one wavefront per pair of
    // the all-states form), surplus wavefronts leave at once
    int64_t blocks = (((total_frames + 63) / 64 + n_utts) * n_cols + EM_WAVES - 1) / EM_WAVES;
    if (blocks > 3 * (int64_t)ctx->n_cus) blocks = 3 * (int64_t)ctx->n_cus;
    void (*kern)(const float*, int64_t, const double*, const double*, float*, int, const int64_t*, int,
                 const unsigned long long*, const int64_t*) =
        ctx->whiten_triangular ? lsr_emission_masked_kernel<true> : lsr_emission_masked_kernel<false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * EM_WAVES), 0, ctx->stream, feats_d, total_frames,
                       ctx->d_wfrag, ctx->d_lognorm, ll_d, lsr_ctx_ll_stride(ctx), frame_off_d, n_utts, utt_mask, pair_first);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, "lsr_emission_masked_kernel");
    return LSR_OK;
}
