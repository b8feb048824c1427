// Viterbi recursion, back-trace and label extraction: one wavefront per utterance, one lane per
// state (at most 64 states; a second, two-wavefront form further down takes 65..127 states, the
// most the reference's int8 back-pointers can address).
//
// Recursion for state s at frame t (all float32 adds, first maximal candidate wins):
//   interior / chain states: best = max_{d=2,1,0} band[d][s] + prev[s-d]
//   word-entry states (loop graph): candidates are  between + prev[last state of word w]  for
//     every word w in ascending order, then the self loop band[0][s] + prev[s]
//   cur[s] = best + ll[t][state(s)]
// Back-pointers go to a global table, four frames of a state to a dword; the back-trace reads it
// back sixty-four frames at a time into registers and follows the pointers with lane reads.  The
// returned path reproduces the reference convention: entry T-1 repeats the predecessor chosen at
// the last frame, the terminal state itself is never written.
#include "lsr_internal.h"

#define VIT_LOOP32 0
#define VIT_LOOP64 1
#define VIT_CHAIN 2
#define VIT_PREFETCH 8

struct vit_args {
    const float* ll;
    int ll_stride;     // floats per row of ll (64, or 128 for models above 64 states)
    const int64_t* frame_off;
    // loop graph
    const float* band;
    const uint8_t* flags;
    const uint8_t* word_of;
    const uint8_t* lower_of;
    const uint8_t* upper_of;
    int n_states;
    int silence;
    float between32;
    double between64;
    // per-utterance chains
    const int64_t* chain_off;
    int64_t total_chain;
    const int32_t* chain_state;
    const float* chain_band;
    const uint8_t* chain_start;
    // outputs
    float* scores;
    int8_t* paths;
    uint8_t* labels;
    int32_t* n_labels;
    int label_cap;
    float* final_ll;
    int8_t* tracer;
    int path_cap;      // bytes reserved for the path in LDS (multiple of 64)
};

// Cross-lane moves through the DPP path of the vector ALU (no LDS round trip).  CTRL: 0x111..0x11F
// row_shr:1..15 (within rows of 16 lanes), 0x138 wave_shr:1, 0x142 row_bcast:15, 0x143 row_bcast:31.
// Lanes without a source keep `fill`.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp(float fill, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xF, false));
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp(double fill, double v) {
    const long long f = __builtin_bit_cast(long long, fill), x = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp((int)f, (int)x, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(x >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ float lane63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ double lane63(double v) {
    const long long x = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)x, 63), hi = __builtin_amdgcn_readlane((int)(x >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double vmax(double a, double b) { return fmax(a, b); }

// maximum over the 64 lanes, returned to every lane: inclusive max-scan inside each row of 16,
// row totals carried to the next rows, grand total read from lane 63
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
    const T ninf = (T)(-INFINITY);
    v = vmax(v, dpp<0x111>(ninf, v));
    v = vmax(v, dpp<0x112>(ninf, v));
    v = vmax(v, dpp<0x114>(ninf, v));
    v = vmax(v, dpp<0x118>(ninf, v));
    v = vmax(v, dpp<0x142, 0xA>(ninf, v));
    v = vmax(v, dpp<0x143, 0xC>(ninf, v));
    return lane63(v);
}

// float32 form: the lane move folded into the maximum (lsr_wave_max_f32); scores are never NaN
template <>
__device__ __forceinline__ float wave_max<float>(float v) {
    return lsr_wave_max_f32(v);
}

// Value of lane - 1 (lane 0 reads 0.0f, which the -inf transition terms of lanes 0 and 1 discard).  With a zero
// fill the compiler can fold the lane move into the instruction that consumes it.
__device__ __forceinline__ float shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}

// The head of a loop-graph step in one block: the candidates v0 = a0 + prev, v1 = a1 + prev[lane-1], v2 = a2 + prev[lane-2],
// the word-exit candidate cand = exit_add + prev and the maximum of cand over the 64 lanes (left in lane 63 of t).
// The maximum exchanges inside quads, half rows and rows, then the row totals travel to the following rows; a DPP
// read needs two wait states after the vector write of its source, and the independent additions fill them where
// there are any.  Lanes 0 / 1 read 0.0f for the missing neighbours (their a1 / a2 are -inf).  exit_add is a quiet NaN
// on the lanes that do not compete: v_max passes over a quiet NaN operand.
__device__ __forceinline__ void vit_head(float prev, float exit_add, float a0, float a1, float a2, float& cand, float& t,
                                         float& v0, float& v1, float& v2) {
    float p1;
    asm("v_add_f32 %[cand], %[xa], %[prev]\n\t"
        "v_add_f32 %[v0], %[a0], %[prev]\n\t"
        "v_mov_b32_dpp %[p1], %[prev] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_max_f32_dpp %[t], %[cand], %[cand] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32 %[v1], %[a1], %[p1]\n\t"
        "s_nop 0\n\t"
        "v_max_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[v2], %[p1], %[a2] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_max_f32_dpp %[t], %[t], %[t] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[t], %[t], %[t] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[t], %[t], %[t] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[t], %[t], %[t] row_bcast:31 row_mask:0xc bank_mask:0xf"
        : [cand] "=&v"(cand), [v0] "=&v"(v0), [p1] "=&v"(p1), [t] "=&v"(t), [v1] "=&v"(v1), [v2] "=&v"(v2)
        : [prev] "v"(prev), [xa] "v"(exit_add), [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2));
}

// per-lane constants of the recursion
struct vit_lane {
    // Quiet NaNs switch a lane out of a comparison or a maximum without an instruction: v_max passes over a quiet NaN
    // operand, and no value compares equal to one.
    float a0, a1, a2;      // log a(s-d -> s), d = 0, 1, 2: -inf where lane - d does not exist; on word-entry lanes of the
                           // float32-penalty form a2 = NaN (v2 never wins) and a1 is unused (the best exit replaces v1)
    float exit_add;        // float32-penalty form: the penalty on word-exit lanes, NaN elsewhere (only exits compete)
    float ninf;            // -inf; NaN on word-entry lanes of the float32-penalty form (their unreachable case points at
                           // the lowest exit through v1, not at state 0)
    int lm2;               // lane - 2
    int lm1z;              // lane - 1; 0 on word-entry lanes
    int emask;             // all ones on word-entry lanes, 0 elsewhere
    bool is_entry, is_exit;
};

// One step of the recursion for every lane: returns the new score, sets the back-pointer (0..63).
template <int MODE>
__device__ __forceinline__ float vit_step(const vit_args& A, const vit_lane& L, float prev, float e, int lane, int& ptr) {
    float v0, v1, v2;
    int via1 = L.lm1z;
    if (MODE == VIT_LOOP32) {
        // A word-entry lane has no predecessor inside its word (v1 = v2 = -inf).  The best word exit takes the place
        // of v1 there: the exits precede the self loop in the reference's candidate list, so they win ties exactly as
        // v1 does.  An entry lane that nothing reaches points at the lowest exit (the first of equal candidates).
        float cand, t;
        vit_head(prev, L.exit_add, L.a0, L.a1, L.a2, cand, t, v0, v1, v2);
        const float m = lane63(t);
        const int first = __builtin_ctzll(__builtin_amdgcn_ballot_w64(cand == m));      // some exit lane holds the maximum
        v1 = L.is_entry ? m : v1;
        via1 |= first & L.emask;
    } else {
        const float p1 = shr1(prev);          // prev of lane - 1
        const float p2 = shr1(p1);            // prev of lane - 2
        v0 = L.a0 + prev;
        v1 = L.a1 + p1;
        v2 = L.a2 + p2;
    }
    if (MODE == VIT_LOOP64) {
        const float best = fmaxf(fmaxf(v2, v1), v0);
        ptr = (v2 == best) ? L.lm2 : ((v1 == best) ? L.lm1z : lane);
        if (best == -INFINITY) ptr = 0;
        const double cand = L.is_exit ? A.between64 + (double)prev : -(double)INFINITY;
        const double m = wave_max(cand);
        const unsigned long long mask = __ballot(L.is_exit && cand == m);
        const int first = __ffsll((long long)mask) - 1;
        double bestd = (double)best;
        if (L.is_entry) {
            const bool cross = m >= (double)v0;
            bestd = cross ? m : (double)v0;
            ptr = cross ? first : lane;
        }
        return (float)(bestd + (double)e);
    }
    const float best = fmaxf(fmaxf(v2, v1), v0);      // a quiet NaN operand (v2 of an entry lane) is passed over
    int p = (v1 == best) ? via1 : lane;
    p = (v2 == best) ? L.lm2 : p;
    ptr = (best == L.ninf) ? 0 : p;
    return best + e;
}

// ModelBoundary.get_labels for one utterance, evaluated by the 64 lanes of a wavefront: the rule is local
// (a word is emitted at frame 0, where the word changes, and where the path jumps from a word's last
// state back to its first), so every lane tests its own frame and the emitted words are compacted with
// ballot / popcount into the fixed-width record.  States outside the graph (-1: no back-pointer) emit nothing.
__device__ __forceinline__ void vit_labels(const vit_args& A, const int8_t* __restrict__ path, int T, int b, int lane) {
    int count = 0;
    for (int base = 0; base < T; base += 64) {
        const int t = base + lane;
        bool emit = false;
        int w = 0;
        if (t < T) {
            const int p = path[t];
            if (p >= 0 && p < A.n_states) {
                w = A.word_of[p];
                if (t == 0) {
                    emit = true;
                } else {
                    const int q = path[t - 1];
                    if (q >= 0 && q < A.n_states && p != q) {
                        const int wq = A.word_of[q];
                        emit = (w != wq) || (q == A.upper_of[q] && p == A.lower_of[q]);
                    }
                }
                if (w == A.silence) emit = false;
            }
        }
        const unsigned long long mask = __ballot(emit);
        if (emit && A.labels) {
            const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
            if (pos < A.label_cap) A.labels[(int64_t)b * A.label_cap + pos] = (uint8_t)w;
        }
        count += __popcll(mask);
    }
    if (A.n_labels && lane == 0) A.n_labels[b] = count;
}

// Back-pointer table of the one-wavefront kernel.  Frames are counted from the END of their utterance (u = T-1-t),
// four to a group; a group is one dword per lane (= state) whose byte u & 3 is the pointer of (frame t, state).
// Utterance b owns the groups from (frame_off[b] >> 2) + b on: [(total_frames >> 2) + n_utts][64] dwords in all.
// The recursion keeps a group in a register and stores it when its last frame (u & 3 == 0) is done; the back-trace
// loads sixteen groups (64 frames) into registers and walks them with lane reads, no LDS staging.
// (LSR_TRACER_BYTES in lsr_internal.h)

// one frame of the walk: the pointer of (row u, state x) becomes x, and is noted as the path entry of the frame before
#define VIT_WALK(k)                                                                                   \
    {                                                                                                 \
        const unsigned word = (unsigned)__builtin_amdgcn_readlane((int)w[(k) >> 2], x);               \
        x = (int)((word >> (8 * ((k) & 3))) & 0xffu);                                                 \
        asm("v_writelane_b32 %0, %1, %2" : "+v"(pv) : "s"(x), "n"(k));                                \
    }

// ROWB: bytes per row of the score matrix (256 or 512).  The score loads run up to fifteen rows past the end of an
// utterance (their values are not used): the score matrix is allocated with that margin.
template <int MODE, int ROWB>
__global__ __launch_bounds__(64) void lsr_viterbi_kernel(vit_args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int8_t* pathbuf = (int8_t*)smem;

    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int64_t off = A.frame_off[b];
    const int T = (int)(A.frame_off[b + 1] - off);
    if (T < 1) return;
    // Sizes the host promised but the device tables contradict (an utterance longer than max_frames would
    // overrun the LDS path buffer; a chain outside 1..64 states has no lane mapping): the utterance gets
    // the sentinels score = NaN and n_labels = -1 and nothing else is written for it.
    bool bad = T > A.path_cap;
    if (MODE == VIT_CHAIN) {
        const int64_t len = A.chain_off[b + 1] - A.chain_off[b];
        bad = bad || len < 1 || len > 64;
    }
    if (bad) {
        if (lane == 0) {
            if (A.scores) A.scores[b] = __builtin_nanf("");
            if (A.n_labels) A.n_labels[b] = -1;
        }
        return;
    }

    // ---- per-lane graph constants -------------------------------------------------------------
    int n, gid;
    vit_lane L;
    bool is_start;
    L.is_entry = L.is_exit = false;
    if (MODE == VIT_CHAIN) {
        const int64_t c0 = A.chain_off[b];
        n = (int)(A.chain_off[b + 1] - c0);
        const bool live = lane < n;
        const int64_t i = c0 + (live ? lane : 0);
        gid = A.chain_state[i];          // idle lanes read the column of the chain's first state: one that has scores
        L.a0 = live ? A.chain_band[i] : -INFINITY;
        L.a1 = live ? A.chain_band[A.total_chain + i] : -INFINITY;
        L.a2 = live ? A.chain_band[2 * A.total_chain + i] : -INFINITY;
        is_start = live && A.chain_start[i] != 0;
    } else {
        n = A.n_states;
        const bool live = lane < n;
        gid = live ? lane : 0;
        L.a0 = A.band[lane];
        L.a1 = A.band[LSR_GRAPH_STRIDE + lane];
        L.a2 = A.band[2 * LSR_GRAPH_STRIDE + lane];
        const unsigned f = A.flags[lane];
        L.is_entry = live && (f & 1);
        L.is_exit = live && (f & 2);
        is_start = L.is_entry;
        if (L.is_entry) L.a1 = L.a2 = -INFINITY;      // an entry lane's candidates are the word exits and its self loop
    }
    if (lane < 1) L.a1 = -INFINITY;
    if (lane < 2) L.a2 = -INFINITY;
    if (MODE == VIT_LOOP32 && L.is_entry) L.a2 = __builtin_nanf("");
    L.exit_add = L.is_exit ? A.between32 : __builtin_nanf("");
    L.ninf = (MODE == VIT_LOOP32 && L.is_entry) ? __builtin_nanf("") : -INFINITY;
    L.lm2 = lane - 2;
    L.lm1z = L.is_entry ? 0 : lane - 1;
    L.emask = L.is_entry ? -1 : 0;

    // Wave-uniform buffer descriptors (base of the utterance's score rows / of its back-pointer groups); the per-lane
    // part of an address is a 32-bit byte offset and the frame / group part a scalar one, so no address arithmetic is
    // left on the vector unit.
    const __amdgpu_buffer_rsrc_t ll_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(A.ll + off * (ROWB / 4)), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t tr_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)((char*)A.tracer + ((off >> 2) + b) * 256), 0, -1, 0x00020000);
    const int gid4 = gid * 4, lane4 = lane * 4;
    // frame t0 + i of the scores: the frame part is scalar, the fixed part i an immediate
#define VIT_LL(t0, i) __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ll_rsrc, gid4 + (i) * ROWB, (t0) * ROWB, 0))
#define VIT_GROUP_LOAD(g) ((unsigned)__builtin_amdgcn_raw_buffer_load_b32(tr_rsrc, lane4, (g) * 256, 0))
#define VIT_GROUP_STORE(g, v) __builtin_amdgcn_raw_buffer_store_b32((int)(v), tr_rsrc, lane4, (g) * 256, 0)
    asm volatile("" : "+v"(L.emask));      // keeps the and-or form of the pointer select (one instruction)

    // frame 0: start states get emission + self-loop term
    float prev = is_start ? VIT_LL(0, 0) + L.a0 : -INFINITY;

    // ---- recursion: frames 1 .. T-1 ---------------------------------------------------------------
    // The first (T-1) & 7 frames are taken one by one, so that the rest is whole chunks of eight whose places in
    // the groups of four are fixed.  Emission scores travel eight frames ahead in a register ring: a place is
    // requested again as soon as the step that used it is done.
    const int head = (T - 1) & 7;
    float eh[7], e[8];
#pragma unroll
    for (int i = 0; i < 7; ++i) eh[i] = VIT_LL(1, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = VIT_LL(1 + head, i);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (i < head) {
            int ptr;
            prev = vit_step<MODE>(A, L, prev, eh[i], lane, ptr);
            acc = (acc << 8) | (unsigned)ptr;
            const int u = T - 2 - i;
            if ((u & 3) == 0) {
                VIT_GROUP_STORE(u >> 2, acc);
                acc = 0;
            }
        }
    }
    for (int t0 = 1 + head; t0 < T; t0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int ptr;
            prev = vit_step<MODE>(A, L, prev, e[i], lane, ptr);
            e[i] = VIT_LL(t0 + 8, i);
            acc = (i & 3) ? ((acc << 8) | (unsigned)ptr) : (unsigned)ptr;
            if ((i & 3) == 3) VIT_GROUP_STORE((T - 1 - (t0 + i)) >> 2, acc);
        }
    }

    // ---- termination ---------------------------------------------------------------------------
    int term;
    float score;
    if (MODE == VIT_CHAIN) {
        term = n - 1;
        score = __shfl(prev, term);
        if (A.final_ll) A.final_ll[(int64_t)b * 64 + lane] = prev;
    } else {
        const float ends = L.is_exit ? prev : -INFINITY;
        score = wave_max(ends);
        const unsigned long long mask = __ballot(L.is_exit && ends == score);
        term = __ffsll((long long)mask) - 1;
        if (term < 0) term = 0;
    }
    if (A.scores && lane == 0) A.scores[b] = score;
    if (!A.paths && !A.labels && !A.n_labels) return;

    // ---- back-trace ----------------------------------------------------------------------------
    // x = state at frame T-1-u; row u maps it to the state at the frame before, which is the path entry of that frame
    // (path[T-1] repeats path[T-2]: the terminal state itself is never written).
    __syncthreads();      // back-pointer stores of this wave are complete and visible to it
    const int rows = T - 1;
    const int last_group = ((rows + 3) >> 2) - 1;
    int x = __builtin_amdgcn_readfirstlane(term);
    if (rows == 0 && lane == 0) pathbuf[0] = -1;
    unsigned w[16], wn[16];
    if (rows > 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = VIT_GROUP_LOAD(j < last_group ? j : last_group);
    }
    for (int u0 = 0; u0 < rows; u0 += 64) {
        if (u0 + 64 < rows) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int g = (u0 >> 2) + 16 + j;
                wn[j] = VIT_GROUP_LOAD(g < last_group ? g : last_group);
            }
        }
        int pv = 0;
        if (u0 + 64 <= rows) {
#pragma unroll
            for (int k = 0; k < 64; ++k) VIT_WALK(k)
        } else {
#pragma unroll
            for (int k = 0; k < 64; ++k)
                if (u0 + k < rows) VIT_WALK(k)
        }
        const int pos = T - 2 - u0 - lane;      // the frame whose state lane k = u - u0 holds
        if (pos >= 0) pathbuf[pos] = (int8_t)pv;
        if (u0 == 0 && lane == 0) pathbuf[T - 1] = (int8_t)pv;
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = wn[j];
    }
    __syncthreads();
    if (A.paths)
        for (int t = lane; t < T; t += 64) A.paths[off + t] = pathbuf[t];

    // ---- path -> word labels (loop graph only) ---------------------------------------------------
    if (MODE != VIT_CHAIN && (A.labels || A.n_labels)) vit_labels(A, pathbuf, T, b, lane);
#undef VIT_LL
#undef VIT_GROUP_LOAD
#undef VIT_GROUP_STORE
}

// ---- graphs and chains of 65..127 states: two wavefronts per utterance -------------------------------------
// Same recursion, back-trace and label pass with thread = state over 128 threads.  The neighbour scores
// prev[s-1], prev[s-2] and the per-wavefront word-exit maxima cross the wavefront boundary through LDS
// (double-buffered by frame parity: one barrier per frame).  The back-pointer table has 128 columns.
#define VW_PREV_LD 132

template <typename T>
__device__ __forceinline__ void vw_combine(T m0, int f0, T m1, int f1, T& m, int& first) {
    // the lowest state among equal maxima wins: wavefront 0 holds the lower states
    const bool take0 = f0 >= 0 && (f1 < 0 || m0 >= m1);
    m = take0 ? m0 : m1;
    first = take0 ? f0 : f1;
}

template <int MODE>
__global__ __launch_bounds__(128) void lsr_viterbi_wide_kernel(vit_args A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int8_t* pathbuf = (int8_t*)smem;
    int8_t* trblk = (int8_t*)smem + A.path_cap;                            // [64][128]
    float* prevbuf = (float*)(smem + A.path_cap + 64 * 128);                // [2][VW_PREV_LD], entry 2 + s = state s
    double* redm = (double*)(prevbuf + 2 * VW_PREV_LD);                     // [2][2] per-wavefront exit maximum
    int* redi = (int*)(redm + 4);                                           // [2][2] state that holds it (-1: none)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x;
    const int64_t off = A.frame_off[b];
    const int T = (int)(A.frame_off[b + 1] - off);
    if (T < 1) return;
    bool bad = T > A.path_cap;
    if (MODE == VIT_CHAIN) {
        const int64_t len = A.chain_off[b + 1] - A.chain_off[b];
        bad = bad || len < 1 || len > 127;
    }
    if (bad) {
        if (tid == 0) {
            if (A.scores) A.scores[b] = __builtin_nanf("");
            if (A.n_labels) A.n_labels[b] = -1;
        }
        return;
    }

    int n, gid;
    float a0, a1, a2;
    bool is_start, is_entry = false, is_exit = false;
    if (MODE == VIT_CHAIN) {
        const int64_t c0 = A.chain_off[b];
        n = (int)(A.chain_off[b + 1] - c0);
        const bool live = tid < n;
        const int64_t i = c0 + (live ? tid : 0);
        gid = A.chain_state[i];          // idle threads read the column of the chain's first state
        a0 = live ? A.chain_band[i] : -INFINITY;
        a1 = live ? A.chain_band[A.total_chain + i] : -INFINITY;
        a2 = live ? A.chain_band[2 * A.total_chain + i] : -INFINITY;
        is_start = live && A.chain_start[i] != 0;
    } else {
        n = A.n_states;
        const bool live = tid < n;
        gid = live ? tid : 0;
        a0 = A.band[tid];
        a1 = A.band[LSR_GRAPH_STRIDE + tid];
        a2 = A.band[2 * LSR_GRAPH_STRIDE + tid];
        const unsigned f = A.flags[tid];
        is_entry = live && (f & 1);
        is_exit = live && (f & 2);
        is_start = is_entry;
    }
    if (tid < 1) a1 = -INFINITY;
    if (tid < 2) a2 = -INFINITY;
    if (tid < 4) prevbuf[(tid >> 1) * VW_PREV_LD + (tid & 1)] = -INFINITY;      // the two entries below state 0

    const float* llrow = A.ll + off * A.ll_stride + gid;
    int8_t* trrow = A.tracer + off * 128 + tid;
    float prev = is_start ? llrow[0] + a0 : -INFINITY;
    float e_next = llrow[(int64_t)(T > 1 ? 1 : 0) * A.ll_stride];

    for (int t = 1; t < T; ++t) {
        const float e = e_next;
        e_next = llrow[(int64_t)(t + 1 < T ? t + 1 : t) * A.ll_stride];          // one frame ahead
        float* pb = prevbuf + (t & 1) * VW_PREV_LD;
        pb[2 + tid] = prev;
        if (MODE == VIT_LOOP32) {
            const float cand = is_exit ? A.between32 + prev : -INFINITY;
            const float m = wave_max(cand);
            const int first = __ffsll((long long)__ballot(is_exit && cand == m)) - 1;
            if (lane == 0) {
                redm[(t & 1) * 2 + wave] = (double)m;
                redi[(t & 1) * 2 + wave] = first >= 0 ? first + 64 * wave : -1;
            }
        } else if (MODE == VIT_LOOP64) {
            const double cand = is_exit ? A.between64 + (double)prev : -(double)INFINITY;
            const double m = wave_max(cand);
            const int first = __ffsll((long long)__ballot(is_exit && cand == m)) - 1;
            if (lane == 0) {
                redm[(t & 1) * 2 + wave] = m;
                redi[(t & 1) * 2 + wave] = first >= 0 ? first + 64 * wave : -1;
            }
        }
        __syncthreads();
        const float p1 = pb[1 + tid], p2 = pb[tid];
        const float v0 = a0 + prev, v1 = a1 + p1, v2 = a2 + p2;
        float best = fmaxf(fmaxf(v2, v1), v0);
        int ptr = (v2 == best) ? tid - 2 : ((v1 == best) ? tid - 1 : tid);
        if (best == -INFINITY) ptr = 0;
        float cur;
        if (MODE == VIT_CHAIN) {
            cur = best + e;
        } else {
            double m;
            int first;
            vw_combine(redm[(t & 1) * 2], redi[(t & 1) * 2], redm[(t & 1) * 2 + 1], redi[(t & 1) * 2 + 1], m, first);
            if (MODE == VIT_LOOP32) {
                if (is_entry) {
                    const float mf = (float)m;                     // exact: it is a float32 value
                    const bool cross = mf >= v0;
                    best = cross ? mf : v0;
                    ptr = cross ? first : tid;
                }
                cur = best + e;
            } else {
                double bestd = (double)best;
                if (is_entry) {
                    const bool cross = m >= (double)v0;
                    bestd = cross ? m : (double)v0;
                    ptr = cross ? first : tid;
                }
                cur = (float)(bestd + (double)e);
            }
        }
        trrow[(int64_t)t * 128] = (int8_t)ptr;
        prev = cur;
    }

    // ---- termination -------------------------------------------------------------------------------
    float* fin = prevbuf + (T & 1) * VW_PREV_LD;
    int term;
    float score;
    if (MODE == VIT_CHAIN) {
        fin[2 + tid] = prev;
        __syncthreads();
        term = n - 1;
        score = fin[2 + term];
    } else {
        const float ends = is_exit ? prev : -INFINITY;
        const float m = wave_max(ends);
        const int first = __ffsll((long long)__ballot(is_exit && ends == m)) - 1;
        if (lane == 0) {
            redm[(T & 1) * 2 + wave] = (double)m;
            redi[(T & 1) * 2 + wave] = first >= 0 ? first + 64 * wave : -1;
        }
        __syncthreads();
        double md;
        vw_combine(redm[(T & 1) * 2], redi[(T & 1) * 2], redm[(T & 1) * 2 + 1], redi[(T & 1) * 2 + 1], md, term);
        score = (float)md;
        if (term < 0) term = 0;
    }
    if (A.scores && tid == 0) A.scores[b] = score;
    if (!A.paths && !A.labels && !A.n_labels) return;

    // ---- back-trace: blocks of 64 frames staged in LDS, walked by every thread alike ---------------------
    __syncthreads();
    int cur = -1;
    for (int blk = (T - 1) >> 6; blk >= 0; --blk) {
        const int base = blk << 6;
        for (int i = tid; i < 64 * 8; i += 128) {
            const int r = i >> 3, c = i & 7;
            if (base + r < T && base + r >= 1)
                ((uint4*)(trblk + r * 128))[c] = ((const uint4*)(A.tracer + (off + base + r) * 128))[c];
        }
        __syncthreads();
        const int top = (base + 63 < T - 1) ? base + 63 : T - 1;
        for (int t = top; t >= base; --t) {
            if (t == T - 1) {
                cur = (T > 1) ? (int)trblk[(t - base) * 128 + term] : -1;
                if (tid == 0) pathbuf[t] = (int8_t)cur;
            } else {
                if (tid == 0) pathbuf[t] = (int8_t)cur;
                if (t > 0) cur = (int)trblk[(t - base) * 128 + cur];
            }
        }
        __syncthreads();
    }
    if (A.paths)
        for (int t = tid; t < T; t += 128) A.paths[off + t] = pathbuf[t];
    if (MODE != VIT_CHAIN && (A.labels || A.n_labels) && wave == 0) vit_labels(A, pathbuf, T, b, lane);
}

// The label pass on its own: paths in, label records out (one wavefront per utterance, the path is read
// from memory instead of the decoder's LDS copy).
__global__ __launch_bounds__(64) void lsr_path_labels_kernel(vit_args A) {
    const int b = blockIdx.x;
    const int64_t off = A.frame_off[b];
    const int T = (int)(A.frame_off[b + 1] - off);
    if (T < 1) {
        if (A.n_labels && threadIdx.x == 0) A.n_labels[b] = 0;
        return;
    }
    vit_labels(A, A.paths + off, T, b, threadIdx.x);
}

// tracer_d: back-pointer table of LSR_TRACER_BYTES(total_frames, n_utts) bytes, 256-byte aligned, supplied by the caller,
// or NULL to use the context's workspace.
// wide: two wavefronts per utterance and a 128-column table (graphs / chains of 65..127 states).
static int launch(lsr_ctx* ctx, int mode, vit_args& A, int n_utts, int64_t total_frames, int max_frames,
                  int8_t* tracer_d = nullptr, bool wide = false) {
    const int cols = wide ? 128 : 64;
    if (!tracer_d) {
        int rc = lsr_reserve(ctx, &ctx->ws_tracer, wide ? (size_t)total_frames * cols : LSR_TRACER_BYTES(total_frames, n_utts),
                             "back-pointers");
        if (rc) return rc;
        tracer_d = (int8_t*)ctx->ws_tracer.ptr;
    } else if (wide) {
        return lsr_fail(ctx, LSR_ERR_LIMIT, "a caller-supplied back-pointer table only serves graphs of at most 64 states");
    }
    A.tracer = tracer_d;
    A.ll_stride = lsr_ctx_ll_stride(ctx);
    A.path_cap = (max_frames + 63) / 64 * 64;
    const size_t lds = (size_t)A.path_cap +
                       (wide ? 64 * cols + 2 * VW_PREV_LD * sizeof(float) + 4 * sizeof(double) + 4 * sizeof(int) : 0);
    if (lds > 160 * 1024)
        return lsr_fail(ctx, LSR_ERR_LIMIT, "utterance of %d frames exceeds the %d-frame limit of the decoder", max_frames,
                        160 * 1024 - 64 * 128 - 2048);
    void (*kern)(vit_args);
    if (wide)
        kern = mode == VIT_LOOP32 ? lsr_viterbi_wide_kernel<VIT_LOOP32>
               : mode == VIT_LOOP64 ? lsr_viterbi_wide_kernel<VIT_LOOP64> : lsr_viterbi_wide_kernel<VIT_CHAIN>;
    else
        kern = A.ll_stride == 64
                   ? (mode == VIT_LOOP32 ? lsr_viterbi_kernel<VIT_LOOP32, 256>
                      : mode == VIT_LOOP64 ? lsr_viterbi_kernel<VIT_LOOP64, 256> : lsr_viterbi_kernel<VIT_CHAIN, 256>)
                   : (mode == VIT_LOOP32 ? lsr_viterbi_kernel<VIT_LOOP32, 512>
                      : mode == VIT_LOOP64 ? lsr_viterbi_kernel<VIT_LOOP64, 512> : lsr_viterbi_kernel<VIT_CHAIN, 512>);
    if (lds > 64 * 1024)
        LSR_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lsr_prof_begin(ctx, LSR_K_VITERBI);
    hipLaunchKernelGGL(kern, dim3((unsigned)n_utts), dim3(wide ? 128 : 64), lds, ctx->stream, A);
    lsr_prof_end(ctx);
    LSR_LAUNCH_CHECK(ctx, wide ? "lsr_viterbi_wide_kernel" : "lsr_viterbi_kernel");
    return LSR_OK;
}

int lsr_launch_viterbi_loop(lsr_ctx* ctx, const float* ll_d, const int64_t* frame_off_d, int n_utts,
                            int64_t total_frames, int max_frames, float* scores_d, int8_t* paths_d, uint8_t* labels_d,
                            int32_t* n_labels_d, int label_cap, int8_t* tracer_d) {
    vit_args A = {};
    A.ll = ll_d;
    A.frame_off = frame_off_d;
    A.band = ctx->d_band;
    A.flags = ctx->d_flags;
    A.word_of = ctx->d_word_of;
    A.lower_of = ctx->d_lower_of;
    A.upper_of = ctx->d_upper_of;
    A.n_states = ctx->g_states;
    A.silence = ctx->g_silence;
    A.between32 = (float)ctx->between;
    A.between64 = ctx->between;
    A.scores = scores_d;
    A.paths = paths_d;
    A.labels = labels_d;
    A.n_labels = n_labels_d;
    A.label_cap = label_cap;
    return launch(ctx, ctx->between_is_f64 ? VIT_LOOP64 : VIT_LOOP32, A, n_utts, total_frames, max_frames, tracer_d,
                  ctx->g_states > 64);
}

int lsr_launch_viterbi_chain(lsr_ctx* ctx, const float* ll_d, const int64_t* frame_off_d, int n_utts,
                             int64_t total_frames, int max_frames, const int64_t* chain_off_d, int64_t total_chain,
                             const int32_t* chain_state_d, const float* chain_band_d, const uint8_t* chain_start_d,
                             float* scores_d, int8_t* paths_d, float* final_d, bool wide) {
    if (wide && final_d) return lsr_fail(ctx, LSR_ERR_ARG, "the last-frame score table is only available for chains of at most 64 states");
    vit_args A = {};
    A.ll = ll_d;
    A.frame_off = frame_off_d;
    A.chain_off = chain_off_d;
    A.total_chain = total_chain;
    A.chain_state = chain_state_d;
    A.chain_band = chain_band_d;
    A.chain_start = chain_start_d;
    A.scores = scores_d;
    A.paths = paths_d;
    A.final_ll = final_d;
    return launch(ctx, VIT_CHAIN, A, n_utts, total_frames, max_frames, nullptr, wide);
}

extern "C" int lsr_labels_from_paths(lsr_ctx* ctx, const int8_t* paths_d, const int64_t* frame_off_d, int n_utts,
                                     uint8_t* labels_d, int32_t* n_labels_d, int label_cap) {
    if (!ctx) return LSR_ERR_ARG;
    if (!paths_d || !frame_off_d || !n_labels_d || n_utts < 1) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_labels_from_paths: bad argument");
    if (labels_d && label_cap < 1) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_labels_from_paths: label_cap < 1");
    if (ctx->g_states == 0) return lsr_fail(ctx, LSR_ERR_STATE, "lsr_labels_from_paths: no loop graph uploaded");
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    vit_args A = {};
    A.frame_off = frame_off_d;
    A.word_of = ctx->d_word_of;
    A.lower_of = ctx->d_lower_of;
    A.upper_of = ctx->d_upper_of;
    A.n_states = ctx->g_states;
    A.silence = ctx->g_silence;
    A.paths = const_cast<int8_t*>(paths_d);
    A.labels = labels_d;
    A.n_labels = n_labels_d;
    A.label_cap = label_cap;
    hipLaunchKernelGGL(lsr_path_labels_kernel, dim3((unsigned)n_utts), dim3(64), 0, ctx->stream, A);
    LSR_LAUNCH_CHECK(ctx, "lsr_path_labels_kernel");
    return LSR_OK;
}
