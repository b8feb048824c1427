// Internal declarations shared by the translation units of liblsr_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "lsr_hip.h"

#define LSR_KPAD 40          // feature dimension padded to a multiple of the MFMA k-step (4)
#define LSR_JBLOCKS 10       // whitened components padded to 10 blocks of 4
#define LSR_TRI_SLOTS 56     // blocks per state of a triangular whitening matrix (55) rounded up
#define LSR_KSTEPS 10        // LSR_KPAD / 4
#define LSR_GRAPH_STRIDE 128 // row stride of the loop-graph tables (band, flags, word maps)
// bytes of the back-pointer table of the one-wavefront Viterbi kernel (graphs / chains of at most 64 states): one
// 256-byte group per four frames, every utterance starting a group of its own
#define LSR_TRACER_BYTES(total_frames, n_utts) ((((size_t)(total_frames) >> 2) + (size_t)(n_utts)) * 256)

struct lsr_buffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

struct lsr_ctx {
    int device = 0;
    int n_cus = 256;                 // compute units of the device (sizes persistent grids)
    hipStream_t stream = nullptr;
    std::string error;

    std::unordered_map<void*, size_t> table_bytes;   // size of every table allocated by upload()

    // front-end tables (device)
    double* d_window = nullptr;      // [320]
    double* d_tw_stage = nullptr;    // [11][16][2]  W320^(n2 k1): twiddles between the two FFT stages
    float* d_slot_w = nullptr;       // [64][6] mel weights of each slot (a run of <= 6 bins of one band)
    int* d_slot_info = nullptr;      // [64] first bin of each slot
    int* d_band_slots = nullptr;     // [40][2] first slot and slot count of each band
    double* d_dct = nullptr;         // [13][40]

    // model (device)
    int n_states = 0;
    double* d_wfrag = nullptr;       // [S][blocks in order of use][4 k][4 rows]  MFMA A operands of [U; -(mean U)]
    bool whiten_triangular = false;  // every whitening matrix is zero above the diagonal
    double* d_lognorm = nullptr;     // [S]
    int n_mix = 1;                   // components per state (n_states counts components)
    double* d_logw = nullptr;        // [S] log mixture weights (n_mix > 1 only)

    // models kept on the device besides the current one (lsr_model_slot): the fields above belong to slot cur_model
    struct model_slot {
        int n_states = 0, n_mix = 1;
        bool whiten_triangular = false;
        double *d_wfrag = nullptr, *d_lognorm = nullptr, *d_logw = nullptr;
    };
    model_slot models[LSR_MODEL_SLOTS];
    int cur_model = 0;

    int chain_emission_mode = 0;     // LSR_CHAIN_EMISSION_*

    // loop graph (device)
    int g_states = 0, g_words = 0, g_silence = -1;
    float* d_band = nullptr;         // [3][128]
    uint8_t* d_flags = nullptr;      // [128] bit0 = word entry state, bit1 = word exit state
    uint8_t* d_word_of = nullptr;    // [128] word index of each state
    uint8_t* d_lower_of = nullptr;   // [128] first state of the state's word
    uint8_t* d_upper_of = nullptr;   // [128] last state of the state's word
    double between = -5.298317366548036;   // log(0.005)
    int between_is_f64 = 1;

    // workspace
    lsr_buffer ws_frame_utt, ws_mel, ws_uttmax, ws_feats, ws_ll, ws_tracer, ws_slab, ws_sep, ws_frame_info, ws_sort, ws_mask;

    int host_check_finite = 1;       // lsr_host_check_finite: bit 0 float32 audio of lsr_recognize_host, bit 1 4-byte arrays of lsr_upload_host

    // host -> device pipeline of lsr_recognize_host (created on first use)
    struct lsr_pipeline* pipeline = nullptr;

    // RCCL communicator of lsr_comm_init (multi-GPU exchange steps)
    struct lsr_comm* comm = nullptr;

    // profiling
    bool profiling = false;
    int profile_mask = 0;            // argument of lsr_profile_enable: bit 0 = every kernel, bit 1 + k = kernel k
    bool prof_open = false;          // lsr_prof_begin recorded a start event that lsr_prof_end has to close
    struct pending { int kernel; hipEvent_t start, stop; };
    std::vector<pending> pend;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[LSR_K_COUNT] = {0};
    int64_t prof_n[LSR_K_COUNT] = {0};
};

int lsr_fail(lsr_ctx* ctx, int code, const char* fmt, ...);
int lsr_reserve(lsr_ctx* ctx, lsr_buffer* buf, size_t bytes, const char* what);
void lsr_prof_begin(lsr_ctx* ctx, int kernel);
void lsr_prof_end(lsr_ctx* ctx);
int lsr_frontend_init(lsr_ctx* ctx);
void lsr_frontend_free(lsr_ctx* ctx);
// grows every workspace buffer of the audio -> labels path to a batch of n_utts utterances / total_frames frames
int lsr_reserve_recognize(lsr_ctx* ctx, int n_utts, int64_t total_frames);
void lsr_pipeline_destroy(lsr_ctx* ctx);
int lsr_noalloc_init(lsr_ctx* ctx);      // resolves the device address of the static workspace (once per device)

#define LSR_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return lsr_fail((ctx), LSR_ERR_HIP, "%s failed: %s (%s:%d)", #expr,               \
                            hipGetErrorString(_e), __FILE__, __LINE__);                       \
    } while (0)

#define LSR_LAUNCH_CHECK(ctx, name)                                                           \
    do {                                                                                      \
        hipError_t _e = hipGetLastError();                                                    \
        if (_e != hipSuccess)                                                                 \
            return lsr_fail((ctx), LSR_ERR_HIP, "launch of %s failed: %s", name,              \
                            hipGetErrorString(_e));                                           \
    } while (0)

#if defined(__HIPCC__)
// Maximum of a float over the 64 lanes of a wavefront, returned to every lane: a running maximum
// inside each row of 16 lanes, row totals carried into the following rows, the grand total read
// from lane 63.  The lane move is the DPP source operand of v_max_f32 (a lane without a source is
// left as it is); the s_nop covers the two wait states between a vector write and a DPP read of
// the same register.  Inputs must not be NaN.
__device__ __forceinline__ float lsr_wave_max_f32(float v) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#endif

// launchers implemented in the kernel translation units
int lsr_launch_mfcc(lsr_ctx* ctx, const void* audio_d, int audio_is_i16, const int64_t* sample_off_d,
                    const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d);
// intermediate buffers of the front-end: frame records int4[frames], utterance of every frame
// int32[frames], mel energies float[frames][40], utterance maxima uint32[utterances]
struct lsr_mfcc_ws {
    void* frame_info;
    int32_t* frame_utt;
    float* mel;
    unsigned int* uttmax;
};
int lsr_launch_mfcc_ws(lsr_ctx* ctx, const void* audio_d, int audio_is_i16, const int64_t* sample_off_d,
                       const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d,
                       const lsr_mfcc_ws& ws);
// Row stride (floats) of emission score matrices for the uploaded model: 64, or 128 when the model has more than
// 64 states (columns beyond the states are -inf either way).
static inline int lsr_ctx_ll_stride(const lsr_ctx* ctx) {
    return ctx->n_states / (ctx->n_mix > 0 ? ctx->n_mix : 1) > 64 ? 128 : 64;
}
int lsr_launch_emission(lsr_ctx* ctx, const float* feats_d, int64_t total_frames, float* ll_d);
// emission scores of the states the alignment chains can visit only (per 64-frame tile); other columns keep
// whatever they held
int lsr_launch_emission_chains(lsr_ctx* ctx, const float* feats_d, int64_t total_frames, float* ll_d,
                               const int64_t* frame_off_d, int n_utts, const int64_t* chain_off_d,
                               const int32_t* chain_state_d);
int lsr_launch_viterbi_loop(lsr_ctx* ctx, const float* ll_d, const int64_t* frame_off_d, int n_utts,
                            int64_t total_frames, int max_frames, float* scores_d, int8_t* paths_d,
                            uint8_t* labels_d, int32_t* n_labels_d, int label_cap, int8_t* tracer_d = nullptr);
int lsr_launch_viterbi_chain(lsr_ctx* ctx, const float* ll_d, const int64_t* frame_off_d, int n_utts,
                             int64_t total_frames, int max_frames, const int64_t* chain_off_d,
                             int64_t total_chain, const int32_t* chain_state_d,
                             const float* chain_band_d, const uint8_t* chain_start_d,
                             float* scores_d, int8_t* paths_d, float* final_d, bool wide = false);
int lsr_launch_utt_index(lsr_ctx* ctx, const int64_t* off_d, int n_utts, int64_t total, int32_t* out_d);
int lsr_launch_stats(lsr_ctx* ctx, const float* feats_d, const int32_t* slot_d, int64_t total_frames,
                     int n_slots, double* count_d, double* sum_x_d, double* sum_xx_d);
