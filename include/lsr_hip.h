/*
 * lsr_hip.h - C ABI of liblsr_hip.so: MFCC front-end, Gaussian emission scores and
 * Viterbi decoding of left-to-right HMM word models on one AMD MI355X (gfx950).
 *
 * The library is the device half of a drop-in for the numeric hot path of the Python
 * package `loe_speech_recognition`.  That package has no FFI of its own; every entry point
 * below names the Python method(s) it replaces (paths relative to
 * src/loe_speech_recognition/ of the reference repository).
 *
 * Conventions
 *   - every function returns 0 on success and a negative LSR_ERR_* code on failure;
 *     lsr_last_error(ctx) returns a human-readable message for the last failure on ctx.
 *   - one context per (process, device).  A context is not thread-safe.
 *   - "_d" parameters are DEVICE pointers owned by the caller (e.g. torch tensors); all other
 *     pointers are host pointers.  Nothing is retained after a call returns except what
 *     lsr_model_upload / lsr_loop_graph_upload copy to the device.
 *   - ragged batches are one concatenated buffer plus int64 offsets[B+1] (offsets[0] = 0,
 *     non-decreasing, offsets[B] = the stated total).  Offsets live on the device and are not
 *     checked on the host; the kernels clamp them to the stated totals, so inconsistent offsets give
 *     meaningless results in the caller's own outputs, not writes outside the library's workspace.
 *     Audio and feature buffers are read through the sample / frame offsets as given.
 *   - kernels are enqueued on the context's stream (lsr_set_stream, default: the null stream);
 *     calls are asynchronous with respect to the host unless stated otherwise.
 *   - internal workspace grows on demand (hipMalloc) and is reused; a steady-state loop over
 *     batches no larger than an earlier one allocates nothing.
 *   - feature dimension is fixed at LSR_DIM = 39, at most LSR_MAX_STATES = 127 states per
 *     decoding graph / alignment chain (the reach of the reference's int8 back-pointers,
 *     hidden_markov_model.py:498-501).  Graphs and chains of at most 64 states run one wavefront
 *     per utterance; larger ones a slower two-wavefront form.
 */
#ifndef LSR_HIP_H
#define LSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSR_DIM 39
#define LSR_MAX_STATES 127
#define LSR_LL_STRIDE 64      /* row stride (floats) of emission score matrices of models with at most 64
                                 states; 128 above that: see lsr_ll_stride */
#define LSR_N_FFT 320
#define LSR_HOP 160
#define LSR_N_MELS 40
#define LSR_N_BINS 161
#define LSR_N_MFCC 13

#define LSR_OK 0
#define LSR_ERR_ARG (-1)      /* invalid argument / shape */
#define LSR_ERR_HIP (-2)      /* a HIP runtime call failed */
#define LSR_ERR_LIMIT (-3)    /* size beyond a documented limit */
#define LSR_ERR_STATE (-4)    /* call order: model or graph not uploaded yet */

typedef struct lsr_ctx lsr_ctx;

/* kernel ids for lsr_profile_get */
#define LSR_K_FRAME_INDEX 0
#define LSR_K_MEL_POWER 1
#define LSR_K_MFCC_FINISH 2
#define LSR_K_EMISSION 3
#define LSR_K_VITERBI 4
#define LSR_K_STATS 5
#define LSR_K_SEPARATION 6
#define LSR_K_COUNT 7

/* ---- context ---------------------------------------------------------------------------- */
int lsr_ctx_create(int device, lsr_ctx** out);
int lsr_ctx_destroy(lsr_ctx* ctx);
const char* lsr_last_error(const lsr_ctx* ctx);
int lsr_set_stream(lsr_ctx* ctx, void* hip_stream);   /* hipStream_t; NULL = null stream */
int lsr_synchronize(lsr_ctx* ctx);                    /* blocks until the stream is idle  */

/* Per-kernel device time, measured with HIP events on the context's stream around every
 * launch while enabled (adds two event records per launch).  on: 0 = off, 1 = every kernel,
 * LSR_PROFILE_ONLY(k) (may be or-ed) = only the launches of kernel id k, so that a timed region can carry the
 * events of one kernel without paying for those of the others.  lsr_profile_get synchronises. */
#define LSR_PROFILE_ONLY(kernel_id) (2 << (kernel_id))
int lsr_profile_enable(lsr_ctx* ctx, int on);
int lsr_profile_reset(lsr_ctx* ctx);
int lsr_profile_get(lsr_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches);

/* ---- host-only table builders (no GPU needed; used by the front-end and by tests) -------- */
void lsr_host_hann_window(double* out320);            /* periodic Hann, n_fft = 320            */
void lsr_host_mel_filterbank(float* out40x161);       /* Slaney mel, 16 kHz, 133.33-6855.4976  */
void lsr_host_dct_matrix(double* out13x40);           /* DCT-II, orthonormal, first 13 rows    */

/* ---- model ------------------------------------------------------------------------------ */
/* Replaces MultivariateNormal.from_means_covariances (hidden_markov_model.py:27-44): the
 * state of n_states frozen full-covariance Gaussians.  mean[S][39]; whiten[S][39][39] is
 * SciPy's U = V * diag(lambda^-1/2) with whiten[s][k][j] multiplying feature k into whitened
 * component j; log_norm[s] = 39*log(2*pi) + log|Sigma_s|.  All float64, host memory.  Any matrix
 * W with W W^T = Sigma^-1 gives the same scores; when every whiten[s] is zero above the diagonal
 * (whiten[s][k][j] == 0 for j > k) the device skips the all-zero 4 x 4 blocks (55 of 100). */
int lsr_model_upload(lsr_ctx* ctx, int n_states, const double* mean, const double* whiten,
                     const double* log_norm);

/* Several models resident at once.  A caller that alternates between a few model sets - the loop decoder's
 * stacked words and the isolated-word collection of scripts/project3_predict_simple.py, say, or the models of
 * successive penalties - selects a slot instead of uploading again: lsr_model_upload and lsr_model_set_mixtures
 * write the CURRENT slot, every compute entry point reads it, lsr_model_slot makes another of the
 * LSR_MODEL_SLOTS slots current (slot 0 at context creation; a slot that was never uploaded holds no model:
 * LSR_ERR_STATE from the compute entry points).  Has no counterpart in the reference, whose objects each hold
 * their own SciPy distributions (hidden_markov_model.py:283-292). */
#define LSR_MODEL_SLOTS 8
int lsr_model_slot(lsr_ctx* ctx, int slot);

/* Gaussian-mixture emissions: regroup the uploaded Gaussians as n_states / n_mix states of n_mix
 * consecutive components with log_weights[n_states] (host, float64).  Emission scores become
 * float32(logaddexp over the components of log w + log N), accumulated sequentially in float64
 * like HMM_GMM.log_emission_prob of the reference's retired GMM code
 * (deprecated/gaussian_mixture_model.py:159-164).  n_mix = 1 restores single Gaussians; a new
 * lsr_model_upload resets to 1. */
int lsr_model_set_mixtures(lsr_ctx* ctx, int n_mix, const double* log_weights);

/* Replaces the state space built by HiddenMarkovModelInference.from_folder
 * (hidden_markov_model.py:421-456): word models stacked in order, band[d][s] =
 * log a(s-d -> s) for d = 0,1,2 (-inf where s-d is not a candidate predecessor: before the
 * word's first state), word_lower/word_upper[w] = first/last state of word w (ascending).
 * silence_word = index of the word dropped from label output (-1: none).  Up to LSR_MAX_STATES
 * states; the decoders pick the two-wavefront form by themselves above 64. */
int lsr_loop_graph_upload(lsr_ctx* ctx, int n_states, int n_words, const float* band,
                          const int32_t* word_lower, const int32_t* word_upper, int silence_word);
/* The writable attribute _log_transition_probability_between_words
 * (hidden_markov_model.py:419).  between_is_f64 != 0 reproduces an np.float64 value (the
 * class default np.log(0.005)): the cross-word candidate is then added in float64;
 * otherwise `between` is rounded to float32 and added in float32 (Python int/float). */
int lsr_loop_set_between(lsr_ctx* ctx, double between, int between_is_f64);

/* ---- front-end -------------------------------------------------------------------------- */
/* Replaces MFCC.batch (mfcc.py:71-84; per clip mfcc.py:24-44,50-69).  audio_d: float32
 * samples of B clips back to back; sample_off_d[B+1]; frame_off_d[B+1] with
 * frames(b) = 1 + samples(b) / 160 (>= 9 each).  feats_d: float32 [total_frames][39]
 * (time-major, the transposed layout MFCC.batch returns). */
int lsr_mfcc_batch(lsr_ctx* ctx, const float* audio_d, const int64_t* sample_off_d,
                   const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d);
/* Same for int16 PCM as read from a wav file: the conversion of DataLoader.lazy_loading
 * (ti_digits.py:130-134, wavfile.read(...).astype(float32)) happens on the device, which halves
 * the bytes uploaded and read. */
int lsr_mfcc_batch_i16(lsr_ctx* ctx, const int16_t* audio_d, const int64_t* sample_off_d,
                       const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d);

/* ---- emission scores -------------------------------------------------------------------- */
/* Replaces MultivariateNormal.log_pdf (hidden_markov_model.py:46-48) for every (frame, state):
 * ll_d[f][s] = float32( -0.5 * (log_norm[s] + |(x_f - mean_s) @ whiten_s|^2) ), float64
 * inside.  ll_d: float32 [round_up(total_frames, 64)][lsr_ll_stride(ctx)]; columns >= n_states
 * are -inf.  The stride is LSR_LL_STRIDE = 64 for models of at most 64 states (mixture components
 * do not count) and 128 for larger ones (at most 128 states). */
int lsr_emission_batch(lsr_ctx* ctx, const float* feats_d, int64_t total_frames, float* ll_d);
int lsr_ll_stride(const lsr_ctx* ctx);

/* ---- decoding --------------------------------------------------------------------------- */
/* Replaces HiddenMarkovModelInference.predict/_viterbi/_viterbi_static
 * (hidden_markov_model.py:458-581) plus ModelBoundary.get_labels (model_boundary.py:107-147)
 * for a batch.  Outputs (any may be NULL): scores_d float32[B]; paths_d int8[total_frames]
 * (ragged like the features); labels_d uint8[B][label_cap] word indices in decode order with
 * the silence word removed; n_labels_d int32[B] = number of labels found (may exceed
 * label_cap; only the first label_cap are stored).  max_frames must be >= the longest utterance: an
 * utterance that is longer gets score NaN and n_labels -1 and no path (the device checks this against
 * frame_off_d; the same holds for every decoder below). */
int lsr_decode_loop_batch(lsr_ctx* ctx, const float* feats_d, const int64_t* frame_off_d,
                          int n_utts, int64_t total_frames, int max_frames, float* scores_d,
                          int8_t* paths_d, uint8_t* labels_d, int32_t* n_labels_d, int label_cap);

/* Replaces ModelBoundary.get_labels (model_boundary.py:107-147) alone, for a batch of state paths over the
 * uploaded loop graph: paths_d int8[total_frames] ragged by frame_off_d[B+1]; labels_d uint8[B][label_cap]
 * (may be NULL) and n_labels_d int32[B] as lsr_decode_loop_batch writes them (silence word dropped).  The
 * same device code produces the label records inside the decoders. */
int lsr_labels_from_paths(lsr_ctx* ctx, const int8_t* paths_d, const int64_t* frame_off_d, int n_utts,
                          uint8_t* labels_d, int32_t* n_labels_d, int label_cap);

/* Same, starting from audio: lsr_mfcc_batch + lsr_decode_loop_batch with the features kept in
 * the workspace (what project5_test_ndigits_with_sil.py:64-77 does per utterance). */
int lsr_recognize_batch(lsr_ctx* ctx, const float* audio_d, const int64_t* sample_off_d,
                        const int64_t* frame_off_d, int n_utts, int64_t total_frames,
                        int max_frames, float* scores_d, int8_t* paths_d, uint8_t* labels_d,
                        int32_t* n_labels_d, int label_cap);
int lsr_recognize_batch_i16(lsr_ctx* ctx, const int16_t* audio_d, const int64_t* sample_off_d,
                            const int64_t* frame_off_d, int n_utts, int64_t total_frames,
                            int max_frames, float* scores_d, int8_t* paths_d, uint8_t* labels_d,
                            int32_t* n_labels_d, int label_cap);

/* The whole per-utterance flow of scripts/project5_test_ndigits_with_sil.py:64-77 starting from HOST
 * arrays, as the reference's loader delivers them (ti_digits.py:130-134: wavfile.read -> int16, or
 * its float32 conversion): clips[b] points to the n_samples[b] samples of clip b (int16 when
 * audio_is_i16 != 0, else float32) anywhere in host memory - no concatenation is needed.  The clips
 * are packed chunk by chunk (about chunk_samples samples each, <= 0: a default of 24 Mi samples of float32, 12 Mi of int16)
 * into pinned staging memory by n_threads worker threads (<= 0: 8), uploaded with asynchronous
 * copies on a separate stream, and decoded (lsr_recognize_batch[_i16]) while later chunks are
 * packed and uploaded; three staging slots rotate.  Outputs are HOST arrays, complete when the call
 * returns: scores float32[B] (may be NULL), labels uint8[B][label_cap], n_labels int32[B], with the
 * same contents lsr_recognize_batch produces for the whole batch at once.  Staging buffers are
 * kept by the context and reused. */
int lsr_recognize_host(lsr_ctx* ctx, const void* const* clips, const int64_t* n_samples, int n_utts,
                       int audio_is_i16, int64_t chunk_samples, int n_threads, float* scores,
                       uint8_t* labels, int32_t* n_labels, int label_cap);

/* Host time of the last lsr_recognize_host call by phase: seconds4 = {waiting for a free staging slot,
 * packing clips into pinned memory, submitting copies and kernels, waiting for the last chunk and
 * copying the results back}; *n_chunks (may be NULL) = number of chunks.  A diagnostic for tuning
 * chunk_samples / n_threads; no counterpart in the reference. */
int lsr_recognize_host_timing(lsr_ctx* ctx, double* seconds4, int* n_chunks);

/* Host arrays -> one device buffer, back to back (the concatenation np.concatenate + one copy would produce): the
 * upload half of lsr_recognize_host alone, for MFCC.batch / predict_batch style callers whose inputs are lists of
 * host arrays (mfcc.py:71-84, ti_digits.py:130-134).  arrays[i] holds n_elems[i] elements of elem_bytes bytes;
 * worker threads pack chunks of about chunk_bytes (0: 96 MiB) into pinned staging memory and a copy stream moves
 * them while later chunks are packed.  The copies are ordered after the work already queued on the context's
 * stream and the context's stream waits for the last copy; the call returns when all arrays have been read. */
int lsr_upload_host(lsr_ctx* ctx, const void* const* arrays, const int64_t* n_elems, int64_t n_arrays,
                    int elem_bytes, int64_t chunk_bytes, int n_threads, void* dst_d);

/* Finiteness of host float32 data, tested while the packing threads copy it (no extra pass over memory).  librosa
 * refuses audio that is not finite everywhere (librosa.util.valid_audio, reached from mfcc.py:31); the device front-end
 * would turn such a clip into NaN features and an arbitrary string.  flags: bit 0 (default on) - float32 clips of
 * lsr_recognize_host; bit 1 (default off) - the arrays of lsr_upload_host when elem_bytes is 4, for callers that
 * upload float32 audio with it.  A non-finite value fails the call with LSR_ERR_ARG and the index of the lowest
 * offending clip / array of the chunk in the message; chunks before it may already have been processed. */
int lsr_host_check_finite(lsr_ctx* ctx, int flags);

/* The other direction, for callers that want the features (or paths) back as host arrays, as MFCC.batch returns them
 * (mfcc.py:71-84): bytes of device memory to host memory through pinned staging chunks (0: 32 MiB) unpacked by
 * worker threads while the next chunk travels.  Ordered after the work queued on the context's stream; dst is
 * complete when the call returns. */
int lsr_download_host(lsr_ctx* ctx, const void* src_d, int64_t bytes, void* dst, int64_t chunk_bytes, int n_threads);

/* lsr_recognize_batch (float32 audio) for callers that cannot allocate device memory: code under
 * HIP stream capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart
 * in the reference.  Same arguments, same kernels and bit-identical outputs; every intermediate
 * buffer lives in one static device array of about 42 MiB per GPU (held until the process ends)
 * instead of the context's growing workspace, and the call performs no hipMalloc / hipFree.
 * Limits: n_utts <= LSR_NOALLOC_MAX_UTTS, total_frames <= LSR_NOALLOC_MAX_FRAMES and a loop graph of
 * at most 64 states (LSR_ERR_LIMIT beyond); profiling must be off (LSR_ERR_STATE).  All calls on one GPU share the
 * array: calls that overlap in time on one GPU (other streams, other contexts) race. */
#define LSR_NOALLOC_MAX_FRAMES 65536
#define LSR_NOALLOC_MAX_UTTS 4096
int lsr_recognize_batch_noalloc(lsr_ctx* ctx, const float* audio_d, const int64_t* sample_off_d,
                                const int64_t* frame_off_d, int n_utts, int64_t total_frames,
                                int max_frames, float* scores_d, int8_t* paths_d, uint8_t* labels_d,
                                int32_t* n_labels_d, int label_cap);

/* Replaces HiddenMarkovModel.predict/_viterbi/_viterbi_static (hidden_markov_model.py:75-91,
 * 160-208) for a batch of per-utterance chains: isolated word models, several independent
 * models side by side (ModelCollection.predict, model_collection.py:23-28) or the stacked
 * alignment chain S d S d ... S of HiddenMarkovModelMultiWord (hidden_markov_model.py:584-664).
 * Chain b has chain_off_d[b+1]-chain_off_d[b] <= 64 states (longer: lsr_decode_chain_batch_wide); for chain position c (ragged
 * index i = chain_off_d[b] + c): chain_state_d[i] = model state whose Gaussian it uses,
 * chain_band_d[d*total_chain + i] = log a(c-d -> c) (float32; 0.0 for the reference's
 * missing cross-word keys, -inf to forbid), chain_start_d[i] != 0 marks states initialised at
 * frame 0.  The path is traced back from the chain's LAST state.  Outputs (any may be NULL):
 * scores_d float32[B] (likelihood of the last state), paths_d int8[total_frames] (chain
 * positions), final_d float32[B][64] (last-frame likelihood of every chain position).  A chain
 * with fewer than 1 or more than 64 states gets score NaN and nothing else. */
int lsr_decode_chain_batch(lsr_ctx* ctx, const float* feats_d, const int64_t* frame_off_d,
                           int n_utts, int64_t total_frames, int max_frames,
                           const int64_t* chain_off_d, int64_t total_chain,
                           const int32_t* chain_state_d, const float* chain_band_d,
                           const uint8_t* chain_start_d, float* scores_d, int8_t* paths_d,
                           float* final_d);

/* lsr_decode_chain_batch for chains of up to 127 states (e.g. the forced alignment of a 9-digit string:
 * 75 states; hidden_markov_model.py:498-501 sizes its pointers for 127): two wavefronts per utterance,
 * about half the speed per utterance.  No final_d output.  Chains outside 1..127 get score NaN. */
int lsr_decode_chain_batch_wide(lsr_ctx* ctx, const float* feats_d, const int64_t* frame_off_d,
                                int n_utts, int64_t total_frames, int max_frames,
                                const int64_t* chain_off_d, int64_t total_chain,
                                const int32_t* chain_state_d, const float* chain_band_d,
                                const uint8_t* chain_start_d, float* scores_d, int8_t* paths_d);

/* Which emission scores lsr_decode_chain_batch[_wide] computes before the recursion.  A chain only reads the
 * columns of its own states for the frames of its own utterance (hidden_markov_model.py:638-664 builds the
 * forced-alignment model S d S d ... S from the models of the utterance's words only, so the reference never
 * evaluates the other words' Gaussians there either).  LSR_CHAIN_EMISSION_REACHABLE scores, per 64-frame tile,
 * the union of the states of the chains whose utterances own frames of the tile; LSR_CHAIN_EMISSION_ALL scores
 * every state of the model; LSR_CHAIN_EMISSION_AUTO (default) takes REACHABLE when the chains hold on average
 * fewer than three quarters of the model's states (chains that span the whole model, like the side-by-side
 * isolated models of ModelCollection.predict, gain nothing).  Scores, paths and last-frame tables are the same
 * bits either way.  Mixture models always take ALL. */
#define LSR_CHAIN_EMISSION_AUTO 0
#define LSR_CHAIN_EMISSION_ALL 1
#define LSR_CHAIN_EMISSION_REACHABLE 2
int lsr_chain_emission_mode(lsr_ctx* ctx, int mode);

/* The penalty sweep of scripts/project5_find_trans_ndigits_with_sil.py:71-88 (the same
 * utterances decoded once per candidate between-word log-probability) as one call: emission scores
 * are computed once and the loop decoder runs once per entry of between[n_between] (host array;
 * between_is_f64[i] as in lsr_loop_set_between).  Outputs are laid out [n_between][B]...:
 * scores_d float32[n_between][B] (may be NULL), labels_d uint8[n_between][B][label_cap],
 * n_labels_d int32[n_between][B].  The penalty set with lsr_loop_set_between is left unchanged. */
int lsr_decode_loop_sweep(lsr_ctx* ctx, const float* feats_d, const int64_t* frame_off_d, int n_utts,
                          int64_t total_frames, int max_frames, const double* between,
                          const int32_t* between_is_f64, int n_between, float* scores_d,
                          uint8_t* labels_d, int32_t* n_labels_d, int label_cap);

/* ---- hard-alignment statistics ---------------------------------------------------------- */
/* Replaces the data reduction of SortedSignals.order_by_state / .transition_probabilities and
 * HiddenMarkovModelTrainable._update_middleware_parameters (signal.py:68-91,
 * hidden_markov_model.py:320-350): given per-frame target slots slot_d[f] in [0, n_slots) or -1
 * (frame not used), accumulates in float64  count[n_slots], sum_x[n_slots][39],
 * sum_xx[n_slots][39][39].  The three outputs are device pointers and are OVERWRITTEN. */
int lsr_accumulate_stats(lsr_ctx* ctx, const float* feats_d, const int32_t* slot_d,
                         int64_t total_frames, int n_slots, double* count_d, double* sum_x_d,
                         double* sum_xx_d);

/* Replaces the segmentation bookkeeping of HiddenMarkovModelMultiWord._remux_path_and_signal
 * (hidden_markov_model.py:602-636), Signal.order_by_state (signal.py:23-47) and the transition
 * counting of SortedSignals.transition_probabilities (signal.py:80-91) for a batch of aligned
 * utterances: paths_d int8[total_frames] holds chain positions (output of
 * lsr_decode_chain_batch), chain_state_d / chain_word_d int32[total_chain] give the model state
 * and the ordinal of the word of every chain position.  slots_d int32[total_frames] receives the
 * model state each frame contributes to, or -1 when drop_last_run != 0 and the frame belongs to
 * the last word run of its utterance (the reference only flushes a run when the label changes, so
 * the final run is never used).  trans_d int64[n_states][n_states] is OVERWRITTEN with the number
 * of consecutive kept frame pairs (same utterance, same word) per (from, to) model state. */
int lsr_remux_slots(lsr_ctx* ctx, const int8_t* paths_d, const int64_t* frame_off_d,
                    const int64_t* chain_off_d, const int32_t* chain_state_d,
                    const int32_t* chain_word_d, int n_utts, int64_t total_frames, int n_states,
                    int drop_last_run, int32_t* slots_d, int64_t* trans_d);

/* ---- multi-GPU exchange steps (RCCL) ---------------------------------------------------------- */
/* The reference fans work out to host processes and collects the results through pickled return
 * values (scripts/project5_test_ndigits_with_sil.py:29-42 for decoding, hidden_markov_model.py:746-750
 * for training).  Here every GPU has its own process and context; utterances are sharded in contiguous
 * ranges and two collectives replace the collection step.  RCCL is loaded on first use (dlopen of
 * librccl.so.1); nothing else in the library needs it. */
#define LSR_COMM_ID_BYTES 128
/* Contiguous shard [lo, hi) of `rank`: sizes differ by at most one, earlier ranks take the extra item. */
void lsr_shard_bounds(int64_t n_items, int rank, int world, int64_t* lo, int64_t* hi);
/* Rank 0 creates the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 * (file, socket, MPI, torch.distributed); then every rank calls lsr_comm_init with its rank. */
int lsr_comm_get_unique_id(lsr_ctx* ctx, void* id_out);
int lsr_comm_init(lsr_ctx* ctx, const void* unique_id, int rank, int world);
int lsr_comm_destroy(lsr_ctx* ctx);
/* Decoded label records of the whole job on every rank, in global utterance order: labels_d
 * uint8[n_local][label_cap] / n_labels_d int32[n_local] are this rank's outputs of lsr_decode_loop_batch
 * or lsr_recognize_batch for its shard lsr_shard_bounds(n_total, rank, world); labels_out_d
 * uint8[n_total][label_cap] and n_labels_out_d int32[n_total] receive all of them.  One ncclAllGather
 * of fixed-width records on the context's stream (asynchronous like the kernels). */
int lsr_allgather_labels(lsr_ctx* ctx, const uint8_t* labels_d, const int32_t* n_labels_d, int64_t n_local,
                         int label_cap, int64_t n_total, uint8_t* labels_out_d, int32_t* n_labels_out_d);
/* In-place sum over all ranks of the packed float64 training statistics (counts, sum_x, sum_xx of
 * lsr_accumulate_stats, transition counts converted to float64): one ncclAllReduce per training
 * iteration, after which every rank runs the same M-step (hidden_markov_model.py:754-770). */
int lsr_allreduce_stats(lsr_ctx* ctx, double* stats_d, int64_t count);
/* The record layout of lsr_allgather_labels on the host (no GPU needed; also used by CPU-side
 * transports): a record is label_cap label bytes followed by the little-endian int32 count
 * (lsr_label_record_bytes); a rank contributes shard_cap = size of the largest shard records, unused
 * ones filled with 0xFF; unpack reads world * shard_cap records back into global order. */
int64_t lsr_label_record_bytes(int label_cap);
int lsr_host_pack_label_records(const uint8_t* labels, const int32_t* n_labels, int64_t n_local, int label_cap,
                                int64_t shard_cap, uint8_t* records);
int lsr_host_unpack_label_records(const uint8_t* records, int world, int64_t shard_cap, int64_t n_total,
                                  int label_cap, uint8_t* labels, int32_t* n_labels);
/* The same two steps on the device (context's stream), for callers that bring their own all-gather
 * (MPI, torch.distributed): records_d holds shard_cap records (pack) / world * shard_cap records with
 * shard_cap = size of shard 0 of n_total (unpack). */
int lsr_pack_label_records(lsr_ctx* ctx, const uint8_t* labels_d, const int32_t* n_labels_d, int64_t n_local,
                           int label_cap, int64_t shard_cap, uint8_t* records_d);
int lsr_unpack_label_records(lsr_ctx* ctx, const uint8_t* records_d, int world, int64_t n_total, int label_cap,
                             uint8_t* labels_out_d, int32_t* n_labels_out_d);

/* ---- silence stripping ------------------------------------------------------------------- */
/* Replaces the per-clip scan of SignalSeparation._remove_empty / .detect_speech
 * (signal_separation.py:102-164) for a batch of clips.  A clip of N samples has
 * N / frame_size full scan frames plus one trailing frame with the remaining samples (possibly
 * none: its energy is NaN and it counts as a frame without speech, as in the reference);
 * scan_off_d int64[B+1] are the prefix sums of those counts.  Frame energy is the float32 mean
 * of |x| rounded exactly as NumPy's float32 mean; thresholds are high * max|clip| and
 * low * max|clip| in float64, and the float32 energy is promoted to float64 for the comparison (the
 * reference's maximum is an np.float64, signal_separation.py:105,71-76).  Outputs per clip: first_d (first kept frame, -1 if speech never starts), end_d (the
 * frame that tripped the silence counter; kept samples are frames [first, end); -1 if it never
 * tripped) and status_d: 0 = kept, 1 = never tripped (the reference raises FailToProcess,
 * :99-100), 2 = fewer than min_kept_frames kept frames (:95-97, 9 in the reference).
 * noise_d uint8[total_scan_frames] flags the frames the reference appends to its noise pool
 * (:139).  Optional outputs (may be NULL): energy_d float32[total_scan_frames],
 * max_volume_d float32[B].  frame_size is limited to 256 samples (LSR_ERR_LIMIT beyond). */
int lsr_separation_scan(lsr_ctx* ctx, const void* audio_d, int audio_is_i16,
                        const int64_t* sample_off_d, const int64_t* scan_off_d, int n_utts,
                        int64_t total_scan_frames, int frame_size, double high, double low,
                        int max_silence_frames, int min_kept_frames, int32_t* first_d,
                        int32_t* end_d, int32_t* status_d, uint8_t* noise_d, float* energy_d,
                        float* max_volume_d);

/* Device-side np.concatenate of slices (signal_separation.py:92-94 builds the kept clip and the
 * noise clip that way): segment i copies dst_off_d[i+1] - dst_off_d[i] elements of elem_bytes
 * (2 or 4) from src_d + src_start_d[i] to dst_d + dst_off_d[i].  max_segment is the longest
 * segment (sizes the launch). */
int lsr_gather_segments(lsr_ctx* ctx, const void* src_d, int elem_bytes, const int64_t* src_start_d,
                        const int64_t* dst_off_d, int n_segments, int64_t max_segment, void* dst_d);

#ifdef __cplusplus
}
#endif
#endif /* LSR_HIP_H */
