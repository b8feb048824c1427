// lsr_recognize_batch_noalloc: the audio -> label records path of lsr_recognize_batch with every
// intermediate buffer (frame records, mel energies, utterance maxima, features, emission scores,
// back-pointers) carved out of one statically allocated device array instead of the context's
// growing workspace.  The call performs no hipMalloc / hipFree, so it can be used under stream
// capture, from serving loops and from harnesses that forbid allocation.
//
// The array exists once per GPU (every device that loads the code object holds its own copy until
// the process ends); its address on a device is resolved once, with that device current, by
// lsr_noalloc_init (called from lsr_ctx_create) and kept per device.  Host code only ever uses the
// resolved address: the array's name in host code is a stand-in symbol, not device memory.
// All calls on one GPU share the array.  Calls made one after the other from the host - on any stream, through
// any context of this process - are ordered on the device by an event kept per GPU: a call first makes its
// stream wait for the event the previous call recorded behind its last kernel.  (Under stream capture the event
// is left alone: the order of the captured launches is the graph's.)  Calls issued CONCURRENTLY from several host
// threads, or from several processes sharing the GPU, still race.
#include "lsr_internal.h"

#define NA_ALIGN(x) (((x) + 255) / 256 * 256)
// byte offsets of the parts (each 256-byte aligned)
#define NA_OFF_LL 0
#define NA_OFF_TRACER (NA_OFF_LL + NA_ALIGN((size_t)LSR_NOALLOC_MAX_FRAMES * LSR_LL_STRIDE * 4))
#define NA_OFF_MEL (NA_OFF_TRACER + NA_ALIGN(LSR_TRACER_BYTES(LSR_NOALLOC_MAX_FRAMES, LSR_NOALLOC_MAX_UTTS)))
#define NA_OFF_FEATS (NA_OFF_MEL + NA_ALIGN((size_t)LSR_NOALLOC_MAX_FRAMES * LSR_N_MELS * 4))
#define NA_OFF_INFO (NA_OFF_FEATS + NA_ALIGN((size_t)LSR_NOALLOC_MAX_FRAMES * LSR_DIM * 4))
#define NA_OFF_FUTT (NA_OFF_INFO + NA_ALIGN((size_t)LSR_NOALLOC_MAX_FRAMES * 16))
#define NA_OFF_UMAX (NA_OFF_FUTT + NA_ALIGN((size_t)LSR_NOALLOC_MAX_FRAMES * 4))
#define NA_BYTES (NA_OFF_UMAX + NA_ALIGN((size_t)LSR_NOALLOC_MAX_UTTS * 4))

static __device__ __align__(256) unsigned char scratch[NA_BYTES];

#define NA_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[NA_MAX_DEVICES];      // device address of `scratch` on each GPU (0 = not resolved)
static hipEvent_t g_scratch_done[NA_MAX_DEVICES];         // recorded behind the last kernel of the latest call on each GPU

int lsr_noalloc_init(lsr_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= NA_MAX_DEVICES) return LSR_OK;      // entry point reports the missing address
    if (g_scratch_ptr[ctx->device]) return LSR_OK;
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    void* ptr = nullptr;
    LSR_HIP_CHECK(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return lsr_fail(ctx, LSR_ERR_HIP, "hipGetSymbolAddress returned a null address for the static workspace");
    // created here, at context creation, so that the entry point itself creates nothing
    LSR_HIP_CHECK(ctx, hipEventCreateWithFlags(&g_scratch_done[ctx->device], hipEventDisableTiming));
    g_scratch_ptr[ctx->device] = (unsigned char*)ptr;
    return LSR_OK;
}

extern "C" int lsr_recognize_batch_noalloc(lsr_ctx* ctx, const float* audio_d, const int64_t* sample_off_d,
                                           const int64_t* frame_off_d, int n_utts, int64_t total_frames,
                                           int max_frames, float* scores_d, int8_t* paths_d, uint8_t* labels_d,
                                           int32_t* n_labels_d, int label_cap) {
    if (!ctx) return LSR_ERR_ARG;
    if (!audio_d || !sample_off_d || !frame_off_d)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_recognize_batch_noalloc: null pointer");
    if (n_utts < 1 || total_frames < n_utts || max_frames < 1 || max_frames > total_frames)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_recognize_batch_noalloc: bad sizes");
    if (labels_d && label_cap < 1) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_recognize_batch_noalloc: label_cap < 1");
    if (n_utts > LSR_NOALLOC_MAX_UTTS || total_frames > LSR_NOALLOC_MAX_FRAMES)
        return lsr_fail(ctx, LSR_ERR_LIMIT,
                        "lsr_recognize_batch_noalloc: %d utterances / %lld frames exceed the static workspace "
                        "(%d utterances, %d frames); use lsr_recognize_batch",
                        n_utts, (long long)total_frames, LSR_NOALLOC_MAX_UTTS, LSR_NOALLOC_MAX_FRAMES);
    if (ctx->n_states == 0 || ctx->g_states == 0)
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_recognize_batch_noalloc: model and loop graph must be uploaded first");
    if (ctx->n_states / ctx->n_mix != ctx->g_states)
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_recognize_batch_noalloc: model has %d states, graph %d",
                        ctx->n_states / ctx->n_mix, ctx->g_states);
    if (ctx->g_states > 64)
        return lsr_fail(ctx, LSR_ERR_LIMIT, "lsr_recognize_batch_noalloc: graphs above 64 states need the growing workspace; "
                                            "use lsr_recognize_batch");
    if (ctx->profiling)      // per-kernel timing creates HIP events on demand
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_recognize_batch_noalloc: disable profiling first (it creates events)");
    if (ctx->device < 0 || ctx->device >= NA_MAX_DEVICES || !g_scratch_ptr[ctx->device])
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_recognize_batch_noalloc: static workspace of device %d was not resolved",
                        ctx->device);
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    unsigned char* base = g_scratch_ptr[ctx->device];
    float* ll = (float*)(base + NA_OFF_LL);
    int8_t* tracer = (int8_t*)(base + NA_OFF_TRACER);
    float* feats = (float*)(base + NA_OFF_FEATS);
    lsr_mfcc_ws ws;
    ws.mel = (float*)(base + NA_OFF_MEL);
    ws.frame_info = base + NA_OFF_INFO;
    ws.frame_utt = (int32_t*)(base + NA_OFF_FUTT);
    ws.uttmax = (unsigned int*)(base + NA_OFF_UMAX);      // cleared by the front-end launcher on every call
    // the previous user of the array on this GPU (any stream) must have finished before this call's kernels start
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &capture) != hipSuccess) capture = hipStreamCaptureStatusNone;
    const bool ordered = capture == hipStreamCaptureStatusNone;
    if (ordered) LSR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, g_scratch_done[ctx->device], 0));
    int rc = lsr_launch_mfcc_ws(ctx, audio_d, 0, sample_off_d, frame_off_d, n_utts, total_frames, feats, ws);
    if (rc == LSR_OK) rc = lsr_launch_emission(ctx, feats, total_frames, ll);
    if (rc == LSR_OK)
        rc = lsr_launch_viterbi_loop(ctx, ll, frame_off_d, n_utts, total_frames, max_frames, scores_d, paths_d, labels_d,
                                     n_labels_d, label_cap, tracer);
    if (ordered) {
        const hipError_t e = hipEventRecord(g_scratch_done[ctx->device], ctx->stream);      // also after a failed launch
        if (e != hipSuccess && rc == LSR_OK)
            rc = lsr_fail(ctx, LSR_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(e));
    }
    return rc;
}
