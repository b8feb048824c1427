// Host-resident clips in, label records out: the upload is cut into chunks and pipelined with the
// decoding of earlier chunks, so that a caller who starts from host arrays (what the reference's
// loader hands over) sees the PCIe rate rather than one synchronous pageable copy followed by the
// kernels.
//
// Per chunk: worker threads copy the clips of the chunk straight from the caller's arrays into a
// pinned staging buffer (no concatenation on the caller's side), one hipMemcpyAsync on the copy
// stream moves the chunk (audio and its two offset tables) to the device, the context's stream waits
// for that copy and runs the front-end, the emission scores and the loop decoder on it.  Three
// staging/device slots rotate: packing of chunk c + 2, upload of chunk c + 1 and decoding of chunk
// c overlap.  Label records and scores of all chunks collect in one device buffer that is copied
// back once at the end.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "lsr_internal.h"

#define PIPE_SLOTS 3

// ---- a small persistent thread pool: run fn(i) for i in [0, n) and wait -----------------------------
struct lsr_pool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(int)> fn;
    int next = 0, total = 0, pending = 0;
    uint64_t generation = 0;
    bool stop = false;

    explicit lsr_pool(int n) {
        for (int t = 0; t < n; ++t) threads.emplace_back([this] { run(); });
    }
    ~lsr_pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto& t : threads) t.join();
    }
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [this] { return stop || next < total; });
            if (stop) return;
            while (next < total) {
                const int i = next++;
                lk.unlock();
                fn(i);
                lk.lock();
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    void parallel_for(int n, std::function<void(int)> f) {
        if (n <= 0) return;
        std::unique_lock<std::mutex> lk(mu);
        fn = std::move(f);
        next = 0;
        total = n;
        pending = n;
        cv_work.notify_all();
        cv_done.wait(lk, [this] { return pending == 0; });
        total = 0;
    }
};

struct lsr_pipe_slot {
    void* pinned = nullptr;          // staging: [audio bytes | sample_off int64[n+1] | frame_off int64[n+1]]
    void* device = nullptr;          // same layout on the device
    size_t bytes = 0;
    hipEvent_t copied = nullptr;     // upload of the slot's chunk has finished
    hipEvent_t decoded = nullptr;    // kernels that read the slot have finished
    bool busy = false;
};

struct lsr_pipeline {
    hipStream_t copy_stream = nullptr;
    lsr_pipe_slot slot[PIPE_SLOTS];
    lsr_buffer out_dev;              // [scores f32[B] | n_labels i32[B] | labels u8[B][cap]]
    void* out_pinned = nullptr;
    size_t out_pinned_bytes = 0;
    lsr_pool* pool = nullptr;
    int pool_threads = 0;
    double t_pack = 0, t_wait = 0, t_submit = 0, t_drain = 0;      // host seconds of the last call, by phase
    int last_chunks = 0;
};

static double pipe_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void pipe_free_slot(lsr_pipe_slot& s) {
    if (s.pinned) (void)hipHostFree(s.pinned);
    if (s.device) (void)hipFree(s.device);
    s.pinned = s.device = nullptr;
    s.bytes = 0;
}

void lsr_pipeline_destroy(lsr_ctx* ctx) {
    lsr_pipeline* p = ctx->pipeline;
    if (!p) return;
    for (auto& s : p->slot) {
        pipe_free_slot(s);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.decoded) (void)hipEventDestroy(s.decoded);
    }
    if (p->out_dev.ptr) (void)hipFree(p->out_dev.ptr);
    if (p->out_pinned) (void)hipHostFree(p->out_pinned);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    delete p->pool;
    delete p;
    ctx->pipeline = nullptr;
}

static int pipe_get(lsr_ctx* ctx, int threads, lsr_pipeline** out) {
    if (!ctx->pipeline) {
        lsr_pipeline* p = new lsr_pipeline();
        ctx->pipeline = p;
        LSR_HIP_CHECK(ctx, hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking));
        for (auto& s : p->slot) {
            LSR_HIP_CHECK(ctx, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
            LSR_HIP_CHECK(ctx, hipEventCreateWithFlags(&s.decoded, hipEventDisableTiming));
        }
    }
    lsr_pipeline* p = ctx->pipeline;
    if (!p->pool || p->pool_threads != threads) {
        delete p->pool;
        p->pool = new lsr_pool(threads);
        p->pool_threads = threads;
    }
    *out = p;
    return LSR_OK;
}

static int pipe_reserve_slot(lsr_ctx* ctx, lsr_pipe_slot& s, size_t bytes) {
    if (bytes <= s.bytes) return LSR_OK;
    pipe_free_slot(s);
    const size_t want = bytes + bytes / 8 + 4096;
    LSR_HIP_CHECK(ctx, hipHostMalloc(&s.pinned, want, hipHostMallocDefault));
    LSR_HIP_CHECK(ctx, hipMalloc(&s.device, want));
    s.bytes = want;
    return LSR_OK;
}

extern "C" int lsr_recognize_host(lsr_ctx* ctx, const void* const* clips, const int64_t* n_samples, int n_utts,
                                  int audio_is_i16, int64_t chunk_samples, int n_threads, float* scores,
                                  uint8_t* labels, int32_t* n_labels, int label_cap) {
    if (!ctx) return LSR_ERR_ARG;
    if (!clips || !n_samples || !labels || !n_labels) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_recognize_host: null pointer");
    if (n_utts < 1 || label_cap < 1) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_recognize_host: bad sizes");
    if (ctx->n_states == 0 || ctx->g_states == 0)
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_recognize_host: model and loop graph must be uploaded first");
    if (ctx->n_states / ctx->n_mix != ctx->g_states)
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_recognize_host: model has %d states, graph %d",
                        ctx->n_states / ctx->n_mix, ctx->g_states);
    if (chunk_samples <= 0) chunk_samples = (int64_t)24 << 20;      // about 48 MB of int16 / 96 MB of float32 per chunk
    if (n_threads <= 0) n_threads = 8;
    if (n_threads > 64) n_threads = 64;
    for (int b = 0; b < n_utts; ++b) {
        if (!clips[b]) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_recognize_host: clip %d is a null pointer", b);
        if (n_samples[b] / LSR_HOP + 1 < 9)
            return lsr_fail(ctx, LSR_ERR_ARG, "lsr_recognize_host: clip %d has %lld samples; the delta filter needs 9 frames",
                            b, (long long)n_samples[b]);
    }
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    lsr_pipeline* p = nullptr;
    int rc = pipe_get(ctx, n_threads, &p);
    if (rc) return rc;

    // chunk boundaries: consecutive clips up to chunk_samples samples (at least one clip per chunk)
    const size_t esz = audio_is_i16 ? 2 : 4;
    std::vector<int> first;
    {
        int64_t acc = 0;
        first.push_back(0);
        for (int b = 0; b < n_utts; ++b) {
            if (acc > 0 && acc + n_samples[b] > chunk_samples) {
                first.push_back(b);
                acc = 0;
            }
            acc += n_samples[b];
        }
        first.push_back(n_utts);
    }
    const int n_chunks = (int)first.size() - 1;
    // sizes of the largest chunk: staging bytes, frames (workspace), longest utterance
    size_t max_bytes = 0;
    int64_t max_frames_chunk = 0;
    int max_utts_chunk = 0;
    for (int c = 0; c < n_chunks; ++c) {
        max_utts_chunk = std::max(max_utts_chunk, first[c + 1] - first[c]);
        int64_t ns = 0, nf = 0;
        for (int b = first[c]; b < first[c + 1]; ++b) {
            ns += n_samples[b];
            nf += n_samples[b] / LSR_HOP + 1;
        }
        const size_t audio_bytes = ((size_t)ns * esz + 255) / 256 * 256;
        const size_t bytes = audio_bytes + 2 * (size_t)(first[c + 1] - first[c] + 1) * sizeof(int64_t);
        max_bytes = std::max(max_bytes, bytes);
        max_frames_chunk = std::max(max_frames_chunk, nf);
    }
    if (max_frames_chunk > ((int64_t)1 << 31) - 64)
        return lsr_fail(ctx, LSR_ERR_LIMIT, "lsr_recognize_host: chunk of %lld frames", (long long)max_frames_chunk);
    // everything that can grow is sized before the pipeline starts, so no chunk stalls on an allocation
    for (int s = 0; s < PIPE_SLOTS && s < n_chunks; ++s)
        if ((rc = pipe_reserve_slot(ctx, p->slot[s], max_bytes))) return rc;
    const size_t out_bytes = (size_t)n_utts * (sizeof(float) + sizeof(int32_t) + (size_t)label_cap);
    if ((rc = lsr_reserve(ctx, &p->out_dev, out_bytes, "pipeline outputs"))) return rc;
    if (out_bytes > p->out_pinned_bytes) {
        if (p->out_pinned) LSR_HIP_CHECK(ctx, hipHostFree(p->out_pinned));
        p->out_pinned = nullptr;
        p->out_pinned_bytes = 0;
        LSR_HIP_CHECK(ctx, hipHostMalloc(&p->out_pinned, out_bytes + out_bytes / 8, hipHostMallocDefault));
        p->out_pinned_bytes = out_bytes + out_bytes / 8;
    }
    if ((rc = lsr_reserve_recognize(ctx, max_utts_chunk, max_frames_chunk))) return rc;

    float* d_scores = (float*)p->out_dev.ptr;
    int32_t* d_count = (int32_t*)(d_scores + n_utts);
    uint8_t* d_labels = (uint8_t*)(d_count + n_utts);

    p->t_pack = p->t_wait = p->t_submit = p->t_drain = 0;
    p->last_chunks = n_chunks;
    for (int c = 0; c < n_chunks; ++c) {
        lsr_pipe_slot& s = p->slot[c % PIPE_SLOTS];
        const int b0 = first[c], nb = first[c + 1] - b0;
        const double t0 = pipe_now();
        if (s.busy) {                                       // the chunk that used this slot has been decoded
            LSR_HIP_CHECK(ctx, hipEventSynchronize(s.decoded));
            s.busy = false;
        }
        const double t1 = pipe_now();
        // pack: offsets first (cheap, serial), then the clips in parallel
        std::vector<int64_t> soff(nb + 1), foff(nb + 1);
        soff[0] = foff[0] = 0;
        int max_frames = 0;
        for (int i = 0; i < nb; ++i) {
            const int64_t ns = n_samples[b0 + i], nf = ns / LSR_HOP + 1;
            soff[i + 1] = soff[i] + ns;
            foff[i + 1] = foff[i] + nf;
            if (nf > max_frames) max_frames = (int)nf;
        }
        const size_t audio_bytes = ((size_t)soff[nb] * esz + 255) / 256 * 256;
        unsigned char* stage = (unsigned char*)s.pinned;
        memcpy(stage + audio_bytes, soff.data(), (nb + 1) * sizeof(int64_t));
        memcpy(stage + audio_bytes + (nb + 1) * sizeof(int64_t), foff.data(), (nb + 1) * sizeof(int64_t));
        // contiguous ranges of clips with about equal bytes per task
        const int tasks = std::min(nb, 4 * n_threads);
        std::vector<int> cut(tasks + 1);
        for (int t = 0, b = 0; t <= tasks; ++t) {
            const int64_t goal = soff[nb] * t / tasks;
            while (b < nb && soff[b] < goal) ++b;
            cut[t] = t == tasks ? nb : b;
        }
        p->pool->parallel_for(tasks, [&](int t) {
            for (int i = cut[t]; i < cut[t + 1]; ++i)
                memcpy(stage + (size_t)soff[i] * esz, clips[b0 + i], (size_t)n_samples[b0 + i] * esz);
        });
        const size_t total_bytes = audio_bytes + 2 * (size_t)(nb + 1) * sizeof(int64_t);
        const double t2 = pipe_now();
        LSR_HIP_CHECK(ctx, hipMemcpyAsync(s.device, s.pinned, total_bytes, hipMemcpyHostToDevice, p->copy_stream));
        LSR_HIP_CHECK(ctx, hipEventRecord(s.copied, p->copy_stream));
        LSR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, s.copied, 0));
        const unsigned char* dev = (const unsigned char*)s.device;
        const int64_t* d_soff = (const int64_t*)(dev + audio_bytes);
        const int64_t* d_foff = d_soff + (nb + 1);
        if (audio_is_i16)
            rc = lsr_recognize_batch_i16(ctx, (const int16_t*)dev, d_soff, d_foff, nb, foff[nb], max_frames, d_scores + b0,
                                         nullptr, d_labels + (size_t)b0 * label_cap, d_count + b0, label_cap);
        else
            rc = lsr_recognize_batch(ctx, (const float*)dev, d_soff, d_foff, nb, foff[nb], max_frames, d_scores + b0,
                                     nullptr, d_labels + (size_t)b0 * label_cap, d_count + b0, label_cap);
        if (rc) return rc;
        LSR_HIP_CHECK(ctx, hipEventRecord(s.decoded, ctx->stream));
        s.busy = true;
        const double t3 = pipe_now();
        p->t_wait += t1 - t0;
        p->t_pack += t2 - t1;
        p->t_submit += t3 - t2;
    }
    const double t4 = pipe_now();
    LSR_HIP_CHECK(ctx, hipMemcpyAsync(p->out_pinned, p->out_dev.ptr, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    LSR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& s : p->slot) s.busy = false;
    const unsigned char* o = (const unsigned char*)p->out_pinned;
    if (scores) memcpy(scores, o, (size_t)n_utts * sizeof(float));
    memcpy(n_labels, o + (size_t)n_utts * sizeof(float), (size_t)n_utts * sizeof(int32_t));
    memcpy(labels, o + (size_t)n_utts * (sizeof(float) + sizeof(int32_t)), (size_t)n_utts * label_cap);
    p->t_drain = pipe_now() - t4;
    return LSR_OK;
}

extern "C" int lsr_recognize_host_timing(lsr_ctx* ctx, double* seconds4, int* n_chunks) {
    if (!ctx || !seconds4) return LSR_ERR_ARG;
    if (!ctx->pipeline) return lsr_fail(ctx, LSR_ERR_STATE, "lsr_recognize_host_timing: lsr_recognize_host has not run");
    seconds4[0] = ctx->pipeline->t_wait;
    seconds4[1] = ctx->pipeline->t_pack;
    seconds4[2] = ctx->pipeline->t_submit;
    seconds4[3] = ctx->pipeline->t_drain;
    if (n_chunks) *n_chunks = ctx->pipeline->last_chunks;
    return LSR_OK;
}
