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
#include <atomic>
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
    void* device = nullptr;          // same layout on the device (lsr_recognize_host only)
    size_t bytes = 0, device_bytes = 0;
    hipEvent_t copied = nullptr;     // upload of the slot's chunk has finished
    hipEvent_t decoded = nullptr;    // whatever reads the slot (kernels, or the copy itself) has finished
    bool busy = false;
};

struct lsr_pipeline {
    hipStream_t copy_stream = nullptr;
    hipEvent_t fence = nullptr;      // position of the context's stream when an upload starts
    lsr_pipe_slot slot[PIPE_SLOTS];
    lsr_buffer out_dev;              // [scores f32[B] | n_labels i32[B] | labels u8[B][cap]]
    void* out_pinned = nullptr;
    size_t out_pinned_bytes = 0;
    lsr_pool* pool = nullptr;
    int pool_threads = 0;
    double t_pack = 0, t_wait = 0, t_submit = 0, t_drain = 0;      // host seconds of the last call, by phase
    int last_chunks = 0;
};

// Copy used by the packing threads.  A clip is read once and the pinned staging chunk is only read back by the DMA
// engine, so the stores go around the cache (non-temporal): memcpy's ordinary stores first read every destination
// line, three memory transfers per byte instead of two (16 threads of a pool host pack 128 GB/s this way, 68 GB/s
// with memcpy).  With check_f32 the bytes are float32 values and the return value tells whether all of them are
// finite (the test rides on the copy: the data passes through registers anyway).
static bool copy_streaming(void* dst, const void* src, size_t n, bool check_f32) {
    unsigned char* d = (unsigned char*)dst;
    const unsigned char* s = (const unsigned char*)src;
    auto finite_scalar = [](const unsigned char* p, size_t bytes) {
        bool ok = true;
        for (size_t i = 0; i + 4 <= bytes; i += 4) {
            uint32_t u;
            memcpy(&u, p + i, 4);
            ok = ok && (u & 0x7f800000u) != 0x7f800000u;
        }
        return ok;
    };
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    typedef long long v2ll __attribute__((vector_size(16)));
    typedef long long v2ll_u __attribute__((vector_size(16), aligned(1)));
    typedef unsigned int v4u __attribute__((vector_size(16)));
    if (n < 512) {
        memcpy(d, s, n);
        return !check_f32 || finite_scalar(s, n);
    }
    bool ok = true;
    size_t head = (16 - ((uintptr_t)d & 15)) & 15;            // non-temporal stores need 16-byte aligned addresses
    if (check_f32) head = (head + 3) & ~(size_t)3;            // whole values (float arrays start on 4-byte boundaries,
                                                              // so a 16-byte boundary is a whole number of values away)
    memcpy(d, s, head);
    if (check_f32) ok = finite_scalar(s, head);
    d += head, s += head, n -= head;
    const size_t blocks = n / 64;
    const v4u expo = {0x7f800000u, 0x7f800000u, 0x7f800000u, 0x7f800000u};
    v4u bad = {0u, 0u, 0u, 0u};                               // lanes that saw an all-ones exponent (infinity or NaN)
    for (size_t i = 0; i < blocks; ++i, s += 64, d += 64) {
        const v2ll a = *(const v2ll_u*)s, b = *(const v2ll_u*)(s + 16), c = *(const v2ll_u*)(s + 32), e = *(const v2ll_u*)(s + 48);
        if (check_f32)
            bad |= (v4u)((((v4u)a & expo) == expo) | (((v4u)b & expo) == expo) | (((v4u)c & expo) == expo) |
                         (((v4u)e & expo) == expo));
        __builtin_nontemporal_store(a, (v2ll*)d);
        __builtin_nontemporal_store(b, (v2ll*)(d + 16));
        __builtin_nontemporal_store(c, (v2ll*)(d + 32));
        __builtin_nontemporal_store(e, (v2ll*)(d + 48));
    }
    memcpy(d, s, n - blocks * 64);
    if (check_f32) ok = ok && !(bad[0] | bad[1] | bad[2] | bad[3]) && finite_scalar(s, n - blocks * 64);
    __builtin_ia32_sfence();      // the stores are globally visible before the chunk is handed to the copy engine
    return ok;
#else
    memcpy(d, s, n);
    return !check_f32 || finite_scalar(s, n);
#endif
}

static double pipe_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void pipe_free_slot(lsr_pipe_slot& s) {
    if (s.pinned) (void)hipHostFree(s.pinned);
    if (s.device) (void)hipFree(s.device);
    s.pinned = s.device = nullptr;
    s.bytes = s.device_bytes = 0;
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
    if (p->fence) (void)hipEventDestroy(p->fence);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    delete p->pool;
    delete p;
    ctx->pipeline = nullptr;
}

static int pipe_get(lsr_ctx* ctx, int threads, lsr_pipeline** out) {
    if (!ctx->pipeline) {
        // the context only sees a pipeline whose stream and events all exist; a creation that fails half way is
        // taken apart again (lsr_pipeline_destroy skips what was never created)
        ctx->pipeline = new lsr_pipeline();
        lsr_pipeline* p = ctx->pipeline;
        hipError_t e = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->fence, hipEventDisableTiming);
        for (auto& s : p->slot) {
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.decoded, hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            lsr_pipeline_destroy(ctx);
            return lsr_fail(ctx, LSR_ERR_HIP, "creating the upload pipeline failed: %s", hipGetErrorString(e));
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

// staging memory is only resized or handed to a new chunk when the copy or kernels that use it have finished
static int pipe_wait_idle(lsr_ctx* ctx, lsr_pipeline* p) {
    for (auto& s : p->slot)
        if (s.busy) {
            LSR_HIP_CHECK(ctx, hipEventSynchronize(s.decoded));
            s.busy = false;
        }
    return LSR_OK;
}

static int pipe_reserve_slot(lsr_ctx* ctx, lsr_pipe_slot& s, size_t bytes, bool with_device) {
    const size_t want = bytes + bytes / 8 + 4096;
    if (bytes > s.bytes) {
        if (s.pinned) LSR_HIP_CHECK(ctx, hipHostFree(s.pinned));
        s.pinned = nullptr;
        s.bytes = 0;
        LSR_HIP_CHECK(ctx, hipHostMalloc(&s.pinned, want, hipHostMallocDefault));
        s.bytes = want;
    }
    if (with_device && bytes > s.device_bytes) {
        if (s.device) LSR_HIP_CHECK(ctx, hipFree(s.device));
        s.device = nullptr;
        s.device_bytes = 0;
        LSR_HIP_CHECK(ctx, hipMalloc(&s.device, want));
        s.device_bytes = want;
    }
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
    // measured optimum on the config-3 batch: 96 MB chunks of float32, 24 MB chunks of int16 (smaller chunks start the
    // first copy earlier, which weighs more when the whole transfer is half as long)
    if (chunk_samples <= 0) chunk_samples = (int64_t)(audio_is_i16 ? 12 : 24) << 20;
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
    if ((rc = pipe_wait_idle(ctx, p))) return rc;
    for (int s = 0; s < PIPE_SLOTS && s < n_chunks; ++s)
        if ((rc = pipe_reserve_slot(ctx, p->slot[s], max_bytes, true))) return rc;
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
        const bool check = !audio_is_i16 && (ctx->host_check_finite & 1);
        std::atomic<int> bad_clip(INT32_MAX);              // lowest clip of the chunk with a non-finite sample
        p->pool->parallel_for(tasks, [&](int t) {
            for (int i = cut[t]; i < cut[t + 1]; ++i)
                if (!copy_streaming(stage + (size_t)soff[i] * esz, clips[b0 + i], (size_t)n_samples[b0 + i] * esz, check)) {
                    int seen = bad_clip.load();
                    while (b0 + i < seen && !bad_clip.compare_exchange_weak(seen, b0 + i)) {}
                }
        });
        if (bad_clip.load() != INT32_MAX)
            return lsr_fail(ctx, LSR_ERR_ARG, "lsr_recognize_host: clip %d holds a non-finite sample (audio must be finite)",
                            bad_clip.load());
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

extern "C" int lsr_host_check_finite(lsr_ctx* ctx, int flags) {
    if (!ctx) return LSR_ERR_ARG;
    if (flags < 0 || flags > 3) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_host_check_finite: flags outside 0..3");
    ctx->host_check_finite = flags;
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

// Host arrays -> one device buffer, back to back: the upload half of the pipeline above without the decoding.
// Worker threads pack the arrays of a chunk into pinned staging memory (no concatenation on the caller's side),
// the copy stream moves chunk after chunk while later chunks are packed.  The copies are ordered after everything
// the context's stream has queued so far (the destination may be memory that earlier kernels still use) and the
// context's stream waits for the last copy, so kernels launched afterwards see the data; the call itself returns
// when every array has been packed, not when the last copy has landed.
extern "C" int lsr_upload_host(lsr_ctx* ctx, const void* const* arrays, const int64_t* n_elems, int64_t n_arrays,
                               int elem_bytes, int64_t chunk_bytes, int n_threads, void* dst_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!arrays || !n_elems || !dst_d) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_upload_host: null pointer");
    if (n_arrays < 1 || n_arrays > INT32_MAX || elem_bytes < 1 || elem_bytes > 16)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_upload_host: bad sizes");
    if (chunk_bytes <= 0) chunk_bytes = (int64_t)96 << 20;
    if (n_threads <= 0) n_threads = 8;
    if (n_threads > 64) n_threads = 64;
    for (int64_t i = 0; i < n_arrays; ++i)
        if (n_elems[i] < 0 || (n_elems[i] > 0 && !arrays[i]))
            return lsr_fail(ctx, LSR_ERR_ARG, "lsr_upload_host: array %lld is null or has a negative size", (long long)i);
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    lsr_pipeline* p = nullptr;
    int rc = pipe_get(ctx, n_threads, &p);
    if (rc) return rc;
    const int n = (int)n_arrays;
    std::vector<int64_t> off(n + 1);                      // byte offset of every array in the destination
    off[0] = 0;
    for (int i = 0; i < n; ++i) off[i + 1] = off[i] + n_elems[i] * elem_bytes;
    std::vector<int> first;                                // chunks of consecutive arrays (at least one per chunk)
    first.push_back(0);
    for (int i = 0; i < n; ++i)
        if (off[i] - off[first.back()] > 0 && off[i + 1] - off[first.back()] > chunk_bytes) first.push_back(i);
    first.push_back(n);
    const int n_chunks = (int)first.size() - 1;
    size_t max_bytes = 0;
    for (int c = 0; c < n_chunks; ++c) max_bytes = std::max(max_bytes, (size_t)(off[first[c + 1]] - off[first[c]]));
    bool grow = false;
    for (int s = 0; s < PIPE_SLOTS && s < n_chunks; ++s) grow = grow || max_bytes > p->slot[s].bytes;
    if (grow && (rc = pipe_wait_idle(ctx, p))) return rc;
    for (int s = 0; s < PIPE_SLOTS && s < n_chunks; ++s)
        if ((rc = pipe_reserve_slot(ctx, p->slot[s], max_bytes, false))) return rc;
    LSR_HIP_CHECK(ctx, hipEventRecord(p->fence, ctx->stream));
    LSR_HIP_CHECK(ctx, hipStreamWaitEvent(p->copy_stream, p->fence, 0));
    hipEvent_t last = nullptr;
    for (int c = 0; c < n_chunks; ++c) {
        lsr_pipe_slot& s = p->slot[c % PIPE_SLOTS];
        if (s.busy) {
            LSR_HIP_CHECK(ctx, hipEventSynchronize(s.decoded));
            s.busy = false;
        }
        const int a0 = first[c], na = first[c + 1] - a0;
        const int64_t base = off[a0], bytes = off[a0 + na] - base;
        if (bytes == 0) continue;
        unsigned char* stage = (unsigned char*)s.pinned;
        const int tasks = std::min(na, 4 * n_threads);
        std::vector<int> cut(tasks + 1);
        for (int t = 0, a = 0; t <= tasks; ++t) {
            const int64_t goal = bytes * t / tasks;
            while (a < na && off[a0 + a] - base < goal) ++a;
            cut[t] = t == tasks ? na : a;
        }
        const bool check = elem_bytes == 4 && (ctx->host_check_finite & 2);
        std::atomic<int> bad_array(INT32_MAX);
        p->pool->parallel_for(tasks, [&](int t) {
            for (int i = a0 + cut[t]; i < a0 + cut[t + 1]; ++i)
                if (n_elems[i] && !copy_streaming(stage + (off[i] - base), arrays[i], (size_t)(off[i + 1] - off[i]), check)) {
                    int seen = bad_array.load();
                    while (i < seen && !bad_array.compare_exchange_weak(seen, i)) {}
                }
        });
        if (bad_array.load() != INT32_MAX) {
            if (last) LSR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, last, 0));      // copies already queued stay ordered
            return lsr_fail(ctx, LSR_ERR_ARG, "lsr_upload_host: array %d holds a non-finite float32 value", bad_array.load());
        }
        LSR_HIP_CHECK(ctx, hipMemcpyAsync((unsigned char*)dst_d + base, s.pinned, (size_t)bytes, hipMemcpyHostToDevice,
                                          p->copy_stream));
        LSR_HIP_CHECK(ctx, hipEventRecord(s.decoded, p->copy_stream));      // the slot is free again after its copy
        s.busy = true;
        last = s.decoded;
    }
    if (last) LSR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, last, 0));
    return LSR_OK;
}

// One device buffer -> host memory: the copy stream moves chunk after chunk into pinned staging memory while worker
// threads copy the previous chunk to its place in the destination (a pageable destination written by one
// synchronous copy reaches a fraction of the link's rate, and fresh pages are faulted in by several threads here).
// The copies are ordered after everything queued on the context's stream; the call returns when dst is complete.
extern "C" int lsr_download_host(lsr_ctx* ctx, const void* src_d, int64_t bytes, void* dst, int64_t chunk_bytes,
                                 int n_threads) {
    if (!ctx) return LSR_ERR_ARG;
    if (!src_d || !dst || bytes < 0) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_download_host: bad argument");
    if (bytes == 0) return LSR_OK;
    if (chunk_bytes <= 0) chunk_bytes = (int64_t)32 << 20;
    if (n_threads <= 0) n_threads = 8;
    if (n_threads > 64) n_threads = 64;
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    lsr_pipeline* p = nullptr;
    int rc = pipe_get(ctx, n_threads, &p);
    if (rc) return rc;
    if ((rc = pipe_wait_idle(ctx, p))) return rc;
    const int64_t n_chunks = (bytes + chunk_bytes - 1) / chunk_bytes;
    const size_t slot_bytes = (size_t)std::min(bytes, chunk_bytes);
    for (int s = 0; s < 2 && s < n_chunks; ++s)
        if ((rc = pipe_reserve_slot(ctx, p->slot[s], slot_bytes, false))) return rc;
    LSR_HIP_CHECK(ctx, hipEventRecord(p->fence, ctx->stream));
    LSR_HIP_CHECK(ctx, hipStreamWaitEvent(p->copy_stream, p->fence, 0));
    auto span = [&](int64_t c, int64_t* off) {
        *off = c * chunk_bytes;
        return std::min(chunk_bytes, bytes - *off);
    };
    for (int64_t c = 0; c <= n_chunks; ++c) {
        if (c < n_chunks) {                                  // chunk c travels into slot c % 2 ...
            int64_t off;
            const int64_t len = span(c, &off);
            lsr_pipe_slot& s = p->slot[c % 2];
            LSR_HIP_CHECK(ctx, hipMemcpyAsync(s.pinned, (const unsigned char*)src_d + off, (size_t)len, hipMemcpyDeviceToHost,
                                              p->copy_stream));
            LSR_HIP_CHECK(ctx, hipEventRecord(s.copied, p->copy_stream));
        }
        if (c >= 1) {                                        // ... while chunk c - 1 is copied to its destination
            int64_t off;
            const int64_t len = span(c - 1, &off);
            lsr_pipe_slot& s = p->slot[(c - 1) % 2];
            LSR_HIP_CHECK(ctx, hipEventSynchronize(s.copied));
            const int tasks = (int)std::min<int64_t>(4 * n_threads, (len + 65535) / 65536);
            const unsigned char* stage = (const unsigned char*)s.pinned;
            unsigned char* out = (unsigned char*)dst + off;
            p->pool->parallel_for(tasks, [&](int t) {
                const int64_t a = len * t / tasks, b = len * (t + 1) / tasks;
                memcpy(out + a, stage + a, (size_t)(b - a));
            });
        }
    }
    return LSR_OK;
}
