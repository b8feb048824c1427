// Context, model/graph upload, workspace, profiling and the public C entry points.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "lsr_internal.h"

int lsr_fail(lsr_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    return code;
}

int lsr_reserve(lsr_ctx* ctx, lsr_buffer* buf, size_t bytes, const char* what) {
    if (bytes <= buf->bytes) return LSR_OK;
    LSR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (buf->ptr) LSR_HIP_CHECK(ctx, hipFree(buf->ptr));
    buf->ptr = nullptr;
    buf->bytes = 0;
    size_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&buf->ptr, want);
    if (e != hipSuccess)
        return lsr_fail(ctx, LSR_ERR_HIP, "hipMalloc(%zu bytes) for %s failed: %s", want, what,
                        hipGetErrorString(e));
    buf->bytes = want;
    return LSR_OK;
}

// ---- profiling ----------------------------------------------------------------------------
static hipEvent_t take_event(lsr_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void lsr_prof_begin(lsr_ctx* ctx, int kernel) {
    ctx->prof_open = false;
    if (!ctx->profiling || !(ctx->profile_mask & (1 | LSR_PROFILE_ONLY(kernel)))) return;
    lsr_ctx::pending p;
    p.kernel = kernel;
    p.start = take_event(ctx);
    p.stop = take_event(ctx);
    if (!p.start || !p.stop) return;
    (void)hipEventRecord(p.start, ctx->stream);
    ctx->pend.push_back(p);
    ctx->prof_open = true;
}

void lsr_prof_end(lsr_ctx* ctx) {
    if (!ctx->prof_open || ctx->pend.empty()) return;
    ctx->prof_open = false;
    (void)hipEventRecord(ctx->pend.back().stop, ctx->stream);
}

static void drain_profile(lsr_ctx* ctx) {
    for (auto& p : ctx->pend) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            ctx->prof_ms[p.kernel] += ms;
            ctx->prof_n[p.kernel] += 1;
        }
        ctx->event_pool.push_back(p.start);
        ctx->event_pool.push_back(p.stop);
    }
    ctx->pend.clear();
}

extern "C" int lsr_profile_enable(lsr_ctx* ctx, int on) {
    if (!ctx) return LSR_ERR_ARG;
    ctx->profiling = on != 0;
    ctx->profile_mask = on;
    return LSR_OK;
}

extern "C" int lsr_profile_reset(lsr_ctx* ctx) {
    if (!ctx) return LSR_ERR_ARG;
    LSR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    drain_profile(ctx);
    for (int k = 0; k < LSR_K_COUNT; ++k) {
        ctx->prof_ms[k] = 0;
        ctx->prof_n[k] = 0;
    }
    return LSR_OK;
}

extern "C" int lsr_profile_get(lsr_ctx* ctx, int kernel_id, double* total_ms, int64_t* launches) {
    if (!ctx || kernel_id < 0 || kernel_id >= LSR_K_COUNT) return LSR_ERR_ARG;
    LSR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    drain_profile(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
    if (launches) *launches = ctx->prof_n[kernel_id];
    return LSR_OK;
}

// ---- host table builders --------------------------------------------------------------------
extern "C" void lsr_host_hann_window(double* out) {
    // scipy.signal.get_window("hann", 320, fftbins=True): 0.5 + 0.5*cos(linspace(-pi, pi, 321)[:-1])
    const double step = (M_PI - (-M_PI)) / 320.0;
    for (int i = 0; i < LSR_N_FFT; ++i) out[i] = 0.5 + 0.5 * cos((double)i * step + (-M_PI));
}

static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3;
    const double min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    if (f >= min_log_hz) return min_log_mel + log(f / min_log_hz) / logstep;
    return f / f_sp;
}

static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3;
    const double min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    if (m >= min_log_mel) return min_log_hz * exp(logstep * (m - min_log_mel));
    return f_sp * m;
}

extern "C" void lsr_host_mel_filterbank(float* out) {
    // librosa.filters.mel(sr=16000, n_fft=320, n_mels=40, fmin=133.33, fmax=6855.4976,
    // htk=False, norm="slaney"): triangles between consecutive mel points evaluated at the
    // 50 Hz bin centres in float64, stored as float32, then scaled by 2/(f[i+2]-f[i]).
    const int nm = LSR_N_MELS, nb = LSR_N_BINS;
    double mel_f[LSR_N_MELS + 2];
    const double lo = hz_to_mel(133.33), hi = hz_to_mel(6855.4976);
    const double step = (hi - lo) / (nm + 1);
    for (int i = 0; i < nm + 2; ++i) mel_f[i] = mel_to_hz(i == nm + 1 ? hi : (double)i * step + lo);
    for (int i = 0; i < nm; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < nb; ++k) {
            const double fk = 50.0 * k;
            const double lower = -(mel_f[i] - fk) / fd0;
            const double upper = (mel_f[i + 2] - fk) / fd1;
            double w = lower < upper ? lower : upper;
            if (!(w > 0)) w = 0;
            const float w32 = (float)w;
            out[i * nb + k] = (float)((double)w32 * enorm);
        }
    }
}

extern "C" void lsr_host_dct_matrix(double* out) {
    // scipy.fftpack.dct(type=2, norm="ortho") rows 0..12 over 40 inputs
    for (int k = 0; k < LSR_N_MFCC; ++k) {
        const double scale = k == 0 ? sqrt(1.0 / LSR_N_MELS) : sqrt(2.0 / LSR_N_MELS);
        for (int n = 0; n < LSR_N_MELS; ++n)
            out[k * LSR_N_MELS + n] = scale * cos(M_PI * k * (2 * n + 1) / (2.0 * LSR_N_MELS));
    }
}

// ---- context ----------------------------------------------------------------------------------
// Device copy of a host table.  The allocation is kept while the size stays the same (a trainer uploads a
// model of the same shape every iteration); the copy is ordered after the work already queued on the
// context's stream, which may still be reading the previous contents.
template <typename T>
static int upload(lsr_ctx* ctx, T** dst, const T* src, size_t n, const char* what) {
    const size_t bytes = n * sizeof(T);
    auto it = ctx->table_bytes.find((void*)*dst);
    if (*dst && (it == ctx->table_bytes.end() || it->second != bytes)) {
        LSR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (it != ctx->table_bytes.end()) ctx->table_bytes.erase(it);
        LSR_HIP_CHECK(ctx, hipFree(*dst));
        *dst = nullptr;
    }
    if (!*dst) {
        hipError_t e = hipMalloc((void**)dst, bytes);
        if (e != hipSuccess)
            return lsr_fail(ctx, LSR_ERR_HIP, "hipMalloc for %s failed: %s", what, hipGetErrorString(e));
        ctx->table_bytes[(void*)*dst] = bytes;
    }
    LSR_HIP_CHECK(ctx, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    LSR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // the source may be a temporary
    return LSR_OK;
}

int lsr_frontend_init(lsr_ctx* ctx) {
    std::vector<double> win(LSR_N_FFT), tw_stage(11 * 16 * 2), dct(LSR_N_MFCC * LSR_N_MELS);
    std::vector<float> melw(LSR_N_MELS * LSR_N_BINS);
    std::vector<float> slot_w(64 * 6, 0.f);
    std::vector<int> slot_info(64, 0), band_slots(LSR_N_MELS * 2, 0);
    lsr_host_hann_window(win.data());
    lsr_host_mel_filterbank(melw.data());
    lsr_host_dct_matrix(dct.data());
    for (int k1 = 0; k1 < 11; ++k1)
        for (int n2 = 0; n2 < 16; ++n2) {
            const double ang = -2.0 * M_PI * (double)(n2 * k1) / 320.0;
            tw_stage[(k1 * 16 + n2) * 2 + 0] = cos(ang);
            tw_stage[(k1 * 16 + n2) * 2 + 1] = sin(ang);
        }
    // The non-zero weights of a band are one run of consecutive bins; cut every run into slots of
    // at most 6 bins.  The kernel gives one lane to each slot, so there must be at most 64.
    int n_slots = 0;
    for (int b = 0; b < LSR_N_MELS; ++b) {
        int first = LSR_N_BINS, last = 0;
        for (int k = 0; k < LSR_N_BINS; ++k)
            if (melw[b * LSR_N_BINS + k] != 0.f) {
                if (k < first) first = k;
                last = k + 1;
            }
        if (first > last) first = last = 0;
        band_slots[2 * b] = n_slots;
        for (int k0 = first; k0 < last; k0 += 6) {
            if (n_slots >= 64) return lsr_fail(ctx, LSR_ERR_LIMIT, "mel filterbank needs more than 64 slots");
            int start = k0;
            if (start + 6 > LSR_N_BINS) start = LSR_N_BINS - 6;      // keep the 6-bin read inside the spectrum
            slot_info[n_slots] = start;
            for (int k = k0; k < last && k < k0 + 6; ++k) slot_w[n_slots * 6 + (k - start)] = melw[b * LSR_N_BINS + k];
            ++n_slots;
        }
        band_slots[2 * b + 1] = n_slots - band_slots[2 * b];
        if (band_slots[2 * b + 1] == 0) return lsr_fail(ctx, LSR_ERR_LIMIT, "empty mel band");
    }
    int rc;
    if ((rc = upload(ctx, &ctx->d_window, win.data(), win.size(), "window"))) return rc;
    if ((rc = upload(ctx, &ctx->d_tw_stage, tw_stage.data(), tw_stage.size(), "twiddles"))) return rc;
    if ((rc = upload(ctx, &ctx->d_slot_w, slot_w.data(), slot_w.size(), "mel slot weights"))) return rc;
    if ((rc = upload(ctx, &ctx->d_slot_info, slot_info.data(), slot_info.size(), "mel slots"))) return rc;
    if ((rc = upload(ctx, &ctx->d_band_slots, band_slots.data(), band_slots.size(), "mel bands"))) return rc;
    if ((rc = upload(ctx, &ctx->d_dct, dct.data(), dct.size(), "dct matrix"))) return rc;
    return LSR_OK;
}

static void free_ptr(void* p) {
    if (p) (void)hipFree(p);
}

void lsr_frontend_free(lsr_ctx* ctx) {
    free_ptr(ctx->d_window);
    free_ptr(ctx->d_tw_stage);
    free_ptr(ctx->d_slot_w);
    free_ptr(ctx->d_slot_info);
    free_ptr(ctx->d_band_slots);
    free_ptr(ctx->d_dct);
}

extern "C" int lsr_ctx_create(int device, lsr_ctx** out) {
    if (!out) return LSR_ERR_ARG;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || device < 0 || device >= count) {
        fprintf(stderr, "lsr_ctx_create(%d): hipGetDeviceCount -> %s, %d device(s)\n", device, hipGetErrorString(e), count);
        return LSR_ERR_HIP;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        fprintf(stderr, "lsr_ctx_create(%d): hipSetDevice -> %s\n", device, hipGetErrorString(e));
        return LSR_ERR_HIP;
    }
    lsr_ctx* ctx = new lsr_ctx();
    ctx->device = device;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
    int rc = lsr_frontend_init(ctx);
    if (rc == LSR_OK) rc = lsr_noalloc_init(ctx);
    if (rc != LSR_OK) {
        fprintf(stderr, "lsr_ctx_create: %s\n", ctx->error.c_str());
        lsr_frontend_free(ctx);
        delete ctx;
        return rc;
    }
    *out = ctx;
    return LSR_OK;
}

extern "C" int lsr_ctx_destroy(lsr_ctx* ctx) {
    if (!ctx) return LSR_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    lsr_pipeline_destroy(ctx);
    lsr_comm_destroy(ctx);
    lsr_frontend_free(ctx);
    free_ptr(ctx->d_wfrag);
    free_ptr(ctx->d_lognorm);
    free_ptr(ctx->d_logw);
    for (int k = 0; k < LSR_MODEL_SLOTS; ++k)
        if (k != ctx->cur_model) {
            free_ptr(ctx->models[k].d_wfrag);
            free_ptr(ctx->models[k].d_lognorm);
            free_ptr(ctx->models[k].d_logw);
        }
    free_ptr(ctx->d_band);
    free_ptr(ctx->d_flags);
    free_ptr(ctx->d_word_of);
    free_ptr(ctx->d_lower_of);
    free_ptr(ctx->d_upper_of);
    lsr_buffer* bufs[] = {&ctx->ws_frame_utt, &ctx->ws_mel, &ctx->ws_uttmax, &ctx->ws_feats,
                          &ctx->ws_ll, &ctx->ws_tracer, &ctx->ws_slab, &ctx->ws_sep, &ctx->ws_frame_info, &ctx->ws_sort,
                          &ctx->ws_mask};
    for (lsr_buffer* b : bufs) free_ptr(b->ptr);
    for (auto& p : ctx->pend) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    delete ctx;
    return LSR_OK;
}

extern "C" const char* lsr_last_error(const lsr_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

extern "C" int lsr_set_stream(lsr_ctx* ctx, void* hip_stream) {
    if (!ctx) return LSR_ERR_ARG;
    ctx->stream = (hipStream_t)hip_stream;
    return LSR_OK;
}

extern "C" int lsr_synchronize(lsr_ctx* ctx) {
    if (!ctx) return LSR_ERR_ARG;
    LSR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return LSR_OK;
}

// ---- model ---------------------------------------------------------------------------------------
extern "C" int lsr_model_upload(lsr_ctx* ctx, int n_states, const double* mean, const double* whiten,
                                const double* log_norm) {
    if (!ctx) return LSR_ERR_ARG;
    if (n_states < 1 || !mean || !whiten || !log_norm) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_model_upload: bad argument");
    if (n_states > 4096) return lsr_fail(ctx, LSR_ERR_LIMIT, "lsr_model_upload: %d states > 4096", n_states);
    const int D = LSR_DIM;
    bool triangular = true;          // zero above the diagonal: feature k only feeds components j <= k
    for (int s = 0; s < n_states && triangular; ++s)
        for (int k = 0; k < D; ++k)
            for (int j = k + 1; j < D; ++j)
                if (whiten[((size_t)s * D + k) * D + j] != 0.0) triangular = false;
    // A operands of v_mfma_f64_4x4x4_4b_f64: for output rows 4 jb .. 4 jb + 3 and features 4 kb .. 4 kb + 3 the
    // 16 values [kk][i] = U[4 kb + kk][4 jb + i] (every lane reads the entry of its k = lane >> 4 and
    // i = lane & 3).  Feature 39 is the constant 1 the kernel appends to every frame and carries
    // -(mean U): the offset of the whitened vector comes out of the matrix product.  The blocks of a
    // state are stored in the order the kernel uses them (a triangular U only has the blocks with
    // kb >= jb, 55 padded to 56); 20 zero blocks past the
    // end let the kernel read ahead without a bounds test.
    const int slots = triangular ? LSR_TRI_SLOTS : LSR_JBLOCKS * LSR_KSTEPS;
    std::vector<double> frag(((size_t)n_states * slots + 20) * 16, 0.0);
    for (int s = 0; s < n_states; ++s) {
        // c[j] = sum_k mean[k] U[k][j] (extended precision, rounded once)
        double c[LSR_DIM];
        for (int j = 0; j < D; ++j) {
            long double acc = 0.0L;
            for (int k = 0; k < D; ++k)
                acc += (long double)mean[(size_t)s * D + k] * (long double)whiten[((size_t)s * D + k) * D + j];
            c[j] = -(double)acc;
        }
        // order of use: row blocks two at a time (jp, jp + 1), their feature blocks alternating
        int v = 0;
        for (int jp = 0; jp < LSR_JBLOCKS; jp += 2) {
            const int first[2] = {triangular ? jp : 0, triangular ? jp + 1 : 0};
            for (int t = 0; t < LSR_KSTEPS; ++t)
                for (int h = 0; h < 2; ++h) {
                    const int jb = jp + h, kb = first[h] + t;
                    if (kb >= LSR_KSTEPS) continue;
                    for (int kk = 0; kk < 4; ++kk)
                        for (int i = 0; i < 4; ++i) {
                            const int k = 4 * kb + kk, j = 4 * jb + i;
                            double val = 0.0;
                            if (j < D) val = k < D ? whiten[((size_t)s * D + k) * D + j] : c[j];
                            // places 2p and 2p + 1 are interleaved: [p][kk][i][place & 1]
                            frag[(((((size_t)s * slots + v) / 2) * 4 + kk) * 4 + i) * 2 + (v & 1)] = val;
                        }
                    ++v;
                }
        }
    }
    ctx->whiten_triangular = triangular;
    int rc;
    if ((rc = upload(ctx, &ctx->d_wfrag, frag.data(), frag.size(), "whitening fragments"))) return rc;
    if ((rc = upload(ctx, &ctx->d_lognorm, log_norm, (size_t)n_states, "log normalisers"))) return rc;
    ctx->n_states = n_states;
    ctx->n_mix = 1;
    return LSR_OK;
}

extern "C" int lsr_model_slot(lsr_ctx* ctx, int slot) {
    if (!ctx) return LSR_ERR_ARG;
    if (slot < 0 || slot >= LSR_MODEL_SLOTS)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_model_slot: slot %d outside 0..%d", slot, LSR_MODEL_SLOTS - 1);
    if (slot == ctx->cur_model) return LSR_OK;
    // only pointers change hands: kernels already queued keep reading the buffers they were launched with
    lsr_ctx::model_slot& out = ctx->models[ctx->cur_model];
    out.n_states = ctx->n_states;
    out.n_mix = ctx->n_mix;
    out.whiten_triangular = ctx->whiten_triangular;
    out.d_wfrag = ctx->d_wfrag;
    out.d_lognorm = ctx->d_lognorm;
    out.d_logw = ctx->d_logw;
    const lsr_ctx::model_slot& in = ctx->models[slot];
    ctx->n_states = in.n_states;
    ctx->n_mix = in.n_mix;
    ctx->whiten_triangular = in.whiten_triangular;
    ctx->d_wfrag = in.d_wfrag;
    ctx->d_lognorm = in.d_lognorm;
    ctx->d_logw = in.d_logw;
    ctx->cur_model = slot;
    return LSR_OK;
}

extern "C" int lsr_model_set_mixtures(lsr_ctx* ctx, int n_mix, const double* log_weights) {
    if (!ctx) return LSR_ERR_ARG;
    if (ctx->n_states == 0) return lsr_fail(ctx, LSR_ERR_STATE, "lsr_model_set_mixtures: no model uploaded");
    if (n_mix < 1 || ctx->n_states % n_mix != 0 || (n_mix > 1 && !log_weights))
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_model_set_mixtures: %d Gaussians cannot form states of %d components", ctx->n_states, n_mix);
    if (n_mix > 1) {
        for (int i = 0; i < ctx->n_states; ++i)
            if (log_weights[i] != log_weights[i]) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_model_set_mixtures: NaN weight");
        int rc = upload(ctx, &ctx->d_logw, log_weights, (size_t)ctx->n_states, "mixture weights");
        if (rc) return rc;
    }
    ctx->n_mix = n_mix;
    return LSR_OK;
}

extern "C" int lsr_loop_graph_upload(lsr_ctx* ctx, int n_states, int n_words, const float* band,
                                     const int32_t* word_lower, const int32_t* word_upper, int silence_word) {
    if (!ctx) return LSR_ERR_ARG;
    if (!band || !word_lower || !word_upper || n_words < 1 || n_states < 1)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_loop_graph_upload: bad argument");
    if (n_states > LSR_MAX_STATES)
        return lsr_fail(ctx, LSR_ERR_LIMIT, "lsr_loop_graph_upload: %d states > %d", n_states, LSR_MAX_STATES);
    if (silence_word >= n_words) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_loop_graph_upload: silence_word out of range");
    const int GS = LSR_GRAPH_STRIDE;
    std::vector<float> b(3 * GS, -INFINITY);
    std::vector<uint8_t> flags(GS, 0), word_of(GS, 0), lower_of(GS, 0), upper_of(GS, 0);
    int expect = 0;
    for (int w = 0; w < n_words; ++w) {
        if (word_lower[w] != expect || word_upper[w] < word_lower[w] || word_upper[w] >= n_states)
            return lsr_fail(ctx, LSR_ERR_ARG, "lsr_loop_graph_upload: words must tile the states in order");
        for (int s = word_lower[w]; s <= word_upper[w]; ++s) {
            word_of[s] = (uint8_t)w;
            lower_of[s] = (uint8_t)word_lower[w];
            upper_of[s] = (uint8_t)word_upper[w];
        }
        flags[word_lower[w]] |= 1;
        flags[word_upper[w]] |= 2;
        expect = word_upper[w] + 1;
    }
    if (expect != n_states) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_loop_graph_upload: words do not cover all states");
    for (int d = 0; d < 3; ++d)
        for (int s = 0; s < n_states; ++s) {
            const float v = band[d * n_states + s];
            if (v != v) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_loop_graph_upload: NaN transition");
            b[d * GS + s] = v;
        }
    int rc;
    if ((rc = upload(ctx, &ctx->d_band, b.data(), b.size(), "band"))) return rc;
    if ((rc = upload(ctx, &ctx->d_flags, flags.data(), flags.size(), "flags"))) return rc;
    if ((rc = upload(ctx, &ctx->d_word_of, word_of.data(), word_of.size(), "word map"))) return rc;
    if ((rc = upload(ctx, &ctx->d_lower_of, lower_of.data(), lower_of.size(), "word map"))) return rc;
    if ((rc = upload(ctx, &ctx->d_upper_of, upper_of.data(), upper_of.size(), "word map"))) return rc;
    ctx->g_states = n_states;
    ctx->g_words = n_words;
    ctx->g_silence = silence_word;
    return LSR_OK;
}

extern "C" int lsr_loop_set_between(lsr_ctx* ctx, double between, int between_is_f64) {
    if (!ctx) return LSR_ERR_ARG;
    if (between != between) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_loop_set_between: NaN");
    ctx->between = between;
    ctx->between_is_f64 = between_is_f64 != 0;
    return LSR_OK;
}

// ---- compute entry points ------------------------------------------------------------------------
static int check_batch(lsr_ctx* ctx, const char* fn, int n_utts, int64_t total_frames) {
    if (n_utts < 1 || total_frames < n_utts) return lsr_fail(ctx, LSR_ERR_ARG, "%s: empty batch", fn);
    if (total_frames > ((int64_t)1 << 31) - 64)
        return lsr_fail(ctx, LSR_ERR_LIMIT, "%s: %lld frames > 2^31", fn, (long long)total_frames);
    return LSR_OK;
}

static int mfcc_any(lsr_ctx* ctx, const void* audio_d, int is_i16, const int64_t* sample_off_d,
                    const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!audio_d || !sample_off_d || !frame_off_d || !feats_d) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_mfcc_batch: null pointer");
    int rc = check_batch(ctx, "lsr_mfcc_batch", n_utts, total_frames);
    if (rc) return rc;
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return lsr_launch_mfcc(ctx, audio_d, is_i16, sample_off_d, frame_off_d, n_utts, total_frames, feats_d);
}

extern "C" int lsr_mfcc_batch(lsr_ctx* ctx, const float* audio_d, const int64_t* sample_off_d,
                              const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d) {
    return mfcc_any(ctx, audio_d, 0, sample_off_d, frame_off_d, n_utts, total_frames, feats_d);
}

extern "C" int lsr_mfcc_batch_i16(lsr_ctx* ctx, const int16_t* audio_d, const int64_t* sample_off_d,
                                  const int64_t* frame_off_d, int n_utts, int64_t total_frames, float* feats_d) {
    return mfcc_any(ctx, audio_d, 1, sample_off_d, frame_off_d, n_utts, total_frames, feats_d);
}

extern "C" int lsr_emission_batch(lsr_ctx* ctx, const float* feats_d, int64_t total_frames, float* ll_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!feats_d || !ll_d) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_emission_batch: null pointer");
    if (ctx->n_states == 0) return lsr_fail(ctx, LSR_ERR_STATE, "lsr_emission_batch: no model uploaded");
    if (ctx->n_states / ctx->n_mix > 128)
        return lsr_fail(ctx, LSR_ERR_LIMIT, "lsr_emission_batch: %d states > 128", ctx->n_states / ctx->n_mix);
    int rc = check_batch(ctx, "lsr_emission_batch", 1, total_frames);
    if (rc) return rc;
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return lsr_launch_emission(ctx, feats_d, total_frames, ll_d);
}

static size_t ll_bytes(const lsr_ctx* ctx, int64_t total_frames) {
    // whole tiles of 64 frames, and one more: the decoder requests scores up to fifteen rows past an utterance's end
    return (size_t)((total_frames + 63) / 64 * 64 + 64) * lsr_ctx_ll_stride(ctx) * sizeof(float);
}

extern "C" int lsr_ll_stride(const lsr_ctx* ctx) { return ctx ? lsr_ctx_ll_stride(ctx) : LSR_LL_STRIDE; }

extern "C" int lsr_decode_loop_batch(lsr_ctx* ctx, const float* feats_d, const int64_t* frame_off_d, int n_utts,
                                     int64_t total_frames, int max_frames, float* scores_d, int8_t* paths_d,
                                     uint8_t* labels_d, int32_t* n_labels_d, int label_cap) {
    if (!ctx) return LSR_ERR_ARG;
    if (!feats_d || !frame_off_d) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_decode_loop_batch: null pointer");
    if (ctx->n_states == 0 || ctx->g_states == 0)
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_decode_loop_batch: model and loop graph must be uploaded first");
    if (ctx->n_states / ctx->n_mix != ctx->g_states)
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_decode_loop_batch: model has %d states, graph %d", ctx->n_states / ctx->n_mix, ctx->g_states);
    if (labels_d && label_cap < 1) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_decode_loop_batch: label_cap < 1");
    int rc = check_batch(ctx, "lsr_decode_loop_batch", n_utts, total_frames);
    if (rc) return rc;
    if (max_frames < 1) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_decode_loop_batch: max_frames < 1");
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if ((rc = lsr_reserve(ctx, &ctx->ws_ll, ll_bytes(ctx, total_frames), "emission scores"))) return rc;
    if ((rc = lsr_launch_emission(ctx, feats_d, total_frames, (float*)ctx->ws_ll.ptr))) return rc;
    return lsr_launch_viterbi_loop(ctx, (const float*)ctx->ws_ll.ptr, frame_off_d, n_utts, total_frames, max_frames,
                                   scores_d, paths_d, labels_d, n_labels_d, label_cap);
}

extern "C" int lsr_decode_loop_sweep(lsr_ctx* ctx, const float* feats_d, const int64_t* frame_off_d, int n_utts,
                                     int64_t total_frames, int max_frames, const double* between,
                                     const int32_t* between_is_f64, int n_between, float* scores_d, uint8_t* labels_d,
                                     int32_t* n_labels_d, int label_cap) {
    if (!ctx) return LSR_ERR_ARG;
    if (!feats_d || !frame_off_d || !between || !between_is_f64 || !labels_d || !n_labels_d)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_decode_loop_sweep: null pointer");
    if (n_between < 1 || label_cap < 1 || max_frames < 1)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_decode_loop_sweep: bad sizes");
    for (int i = 0; i < n_between; ++i)
        if (between[i] != between[i]) return lsr_fail(ctx, LSR_ERR_ARG, "lsr_decode_loop_sweep: NaN penalty");
    if (ctx->n_states == 0 || ctx->g_states == 0)
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_decode_loop_sweep: model and loop graph must be uploaded first");
    if (ctx->n_states / ctx->n_mix != ctx->g_states)
        return lsr_fail(ctx, LSR_ERR_STATE, "lsr_decode_loop_sweep: model has %d states, graph %d",
                        ctx->n_states / ctx->n_mix, ctx->g_states);
    int rc = check_batch(ctx, "lsr_decode_loop_sweep", n_utts, total_frames);
    if (rc) return rc;
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if ((rc = lsr_reserve(ctx, &ctx->ws_ll, ll_bytes(ctx, total_frames), "emission scores"))) return rc;
    if ((rc = lsr_launch_emission(ctx, feats_d, total_frames, (float*)ctx->ws_ll.ptr))) return rc;
    const double saved = ctx->between;
    const int saved_wide = ctx->between_is_f64;
    for (int i = 0; i < n_between && rc == LSR_OK; ++i) {
        ctx->between = between[i];
        ctx->between_is_f64 = between_is_f64[i] != 0;
        rc = lsr_launch_viterbi_loop(ctx, (const float*)ctx->ws_ll.ptr, frame_off_d, n_utts, total_frames, max_frames,
                                     scores_d ? scores_d + (size_t)i * n_utts : nullptr, nullptr,
                                     labels_d + (size_t)i * n_utts * label_cap, n_labels_d + (size_t)i * n_utts,
                                     label_cap);
    }
    ctx->between = saved;
    ctx->between_is_f64 = saved_wide;
    return rc;
}

int lsr_reserve_recognize(lsr_ctx* ctx, int n_utts, int64_t total_frames) {
    int rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_frame_utt, (size_t)total_frames * sizeof(int32_t), "frame index"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_mel, (size_t)total_frames * LSR_N_MELS * sizeof(float), "mel energies"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_uttmax, (size_t)n_utts * sizeof(unsigned int), "utterance maxima"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_frame_info, (size_t)total_frames * 16, "frame records"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_feats, (size_t)total_frames * LSR_DIM * sizeof(float), "features"))) return rc;
    if ((rc = lsr_reserve(ctx, &ctx->ws_ll, ll_bytes(ctx, total_frames), "emission scores"))) return rc;
    return lsr_reserve(ctx, &ctx->ws_tracer,
                       ctx->g_states > 64 ? (size_t)total_frames * 128 : LSR_TRACER_BYTES(total_frames, n_utts), "back-pointers");
}

static int recognize_any(lsr_ctx* ctx, const void* audio_d, int is_i16, const int64_t* sample_off_d,
                         const int64_t* frame_off_d, int n_utts, int64_t total_frames, int max_frames,
                         float* scores_d, int8_t* paths_d, uint8_t* labels_d, int32_t* n_labels_d, int label_cap) {
    if (!ctx) return LSR_ERR_ARG;
    int rc = check_batch(ctx, "lsr_recognize_batch", n_utts, total_frames);
    if (rc) return rc;
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if ((rc = lsr_reserve(ctx, &ctx->ws_feats, (size_t)total_frames * LSR_DIM * sizeof(float), "features"))) return rc;
    if ((rc = mfcc_any(ctx, audio_d, is_i16, sample_off_d, frame_off_d, n_utts, total_frames, (float*)ctx->ws_feats.ptr)))
        return rc;
    return lsr_decode_loop_batch(ctx, (const float*)ctx->ws_feats.ptr, frame_off_d, n_utts, total_frames, max_frames,
                                 scores_d, paths_d, labels_d, n_labels_d, label_cap);
}

extern "C" int lsr_recognize_batch(lsr_ctx* ctx, const float* audio_d, const int64_t* sample_off_d,
                                   const int64_t* frame_off_d, int n_utts, int64_t total_frames, int max_frames,
                                   float* scores_d, int8_t* paths_d, uint8_t* labels_d, int32_t* n_labels_d,
                                   int label_cap) {
    return recognize_any(ctx, audio_d, 0, sample_off_d, frame_off_d, n_utts, total_frames, max_frames, scores_d, paths_d,
                         labels_d, n_labels_d, label_cap);
}

extern "C" int lsr_recognize_batch_i16(lsr_ctx* ctx, const int16_t* audio_d, const int64_t* sample_off_d,
                                       const int64_t* frame_off_d, int n_utts, int64_t total_frames, int max_frames,
                                       float* scores_d, int8_t* paths_d, uint8_t* labels_d, int32_t* n_labels_d,
                                       int label_cap) {
    return recognize_any(ctx, audio_d, 1, sample_off_d, frame_off_d, n_utts, total_frames, max_frames, scores_d, paths_d,
                         labels_d, n_labels_d, label_cap);
}

static int decode_chain_any(lsr_ctx* ctx, const char* fn, bool wide, const float* feats_d, const int64_t* frame_off_d,
                            int n_utts, int64_t total_frames, int max_frames, const int64_t* chain_off_d,
                            int64_t total_chain, const int32_t* chain_state_d, const float* chain_band_d,
                            const uint8_t* chain_start_d, float* scores_d, int8_t* paths_d, float* final_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!feats_d || !frame_off_d || !chain_off_d || !chain_state_d || !chain_band_d || !chain_start_d)
        return lsr_fail(ctx, LSR_ERR_ARG, "%s: null pointer", fn);
    if (ctx->n_states == 0) return lsr_fail(ctx, LSR_ERR_STATE, "%s: no model uploaded", fn);
    if (ctx->n_states / ctx->n_mix > 128)
        return lsr_fail(ctx, LSR_ERR_LIMIT, "%s: %d model states > 128", fn, ctx->n_states / ctx->n_mix);
    int rc = check_batch(ctx, fn, n_utts, total_frames);
    if (rc) return rc;
    if (max_frames < 1 || total_chain < n_utts) return lsr_fail(ctx, LSR_ERR_ARG, "%s: bad sizes", fn);
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if ((rc = lsr_reserve(ctx, &ctx->ws_ll, ll_bytes(ctx, total_frames), "emission scores"))) return rc;
    // chains that hold only part of the model's states (forced alignments) need only those columns
    const int n_cols = ctx->n_states / ctx->n_mix;
    bool reachable = ctx->chain_emission_mode == LSR_CHAIN_EMISSION_REACHABLE ||
                     (ctx->chain_emission_mode == LSR_CHAIN_EMISSION_AUTO && 4 * total_chain < 3 * (int64_t)n_utts * n_cols);
    if (ctx->n_mix > 1) reachable = false;
    if (reachable)
        rc = lsr_launch_emission_chains(ctx, feats_d, total_frames, (float*)ctx->ws_ll.ptr, frame_off_d, n_utts, chain_off_d,
                                        chain_state_d);
    else
        rc = lsr_launch_emission(ctx, feats_d, total_frames, (float*)ctx->ws_ll.ptr);
    if (rc) return rc;
    return lsr_launch_viterbi_chain(ctx, (const float*)ctx->ws_ll.ptr, frame_off_d, n_utts, total_frames, max_frames,
                                    chain_off_d, total_chain, chain_state_d, chain_band_d, chain_start_d, scores_d,
                                    paths_d, final_d, wide);
}

extern "C" int lsr_decode_chain_batch(lsr_ctx* ctx, const float* feats_d, const int64_t* frame_off_d, int n_utts,
                                      int64_t total_frames, int max_frames, const int64_t* chain_off_d,
                                      int64_t total_chain, const int32_t* chain_state_d, const float* chain_band_d,
                                      const uint8_t* chain_start_d, float* scores_d, int8_t* paths_d, float* final_d) {
    return decode_chain_any(ctx, "lsr_decode_chain_batch", false, feats_d, frame_off_d, n_utts, total_frames, max_frames,
                            chain_off_d, total_chain, chain_state_d, chain_band_d, chain_start_d, scores_d, paths_d,
                            final_d);
}

extern "C" int lsr_decode_chain_batch_wide(lsr_ctx* ctx, const float* feats_d, const int64_t* frame_off_d, int n_utts,
                                           int64_t total_frames, int max_frames, const int64_t* chain_off_d,
                                           int64_t total_chain, const int32_t* chain_state_d, const float* chain_band_d,
                                           const uint8_t* chain_start_d, float* scores_d, int8_t* paths_d) {
    return decode_chain_any(ctx, "lsr_decode_chain_batch_wide", true, feats_d, frame_off_d, n_utts, total_frames,
                            max_frames, chain_off_d, total_chain, chain_state_d, chain_band_d, chain_start_d, scores_d,
                            paths_d, nullptr);
}

extern "C" int lsr_chain_emission_mode(lsr_ctx* ctx, int mode) {
    if (!ctx) return LSR_ERR_ARG;
    if (mode != LSR_CHAIN_EMISSION_AUTO && mode != LSR_CHAIN_EMISSION_ALL && mode != LSR_CHAIN_EMISSION_REACHABLE)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_chain_emission_mode: unknown mode %d", mode);
    ctx->chain_emission_mode = mode;
    return LSR_OK;
}

extern "C" int lsr_accumulate_stats(lsr_ctx* ctx, const float* feats_d, const int32_t* slot_d, int64_t total_frames,
                                    int n_slots, double* count_d, double* sum_x_d, double* sum_xx_d) {
    if (!ctx) return LSR_ERR_ARG;
    if (!feats_d || !slot_d || !count_d || !sum_x_d || !sum_xx_d)
        return lsr_fail(ctx, LSR_ERR_ARG, "lsr_accumulate_stats: null pointer");
    if (n_slots < 1 || n_slots > 4096) return lsr_fail(ctx, LSR_ERR_LIMIT, "lsr_accumulate_stats: n_slots out of range");
    int rc = check_batch(ctx, "lsr_accumulate_stats", 1, total_frames);
    if (rc) return rc;
    LSR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    return lsr_launch_stats(ctx, feats_d, slot_d, total_frames, n_slots, count_d, sum_x_d, sum_xx_d);
}
