/* A plain C host of liblsr_hip.so: no Python, no PyTorch, no HIP calls of its own.
 *
 *   recognize_host <model+audio file> [chunk_samples]
 *
 * reads one binary file (layout below, written by tests/test_gpu_capi_host.py or any other producer), uploads the
 * stacked Gaussians and the word-loop graph, decodes every clip from host memory with lsr_recognize_host and prints
 * one line per clip: "<score as hex float bits> <n_labels> <label indices ...>".
 *
 * File layout (little endian): int32 S, W, silence_word, between_is_f64, B, label_cap; float64 between;
 * float64 mean[S][39], whiten[S][39][39], log_norm[S]; float32 band[3][S]; int32 word_lower[W], word_upper[W];
 * int64 n_samples[B]; int16 audio[sum n_samples].
 *
 * Build: gcc -O2 -Iinclude examples/recognize_host.c -Lcs-304-speech-recognition-code_amd/lib -llsr_hip -o recognize_host
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lsr_hip.h"

static void* read_exact(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "short read (%zu bytes wanted)\n", bytes);
        exit(2);
    }
    return p;
}

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != LSR_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, lsr_last_error(ctx)); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <file> [chunk_samples]\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    int32_t* head = (int32_t*)read_exact(f, 6 * sizeof(int32_t));
    const int S = head[0], W = head[1], silence = head[2], wide = head[3], B = head[4], cap = head[5];
    double* between = (double*)read_exact(f, sizeof(double));
    double* mean = (double*)read_exact(f, (size_t)S * LSR_DIM * sizeof(double));
    double* whiten = (double*)read_exact(f, (size_t)S * LSR_DIM * LSR_DIM * sizeof(double));
    double* log_norm = (double*)read_exact(f, (size_t)S * sizeof(double));
    float* band = (float*)read_exact(f, (size_t)3 * S * sizeof(float));
    int32_t* lower = (int32_t*)read_exact(f, (size_t)W * sizeof(int32_t));
    int32_t* upper = (int32_t*)read_exact(f, (size_t)W * sizeof(int32_t));
    int64_t* n_samples = (int64_t*)read_exact(f, (size_t)B * sizeof(int64_t));
    int64_t total = 0;
    for (int b = 0; b < B; ++b) total += n_samples[b];
    int16_t* audio = (int16_t*)read_exact(f, (size_t)total * sizeof(int16_t));
    fclose(f);

    lsr_ctx* ctx = NULL;
    if (lsr_ctx_create(0, &ctx) != LSR_OK) {
        fprintf(stderr, "no usable GPU\n");
        return 1;
    }
    CHECK(lsr_model_upload(ctx, S, mean, whiten, log_norm));
    CHECK(lsr_loop_graph_upload(ctx, S, W, band, lower, upper, silence));
    CHECK(lsr_loop_set_between(ctx, *between, wide));

    const void** clips = (const void**)malloc((size_t)B * sizeof(void*));
    int64_t at = 0;
    for (int b = 0; b < B; ++b) {          /* the clips may live anywhere in host memory; here they sit back to back */
        clips[b] = audio + at;
        at += n_samples[b];
    }
    float* scores = (float*)malloc((size_t)B * sizeof(float));
    uint8_t* labels = (uint8_t*)malloc((size_t)B * cap);
    int32_t* n_labels = (int32_t*)malloc((size_t)B * sizeof(int32_t));
    const int64_t chunk = argc > 2 ? atoll(argv[2]) : 0;
    CHECK(lsr_recognize_host(ctx, clips, n_samples, B, /*audio_is_i16=*/1, chunk, /*n_threads=*/4, scores, labels,
                             n_labels, cap));
    for (int b = 0; b < B; ++b) {
        uint32_t bits;
        memcpy(&bits, &scores[b], 4);
        printf("%08x %d", bits, n_labels[b]);
        for (int i = 0; i < n_labels[b] && i < cap; ++i) printf(" %d", labels[(size_t)b * cap + i]);
        printf("\n");
    }
    lsr_ctx_destroy(ctx);
    return 0;
}
