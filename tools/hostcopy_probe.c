/* Host-side packing rate of the upload pipeline: N clips (separate heap arrays) copied into one staging buffer by
 * worker threads, with memcpy and with non-temporal stores.  gcc -O2 -pthread tools/hostcopy_probe.c -o build/hostcopy_probe
 *   usage: hostcopy_probe [threads] [clips] [bytes per clip] */
#define _GNU_SOURCE
#include <emmintrin.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static void copy_nt(void* dst, const void* src, size_t n) {
    unsigned char* d = (unsigned char*)dst;
    const unsigned char* s = (const unsigned char*)src;
    size_t head = (16 - ((uintptr_t)d & 15)) & 15;
    if (head > n) head = n;
    memcpy(d, s, head);
    d += head, s += head, n -= head;
    size_t blocks = n / 64;
    for (size_t i = 0; i < blocks; ++i) {
        __m128i a = _mm_loadu_si128((const __m128i*)s), b = _mm_loadu_si128((const __m128i*)(s + 16));
        __m128i c = _mm_loadu_si128((const __m128i*)(s + 32)), e = _mm_loadu_si128((const __m128i*)(s + 48));
        _mm_stream_si128((__m128i*)d, a);
        _mm_stream_si128((__m128i*)(d + 16), b);
        _mm_stream_si128((__m128i*)(d + 32), c);
        _mm_stream_si128((__m128i*)(d + 48), e);
        s += 64, d += 64;
    }
    memcpy(d, s, n - blocks * 64);
    _mm_sfence();
}

struct job {
    unsigned char** clips;
    size_t bytes;
    unsigned char* stage;
    int first, last, nt;
};

static void* run(void* p) {
    struct job* j = (struct job*)p;
    for (int i = j->first; i < j->last; ++i) {
        if (j->nt) copy_nt(j->stage + (size_t)i * j->bytes, j->clips[i], j->bytes);
        else memcpy(j->stage + (size_t)i * j->bytes, j->clips[i], j->bytes);
    }
    return 0;
}

int main(int argc, char** argv) {
    int threads = argc > 1 ? atoi(argv[1]) : 16, clips = argc > 2 ? atoi(argv[2]) : 10000;
    size_t bytes = argc > 3 ? (size_t)atol(argv[3]) : 142360;
    unsigned char** c = malloc(sizeof(*c) * clips);
    for (int i = 0; i < clips; ++i) {
        c[i] = malloc(bytes);
        memset(c[i], i, bytes);
    }
    unsigned char* stage = aligned_alloc(4096, (size_t)clips * bytes + 4096);
    memset(stage, 0, (size_t)clips * bytes);
    pthread_t th[64];
    struct job jobs[64];
    for (int nt = 0; nt < 2; ++nt)
        for (int rep = 0; rep < 4; ++rep) {
            double t0 = now();
            for (int t = 0; t < threads; ++t) {
                jobs[t] = (struct job){c, bytes, stage, (int)((long)clips * t / threads), (int)((long)clips * (t + 1) / threads), nt};
                pthread_create(&th[t], 0, run, &jobs[t]);
            }
            for (int t = 0; t < threads; ++t) pthread_join(th[t], 0);
            double dt = now() - t0;
            printf("%s rep %d: %.1f ms, %.1f GB/s\n", nt ? "non-temporal" : "memcpy", rep, dt * 1e3, clips * (double)bytes / dt / 1e9);
        }
    return stage[12345] == 77;
}
