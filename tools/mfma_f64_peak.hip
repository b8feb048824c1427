// Micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 on every SIMD of the chip
// (register operands only, NACC independent accumulators per wave, WPS waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/mfma_f64_peak.hip -o /tmp/mfma_f64_peak && /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
// 64 KiB of (unused) dynamic LDS per workgroup: at most two workgroups fit a compute unit, so a grid of
// 2 x CUs workgroups is spread evenly (without it the dispatcher stacks 3-4 on some units)
#define LDS_PER_BLOCK 65536

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));   // accumulators stay in VGPRs
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Same loop with NV float64 FMAs (or NV 32-bit integer adds when INTOPS) per 4 MFMAs on independent
// registers: shows whether ordinary vector instructions take issue slots from the float64 MFMAs.
template <int NV, bool INTOPS>
__global__ __launch_bounds__(256) void kmix(double* out, int iters, double a0, double b0) {
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    double f[8];
    int n[8];
    for (int i = 0; i < 8; ++i) { f[i] = a0 * i; n[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));   // accumulators stay in VGPRs
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (INTOPS) {
                n[v & 7] = n[v & 7] * 3 + it;
                asm volatile("" : "+v"(n[v & 7]));
            } else {
                f[v & 7] = __builtin_fma(f[v & 7], b, a);
                asm volatile("" : "+v"(f[v & 7]));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += f[i] + n[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool INTOPS>
void run_mix(int blocks_per_cu) {
    const int cus = 256, iters = 40000;
    double* out;
    hipMalloc(&out, sizeof(double) * cus * blocks_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kmix<NV, INTOPS>), dim3(cus * blocks_per_cu), dim3(256), LDS_PER_BLOCK, 0, out, iters, 1.0, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = 2048.0 * 4 * iters * (double)cus * blocks_per_cu * 4;
        if (rep == 2)
            printf("4 MFMA + %d %s per iteration, waves/SIMD=%d  %.3f ms  %.1f TFLOP/s (MFMA only)  %.1f cycles per iteration per SIMD at 2.4 GHz\n",
                   NV, INTOPS ? "int mul-add" : "f64 fma", blocks_per_cu, ms, flops / ms / 1e9,
                   ms * 1e-3 * 2.4e9 / ((double)iters * blocks_per_cu));
    }
    hipFree(out);
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction (256 MACs), one accumulator value per lane
template <int NACC>
__global__ __launch_bounds__(256) void k4(double* out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run4(int blocks_per_cu) {
    const int cus = 256, iters = 40000;
    double* out;
    hipMalloc(&out, sizeof(double) * cus * blocks_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k4<NACC>, dim3(cus * blocks_per_cu), dim3(256), LDS_PER_BLOCK, 0, out, iters, 1.0, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = 512.0 * NACC * iters * (double)cus * blocks_per_cu * 4;
        if (rep == 2)
            printf("4x4x4_4b nacc=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NACC,
                   blocks_per_cu, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (NACC * (double)iters * blocks_per_cu));
    }
    hipFree(out);
}

template <int NACC>
void run(int blocks_per_cu, int cus = 256) {
    const int iters = 40000;
    double* out;
    hipMalloc(&out, sizeof(double) * cus * blocks_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(cus * blocks_per_cu), dim3(256), LDS_PER_BLOCK, 0, out, iters, 1.0, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = 2048.0 * NACC * iters * (double)cus * blocks_per_cu * 4;
        if (rep == 2)
            printf("CUs=%d nacc=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", cus, NACC,
                   blocks_per_cu, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (NACC * (double)iters * blocks_per_cu));
    }
    hipFree(out);
}

int main() {
    run<1>(1);
    run<4>(1);
    run<8>(1);
    run<4>(2);
    run<8>(2);
    run4<2>(2);
    run4<4>(1);
    run4<4>(2);
    run4<8>(1);
    run4<8>(2);
    run4<16>(2);
    run_mix<0, false>(2);
    run_mix<2, false>(2);
    run_mix<4, false>(2);
    run_mix<8, false>(2);
    run_mix<2, true>(2);
    run_mix<4, true>(2);
    run_mix<8, true>(2);
    return 0;
}
