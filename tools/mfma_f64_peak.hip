// Micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 on every SIMD of the chip
// (register operands only, NACC independent accumulators per wave, WPS waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_f64_peak && /tmp/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu) {
    const int cus = 256, iters = 4000;
    double* out;
    hipMalloc(&out, sizeof(double) * cus * blocks_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = 2048.0 * NACC * iters * (double)cus * blocks_per_cu * 4;
        if (rep == 2)
            printf("nacc=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NACC,
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
    return 0;
}
