// Operand layout of v_mfma_f64_4x4x4_4b_f64 found by experiment: for every pair of lanes (la, lb) the
// product of A = [lane == la] and B = [lane == lb] is computed; a non-zero output lane ld means that
// A's element at la and B's element at lb meet (same block, same k) and land at D's element ld.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-result tools/mfma_4x4x4_probe.hip -o tools/bin/mfma_4x4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(int* table) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0, d = 0.0;
            asm volatile("s_nop 4\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7" : "+v"(d) : "v"(a), "v"(b));
            if (d != 0.0) table[la * 64 + lb] = lane;
        }
}

int main() {
    int* t;
    hipMalloc(&t, 4096 * sizeof(int));
    hipMemset(t, 0xff, 4096 * sizeof(int));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, t);
    static int h[4096];
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    // print, for each A lane, the B lanes it meets and where the product lands
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (h[la * 64 + lb] >= 0) printf("  B%2d->D%2d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
