// Microbenchmark: issue cost of v_min_f64 / v_fma_f64 / v_med3_f32 / v_cmp+cndmask on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 65536
template <int MODE>
__global__ void k(double* out, double seed, long long* cyc) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = seed * 3;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, g = (float)b;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) {
            asm volatile("v_min_f64 %0, %0, %4\n v_min_f64 %1, %1, %4\n v_min_f64 %2, %2, %4\n v_min_f64 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (MODE == 1) {
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (MODE == 2) {
            asm volatile("v_med3_f32 %0, %0, %4, %1\n v_med3_f32 %1, %1, %4, %2\n v_med3_f32 %2, %2, %4, %3\n v_med3_f32 %3, %3, %4, %0"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));
        } else {
            asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4"
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; long long h[1024];
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 1024 * 8);
    const char* names[4] = {"v_min_f64", "v_fma_f64", "v_med3_f32", "v_min_f32"};
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int m = 0; m < 4; ++m) {
            dim3 g(256), b(256 * waves);   // waves per SIMD (4 SIMDs per CU, one block per CU)
            if (m == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, 1.5, cyc);
            if (m == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, 1.5, cyc);
            if (m == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, 1.5, cyc);
            if (m == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, 1.5, cyc);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (m == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, 1.5, cyc);
            if (m == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, 1.5, cyc);
            if (m == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, 1.5, cyc);
            if (m == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, 1.5, cyc);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
            // cycles per wave-instruction per SIMD = elapsed / (instr per wave) / waves-per-SIMD
            // wall: instructions per SIMD = N*4*waves; ns per instruction per SIMD
            printf("%-11s waves/SIMD=%d : %.2f ticks/instr/wave (counter)  |  wall %.3f ms -> %.3f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n",
                   names[m], waves, s / 256 / (N * 4.0), ms, ms * 1e6 / (N * 4.0 * waves), ms * 1e6 / (N * 4.0 * waves) * 2.4);
        }
    return 0;
}
