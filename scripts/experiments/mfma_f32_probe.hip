// Ceiling probe for the exact-f32 MFMA (round 5): how many TFLOP/s does v_mfma_f32_32x32x2_f32 sustain on this box when nothing but the
// matrix pipe is busy -- NACC independent accumulators per wave, W waves per SIMD, long enough for the clock to settle?
// Build: hipcc -O3 --offload-arch=gfx950 scripts/experiments/mfma_f32_probe.hip -o scripts/experiments/bin/mfma_f32_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(int wgs_per_cu, int iters) {
    const int cus = 256, wgs = cus * wgs_per_cu;
    float *out;
    hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<NACC>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)wgs * 4 * iters * 8 * NACC * (2.0 * 32 * 32 * 2);
        printf("NACC=%d waves/SIMD=%d iters=%d rep %d: %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, iters, rep, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<1>(1, 20000);
    run<2>(1, 10000);
    run<4>(1, 5000);
    run<1>(2, 20000);
    run<3>(2, 6000);
    run<3>(4, 6000);
    run<4>(2, 20000);      // ~60 ms: long enough for the clock to settle under load
    return 0;
}
