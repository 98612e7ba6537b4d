// mfma_probe_bf16.hip -- issue rate of v_mfma_f32_16x16x32_bf16 / 32x32x16 / 16x16x16 on gfx950 with 32 independent accumulators
// (the shape of k_predict_bf16's inner loop), no memory traffic.   hipcc -O3 --offload-arch=gfx950 -o /tmp/p tools/mfma_probe_bf16.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + threadIdx.x + i); b[i] = (__bf16)(1.f + i); }
    float s = 0.f;
    if (KIND == 0) {          // 16x16x32
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    } else if (KIND == 1) {   // 32x32x16
        f32x16 acc[NACC / 4];
        for (int i = 0; i < NACC / 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC / 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC / 4; ++i) s += acc[i][0] + acc[i][15];
    } else {                  // 16x16x16 (_1k)
        f32x4 acc[NACC];
        s16x4 a4 = {1, 2, 3, (short)threadIdx.x}, b4 = {4, 5, 6, 7};
        for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, double flop_per_mfma, int per_iter) {
    float* out; hipMalloc(&out, 1 << 24);
    const int iters = 4000;
    for (int wpc = 1; wpc <= 2; ++wpc) {
        const int grid = 256 * wpc;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<KIND, 32>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-10s %d wave(s)/SIMD: %.3f ms -> %.1f TFLOP/s\n", name, wpc, ms, flop_per_mfma * per_iter * iters * 4.0 * grid / (ms * 1e-3) / 1e12);
        }
    }
    hipFree(out);
}
int main() {
    run<0>("16x16x32", 2.0 * 16 * 16 * 32, 32);
    run<1>("32x32x16", 2.0 * 32 * 32 * 16, 8);
    run<2>("16x16x16", 2.0 * 16 * 16 * 16, 32);
    return 0;
}
