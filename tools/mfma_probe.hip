// mfma_probe.hip -- issue rate of v_mfma_f32_16x16x4_f32 on gfx950: clocks per MFMA for one wave per SIMD and for
// two, with 16 independent accumulators (the shape of k_predict's inner loop), no memory traffic at all.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_probe tools/mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 1 << 24); hipMalloc(&clk, 8 * 4096);
    const int iters = 2000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.f, 2.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[512]; hipMemcpy(h, clk, 8 * grid, hipMemcpyDeviceToHost);
            double mean = 0; for (int i = 0; i < grid; ++i) mean += (double)h[i]; mean /= grid;
            const double n_mfma = (double)iters * 64;   // per wave
            printf("%d wave(s)/SIMD: %.1f memtime ticks per MFMA per wave; kernel %.3f ms -> %.2f ns per MFMA per SIMD -> %.1f TFLOP/s\n", wgs_per_cu, mean / n_mfma,
                   ms, 1e6 * ms / (n_mfma * wgs_per_cu), 2048.0 * n_mfma * 4 * grid / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
