// bw_probe2.hip -- which ingredient of k_w1_update_fwd costs HBM bandwidth?  Adam-like 3R/3W
// stream (8 waves/WG, 2 x 1KiB tiles per wave per array per chunk, prefetch 1) plus optional:
// NM f32 MFMAs per wave per chunk, a workgroup barrier per chunk, LDS operand reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void upd(f32x4& w, f32x4& m, f32x4& v, f32x4 g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] += (g[r] - m[r]) * 0.1f; v[r] += (g[r] * g[r] - v[r]) * 0.001f;
        w[r] -= m[r] * 1e-4f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v[r]) + 1e-7f); }
}

template <int WAVES, int NM, bool BAR, bool LDSR, int CHAINS>
__global__ __launch_bounds__(WAVES * 64) void k(float* W, float* M, float* V, int cpw, float seed) {
    constexpr int T = 16 / WAVES;
    __shared__ float sm[2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (LDSR) { for (int i = threadIdx.x; i < 2048; i += WAVES * 64) sm[i] = seed * i; __syncthreads(); }
    const size_t base = (size_t)blockIdx.x * cpw * 4096 + (size_t)wave * T * 256 + lane * 4;
    f32x4 w[2][T], m[2][T], v[2][T];
    f32x4 acc[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) acc[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < T; ++t) { const size_t i = base + t * 256; w[0][t] = *(f32x4*)(W + i); m[0][t] = *(f32x4*)(M + i); v[0][t] = *(f32x4*)(V + i); }
    for (int c = 0; c < cpw; ++c) {
        const int cn = c + 1 < cpw ? c + 1 : cpw - 1;
#pragma unroll
        for (int t = 0; t < T; ++t) { const size_t i = base + (size_t)cn * 4096 + t * 256; w[1][t] = *(f32x4*)(W + i); m[1][t] = *(f32x4*)(M + i); v[1][t] = *(f32x4*)(V + i); }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 g[T];
#pragma unroll
        for (int t = 0; t < T; ++t) g[t] = (f32x4){seed, seed, seed, seed};
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const float a = LDSR ? sm[(64 * i + lane) & 2047] : seed + i;
            acc[i % CHAINS] = MFMA16(a, seed, acc[i % CHAINS]);
        }
        if (NM) {
#pragma unroll
            for (int t = 0; t < T; ++t) g[t] += acc[t % CHAINS];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const size_t i = base + (size_t)c * 4096 + t * 256;
            upd(w[0][t], m[0][t], v[0][t], g[t]);
            *(f32x4*)(W + i) = w[0][t]; *(f32x4*)(M + i) = m[0][t]; *(f32x4*)(V + i) = v[0][t];
        }
#pragma unroll
        for (int t = 0; t < T; ++t) { w[0][t] = w[1][t]; m[0][t] = m[1][t]; v[0][t] = v[1][t]; }
        if (BAR) __syncthreads();
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename F> static void timeit(const char* name, F launch, double bytes, double flops) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a)); const int R = 20;
    for (int i = 0; i < R; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-46s %8.1f us  %7.0f GB/s %6.1f TF\n", name, 1e3 * ms / R, bytes / (ms / R * 1e-3) / 1e9, flops / (ms / R * 1e-3) / 1e12);
}
int main() {
    const size_t nfl = (size_t)6144 * 4096;
    float *W, *M, *V; CK(hipMalloc(&W, nfl * 4)); CK(hipMalloc(&M, nfl * 4)); CK(hipMalloc(&V, nfl * 4));
    CK(hipMemset(W, 0, nfl * 4)); CK(hipMemset(M, 0, nfl * 4)); CK(hipMemset(V, 0, nfl * 4));
    const double bytes = 6.0 * nfl * 4;
#define RUN(WV, NM, BAR, LDSR, CH, NWG) { char nm[96]; snprintf(nm, 96, "waves=%d mfma=%d bar=%d lds=%d chains=%d wgs=%d", WV, NM, BAR, LDSR, CH, NWG); \
    timeit(nm, [&] { hipLaunchKernelGGL((k<WV, NM, BAR, LDSR, CH>), dim3(NWG), dim3(WV * 64), 0, 0, W, M, V, 6144 / NWG, 0.f); }, bytes, 6144.0 * WV * NM * 2048); }
    RUN(8, 0, false, false, 2, 256) RUN(8, 0, true, false, 2, 256)
    RUN(8, 32, false, false, 2, 256) RUN(8, 64, false, false, 2, 256) RUN(8, 64, false, false, 8, 256) RUN(8, 64, true, false, 8, 256)
    RUN(8, 64, false, true, 8, 256) RUN(8, 64, true, true, 8, 256) RUN(8, 64, true, true, 2, 256)
    RUN(16, 32, false, false, 1, 256) RUN(16, 32, false, false, 4, 256) RUN(16, 32, true, true, 4, 256) RUN(16, 32, true, true, 1, 256)
    RUN(8, 64, false, true, 8, 512) RUN(16, 32, false, true, 4, 512) RUN(4, 128, false, true, 8, 512) RUN(4, 128, false, true, 8, 768) RUN(4, 128, false, true, 8, 1024)
    RUN(8, 128, false, false, 8, 256) RUN(16, 0, false, false, 1, 256)
    return 0;
}
