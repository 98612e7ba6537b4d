// bw_probe3.hip -- what do the scattered X-tile reads of the B1F1 kernel cost next to its
// 3R/3W state stream, and does the request granularity matter?  16 waves/WG, 1 WG/CU, 1 KiB tile
// per wave per array per chunk, two chunks in flight; side reads: 64 "batch rows" (stride 9600 B, random
// rows of a 50000-row matrix), G bytes per row every G/64 chunks (same bytes for every G), NS row sets.
// CAVEAT: the side loads sit under a runtime condition, so the waitcnt pass is conservative here; the
// differences this probe shows did NOT carry over to the real kernel (see tools/k_probe_tl.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int G, int NS>   // G: bytes per row per request group (64, 128, 256, 512); NS: 0, 1 or 2 scattered tiles
__global__ __launch_bounds__(1024) void k(float* W, float* M, float* V, const float* X, const int* rows, int cpw, float* sink) {
    constexpr int PER = G / 64;                 // chunks per group
    constexpr int F4 = 64 * G / 16;             // float4 per tile group
    constexpr int LPT = (F4 + 1023) / 1024;     // loads per thread per tile group
    __shared__ float sm[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * cpw * 4096 + (size_t)wave * 256 + lane * 4;
    f32x4 w[3], m[3], v[3], xs[2][LPT];
    const float* xp[2][LPT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int e = (tid + 1024 * j) % F4, row = e / (G / 16), q = e % (G / 16);
            xp[s][j] = X + (size_t)(blockIdx.x / 6) * 50000 * 2400 + (size_t)rows[64 * s + row] * 2400 + (size_t)(blockIdx.x % 6) * cpw * 16 + 4 * q;   // 6 WGs share a sub-net's rows, disjoint column ranges
            xs[s][j] = (f32x4){0, 0, 0, 0};
        }
    auto ld = [&](int i, int c) { const size_t a = base + (size_t)(c < cpw ? c : cpw - 1) * 4096; w[i] = *(f32x4*)(W + a); m[i] = *(f32x4*)(M + a); v[i] = *(f32x4*)(V + a); };
    ld(0, 0); ld(1, 1);
    f32x4 acc = (f32x4){0, 0, 0, 0};
    for (int c = 0; c < cpw; c += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int cc = c + u;
            ld((u + 2) % 3, cc + 2);
            if (NS > 0 && (cc % PER) == 0) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int j = 0; j < LPT; ++j) xs[s][j] = *(const f32x4*)(xp[s][j] + (cc / PER) * (G / 4));
            }
            __builtin_amdgcn_sched_barrier(0);
            const size_t a = base + (size_t)(cc < cpw ? cc : cpw - 1) * 4096;
#pragma unroll
            for (int r = 0; r < 4; ++r) { m[u][r] += (acc[r] - m[u][r]) * 0.1f; v[u][r] += (acc[r] * acc[r] - v[u][r]) * 0.001f;
                w[u][r] -= m[u][r] * 1e-4f * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v[u][r]) + 1e-7f); }
            *(f32x4*)(W + a) = w[u]; *(f32x4*)(M + a) = m[u]; *(f32x4*)(V + a) = v[u];
            if (NS > 0 && (cc % PER) == PER - 1) {
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int j = 0; j < LPT; ++j) *(f32x4*)(sm + 4 * ((tid + 64 * j) & 1023)) = xs[s][j];
            }
            __syncthreads();
            acc[0] += sm[lane];
        }
    }
    if (acc[0] == 123.f) sink[0] = acc[0];
}
template <typename F> static double timeit(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a)); const int R = 20;
    for (int i = 0; i < R; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1e3 * ms / R;
}
int main() {
    const int cpw = 24, wgs = 256;
    const size_t nfl = (size_t)wgs * cpw * 4096;     // 25.2 M floats per array (~ cfg3's 24.6 M)
    float *W, *M, *V, *X, *sink; int* rows;
    CK(hipMalloc(&W, nfl * 4)); CK(hipMalloc(&M, nfl * 4)); CK(hipMalloc(&V, nfl * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(W, 0, nfl * 4)); CK(hipMemset(M, 0, nfl * 4)); CK(hipMemset(V, 0, nfl * 4));
    const size_t xfl = (size_t)43 * 50000 * 2400 + 8192; CK(hipMalloc(&X, xfl * 4)); CK(hipMemset(X, 0, xfl * 4)); CK(hipDeviceSynchronize());
    std::vector<int> hr(128); for (int i = 0; i < 128; ++i) hr[i] = (int)((i * 7919LL + 13) % 50000);
    CK(hipMalloc(&rows, 512)); CK(hipMemcpy(rows, hr.data(), 512, hipMemcpyHostToDevice));
    const double bytes = 6.0 * nfl * 4;
#define RUN(G, NS) { double us = timeit([&] { hipLaunchKernelGGL((k<G, NS>), dim3(wgs), dim3(1024), 0, 0, W, M, V, X, rows, cpw, sink); }); CK(hipGetLastError()); \
    printf("G=%3d B/row/request, %d scattered tile(s): %7.1f us  state stream %5.0f GB/s\n", G, NS, us, bytes / us / 1e3); }
    RUN(64, 0) RUN(64, 1) RUN(64, 2) RUN(128, 1) RUN(128, 2) RUN(256, 1) RUN(256, 2) RUN(512, 2) RUN(64, 0)
    return 0;
}
