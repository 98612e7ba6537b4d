// bw_probe.hip -- what HBM bandwidth can an Adam-like 3-read/3-write stream reach on MI355X?
// Variants: (0) grid-stride float4 over three arrays, (1) the blocked pattern of k_w1_update_fwd
// (workgroup = contiguous chunk range, wave = 1 KiB tiles, 16 KB chunk stride), with optional
// register prefetch depth.  Build: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void upd(f32x4& w, f32x4& m, f32x4& v) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] += (0.5f - m[r]) * 0.1f; v[r] += (0.25f - v[r]) * 0.001f; w[r] -= m[r] * 1e-4f / (sqrtf(v[r]) + 1e-7f); }
}

__global__ __launch_bounds__(256) void k_linear(float* W, float* M, float* V, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 w = ((f32x4*)W)[i], m = ((f32x4*)M)[i], v = ((f32x4*)V)[i];
        upd(w, m, v);
        ((f32x4*)W)[i] = w; ((f32x4*)M)[i] = m; ((f32x4*)V)[i] = v;
    }
}

// blocked: workgroup b owns chunks [b*cpw, (b+1)*cpw); chunk = 16 KB per array = 4096 floats;
// WAVES waves, each wave owns 16KB/WAVES per chunk in 1 KiB pieces; PF = prefetch depth (0,1,2)
template <int WAVES, int PF>
__global__ __launch_bounds__(WAVES * 64) void k_blocked(float* W, float* M, float* V, int cpw) {
    constexpr int T = 16 / WAVES;   // 1 KiB tiles per wave per chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * cpw * 4096 + (size_t)wave * T * 256 + lane * 4;
    f32x4 w[PF + 1][T], m[PF + 1][T], v[PF + 1][T];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const size_t i = base + (size_t)(p < cpw ? p : cpw - 1) * 4096 + t * 256;
            w[p][t] = *(f32x4*)(W + i); m[p][t] = *(f32x4*)(M + i); v[p][t] = *(f32x4*)(V + i);
        }
    for (int c = 0; c < cpw; ++c) {
        const int cn = c + PF < cpw ? c + PF : cpw - 1;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const size_t i = base + (size_t)cn * 4096 + t * 256;
            w[PF][t] = *(f32x4*)(W + i); m[PF][t] = *(f32x4*)(M + i); v[PF][t] = *(f32x4*)(V + i);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const size_t i = base + (size_t)c * 4096 + t * 256;
            upd(w[0][t], m[0][t], v[0][t]);
            *(f32x4*)(W + i) = w[0][t]; *(f32x4*)(M + i) = m[0][t]; *(f32x4*)(V + i) = v[0][t];
        }
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int t = 0; t < T; ++t) { w[p][t] = w[p + 1][t]; m[p][t] = m[p + 1][t]; v[p][t] = v[p + 1][t]; }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static void timeit(const char* name, F launch, double bytes) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a)); const int R = 20;
    for (int i = 0; i < R; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-40s %8.1f us  %7.0f GB/s\n", name, 1e3 * ms / R, bytes / (ms / R * 1e-3) / 1e9);
}

int main() {
    const size_t nfl = (size_t)6144 * 4096;          // 100.7 MB per array (cfg3: 98 MB)
    float *W, *M, *V;
    CK(hipMalloc(&W, nfl * 4)); CK(hipMalloc(&M, nfl * 4)); CK(hipMalloc(&V, nfl * 4));
    CK(hipMemset(W, 0, nfl * 4)); CK(hipMemset(M, 0, nfl * 4)); CK(hipMemset(V, 0, nfl * 4));
    const double bytes = 6.0 * nfl * 4;
    for (int g : {512, 1024, 2048, 4096})
        { char nm[64]; snprintf(nm, 64, "linear grid=%d", g); timeit(nm, [&] { hipLaunchKernelGGL(k_linear, dim3(g), dim3(256), 0, 0, W, M, V, nfl / 4); }, bytes); }
#define BLK(WV, PF, NWG) { char nm[64]; snprintf(nm, 64, "blocked waves=%d pf=%d wgs=%d", WV, PF, NWG); \
        timeit(nm, [&] { hipLaunchKernelGGL((k_blocked<WV, PF>), dim3(NWG), dim3(WV * 64), 0, 0, W, M, V, 6144 / NWG); }, bytes); }
    BLK(8, 0, 256) BLK(8, 1, 256) BLK(8, 2, 256) BLK(8, 0, 512) BLK(8, 1, 512) BLK(8, 2, 512)
    BLK(4, 0, 512) BLK(4, 1, 512) BLK(4, 0, 768) BLK(4, 1, 768) BLK(4, 0, 1024) BLK(4, 1, 1024) BLK(4, 0, 1536) BLK(4, 0, 2048)
    BLK(16, 0, 256) BLK(16, 1, 256) BLK(16, 2, 256) BLK(16, 1, 512)
    return 0;
}
