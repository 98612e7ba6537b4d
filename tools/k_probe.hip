// k_probe.hip -- standalone timing harness for the W1-update kernels of dimn_kernels.h on
// cfg3-shaped synthetic state (K=40, D=2400, H=256, n=50000).  Diagnostics only.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/k_probe.hip -o /tmp/k_probe && /tmp/k_probe
#include "../deepimpute_amd/csrc/dimn_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(float* p, size_t n, float scale, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = scale * ((x >> 8) * (1.0f / 16777216.0f) - 0.5f);
    }
}
template <typename F> static double timeit(F launch, int R = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < R; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1e3 * ms / R;
}
int main(int argc, char** argv) {
    const int K = 40, D = 2400, H = 256, O = 512; const int64_t n = argc > 1 ? atoll(argv[1]) : 50000;
    const float vscale = argc > 2 ? atof(argv[2]) : 1e-6f;
    Dims dm; dm.K = K; dm.H = H; dm.O = O; dm.Hp = 256; dm.Op = 512; dm.HT = 16; dm.OT = 32; dm.ldd = 258; dm.OS = 8; dm.LS = 8;
    std::vector<SubnetDev> sn(K);
    int64_t w1 = 0, x = 0;
    for (int k = 0; k < K; ++k) { sn[k].D = D; sn[k].Dp = D; sn[k].nchunk = D / 16; sn[k].kg = k; sn[k].xoff = x; sn[k].w1off = w1; w1 += (int64_t)D * 256; x += n * D; }
    float *X, *W, *M, *V, *dA, *P; CK(hipMalloc(&X, x * 4)); CK(hipMalloc(&W, w1 * 4)); CK(hipMalloc(&M, w1 * 4)); CK(hipMalloc(&V, w1 * 4));
    CK(hipMalloc(&dA, (size_t)K * 64 * 256 * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, X, (size_t)x, 4.f, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, W, (size_t)w1, 0.05f, 2u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, M, (size_t)w1, vscale, 3u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, V, (size_t)w1, vscale * vscale, 4u);   // may be negative: abs below
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, dA, (size_t)K * 64 * 256, vscale, 5u);
    std::vector<int32_t> rows(128); for (int i = 0; i < 128; ++i) rows[i] = (int32_t)((i * 7919LL + 13) % n);
    int32_t* drows; CK(hipMalloc(&drows, 512)); CK(hipMemcpy(drows, rows.data(), 512, hipMemcpyHostToDevice));
    AdamP ap{1e-4f, 0.1f, 0.001f, 1e-7f};
    const double bytes = 24.0 * w1 + 2 * 4.0 * 64 * D * K;
    for (int wgs : {256}) {
        std::vector<Work> work; int slot = 0;
        for (int k = 0; k < K; ++k) { const int nc = D / 16, ns = wgs / K + (k < wgs % K ? 1 : 0); sn[k].slot0 = slot; sn[k].nslice = ns;
            for (int i = 0; i < ns; ++i) work.push_back(Work{k, nc * i / ns, nc * (i + 1) / ns, slot++}); }
        SubnetDev* dsn; Work* dwk; CK(hipMalloc(&dsn, K * sizeof(SubnetDev))); CK(hipMalloc(&dwk, work.size() * sizeof(Work)));
        CK(hipMemcpy(dsn, sn.data(), K * sizeof(SubnetDev), hipMemcpyHostToDevice)); CK(hipMemcpy(dwk, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
        CK(hipMalloc(&P, (size_t)slot * 64 * 256 * 4));
        const unsigned g = (unsigned)work.size();
#define T(name, ...) { double us = timeit([&] { hipLaunchKernelGGL(__VA_ARGS__); }); CK(hipGetLastError()); printf("wgs=%4u %-34s %8.1f us %6.0f GB/s\n", g, name, us, bytes / us / 1e3); }
        T("sh<16,1> next", (k_w1_update_fwd_sh<16, 1>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        T("sh<16,1> no-next", (k_w1_update_fwd_sh<16, 1>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, (const int32_t*)nullptr, 0, dA, P, dm, ap)
        T("ring<16,1> next", (k_w1_update_fwd_ring<16, 1>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        T("ring<16,1> next, X_n rows == X_t rows", (k_w1_update_fwd_ring<16, 1>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows, 64, dA, P, dm, ap)
        T("ring4<16,1> next", (k_w1_update_fwd_ring<16, 1, 4>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        T("ring4<16,1> no-next", (k_w1_update_fwd_ring<16, 1, 4>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, (const int32_t*)nullptr, 0, dA, P, dm, ap)
        T("ring<16,1> no-next", (k_w1_update_fwd_ring<16, 1>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, (const int32_t*)nullptr, 0, dA, P, dm, ap)
        // half of the sub-nets on half of the CUs (what one of two lanes launches; the GB/s column assumes the full byte count):
        // measured 82 us for 320 MB = 3.9 TB/s -- 128 CUs cannot pull the whole HBM rate, so two serialised half launches
        // (164 us) lose to one full launch (110-120 us) and sub-net lanes cannot pay off
        T("ring<16,1> next, first 128 WGs only", (k_w1_update_fwd_ring<16, 1>), dim3(g / 2), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        T("ring<8,1> x2 halves next", (k_w1_update_fwd_ring<8, 1>), dim3(g, 2), dim3(512), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        // T("ring<8,2> next", (k_w1_update_fwd_ring<8, 2>), dim3(g), dim3(512), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        T("sh<8,1> x2 halves next", (k_w1_update_fwd_sh<8, 1>), dim3(g, 2), dim3(512), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        T("sh<8,1> x2 halves no-next", (k_w1_update_fwd_sh<8, 1>), dim3(g, 2), dim3(512), 0, 0, dwk, dsn, X, W, M, V, drows, 64, (const int32_t*)nullptr, 0, dA, P, dm, ap)
        T("sh<8,2> next", (k_w1_update_fwd_sh<8, 2>), dim3(g), dim3(512), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        // T("priv<2> next", (k_w1_update_fwd<2, true, 0>), dim3(g), dim3(512), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap)
        CK(hipFree(P)); CK(hipFree(dsn)); CK(hipFree(dwk));
    }
    return 0;
}
