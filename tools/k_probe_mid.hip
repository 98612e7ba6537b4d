// k_probe_mid.hip -- standalone timing of the middle kernels (RED / MF / MB) on cfg3-shaped
// synthetic state (K=40, H=256, O=512, n=50000).  Diagnostics only.
#define DIMN_MID_TL 1
#define DIMN_MIDP_TL 1
#ifndef PROBE_KEEP
#define PROBE_KEEP true
#endif
#include "../deepimpute_amd/csrc/dimn_kernels.h"
#include "../deepimpute_amd/csrc/dimn_mid_pipe.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_fill(float* p, size_t n, float scale, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = scale * ((x >> 8) * (1.0f / 16777216.0f) - 0.5f);
    }
}
__global__ void k_abs(float* p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = fabsf(p[i]); }
template <typename F> static double timeit(F launch, int R = 30) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < R; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return 1e3 * ms / R;
}
static float* dalloc(size_t n, float scale, unsigned seed) {
    float* p; CK(hipMalloc(&p, n * 4));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, p, n, scale, seed);
    return p;
}
int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 40, H = 256, O = 512, S = 19; const int64_t n = 50000;
    printf("K = %d\n", K);
    Dims dm; dm.K = K; dm.H = H; dm.O = O; dm.Hp = 256; dm.Op = 512; dm.HT = 16; dm.OT = 32; dm.ldd = 258; dm.OS = 8; dm.LS = 8;
    std::vector<SubnetDev> sn(K);
    for (int k = 0; k < K; ++k) { sn[k].D = 2400; sn[k].Dp = 2400; sn[k].nchunk = 150; sn[k].kg = k; sn[k].slot0 = k * S; sn[k].nslice = S; sn[k].xoff = 0; sn[k].w1off = 0; }
    SubnetDev* dsn; CK(hipMalloc(&dsn, K * sizeof(SubnetDev))); CK(hipMemcpy(dsn, sn.data(), K * sizeof(SubnetDev), hipMemcpyHostToDevice));
    float* P = dalloc((size_t)K * S * 64 * 256, 0.1f, 1);
    float* b1 = dalloc((size_t)3 * K * 256, 0.01f, 2); float* b2 = dalloc((size_t)3 * K * 512, 0.01f, 3);
    float* Dd = dalloc((size_t)K * 64 * 256, 1.f, 4); float* dZ = dalloc((size_t)K * 64 * 512, 1e-3f, 5); float* dA = dalloc((size_t)K * 64 * 256, 1e-3f, 6);
    float* W2 = dalloc((size_t)K * 256 * 512, 0.1f, 7); float* M2 = dalloc((size_t)K * 256 * 512, 1e-6f, 8); float* V2 = dalloc((size_t)K * 256 * 512, 1e-6f, 9);
    hipLaunchKernelGGL(k_abs, dim3(2048), dim3(256), 0, 0, V2, (size_t)K * 256 * 512);       // second moments are non-negative
    hipLaunchKernelGGL(k_abs, dim3(64), dim3(256), 0, 0, b2 + 2 * (size_t)K * 512, (size_t)K * 512);
    float* Y = dalloc((size_t)K * n * 512, 4.f, 10);
    float* ls; CK(hipMalloc(&ls, K * 8 * 4)); double* la; CK(hipMalloc(&la, K * 8 * 8)); CK(hipMemset(la, 0, K * 8 * 8));
    std::vector<int32_t> rows(64); for (int i = 0; i < 64; ++i) rows[i] = (int32_t)((i * 7919LL + 13) % n);
    int32_t* drows; CK(hipMalloc(&drows, 256)); CK(hipMemcpy(drows, rows.data(), 256, hipMemcpyHostToDevice));
    AdamP ap{1e-4f, 0.1f, 0.001f, 1e-7f};
    const size_t kh = (size_t)K * 256, ko = (size_t)K * 512;
    CK(hipDeviceSynchronize());
#define T(name, ...) { double us = timeit([&] { hipLaunchKernelGGL(__VA_ARGS__); }); CK(hipGetLastError()); printf("%-34s %8.1f us\n", name, us); }
    T("k_reduce_act", k_reduce_act, dim3(16, K), dim3(256), 0, 0, dsn, P, b1, (const uint8_t*)nullptr, Dd, dm, 64, 0.2f, 1.25f, 1234ull, 0u, 0u, 0, 0, (float*)nullptr)
    T("k_mid_fwd<16>", k_mid_fwd<16>, dim3(8, K), dim3(512), 64 * 258 * 4, 0, W2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, dZ, ls, la, dm, ap, 1.f / (64 * 512), 0, 0)
    T("k_mid_fwd<0>", k_mid_fwd<0>, dim3(8, K), dim3(512), 64 * 258 * 4, 0, W2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, dZ, ls, la, dm, ap, 1.f / (64 * 512), 0, 0)
    T("k_mid_bwd<true,1>", (k_mid_bwd<true, 1, 8>), dim3(16, K), dim3(512), 0, 0, Dd, dZ, W2, M2, V2, b1, b1 + kh, b1 + 2 * kh, dA, dm, ap, 1.25f, 4, 0, (const float*)nullptr)
    T("k_mid_bwd<true,1,4>", (k_mid_bwd<true, 1, 4>), dim3(16, K), dim3(256), 0, 0, Dd, dZ, W2, M2, V2, b1, b1 + kh, b1 + 2 * kh, dA, dm, ap, 1.25f, 8, 0, (const float*)nullptr)
    T("k_mid_bwd<true,1,16>", (k_mid_bwd<true, 1, 16>), dim3(16, K), dim3(1024), 0, 0, Dd, dZ, W2, M2, V2, b1, b1 + kh, b1 + 2 * kh, dA, dm, ap, 1.25f, 2, 0, (const float*)nullptr)
    T("k_mid_bwd<true,2>", (k_mid_bwd<true, 2, 8>), dim3(8, K), dim3(512), 0, 0, Dd, dZ, W2, M2, V2, b1, b1 + kh, b1 + 2 * kh, dA, dm, ap, 1.25f, 4, 0, (const float*)nullptr)
    {   // fused second layer: work table as dimn.hip's build_mid
        const int Sm = std::max(4, std::min(8, 256 / K));     // K = 40: 6 slices of 5-6 tiles (KEEP eligible)
        std::vector<MidWork> mw; std::vector<int32_t> midk(2 * K);
        for (int k = 0; k < K; ++k) { midk[2 * k] = k * Sm; midk[2 * k + 1] = Sm;
            for (int i = 0; i < Sm; ++i) mw.push_back(MidWork{k, 32 * i / Sm, 32 * (i + 1) / Sm, k * Sm + i, i}); }
        MidWork* dmw; int32_t* dmk; CK(hipMalloc(&dmw, mw.size() * sizeof(MidWork))); CK(hipMalloc(&dmk, midk.size() * 4));
        CK(hipMemcpy(dmw, mw.data(), mw.size() * sizeof(MidWork), hipMemcpyHostToDevice)); CK(hipMemcpy(dmk, midk.data(), midk.size() * 4, hipMemcpyHostToDevice));
        float* P2 = dalloc(mw.size() * 64 * 256, 0.f, 11);
        const size_t ldsb = ((size_t)64 * DIMN_MID_LDD + 8 * 1024 + 8 * 1024 + 8 + 64) * 4;
        CK(hipFuncSetAttribute((const void*)k_mid_fused<PROBE_KEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        printf("fused: %zu workgroups (%d slices per sub-net)\n", mw.size(), Sm);
        T("k_mid_fused (warm)", k_mid_fused<PROBE_KEEP>, dim3((unsigned)mw.size()), dim3(512), ldsb, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0)
        T("k_reduce_dd", k_reduce_dd, dim3(4, K), dim3(1024), 0, 0, dmk, P2, Dd, b1, b1 + kh, b1 + 2 * kh, dA, dm, ap, 1.25f, 0, (const float*)nullptr)

        {   // k_mid_pipe (dimn_mid_pipe.h): same state, same outputs within summation order; then its timings
            CK(hipFuncSetAttribute((const void*)k_mid_pipe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            const size_t ldsp = (size_t)DIMN_MIDP_LDS_FLOATS * 4;
            const size_t nw2 = (size_t)K * 256 * 512, nb2 = (size_t)3 * K * 512, np2 = mw.size() * 64 * 256;
            std::vector<float> w0(nw2), m0(nw2), v0(nw2), bb0(nb2);
            CK(hipMemcpy(w0.data(), W2, nw2 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(m0.data(), M2, nw2 * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(v0.data(), V2, nw2 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(bb0.data(), b2, nb2 * 4, hipMemcpyDeviceToHost));
            std::vector<float> out[2][6];
            for (int which = 0; which < 2; ++which) {
                CK(hipMemcpy(W2, w0.data(), nw2 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(M2, m0.data(), nw2 * 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(V2, v0.data(), nw2 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b2, bb0.data(), nb2 * 4, hipMemcpyHostToDevice));
                CK(hipMemset(P2, 0, np2 * 4)); CK(hipMemset(ls, 0, K * 8 * 4));
                if (which == 0) hipLaunchKernelGGL(k_mid_fused<PROBE_KEEP>, dim3((unsigned)mw.size()), dim3(512), ldsb, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 61, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0);
                else hipLaunchKernelGGL(k_mid_pipe<false>, dim3((unsigned)mw.size()), dim3(512), ldsp, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 61, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0);
                CK(hipDeviceSynchronize());
                const float* src[6] = {W2, M2, V2, b2, P2, ls}; const size_t cnt[6] = {nw2, nw2, nw2, nb2, np2, (size_t)K * 8};
                for (int j = 0; j < 6; ++j) { out[which][j].resize(cnt[j]); CK(hipMemcpy(out[which][j].data(), src[j], cnt[j] * 4, hipMemcpyDeviceToHost)); }
            }
            const char* nm[6] = {"W2", "M2", "V2", "b2 (w, m, v)", "P2", "loss"};
            for (int j = 0; j < 6; ++j) {
                double md = 0, mx = 0; size_t bad = 0;
                for (size_t i = 0; i < out[0][j].size(); ++i) { const double a = out[0][j][i], b = out[1][j][i]; md = std::max(md, fabs(a - b)); mx = std::max(mx, fabs(a)); if (!(fabs(a - b) <= 1e-5 * (fabs(a) + 1e-3))) ++bad; }
                printf("pipe vs fused  %-14s max |delta| %.3e  (max |value| %.3e)  beyond 1e-5 rel: %zu of %zu\n", nm[j], md, mx, bad, out[0][j].size());
            }
            T("k_mid_pipe (warm)", k_mid_pipe<false>, dim3((unsigned)mw.size()), dim3(512), ldsp, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0)
            float* big2 = dalloc((size_t)256 << 20, 1.f, 12);
            hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
            double cold_p = 0, cold_f2 = 0; const int R = 10;
            for (int it = 0; it < R; ++it) {
                float ms;
                hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, big2, (size_t)256 << 20, 1.f, 13u + it);
                CK(hipEventRecord(ea));
                hipLaunchKernelGGL(k_mid_pipe<false>, dim3((unsigned)mw.size()), dim3(512), ldsp, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0);
                CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb)); cold_p += ms;
                hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, big2, (size_t)256 << 20, 1.f, 13u + it);
                CK(hipEventRecord(ea));
                hipLaunchKernelGGL(k_mid_fused<PROBE_KEEP>, dim3((unsigned)mw.size()), dim3(512), ldsb, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0);
                CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb)); cold_f2 += ms;
            }
            printf("cold (after 1 GB of other traffic): k_mid_pipe %.1f us   k_mid_fused %.1f us\n", 1e3 * cold_p / R, 1e3 * cold_f2 / R);
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, big2, (size_t)256 << 20, 1.f, 99u);
            hipLaunchKernelGGL(k_mid_pipe<false>, dim3((unsigned)mw.size()), dim3(512), ldsp, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> tl(512 * 8 * 12); CK(hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_midp_tl), tl.size() * 8));
            double ph[4] = {0, 0, 0, 0}; unsigned long long tmin = ~0ull, tmax = 0; const size_t nw = mw.size() * 8;
            for (size_t i = 0; i < nw; ++i) { const unsigned long long* o = &tl[i * 12]; for (int j = 0; j < 4; ++j) ph[j] += (double)(o[j + 1] - o[j]); if (o[0] < tmin) tmin = o[0]; if (o[4] > tmax) tmax = o[4]; }
            { double bl[6] = {0, 0, 0, 0, 0, 0}; size_t n6 = 0;       // blocks of the six-tile slices: stamps 5 .. 10 follow stamp 2
              for (size_t i = 0; i < nw; ++i) { const unsigned long long* o = &tl[i * 12]; const MidWork& m = mw[((i / 8) & 7) * (mw.size() / 8) + (i / 8) / 8]; if (m.ot1 - m.ot0 != 6) continue; ++n6;
                  bl[0] += (double)(o[5] - o[2]); for (int j = 1; j < 6; ++j) bl[j] += (double)(o[5 + j] - o[4 + j]); }
              if (n6) printf("pipe blocks of the 6-tile slices, mean clk per wave: %.0f %.0f %.0f %.0f %.0f %.0f\n", bl[0] / n6, bl[1] / n6, bl[2] / n6, bl[3] / n6, bl[4] / n6, bl[5] / n6); }
            printf("pipe timeline, mean clk (100 MHz) per wave: Dd staged %.0f | forward(0)+softplus(0) %.0f | tile blocks %.0f | P2, b2, loss %.0f | first start -> last end %.0f\n",
                   ph[0] / nw, ph[1] / nw, ph[2] / nw, ph[3] / nw, (double)(tmax - tmin));
        }
        // cold: 1 GB of unrelated traffic between launches, as the W1 update does in a real step
        float* big = dalloc((size_t)256 << 20, 1.f, 12);
        hipEvent_t ea, eb; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
        double cold_f = 0, cold_mf = 0, cold_mb = 0; const int R = 10;
        for (int it = 0; it < R; ++it) {
            float ms;
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, big, (size_t)256 << 20, 1.f, 13u + it);
            CK(hipEventRecord(ea));
            hipLaunchKernelGGL(k_mid_fused<PROBE_KEEP>, dim3((unsigned)mw.size()), dim3(512), ldsb, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0);
            CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb)); cold_f += ms;
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, big, (size_t)256 << 20, 1.f, 33u + it);
            CK(hipEventRecord(ea));
            hipLaunchKernelGGL(k_mid_fwd<16>, dim3(8, K), dim3(512), 64 * 258 * 4, 0, W2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, dZ, ls, la, dm, ap, 1.f / (64 * 512), 0, 0);
            CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb)); cold_mf += ms;
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, big, (size_t)256 << 20, 1.f, 53u + it);
            CK(hipEventRecord(ea));
            hipLaunchKernelGGL((k_mid_bwd<true, 1, 4>), dim3(16, K), dim3(256), 0, 0, Dd, dZ, W2, M2, V2, b1, b1 + kh, b1 + 2 * kh, dA, dm, ap, 1.25f, 8, 0, (const float*)nullptr);
            CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb)); cold_mb += ms;
        }
        printf("cold (after 1 GB of other traffic): k_mid_fused %.1f us   k_mid_fwd<16> %.1f us   k_mid_bwd<1,4> %.1f us\n", 1e3 * cold_f / R, 1e3 * cold_mf / R, 1e3 * cold_mb / R);
        // phase timeline of one cold launch
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, big, (size_t)256 << 20, 1.f, 99u);
        hipLaunchKernelGGL(k_mid_fused<PROBE_KEEP>, dim3((unsigned)mw.size()), dim3(512), ldsb, 0, dmw, W2, M2, V2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, P2, ls, la, dm, ap, 1.f / (64 * 512), 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> tl(512 * 8 * 8); CK(hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_mid_tl), tl.size() * 8));
        double ph[6] = {0, 0, 0, 0, 0, 0}; unsigned long long tmin = ~0ull, tmax = 0; const size_t nw = mw.size() * 8;
        for (size_t i = 0; i < nw; ++i) { const unsigned long long* o = &tl[i * 8]; for (int j = 0; j < 6; ++j) ph[j] += (double)(o[j + 1] - o[j]); if (o[0] < tmin) tmin = o[0]; if (o[6] > tmax) tmax = o[6]; }
        printf("timeline, mean clk per wave: prologue+Dd stage %.0f | phase1 %.0f | barrier %.0f | Adam(b2)+ddf %.0f | phase2 %.0f | P2 store %.0f | first start -> last end %.0f clk\n",
               ph[0] / nw, ph[1] / nw, ph[2] / nw, ph[3] / nw, ph[4] / nw, ph[5] / nw, (double)(tmax - tmin));
    }
    // chained like a real step
    T("RED+MF+MB chain", k_reduce_act, dim3(16, K), dim3(256), 0, 0, dsn, P, b1, (const uint8_t*)nullptr, Dd, dm, 64, 0.2f, 1.25f, 1234ull, 0u, 0u, 0, 0, (float*)nullptr);
    { double us = timeit([&] {
        hipLaunchKernelGGL(k_reduce_act, dim3(16, K), dim3(256), 0, 0, dsn, P, b1, (const uint8_t*)nullptr, Dd, dm, 64, 0.2f, 1.25f, 1234ull, 0u, 0u, 0, 0, (float*)nullptr);
        hipLaunchKernelGGL(k_mid_fwd<16>, dim3(8, K), dim3(512), 64 * 258 * 4, 0, W2, b2, b2 + ko, b2 + 2 * ko, Y, n, drows, 64, Dd, dZ, ls, la, dm, ap, 1.f / (64 * 512), 0, 0);
        hipLaunchKernelGGL((k_mid_bwd<true, 1, 8>), dim3(16, K), dim3(512), 0, 0, Dd, dZ, W2, M2, V2, b1, b1 + kh, b1 + 2 * kh, dA, dm, ap, 1.25f, 4, 0, (const float*)nullptr); });
      printf("%-34s %8.1f us\n", "RED+MF+MB back-to-back", us); }
    return 0;
}
