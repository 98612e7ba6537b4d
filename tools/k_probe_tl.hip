// k_probe_tl.hip -- ablation + phase-timeline probe for the shipped B1F1 kernel
// (k_w1_update_fwd_ring<16,1,3> of dimn_kernels.h), cfg3-shaped synthetic state.  Diagnostics only:
// the kernel below is a COPY of the product kernel's loop with switchable pieces (VAR bits) and
// optional s_memtime stamps; it never ships.
// Findings (MI355X, round 1): no compute ablation changes the time (the loop is memory-bound: ~35 % of a
// wave's chunk time is VMEM issue back-pressure, ~33 % waiting at the barrier for the slowest wave); the
// fused forward costs ~9 us for the distinct X_{t+1} rows (24.6 MB of 64-byte row pieces), ~5 us for the
// P store (16.8 MB at the kernel's tail, layout-insensitive) and ~2 us of compute; staging the X tiles as
// 128-byte chunk PAIRS (whole L2 lines) was built and A/B-ed in one process: no change, reverted; the
// prologue (work -> sub-net -> rows -> first loads: three dependent round trips) is ~9 us per launch.
// With the non-temporal state stores (DIMN_NT=2, also used by this copy): next 109 us vs no-next 102 us on one
// box -- the fused forward now costs ~7 us (X_{t+1} rows 4, P store 2, MFMAs 2); a barrier only after every
// second chunk (bit 8, racy): no gain.
//   VAR bit0: no gradient MFMAs   bit1: no forward MFMAs   bit2: no forward LDS reads
//       bit3: no Adam arithmetic  bit4: no per-chunk barrier (racy, timing only)  bit5: stamps
//       bit6: no P store          bit7: P stored in the MFMA-fragment-native layout (float4 per lane)
//       bit8: barrier only after every second chunk (racy, timing only)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/k_probe_tl.hip -o /tmp/k_probe_tl && /tmp/k_probe_tl
#include "../deepimpute_amd/csrc/dimn_kernels.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VAR>
__global__ __launch_bounds__(1024) void k_ring_var(const Work* __restrict__ work, const SubnetDev* __restrict__ sn,
                                                   const float* __restrict__ X, float* __restrict__ W1,
                                                   float* __restrict__ M1, float* __restrict__ V1,
                                                   const int32_t* __restrict__ rows_t, int b_act,
                                                   const int32_t* __restrict__ rows_n, int b_next,
                                                   const float* __restrict__ dA, float* __restrict__ P, Dims dm, AdamP ap,
                                                   unsigned long long* __restrict__ tl) {
    constexpr int WAVES = 16, NT2 = 1;
    constexpr int XT = DIMN_TB * 16, XN = DIMN_TB * 20;
    constexpr int DUMMY = 2 * (XT + XN);
    __shared__ __attribute__((aligned(16))) float sm[2 * (XT + XN) + 4 * WAVES * 64];
    const Work wk = work[blockIdx.x];
    const SubnetDev s = sn[wk.k];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int nt0 = wave;
    const int Hp = dm.Hp;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, tstart = 0, rstart = 0;
    if (VAR & 32) { tstart = __builtin_amdgcn_s_memtime(); rstart = __builtin_amdgcn_s_memrealtime(); }

    float bfr[16];
    const float* dak = dA + (int64_t)wk.k * DIMN_TB * Hp;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) bfr[kb] = dak[(4 * kb + lj) * Hp + 16 * nt0 + li];

    const bool stager = tid < 512;
    const bool stage_next = (tid & 511) >= 256;
    const bool have_next = b_next > 0;
    const int sb = (tid & 255) >> 2, sq = tid & 3;
    const bool svalid = stage_next ? (sb < b_next) : (sb < b_act);
    const int32_t* srows = (stage_next && have_next) ? rows_n : rows_t;
    const float* xsrc = X + s.xoff + (int64_t)srows[svalid ? sb : 0] * s.Dp + 4 * sq;
    const int sdst0 = stage_next ? (2 * XT + sb * 20 + 4 * sq) : (sb * 16 + 4 * sq);
    const int sbuf = stager ? (stage_next ? XN : XT) : 0;
    const int sdst = stager ? sdst0 : DUMMY + 4 * tid;

    const int64_t cstride = (int64_t)Hp * 16;
    const int64_t wb = s.w1off + (int64_t)(16 * nt0 + li) * 16 + 4 * lj;
    f32x4 pacc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) pacc[mt] = zero4;

    const int clast = wk.c1 - 1;
    auto fetch = [&](W1Set<1>& st, int c) {
        const int cc = c < clast ? c : clast;
        st.x = *(const f32x4*)(xsrc + 16 * cc);
        const int64_t idx = wb + cc * cstride;
        st.w[0] = *(const f32x4*)(W1 + idx); st.m[0] = *(const f32x4*)(M1 + idx); st.v[0] = *(const f32x4*)(V1 + idx);
    };
#define STAMP(i) if (VAR & 32) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); ph[i] += t1 - t0; t0 = t1; }
    auto step = [&](W1Set<1>& cur, W1Set<1>& nx1, W1Set<1>& nx2, int c) {
        const int par = (c - wk.c0) & 1;
        if (VAR & 32) t0 = __builtin_amdgcn_s_memtime();
        fetch(nx2, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        const float* xt = sm + par * XT;
        f32x4 g = zero4;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float a = xt[64 * kb + lane];
            if (VAR & 1) { g[kb & 3] += a * bfr[kb]; } else g = MFMA16(a, bfr[kb], g);
        }
        if (VAR & 32) { asm volatile("" : "+v"(g)); }
        STAMP(0)                                              // gradient: LDS reads + 16 MFMAs
        if (VAR & 8) { cur.w[0] += g * 1e-30f; cur.m[0] += g * 1e-30f; cur.v[0] += g * 1e-30f; }
        else adam4(cur.w[0], cur.m[0], cur.v[0], g, ap);
        if (VAR & 32) { asm volatile("" : "+v"(cur.w[0]), "+v"(cur.m[0]), "+v"(cur.v[0])); }
        STAMP(1)                                              // wait for the chunk's state + Adam
        *(f32x4*)(sm + sdst + (par ^ 1) * sbuf) = svalid ? nx1.x : zero4;
        const int64_t idx = wb + c * cstride;
        DIMN_ST_STATE(W1 + idx, cur.w[0]); DIMN_ST_STATE(M1 + idx, cur.m[0]); DIMN_ST_STATE(V1 + idx, cur.v[0]);
        STAMP(2)                                              // wait for next X tile, LDS staging, stores issued
        if (have_next) {
            const float* xn = sm + 2 * XT + par * XN;
            f32x4 af[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (VAR & 4) af[mt] = (f32x4){1.f + mt, 2.f, 3.f, 4.f};
                else af[mt] = *(const f32x4*)(xn + (16 * mt + li) * 20 + 4 * lj);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (VAR & 2) pacc[mt][r] += af[mt][r] * cur.w[0][r]; else pacc[mt] = MFMA16(af[mt][r], cur.w[0][r], pacc[mt]);
                }
            if (VAR & 32) { asm volatile("" : "+v"(pacc[0]), "+v"(pacc[1]), "+v"(pacc[2]), "+v"(pacc[3])); }
        }
        STAMP(3)                                              // forward: LDS reads + 16 MFMAs
        if (!(VAR & 16) && (!(VAR & 256) || ((c - wk.c0) & 1))) __syncthreads();
        STAMP(4)                                              // barrier
    };

    W1Set<1> A, B, C;
    fetch(A, wk.c0);
    fetch(B, wk.c0 + 1);
    *(f32x4*)(sm + sdst) = svalid ? A.x : zero4;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) asm volatile("" : "+v"(bfr[kb]));
    asm volatile("" : "+v"(A.w[0]), "+v"(A.m[0]), "+v"(A.v[0]));
    asm volatile("" : "+v"(B.w[0]), "+v"(B.m[0]), "+v"(B.v[0]));
    asm volatile("" : "+v"(B.x));
    __syncthreads();
    unsigned long long tloop = 0;
    if (VAR & 32) tloop = __builtin_amdgcn_s_memtime();

    int c = wk.c0;
    for (; c + 3 <= wk.c1; c += 3) {
        step(A, B, C, c);
        step(B, C, A, c + 1);
        step(C, A, B, c + 2);
    }
    if (c < wk.c1) {
        step(A, B, C, c);
        if (c + 1 < wk.c1) step(B, C, A, c + 1);
    }
    if (have_next) {
        float* p = P + (int64_t)wk.slot * DIMN_TB * Hp;
        if (VAR & 64) { asm volatile("" : "+v"(pacc[0]), "+v"(pacc[1]), "+v"(pacc[2]), "+v"(pacc[3])); }
        else if (VAR & 128) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) *(f32x4*)(p + ((nt0 * 4 + mt) * 64 + lane) * 4) = pacc[mt];
        } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[(16 * mt + 4 * lj + r) * Hp + 16 * nt0 + li] = pacc[mt][r];
        }
    }
    if ((VAR & 32) && lane == 0) {
        const unsigned long long tend = __builtin_amdgcn_s_memtime(), rend = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = tl + ((int64_t)blockIdx.x * WAVES + wave) * 12;
        for (int i = 0; i < 5; ++i) o[i] = ph[i];
        o[5] = tloop - tstart; o[6] = tend - tstart; o[7] = rend - rstart; o[8] = wk.c1 - wk.c0; o[9] = tstart; o[10] = tend;
    }
}

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
    const int K = 40, D = 2400, H = 256, O = 512; const int64_t n = 50000;
    Dims dm; dm.K = K; dm.H = H; dm.O = O; dm.Hp = 256; dm.Op = 512; dm.HT = 16; dm.OT = 32; dm.ldd = 258; dm.OS = 8; dm.LS = 8;
    std::vector<SubnetDev> sn(K);
    int64_t w1 = 0, x = 0;
    for (int k = 0; k < K; ++k) { sn[k].D = D; sn[k].Dp = D; sn[k].nchunk = D / 16; sn[k].kg = k; sn[k].xoff = x; sn[k].w1off = w1; w1 += (int64_t)D * 256; x += n * D; }
    float *X, *W, *M, *V, *dA, *P; CK(hipMalloc(&X, x * 4)); CK(hipMalloc(&W, w1 * 4)); CK(hipMalloc(&M, w1 * 4)); CK(hipMalloc(&V, w1 * 4));
    CK(hipMalloc(&dA, (size_t)K * 64 * 256 * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, X, (size_t)x, 4.f, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, W, (size_t)w1, 0.05f, 2u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, M, (size_t)w1, 1e-6f, 3u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, V, (size_t)w1, 1e-12f, 4u);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, dA, (size_t)K * 64 * 256, 1e-6f, 5u);
    std::vector<int32_t> rows(128); for (int i = 0; i < 128; ++i) rows[i] = (int32_t)((i * 7919LL + 13) % n);
    int32_t* drows; CK(hipMalloc(&drows, 512)); CK(hipMemcpy(drows, rows.data(), 512, hipMemcpyHostToDevice));
    AdamP ap{1e-4f, 0.1f, 0.001f, 1e-7f};
    const double bytes = 24.0 * w1 + 2 * 4.0 * 64 * D * K;
    const int wgs = 256;
    std::vector<Work> work; int slot = 0;
    for (int k = 0; k < K; ++k) { const int nc = D / 16, ns = wgs / K + (k < wgs % K ? 1 : 0); sn[k].slot0 = slot; sn[k].nslice = ns;
        for (int i = 0; i < ns; ++i) work.push_back(Work{k, nc * i / ns, nc * (i + 1) / ns, slot++}); }
    SubnetDev* dsn; Work* dwk; CK(hipMalloc(&dsn, K * sizeof(SubnetDev))); CK(hipMalloc(&dwk, work.size() * sizeof(Work)));
    CK(hipMemcpy(dsn, sn.data(), K * sizeof(SubnetDev), hipMemcpyHostToDevice)); CK(hipMemcpy(dwk, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
    CK(hipMalloc(&P, (size_t)slot * 64 * 256 * 4));
    unsigned long long* tl; CK(hipMalloc(&tl, (size_t)wgs * 16 * 12 * 8)); CK(hipMemset(tl, 0, (size_t)wgs * 16 * 12 * 8));
    const unsigned g = (unsigned)work.size();
#define T(name, V_, NX, BN) { double us = timeit([&] { hipLaunchKernelGGL((k_ring_var<V_>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, NX, BN, dA, P, dm, ap, tl); }); CK(hipGetLastError()); printf("%-52s %8.1f us %6.0f GB/s\n", name, us, bytes / us / 1e3); }
    const int32_t* none = nullptr;
    T("next  baseline", 0, drows + 64, 64)
    T("nonext baseline", 0, none, 0)
    T("next  -gradMFMA", 1, drows + 64, 64)
    T("next  -fwdMFMA", 2, drows + 64, 64)
    T("next  -fwdLDS", 4, drows + 64, 64)
    T("next  -fwdMFMA -fwdLDS", 6, drows + 64, 64)
    T("next  -Adam", 8, drows + 64, 64)
    T("next  -barrier", 16, drows + 64, 64)
    T("next  barrier every 2nd chunk (racy, timing only)", 256, drows + 64, 64)
    T("next  baseline again", 0, drows + 64, 64)
    T("next  barrier every 2nd chunk again", 256, drows + 64, 64)
    T("next  -gradMFMA -fwdMFMA -fwdLDS -Adam (stream+barrier)", 15, drows + 64, 64)
    T("next  everything off incl. barrier", 31, drows + 64, 64)
    T("next  no P store", 64, drows + 64, 64)
    T("next  native-layout P store", 128, drows + 64, 64)
    T("next  X_n rows == X_t rows", 0, drows, 64)
    T("next  X_n rows == X_t rows, no P store", 64, drows, 64)
    T("next  X_n == X_t, no P store, all compute off", 64 + 15, drows, 64)
    T("nonext -gradMFMA", 1, none, 0)
    T("nonext -Adam", 8, none, 0)
    T("nonext -gradMFMA -Adam", 9, none, 0)
    T("nonext -barrier", 16, none, 0)
    T("nonext -gradMFMA -Adam -barrier", 25, none, 0)
    // phase timeline (stamped variants perturb: every stamp drains lgkmcnt)
    for (int pass = 0; pass < 2; ++pass) {
        CK(hipMemset(tl, 0, (size_t)wgs * 16 * 12 * 8));
        if (pass == 0) hipLaunchKernelGGL((k_ring_var<32>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, drows + 64, 64, dA, P, dm, ap, tl);
        else hipLaunchKernelGGL((k_ring_var<32>), dim3(g), dim3(1024), 0, 0, dwk, dsn, X, W, M, V, drows, 64, none, 0, dA, P, dm, ap, tl);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)wgs * 16 * 12); CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
        double ph[5] = {0, 0, 0, 0, 0}, pro = 0, tot = 0, real = 0, chunks = 0; unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t i = 0; i < (size_t)wgs * 16; ++i) { const unsigned long long* o = &h[i * 12];
            for (int j = 0; j < 5; ++j) ph[j] += o[j]; pro += o[5]; tot += o[6]; real += o[7]; chunks += o[8];
            if (o[9] < tmin) tmin = o[9]; if (o[10] > tmax) tmax = o[10]; }
        const double nw = wgs * 16.0;
        printf("%s: per wave avg: total %.0f clk (%.2f us realtime => %.0f MHz shader clock), prologue %.0f clk, chunks %.1f\n",
               pass ? "nonext" : "next", tot / nw, real / nw / 100.0, (tot / nw) / (real / nw / 100.0), pro / nw, chunks / nw);
        printf("   per chunk clk: grad %.0f | state-wait+Adam %.0f | xwait+stage+stores %.0f | fwd %.0f | barrier %.0f | sum %.0f ; grid span %.0f clk\n",
               ph[0] / chunks, ph[1] / chunks, ph[2] / chunks, ph[3] / chunks, ph[4] / chunks, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / chunks, (double)(tmax - tmin));
    }
    return 0;
}
