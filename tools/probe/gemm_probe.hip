// Probe of k_gen_rowgemm on synthetic dense operands: time per launch against the number of descriptors (sub-nets), k-ranges S and the inner
// dimension K -- args: nd K N S M.  K = 16 is one chunk per workgroup: the launch's fixed cost (round 6: 32 us with every guarded form and
// epilogue compiled in, 6 us for the aligned, epilogue-templated kernel; an empty kernel of the same footprint: 2.7 us).
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -I deepimpute_amd/csrc -I include -o /tmp/gemm_probe tools/probe/gemm_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "dimn.h"
#include "dimn_general.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int nd = argc > 1 ? atoi(argv[1]) : 40, K = argc > 2 ? atoi(argv[2]) : 2400, N = argc > 3 ? atoi(argv[3]) : 256, S = argc > 4 ? atoi(argv[4]) : 4;
    const int M = argc > 5 ? atoi(argv[5]) : 64, reps = 20;
    const unsigned dyn = argc > 6 ? (unsigned)atoi(argv[6]) : 0u;        // unused dynamic LDS bytes per workgroup: caps the workgroups a CU takes
    float *A, *B, *C, *part;
    CK(hipMalloc(&A, (size_t)nd * M * K * 4)); CK(hipMalloc(&B, (size_t)nd * K * N * 4)); CK(hipMalloc(&C, (size_t)nd * M * N * 4)); CK(hipMalloc(&part, (size_t)nd * S * M * N * 4));
    CK(hipMemset(A, 0, (size_t)nd * M * K * 4)); CK(hipMemset(B, 0, (size_t)nd * K * N * 4));
    std::vector<GDesc> ds(nd);
    for (int i = 0; i < nd; ++i) { GDesc d; memset(&d, 0, sizeof d); d.A = A + (size_t)i * M * K; d.lda = K; d.B = B + (size_t)i * K * N; d.ldb = N; d.C = C + (size_t)i * M * N; d.ldc = N; d.bias = C; d.N = N; d.K = K; ds[i] = d; }
    GDesc* dd; CK(hipMalloc(&dd, nd * sizeof(GDesc))); CK(hipMemcpy(dd, ds.data(), nd * sizeof(GDesc), hipMemcpyHostToDevice));
    GEpi ep; memset(&ep, 0, sizeof ep); ep.mode = 0;
    if (S > 1) { ep.ksplit = S; ep.part = part; ep.part_stride = (int64_t)M * N; }
    const int nblk = (N + 63) / 64, mblk = (M + 63) / 64;
    const unsigned grid = (unsigned)(nd * nblk * mblk * S);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) {
#define LAUNCH(EV) hipLaunchKernelGGL((k_gen_rowgemm<false, EV>), dim3(grid), dim3(256), dyn, 0, dd, nd, M, nblk, mblk, nd % 8 == 0 ? 1 : 0, ep)
            if (S > 1) LAUNCH(9); else LAUNCH(2);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (w) printf("dynLDS %u nd %d K %d N %d S %d M %d grid %u: %.2f us per launch  (%.1f MB of B -> %.2f TB/s; %.2f TFLOP/s)\n", dyn, nd, K, N, S, M, grid, ms * 1000 / reps, (double)nd * K * N * 4 / 1e6,
                      (double)nd * K * N * 4 / (ms * 1e-3 / reps) / 1e12, 2.0 * nd * M * K * N / (ms * 1e-3 / reps) / 1e12);
    }
    return 0;
}
