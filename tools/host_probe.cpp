// Host-side rates of the drop-in's PCIe pipelines on the GPU box (no GPU work): the count scan of dimn_counts_create (plain C++ loop
// vs the AVX2 form), a threaded memcpy of ~128 MB blocks (pinned -> frame copies of dimn_impute_finish), first-touch cost of a fresh
// destination.  Build + run:  /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -march=x86-64-v3 -pthread tools/host_probe.cpp -o /tmp/host_probe && /tmp/host_probe
#include <immintrin.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
static inline uint64_t mix(uint64_t x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; }
static uint64_t scan_scalar(const double* src, float* out, int64_t n, uint64_t base) {
    uint64_t h = 0; double m = -INFINITY; bool fine = true;
    for (int64_t j = 0; j < n; ++j) {
        const double x = src[j]; uint64_t bits; memcpy(&bits, &x, 8);
        h += mix(bits + 0x9e3779b97f4a7c15ull * (base + (uint64_t)j + 1));
        m = x > m ? x : m;
        const bool in = x >= 0.0 && x <= 4194304.0; const double xr = in ? x : 0.5;
        fine &= in & ((double)(int32_t)xr == xr) & ((bits >> 63) == 0);
        if (out) out[j] = (float)x;
    }
    return h + (uint64_t)m + fine;
}
static inline __m256i mul64(__m256i v, __m256i clo, __m256i chi) {
    const __m256i lo = _mm256_mul_epu32(v, clo);
    const __m256i t = _mm256_add_epi64(_mm256_mul_epu32(_mm256_srli_epi64(v, 32), clo), _mm256_mul_epu32(v, chi));
    return _mm256_add_epi64(lo, _mm256_slli_epi64(t, 32));
}
static uint64_t scan_avx2(const double* src, float* out, int64_t n, uint64_t base) {
    const uint64_t G = 0x9e3779b97f4a7c15ull, C1 = 0xbf58476d1ce4e5b9ull, C2 = 0x94d049bb133111ebull;
    const __m256i c1lo = _mm256_set1_epi64x((long long)(C1 & 0xffffffffull)), c1hi = _mm256_set1_epi64x((long long)(C1 >> 32));
    const __m256i c2lo = _mm256_set1_epi64x((long long)(C2 & 0xffffffffull)), c2hi = _mm256_set1_epi64x((long long)(C2 >> 32));
    __m256i kv = _mm256_set_epi64x((long long)(G * (base + 4)), (long long)(G * (base + 3)), (long long)(G * (base + 2)), (long long)(G * (base + 1)));
    const __m256i ks = _mm256_set1_epi64x((long long)(G * 4));
    __m256i hv = _mm256_setzero_si256(), orv = hv;
    __m256d mv = _mm256_set1_pd(-INFINITY), gv = _mm256_castsi256_pd(_mm256_set1_epi64x(-1));
    const __m256d zero = _mm256_setzero_pd(), top = _mm256_set1_pd(4194304.0), half = _mm256_set1_pd(0.5);
    for (int64_t j = 0; j + 4 <= n; j += 4) {
        const __m256d x = _mm256_loadu_pd(src + j);
        const __m256i bits = _mm256_castpd_si256(x);
        __m256i v = _mm256_add_epi64(bits, kv); kv = _mm256_add_epi64(kv, ks);
        v = _mm256_xor_si256(v, _mm256_srli_epi64(v, 30)); v = mul64(v, c1lo, c1hi);
        v = _mm256_xor_si256(v, _mm256_srli_epi64(v, 27)); v = mul64(v, c2lo, c2hi);
        v = _mm256_xor_si256(v, _mm256_srli_epi64(v, 31));
        hv = _mm256_add_epi64(hv, v); mv = _mm256_max_pd(x, mv);
        const __m256d in = _mm256_and_pd(_mm256_cmp_pd(x, zero, _CMP_GE_OQ), _mm256_cmp_pd(x, top, _CMP_LE_OQ));
        const __m256d xr = _mm256_blendv_pd(half, x, in);
        gv = _mm256_and_pd(gv, _mm256_and_pd(in, _mm256_cmp_pd(_mm256_cvtepi32_pd(_mm256_cvttpd_epi32(xr)), xr, _CMP_EQ_OQ)));
        orv = _mm256_or_si256(orv, bits);
        if (out) _mm_storeu_ps(out + j, _mm256_cvtpd_ps(x));
    }
    alignas(32) uint64_t hl[4]; _mm256_store_si256((__m256i*)hl, _mm256_add_epi64(hv, orv));
    return hl[0] + hl[1] + hl[2] + hl[3] + (uint64_t)_mm256_movemask_pd(gv) + (uint64_t)_mm256_cvtsd_f64(mv);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par(int nt, const std::function<void(int)>& f) { std::vector<std::thread> th; for (int t = 1; t < nt; ++t) th.emplace_back(f, t); f(0); for (auto& x : th) x.join(); }
int main() {
    const int64_t n = (int64_t)1 << 28;                       // 2 GiB of float64
    double* a = (double*)malloc(n * 8); float* o = (float*)malloc(n * 4); double* d = (double*)malloc(n * 8);
    par(32, [&](int t) { for (int64_t i = n * t / 32; i < n * (t + 1) / 32; ++i) { a[i] = (double)((i * 2654435761u) % 37); o[i] = 0; } });
    printf("host threads: %u\n", std::thread::hardware_concurrency());
    volatile uint64_t sink = 0;
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); sink += scan_scalar(a, o, 1 << 24, 0); double t1 = now(); sink += scan_avx2(a, o, 1 << 24, 0); double t2 = now();
        printf("one thread, 16M elements: scalar %.2f ns/elem, avx2 %.2f ns/elem\n", (t1 - t0) / (1 << 24) * 1e9, (t2 - t1) / (1 << 24) * 1e9);
    }
    for (int nt : {16, 32, 64, 128}) for (int which = 0; which < 3; ++which) {
        double t0 = now();
        par(nt, [&](int t) { const int64_t x0 = n * t / nt, x1 = n * (t + 1) / nt; uint64_t s = which == 0 ? scan_scalar(a + x0, o + x0, x1 - x0, x0) : which == 1 ? scan_avx2(a + x0, o + x0, x1 - x0, x0) : scan_avx2(a + x0, nullptr, x1 - x0, x0); sink += s; });
        double dt = now() - t0;
        printf("%3d threads: %-22s %6.1f GB/s read (%.3f s for 2 GiB -> 8 GB in %.3f s)\n", nt, which == 0 ? "scalar scan + f32 copy" : which == 1 ? "avx2 scan + f32 copy" : "avx2 scan (checksum)", n * 8 / dt / 1e9, dt, dt * 4);
    }
    // memcpy: first touch of a fresh destination, then again (128 MB pieces like the finish pipeline would be the same rates)
    for (int nt : {24, 48, 96}) for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) { free(d); d = (double*)malloc(n * 8); }
        double t0 = now();
        par(nt, [&](int t) { const int64_t x0 = n * t / nt, x1 = n * (t + 1) / nt; memcpy(d + x0, a + x0, (x1 - x0) * 8); });
        double dt = now() - t0;
        printf("%3d threads: memcpy 2 GiB %-26s %6.1f GB/s (8 GB in %.3f s)\n", nt, pass == 0 ? "into a fresh allocation" : "into touched pages", n * 8 / dt / 1e9, dt * 4);
    }
    return (int)(sink & 1);
}
