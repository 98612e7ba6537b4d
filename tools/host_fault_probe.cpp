// What the first touch of predict()'s 8 GB result frame costs on the GPU box's host: a threaded memcpy into a fresh anonymous mapping,
// plain / with MADV_HUGEPAGE (what numpy asks for) / pre-faulted with MADV_POPULATE_WRITE on T threads.
// /opt/rocm/lib/llvm/bin/clang++ -O3 -std=c++17 -pthread tools/host_fault_probe.cpp -o /tmp/hfp && /tmp/hfp
#include <sys/mman.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par(int nt, const std::function<void(int)>& f) { std::vector<std::thread> th; for (int t = 1; t < nt; ++t) th.emplace_back(f, t); f(0); for (auto& x : th) x.join(); }
int main() {
    const size_t n = (size_t)2 << 30;
    char* src = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    par(32, [&](int t) { memset(src + n * t / 32, 1, n / 32); });
    for (int mode = 0; mode < 6; ++mode) {
        char* d = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (mode == 1 || mode >= 3) madvise(d, n, MADV_HUGEPAGE);
        double tp = 0;
        if (mode >= 2) {
            const int pt = mode == 2 || mode == 3 ? 1 : (mode == 4 ? 8 : 32);
            double t0 = now(); int bad = 0;
            par(pt, [&](int t) { const size_t a = (n / pt * t) & ~((size_t)(2 << 20) - 1), b = t + 1 == pt ? n : (n / pt * (t + 1)) & ~((size_t)(2 << 20) - 1); if (madvise(d + a, b - a, MADV_POPULATE_WRITE) != 0) bad = 1; });
            tp = now() - t0;
            if (bad) printf("(MADV_POPULATE_WRITE not supported here)\n");
        }
        double t0 = now();
        par(24, [&](int t) { memcpy(d + n * t / 24, src + n * t / 24, n / 24); });
        double dt = now() - t0;
        const char* names[] = {"plain 4 KB pages", "MADV_HUGEPAGE", "plain + POPULATE_WRITE x1", "HUGEPAGE + POPULATE_WRITE x1", "HUGEPAGE + POPULATE_WRITE x8", "HUGEPAGE + POPULATE_WRITE x32"};
        printf("%-32s populate %.3f s (8 GB: %.3f s)   24-thread memcpy %.3f s = %.1f GB/s (8 GB: %.3f s)\n", names[mode], tp, tp * 4, dt, n / dt / 1e9, dt * 4);
        munmap(d, n);
    }
    return 0;
}
