// dimn.hip -- host side of libdimn: the C ABI of include/dimn.h on top of the gfx950 kernels
// in dimn_kernels.h.  One handle = one GPU = one HIP stream; RCCL is bound lazily (dlopen).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>      // host pass only: the AVX2 form of counts_scan's inner loop (the Makefile builds the host side with -march=x86-64-v3)
#endif

#include "../../include/dimn.h"
#include "dimn_kernels.h"
#include "dimn_mid_pipe.h"
#include "dimn_corr.h"
#include "dimn_resident.h"
#include "dimn_general.h"
#include "dimn_csv.h"
#include "dimn_hoststats.h"
#include "dimn_counts_dev.h"

#define DIMN_ABI_VERSION 9

// DIMN_TRACE=1: stage times of the host-heavy entry points on stderr (diagnostic)
struct Trace {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = getenv("DIMN_TRACE") && atoi(getenv("DIMN_TRACE")) != 0;
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[dimn] %-32s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
static thread_local char g_err[1024];
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(DIMN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHK(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_) return rc_;   \
    } while (0)

extern "C" const char* dimn_last_error(void) { return g_err; }
extern "C" int dimn_abi_version(void) { return DIMN_ABI_VERSION; }
extern "C" int dimn_device_count(int32_t* n) {
    if (!n) return fail(DIMN_ERR_ARG, "dimn_device_count: null argument");
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess) nd = 0;
    *n = nd;
    return DIMN_OK;
}

// ---- RCCL, bound at first use so that the library loads on machines without it ----------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(ncclComm_t, int*) = nullptr;
    int (*CommUserRank)(ncclComm_t, int*) = nullptr;
};
static Rccl g_rccl;
static int rccl_bind() {
    if (g_rccl.lib) return DIMN_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return fail(DIMN_ERR_COMM, "cannot dlopen librccl: %s", dlerror());
#define BIND(field, sym)                                                         \
    *(void**)(&g_rccl.field) = dlsym(lib, sym);                                  \
    if (!g_rccl.field) return fail(DIMN_ERR_COMM, "librccl lacks symbol %s", sym)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(AllReduce, "ncclAllReduce");
    BIND(Send, "ncclSend");
    BIND(Recv, "ncclRecv");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(GetErrorString, "ncclGetErrorString");
    BIND(CommCount, "ncclCommCount");
    BIND(CommUserRank, "ncclCommUserRank");
#undef BIND
    g_rccl.lib = lib;
    return DIMN_OK;
}
#define NCCLCHK(expr)                                                                          \
    do {                                                                                       \
        int r_ = (expr);                                                                       \
        if (r_ != 0) return fail(DIMN_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)
enum { kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0 };

// the raw count matrix resident on the device (dimn_counts_*, below)
// Process-wide pinned bounce buffers.  Pinning 4 x 128 MB costs ~90 ms, as much as moving 4 GB over PCIe, and every fit() makes a
// new handle: the buffers are allocated once per process and LEASED to one pipeline at a time (dimn_counts_create, dimn_impute_finish);
// a second pipeline running at the same moment allocates its own and frees them again.
static std::mutex g_pin_mu;
static void* g_pin_buf[4] = {nullptr, nullptr, nullptr, nullptr};
static size_t g_pin_cap = 0;
static void* g_fin_res[2] = {nullptr, nullptr};   // the two device result blocks of dimn_impute_finish's pipeline: they belong to whoever holds the shared lease
static int g_fin_dev = -1;                         // the device they live on
static hipStream_t g_fin_st[2] = {nullptr, nullptr};     // ... and the pipeline's two streams / events (an HSA queue per stream: ~10-25 ms to create)
static hipEvent_t g_fin_ev[2] = {nullptr, nullptr};
static size_t g_fin_cap = 0;                       // (the first hipMalloc of that size in a process cost 17-86 ms inside predict(); dimn_warm_up makes them)
static std::atomic<int> g_pin_warming{0};         // dimn_warm_up holds the lock while it pins the shared set: a pipeline that arrives meanwhile waits for it
struct PinLease {
    std::unique_lock<std::mutex> lock;
    void* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    bool own = false;
    // the first `count` (<= 4) buffers, `bytes` each; false (with the HIP error text in *err) when pinning fails
    bool take(int count, size_t bytes, const char** err) {
        lock = std::unique_lock<std::mutex>(g_pin_mu, std::try_to_lock);
        if (!lock.owns_lock() && g_pin_warming) lock.lock();      // (the warm-up is pinning exactly these buffers: waiting is cheaper than pinning a second set)
        own = !lock.owns_lock();
        if (!own && g_pin_cap < bytes) {                 // the shared set grows: start again at the new size
            for (auto& pb : g_pin_buf) if (pb) { (void)hipHostFree(pb); pb = nullptr; }
            g_pin_cap = bytes;
        }
        void** dst = own ? buf : g_pin_buf;
        for (int i = 0; i < count; ++i) {
            if (!dst[i]) {
                const hipError_t e = hipHostMalloc(&dst[i], own ? bytes : g_pin_cap, hipHostMallocDefault);
                if (e != hipSuccess) { dst[i] = nullptr; *err = hipGetErrorString(e); return false; }
            }
            buf[i] = dst[i];
        }
        return true;
    }
    ~PinLease() {
        if (own) for (auto& pb : buf) if (pb) (void)hipHostFree(pb);
    }
};

struct dimn_counts_s {
    int device = 0; int64_t n = 0, g = 0; float* d = nullptr; double vmax = 0; uint64_t checksum = 0;
    double* d_corr = nullptr; int64_t corr_g = 0;     // |corr| of the last dimn_counts_corr pool, until dimn_counts_topk has used it
};

// ---- handle --------------------------------------------------------------------------------
struct dimn_handle_s {
    dimn_config cfg;
    Dims dm;
    int K, H, O, B, NT, NT2, OTW, HS;   // NT/NT2 hidden tiles per wave (4-/8-wave kernels); OTW out tiles per wave; HS = ceil(HT/2)
    int ncu = 256;
    int w1_waves = 0;                      // B1F1 as k_w1_update_fwd_ring<w1_waves, 1, 4> (0: the width's older kernel): one hidden tile per wave, four-set register ring
    int w1_wpc = 1;                        // ... workgroups per CU (2 for 8 waves)
    int w1_split = 1;                      // B1F1: the hidden tiles of a D-slice over this many workgroups (grid.y); 2: 18 .. 24 hidden tiles on the ring (build_work)
    std::vector<SubnetDev> sn;
    std::vector<Work> work;
    std::vector<MidWork> midwork;          // work table of the fused second-layer kernel (k_mid_pipe)
    int mid_fused = 0, mid_slices = 0;     // 1: RED -> MFB -> RED2 -> B1F1; 0: RED -> MF -> MB -> B1F1
    int train_bf16 = 0;                    // 1: precision bf16 and the fused second layer runs its three GEMMs on the bf16 matrix cores
    MidWork* d_midwork = nullptr; int32_t* d_midk = nullptr; float* d_P2 = nullptr;
    std::vector<std::vector<int32_t>> pred, targ;
    int nslots = 0;
    int64_t w1_total = 0, x_total = 0, y_total = 0;
    int64_t n = 0, g = 0, n_tr = 0, n_val = 0;
    bool gathered = false, gathered_targets = false, have_idx = false, streamed = false;
    int32_t stream_part = 0, stream_parts = 1;             // dimn_set_stream_order: where this handle's streamed hand-over starts (rank r of w ranks: block r NB / w)
    // device
    SubnetDev* d_sn = nullptr; Work* d_work = nullptr;
    float *d_norm = nullptr, *d_X = nullptr, *d_Y = nullptr;
    int32_t *d_pred = nullptr, *d_targ = nullptr; int64_t* d_pred_off = nullptr;
    float *d_W1 = nullptr, *d_M1 = nullptr, *d_V1 = nullptr;
    float *d_W2 = nullptr, *d_M2 = nullptr, *d_V2 = nullptr;
    float *d_b1 = nullptr, *d_b2 = nullptr;   // [3][K][Hp|Op]: w, m, v
    float *d_P = nullptr, *d_Dd = nullptr, *d_dZ = nullptr, *d_dA = nullptr;
    int act = 0; float* d_G = nullptr;     // hidden activation (DIMN_ACT_*), gate buffer f'(A)*keep*scale for act != relu
    float* d_loss_step = nullptr; double* d_loss_acc = nullptr;
    uint8_t* d_mask = nullptr;
    int32_t *d_rows_step = nullptr, *d_epoch_rows = nullptr, *d_val_rows = nullptr, *d_pred_rows = nullptr;
    int rows_step_cap = 0;
    int64_t pred_rows_cap = 0;
    std::vector<int32_t> train_rows, val_rows;
    std::vector<int32_t> next_perm;        // the permutation of epoch next_perm_epoch, made on a helper thread while the epoch before it ran (0.7 ms of Philox
    int64_t next_perm_epoch = -1;          // Fisher-Yates per epoch at 47 500 rows, with the GPU idle: 1.3 ms per epoch of host work in front of the first launch)
    float* d_out = nullptr; int64_t out_cap = 0; int64_t out_rows = 0;
    // dimn_predict_device over all cells runs as a few row chunks with an event behind each, so that dimn_impute_finish* can start on the
    // first rows while the forward still computes the last ones (round 5): pred_ev_rows[c] = first row NOT covered by chunks 0 .. c
    std::vector<hipEvent_t> pred_ev; std::vector<int64_t> pred_ev_rows; int32_t* d_pred_iota = nullptr; int64_t pred_iota_n = 0;
    float* d_loss_part = nullptr; int64_t loss_part_cap = 0;
    float *d_full = nullptr, *d_stage = nullptr; int64_t full_cap = 0;   // root's gathered predictions
    int64_t full_rows = 0, full_width = 0;                               // shape of the last gathered matrix (rows, K_global * O)
    double* d_red = nullptr; int red_cap = 0;                            // all-reduce scratch
    // register-resident epoch kernel (dimn_resident.h): chosen at create when the sub-nets of this handle fit the CUs
    int res_G = 0, res_S1 = 0, res_T1 = 0, res_Kg = 0;                    // 0: not eligible; Kg: sub-nets per epoch launch
    float *d_res_P = nullptr, *d_res_D = nullptr, *d_res_T = nullptr, *d_res_A = nullptr, *d_res_b1 = nullptr, *d_res_alpha = nullptr;
    unsigned* d_res_flags = nullptr; double* d_res_loss = nullptr; int64_t res_alpha_cap = 0;
    float* d_res_snap = nullptr;           // the optimiser state before the running epoch launch (restored if the launch aborts)
    float *d_res_Xe = nullptr, *d_res_Ye = nullptr; int32_t* d_res_iota = nullptr; int64_t res_iota_n = 0; bool res_erows_off = false;   // epoch-ordered copies of the training rows (large arenas)
    int res_checked = 0;                   // 1: co-residency of a launch's workgroups verified against the occupancy of the kernel
    int res_bf16 = 0;                      // 1: precision bf16 -> the resident kernel runs EVERY training GEMM on the bf16 matrix cores (template BF)
    double tm_res_ms = 0; int64_t tm_res_steps = 0;
    hipStream_t stream = nullptr;          // lane 0's stream; also used by every non-training call
    struct Lane {                          // sub-nets [k0,k1), work items [w0,w1)
        hipStream_t stream; int k0, k1, w0, w1;
    };
    std::vector<Lane> lanes;               // sub-net groups with a stream each (one lane: all sub-nets on the handle's stream)
    int64_t t = 0;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev;   // pairs around k_w1_update, plus step brackets
    size_t ev_used = 0;
    double tm_step_ms = 0, tm_w1_ms = 0, tm_w1_bytes = 0; int64_t tm_steps = 0, tm_w1 = 0;
    std::vector<double> ev_bytes;   // algorithmic bytes of the W1 launch bracketed by each event triple
    // comm
    ncclComm_t comm = nullptr; int n_ranks = 1, rank = 0;
    bf16_t* d_zero1k = nullptr;            // 1 KB of zeros (k_predict_bf16)
    bf16_t *d_W1b = nullptr, *d_W2t = nullptr; bool predict_bf16 = false;   // bf16 images of the weights for k_predict_bf16
    float* d_W2tf = nullptr;                                                // W2 in the operand form of k_predict's second layer (k_prep_w2t)
    int prec = 0;                          // DIMN_PREC_*: 1 = X arena in bfloat16, inference GEMMs on the bf16 matrix cores
    struct GenNet* gen = nullptr;          // != NULL: the general path (dimn_general.h) owns the network of this handle
    struct dimn_counts_s* counts = nullptr;   // borrowed: the resident count matrix this handle's matrix came from (dimn_set_matrix_counts)
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// run a statement with XT = the element type of this handle's X arena
#define WITH_XT(h_, ...)                                   \
    do {                                                   \
        if ((h_)->prec) { using XT = bf16_t; __VA_ARGS__; } \
        else { using XT = float; __VA_ARGS__; }             \
    } while (0)
#define XBYTES(h_) ((h_)->prec ? 2 : 4)



static int use_device(dimn_handle h) {
    HIPCHK(hipSetDevice(h->cfg.device_id));
    return DIMN_OK;
}

// ---- process-wide cache of the LARGE device allocations ------------------------------------
// hipMalloc of a multi-GB block usually returns in 0.3 ms and SOMETIMES in 0.5-3.4 s -- whenever the driver has to wipe the VRAM it
// hands out, which depends on what earlier processes left behind, not on this process's history (tools/malloc_probe.py,
// profiles/r04_malloc_probe.txt: 19.6 GB in 0.3 ms / 483 ms / 1 071 ms / 2 295 ms in one process).  That is the 0.02 / 0.19 / 0.62 s
// of the drop-in fit()'s hand-over (BENCH_r03 config.dropin.stages_s).  Blocks of >= 32 MB (the matrix, the gathered X / Y arenas,
// the resident counts, predictions, correlation temporaries) are therefore never given back while the process lives: a freed
// block waits here for the next request it fits (at most 25 % larger than asked for), and requests are rounded up to an eighth of
// their power of two (19.49 and 19.55 GB both take a 20 GiB block: the arena of the next fit(), whose predictor lists differ by
// a few columns, fits the previous one's).  dimn_release_cached_memory() empties the cache; an allocation that fails empties it
// and tries once more WITH THE EXACT SIZE (the rounding must never turn a request that fits into one that does not); the cache never holds
// more than DIMN_ARENA_CACHE_GB (default 48: the ~37 GB of arenas of the 50k x 20k job -- a sixth of the device; 0 = no cache, no
// rounding).  MultiNet.close() and deepimpute_amd.release_cached_memory() give everything back (other tenants of the GPU, RCCL, other libraries in the process).
// hipFree() synchronises the whole device before it returns and callers relied on that (a block may still be read or written by queued
// kernels of its previous owner on streams the caller does not know about), so put() does the same before a block becomes visible to the
// next owner: one hipDeviceSynchronize() on the block's device -- microseconds on an idle device, and only for blocks of >= 32 MB.
static const size_t kArenaMin = (size_t)32 << 20;
struct ArenaPool {
    struct Blk { void* p; size_t bytes; int dev; };
    std::mutex mu;
    std::vector<Blk> idle, live;
    size_t idle_bytes = 0;
    double cap_gb() { const char* e = getenv("DIMN_ARENA_CACHE_GB"); return e ? atof(e) : 48.0; }
    hipError_t get(void** out, size_t bytes) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const size_t asked = bytes;
        if (cap_gb() > 0.0) {                                    // size classes: multiples of 2^floor(log2(bytes)) / 8
            size_t p2 = (size_t)1 << 25;
            while ((p2 << 1) <= bytes) p2 <<= 1;
            const size_t gran = p2 >> 3;
            bytes = (bytes + gran - 1) / gran * gran;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].dev == dev && idle[i].bytes >= bytes && idle[i].bytes <= bytes + bytes / 4 && (best == idle.size() || idle[i].bytes < idle[best].bytes)) best = i;
            if (best < idle.size()) {
                *out = idle[best].p;
                live.push_back(idle[best]);
                idle_bytes -= idle[best].bytes;
                idle.erase(idle.begin() + (long)best);
                return hipSuccess;
            }
        }
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); trim(0); bytes = asked; e = hipMalloc(out, bytes); }   // (the exact size: what fit without the cache still fits)
        if (e == hipSuccess) { std::lock_guard<std::mutex> lk(mu); live.push_back({*out, bytes, dev}); }
        return e;
    }
    // true: p was one of ours (now idle, or freed when the cache is full / off)
    bool put(void* p) {
        Blk b{nullptr, 0, 0};
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < live.size(); ++i)
                if (live[i].p == p) { b = live[i]; live.erase(live.begin() + (long)i); break; }
            if (!b.p) return false;
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != b.dev) (void)hipSetDevice(b.dev);
        bool kept = false;
        if ((double)b.bytes <= cap_gb() * 1073741824.0) {
            (void)hipDeviceSynchronize();                           // what hipFree() would have done: nothing queued still touches the block
            std::lock_guard<std::mutex> lk(mu);
            if ((double)(idle_bytes + b.bytes) <= cap_gb() * 1073741824.0) { idle.push_back(b); idle_bytes += b.bytes; kept = true; }
        }
        if (!kept) (void)hipFree(p);
        if (cur != b.dev) (void)hipSetDevice(cur);
        return true;
    }
    void trim(size_t keep_bytes) {
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            while (!idle.empty() && idle_bytes > keep_bytes) { drop.push_back(idle.front()); idle_bytes -= idle.front().bytes; idle.erase(idle.begin()); }
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto& b : drop) { (void)hipSetDevice(b.dev); (void)hipFree(b.p); }
        if (!drop.empty()) (void)hipSetDevice(cur);
    }
};
static ArenaPool g_arena;
static hipError_t dev_malloc_bytes(void** p, size_t bytes) {
    if (bytes >= kArenaMin) return g_arena.get(p, bytes);
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); g_arena.trim(0); e = hipMalloc(p, bytes); }
    return e;
}
static void dev_free_any(void* p) {
    if (p && !g_arena.put(p)) (void)hipFree(p);
}
// Bring the device up ahead of the first real call: the HIP context of `device_id` and the four shared 128 MB pinned bounce buffers
// (~90 ms of pinning that the first dimn_counts_create / dimn_impute_finish of a process would otherwise pay inside fit() / predict()).
// Idempotent; meant to be called from a helper thread while the caller still reads its input.
extern "C" int dimn_warm_up(int32_t device_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(DIMN_ERR_HIP, "dimn_warm_up: no HIP device visible");
    if (device_id < 0 || device_id >= ndev) return fail(DIMN_ERR_ARG, "dimn_warm_up: device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    HIPCHK(hipFree(nullptr));                                      // (forces the context)
    std::unique_lock<std::mutex> lock(g_pin_mu);
    g_pin_warming = 1;
    const size_t bytes = (size_t)128u << 20;
    hipError_t e = hipSuccess;
    if (g_pin_cap <= bytes) {
        g_pin_cap = bytes;
        for (auto& pb : g_pin_buf)
            if (!pb && e == hipSuccess) { e = hipHostMalloc(&pb, bytes, hipHostMallocDefault); if (e != hipSuccess) pb = nullptr; }
    }
    // ... and the two 128 MB device blocks of dimn_impute_finish's row-block pipeline (DIMN_TRACE, "finish: allocations": 17-86 ms on
    // the first predict() of a process, the hipMalloc lottery of tools/malloc_probe.py)
    if (e == hipSuccess && g_fin_cap < bytes && (g_fin_dev < 0 || g_fin_dev == device_id)) {
        for (auto& p : g_fin_res) { if (p) (void)hipFree(p); p = nullptr; }
        g_fin_cap = 0;
        if (hipMalloc(&g_fin_res[0], bytes) == hipSuccess && hipMalloc(&g_fin_res[1], bytes) == hipSuccess) { g_fin_cap = bytes; g_fin_dev = device_id; }
        else { (void)hipGetLastError(); for (auto& p : g_fin_res) { if (p) (void)hipFree(p); p = nullptr; } }
    }
    if (e == hipSuccess && g_fin_dev == device_id)
        for (int b = 0; b < 2; ++b) {
            if (!g_fin_st[b] && hipStreamCreateWithFlags(&g_fin_st[b], hipStreamNonBlocking) != hipSuccess) { g_fin_st[b] = nullptr; (void)hipGetLastError(); }
            if (!g_fin_ev[b] && hipEventCreateWithFlags(&g_fin_ev[b], hipEventDisableTiming) != hipSuccess) { g_fin_ev[b] = nullptr; (void)hipGetLastError(); }
        }
    g_pin_warming = 0;
    lock.unlock();
    if (e != hipSuccess) return fail(DIMN_ERR_HIP, "dimn_warm_up: pinning the bounce buffers failed: %s", hipGetErrorString(e));
    return DIMN_OK;
}

extern "C" int dimn_release_cached_memory(void) {
    g_arena.trim(0);
    return DIMN_OK;
}
extern "C" int dimn_cached_memory_info(int64_t* out2) {
    if (!out2) return fail(DIMN_ERR_ARG, "dimn_cached_memory_info: null");
    std::lock_guard<std::mutex> lk(g_arena.mu);
    size_t live = 0;
    for (auto& b : g_arena.live) live += b.bytes;
    out2[0] = (int64_t)g_arena.idle_bytes;
    out2[1] = (int64_t)live;
    return DIMN_OK;
}

template <typename T>
static int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(dev_malloc_bytes((void**)p, count * sizeof(T)));
    return DIMN_OK;
}
#define DEV_FREE(p)            \
    do {                       \
        if (p) dev_free_any(p); \
        p = nullptr;           \
    } while (0)

#include "dimn_general_host.inc"

// The upper-triangular block pairs (i <= j < nb) of a Gram-matrix kernel (one workgroup per pair, pairs[blockIdx.x]) in an order that lets the workgroups
// RUNNING TOGETHER ON ONE XCD share operand blocks in that XCD's L2.  Workgroup b runs on XCD b % 8, so XCD c works through positions c, c + 8, c + 16, ...: it
// gets the c-th eighth of a walk over 6 x 6 SUPER-TILES of pairs (36 pairs, ~ the 32 workgroups of an XCD's CUs: 12 distinct operand blocks instead of 33).
// Round 5, PMC (20k genes x 50k cells, two planes): row-major order 203 GB fetched from memory per launch for 2 GB of operand planes, this order 138 GB.  The time
// does not move (35.0 -> 34.1 ms): with two planes the kernel issues four int8 products per pair -- 160 TOP in 34 ms = 0.94 of the int8 matrix peak.
static std::vector<int2> xcd_tiled_pairs(int nb) {
    constexpr int T = 6;
    std::vector<int2> walk;
    const int nsb = (nb + T - 1) / T;
    for (int si = 0; si < nsb; ++si)
        for (int sj = si; sj < nsb; ++sj)
            for (int i = si * T; i < std::min(nb, si * T + T); ++i)
                for (int j = std::max(i, sj * T); j < std::min(nb, sj * T + T); ++j) walk.push_back(make_int2(i, j));
    const int total = (int)walk.size(), q = total >> 3, r = total & 7;
    std::vector<int2> out((size_t)total);
    for (int b = 0; b < total; ++b) {
        const int xcd = b & 7;
        out[(size_t)b] = walk[(size_t)(xcd * q + std::min(xcd, r) + (b >> 3))];
    }
    return out;
}

static void build_work(dimn_handle h) {
    // Split every sub-net's chunk range into slices = workgroups of the W1 kernels.  The total is made
    // EXACTLY ncu (a partially filled last round of workgroups costs a whole round), shared
    // out in proportion to the chunk counts (largest remainder), subject to a minimum slice length: every
    // workgroup writes a 64-row split-K partial, so very fine slicing would drown the step in partials.
    std::vector<int> ns((size_t)h->K);
    {
        const int k0 = 0, k1 = h->K;
        int64_t total_chunks = 0;
        for (int k = k0; k < k1; ++k) total_chunks += h->sn[k].nchunk;
        // Round 5, every width of 8 .. 24 hidden tiles other than 16 (which has k_w1_update_fwd_ring<16, 1, 3>): the same ring with ONE tile per wave and four
        // register sets -- HT <= 15: HT waves (8 waves: two workgroups per CU); HT = 18 .. 24: two halves of HT / 2 waves.  The generic kernels of those widths
        // (8 waves x 1-3 tiles, one or two chunks in flight, 256 registers + spills at 3 tiles) ran at 0.47-0.49 of the HBM peak: profiles/r05_hidden_widths.txt
        h->w1_split = 1; h->w1_waves = 0; h->w1_wpc = 1;
        const int HT = h->dm.HT;
        // (Same-box A/B at hidden 300, 5 / 10 / 20 / 30 sub-nets of D ~ 2 400 and configs[1]: the ring wins at every size, 3-9 % per step:
        //  profiles/r05_hidden_widths.txt.  Round 6: at every chunk count -- the two-set shared-staging kernel k_w1_update_fwd_sh<10, 2>, which rounds 2-5 kept
        //  for fewer than two chunks per CU, is retired.)
        if (HT >= 8 && HT <= 24 && HT != 16) {
            h->w1_split = HT > 16 ? 2 : 1;
            h->w1_waves = HT / h->w1_split;
            h->w1_wpc = h->w1_waves == 8 ? 2 : 1;
        }
        // (16 tiles keep k_w1_update_fwd_ring<16, 1, 3>: with four sets -- 144 KB in flight per CU instead of 96 -- it takes 120 us where the three-set ring takes
        //  108, as two halves of 8 waves with two workgroups per CU 117: profiles/r05_hidden_widths.txt)
        const int64_t target = (int64_t)h->ncu * h->w1_wpc / h->w1_split;
        std::vector<std::pair<double, int>> frac;
        int64_t assigned = 0;
        // minimum chunks per slice: 8 when there is plenty of work per CU; down to 2 when a GPU owns only a
        // few sub-nets (8-GPU sharding): the step is then latency-bound and parallelism beats partial traffic
        const int min_chunks = (int)std::min<int64_t>(8, std::max<int64_t>(2, total_chunks / std::max<int64_t>(1, target)));
        for (int k = k0; k < k1; ++k) {
            const double share = (double)target * h->sn[k].nchunk / (double)total_chunks;
            const int cap = std::max(1, h->sn[k].nchunk / min_chunks);
            ns[(size_t)k] = std::min(cap, std::max(1, (int)share));
            assigned += ns[(size_t)k];
            frac.push_back({share - (int)share, k});
        }
        std::sort(frac.begin(), frac.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
        for (size_t i = 0; assigned < target && i < frac.size(); ++i) {
            const int k = frac[i].second;
            if (ns[(size_t)k] < std::max(1, h->sn[k].nchunk / min_chunks)) { ns[(size_t)k]++; assigned++; }
        }
    }
    h->work.clear();
    int slot = 0;
    for (int k = 0; k < h->K; ++k) {
        SubnetDev& s = h->sn[k];
        s.slot0 = slot;
        s.nslice = ns[(size_t)k];
        for (int i = 0; i < s.nslice; ++i) {
            Work w;
            w.k = k;
            w.c0 = (int)((int64_t)s.nchunk * i / s.nslice);
            w.c1 = (int)((int64_t)s.nchunk * (i + 1) / s.nslice);
            w.slot = slot++;
            h->work.push_back(w);
        }
    }
    h->nslots = slot;
}

static int sync_lanes_fwd(dimn_handle h);
static void build_mid(dimn_handle h) {
    // Fused second layer (H = 256): every sub-net's OT output tiles are cut into S slices, S*K <= ncu so that
    // each CU runs at most one workgroup (S <= OT: one tile per slice when a GPU owns few sub-nets); a slice
    // holds at most DIMN_MID_TMAX tiles (LDS).  DIMN_MID=0 keeps the two-kernel path (MF + MB).
    const Dims& dm = h->dm;
    h->mid_fused = 0;
    if (dm.HT != 16) return;
    int force = -1;
    int want_slices = 0;
    if (const char* e = getenv("DIMN_MID")) {                       // "0": the two-kernel second layer; "1": the fused one whatever the size; "1:S": with S slices per sub-net (tests)
        force = atoi(e) != 0;
        if (const char* c = strchr(e, ':')) want_slices = atoi(c + 1);
    }
    if (force == 0) return;
    // S <= 8: finer slices (down to one tile per workgroup) were measured for GPUs that own few sub-nets and
    // bring nothing (K=5: MFB 12.8 + RED2 8.5 us vs MF 11.3 + MB 10.6), so small K keeps the two-kernel path:
    // fused from ~0.6 workgroups per CU up (per step: K=5 62 vs 54 us, K=10 77 vs 68, K=20 106 vs 107, K=40 170 vs 186)
    int S = std::max(1, std::min(h->ncu / std::max(1, h->K), std::min(8, (int)dm.OT)));
    S = std::max(S, ceil_div(dm.OT, DIMN_MID_TMAX));
    if (want_slices > 0) S = std::max(ceil_div(dm.OT, DIMN_MID_TMAX), std::min(want_slices, (int)dm.OT));
    if (S > dm.OT) return;
    // (precision bf16: the fused kernel has the bf16 matrix-core variant and wins from a quarter-filled GPU on -- configs[4]'s 8 sub-nets
    //  per rank: 62.0 vs 63.8 us per step)
    if (force < 0 && (h->prec == DIMN_PREC_BF16 ? 4 * S * h->K < h->ncu : 5 * S * h->K < 3 * h->ncu)) return;
    h->mid_slices = S;
    h->dm.LS = std::max((int)dm.OS, S);
    h->midwork.clear();
    int slot = 0;
    for (int k = 0; k < h->K; ++k)
        for (int i = 0; i < S; ++i) {
            MidWork m;
            m.k = k; m.ot0 = dm.OT * i / S; m.ot1 = dm.OT * (i + 1) / S; m.slot = slot++; m.sidx = i;
            h->midwork.push_back(m);
        }
    h->mid_fused = 1;
    h->train_bf16 = h->prec == DIMN_PREC_BF16 && !(getenv("DIMN_TRAIN_BF16") && atoi(getenv("DIMN_TRAIN_BF16")) == 0);   // (the pipeline has the bf16 form at any slice size)
}

// Test knobs of the register-resident path in ONE variable: DIMN_RES_TEST="s1=2,groups=2,split=1,erows=0,abort=3" (any subset; tests/test_gpu_*.py).
//   s1      cap on the D-splits per hidden tile (other decompositions on small problems)     groups  minimum number of sub-net groups
//   split   tile order of the kernel's loop (0 alternating, 1 all gradient tiles first)      erows   epoch-ordered row copies on / off
//   abort   pretend the epoch launch number N (1-based) timed out
static int res_test_knob(const char* key, int fallback) {
    const char* e = getenv("DIMN_RES_TEST");
    if (!e) return fallback;
    const size_t kl = strlen(key);
    for (const char* p = e; *p;) {
        if (strncmp(p, key, kl) == 0 && p[kl] == '=') return atoi(p + kl + 1);
        const char* c = strchr(p, ',');
        if (!c) break;
        p = c + 1;
    }
    return fallback;
}
static bool resident_plan(dimn_handle h, int Kg, int& S1o, int& T1o) {
    // the decomposition of one launch over Kg sub-nets: D-splits per hidden tile, W1 tiles per wave; false: not eligible
    const Dims& dm = h->dm;
    int S1 = std::min(8, h->ncu / std::max(1, Kg) / 16);
    if (const int cap = res_test_knob("s1", 0)) S1 = std::min(S1, std::max(1, cap));         // tests: other decompositions
    if (S1 < 1 || dm.OT > 16 * S1) return false;
    int maxchunk = 0, minchunk = 1 << 30;
    for (auto& s : h->sn) { maxchunk = std::max(maxchunk, s.nchunk); minchunk = std::min(minchunk, s.nchunk); }
    if (minchunk < S1) S1 = std::max(1, minchunk);
    if (dm.OT > 16 * S1) return false;
    const int per_wg = ceil_div(maxchunk, S1);
    const int T1 = ceil_div(per_wg + 1, 8);                  // +1: the integer split of nchunk may give one workgroup one more
    if (T1 > 7) return false;
    const int T1c = T1 <= 2 ? 2 : (T1 <= 4 ? 4 : 7);         // the kernel instance; res_chunk_range() never gives a split more than 8 * T1c chunks
    for (auto& s : h->sn)
        for (int sp = 0; sp < S1; ++sp) {
            int cb, ce;
            res_chunk_range(s.nchunk, S1, sp, cb, ce);
            if (ce - cb > 8 * T1c || ce - cb < 1 || cb < 0 || ce > s.nchunk) return false;
        }
    S1o = S1; T1o = T1c;
    return true;
}

static void build_resident(dimn_handle h) {
    // Register-resident epoch kernel (dimn_resident.h): every sub-net gets G = 16*S1 co-resident workgroups (hidden tile x
    // D-split), one per CU; eligible when the workgroups of a launch fit the CUs, the W1 slice of a wave is at most 7 tiles
    // (register budget), the output tiles fit the G workgroups, and the shapes are the ones the kernel is written for
    // (H padded to 256, relu, H % 4 == 0).  Sub-nets share nothing, so a handle whose state does not fit at once trains its
    // sub-nets in GROUPS, one epoch launch per group after the other: at 10 sub-nets of D ~ 2400 (one rank of a 4-GPU job)
    // two launches of 5 cost 2 x 26 us per optimiser step against 70 us for the four streaming launches, three launches at
    // 15 sub-nets 77 against 87 us; from four groups on the streaming kernels are as fast (DIMN_RES_GROUPS: the largest group
    // count taken, default 3).
    // DIMN_RESIDENT=0 disables the kernel, =1 is the default (auto).
    h->res_G = h->res_S1 = h->res_T1 = 0; h->res_Kg = 0; h->res_bf16 = 0;
    const Dims& dm = h->dm;
    if (const char* e = getenv("DIMN_RESIDENT")) if (atoi(e) == 0) return;
    if (dm.HT != 16 || (dm.H & 3) != 0 || h->B > DIMN_TB) return;
    const int max_groups = 3;
    int min_groups = 1;
    min_groups = std::max(1, res_test_knob("groups", 1));                                      // tests: groups on small problems
    min_groups = std::min(min_groups, h->K);
    // (round 3: with the manager protocol a launch of five sub-nets costs 23.7 us per step, so FOUR groups of five -- the 2-GPU share of the
    //  50k x 20k job -- take 94 us against 104 us for the streaming kernels; four groups of four (K = 16) only draw: 86 vs 84-88 us)
    const bool four_of_five = ceil_div(h->K, 4) == 5;
    for (int groups = min_groups; groups <= std::min(std::max(four_of_five ? 4 : max_groups, min_groups), h->K); ++groups) {
        const int Kg = ceil_div(h->K, groups);
        int S1 = 0, T1c = 0;
        if (!resident_plan(h, Kg, S1, T1c)) continue;
        h->res_G = 16 * S1; h->res_S1 = S1; h->res_T1 = T1c; h->res_Kg = Kg;
        h->res_bf16 = h->prec == DIMN_PREC_BF16 && !(getenv("DIMN_TRAIN_BF16") && atoi(getenv("DIMN_TRAIN_BF16")) == 0);
        return;
    }
}

static int create_common(const dimn_config* cfg, const int32_t* D, bool general, dimn_handle* out) {
    if (!cfg || !D || !out) return fail(DIMN_ERR_ARG, "dimn_create: null argument");
    if (cfg->n_subnets < 1 || cfg->hidden < 1 || cfg->out_dim < 1)
        return fail(DIMN_ERR_ARG, "dimn_create: n_subnets/hidden/out_dim must be >= 1");
    if (cfg->batch_size < 1 || (!general && cfg->batch_size > DIMN_MAX_BATCH))
        return fail(DIMN_ERR_UNSUP, "dimn_create: batch_size %d not in 1..%d (dimn_create_general takes any batch size)", cfg->batch_size, DIMN_MAX_BATCH);
    if (!(cfg->dropout_rate >= 0.f && cfg->dropout_rate < 1.f))
        return fail(DIMN_ERR_ARG, "dimn_create: dropout_rate must be in [0,1)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(DIMN_ERR_HIP, "dimn_create: no HIP device visible (libdimn has no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(DIMN_ERR_ARG, "dimn_create: device_id %d out of range (%d devices)", cfg->device_id, ndev);

    if (cfg->precision != DIMN_PREC_F32 && cfg->precision != DIMN_PREC_BF16)
        return fail(DIMN_ERR_ARG, "dimn_create: precision must be DIMN_PREC_F32 or DIMN_PREC_BF16");
    dimn_handle h = new dimn_handle_s();
    h->cfg = *cfg;
    h->prec = cfg->precision;
    h->K = cfg->n_subnets; h->H = cfg->hidden; h->O = cfg->out_dim; h->B = cfg->batch_size;
    Dims& dm = h->dm;
    dm.K = h->K; dm.H = h->H; dm.O = h->O;
    dm.Hp = ceil_div(h->H, 16) * 16; dm.HT = dm.Hp / 16;
    // hidden = 300 is the reference CLI's default (parser.py): 19 tiles -> 20 (zero-padded, provably inert), so that
    // the shared-staging B1F1 kernel can run it as 10 waves x 2 whole tiles with no predicated memory op: 165 vs 204 us
    // per launch, step 0.276 vs 0.298 ms at 50k x 20k (DIMN_HT20=0: off).  The three-set ring needs 168 VGPRs + 50
    // spilled at 10 waves x 2 tiles, and two co-resident 10 x 1 workgroups spill 16: both no faster than the generic kernel
    // (Round 5 measured the 20th tile as an ALIAS of the 19th instead of padding -- Hp = 304, the owning wave updating that tile twice with identical operands:
    //  neutral in time, profiles/r05_h300_ab.txt, and a second copy's load is only ordered before the first copy's store by timing; removed.)
    if (dm.HT == 19) { dm.Hp = 320; dm.HT = 20; }
    // the same padding for every odd tile count above 16 (round 5): the first-layer ring kernel takes the hidden tiles of a D-slice in two halves of HT / 2 waves
    if (!general && dm.HT > 16 && dm.HT <= 24 && (dm.HT & 1)) { dm.HT += 1; dm.Hp = 16 * dm.HT; }
    dm.Op = ceil_div(h->O, 16) * 16; dm.OT = dm.Op / 16;
    dm.ldd = dm.Hp + ((dm.Hp % 32 == 0) ? 2 : 18);   // LDS row stride = 2 (mod 32) words: conflict-free b32 column reads
    dm.ldp = dm.Hp + ((dm.Hp % 32 == 0) ? 4 : 20);   // k_predict: 4 (mod 32) words, rows 16-byte aligned: conflict-free b128 row reads
    dm.OS = ceil_div(dm.OT, 4);
    dm.LS = dm.OS;
    h->NT = ceil_div(dm.HT, 4);
    h->NT2 = ceil_div(dm.HT, 8);
    h->OTW = ceil_div(dm.OT, 4);   // output tiles per wave of the 4-wave middle-backward kernel (k_mid_bwd)
    h->HS = ceil_div(dm.HT, 2);
    if (!general && h->NT > 6) {
        delete h;
        return fail(DIMN_ERR_UNSUP, "dimn_create: hidden=%d > 384 is outside the tuned kernels (use dimn_create_general)", cfg->hidden);
    }
    if (!general && ((size_t)DIMN_TB * dm.ldp + DIMN_PRED_XS) * sizeof(float) > 160 * 1024) {
        delete h;
        return fail(DIMN_ERR_UNSUP, "dimn_create: hidden too large for LDS staging");
    }
    hipDeviceProp_t prop;
    if (hipSetDevice(cfg->device_id) != hipSuccess || hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess) {
        delete h;
        return fail(DIMN_ERR_HIP, "dimn_create: cannot select device %d", cfg->device_id);
    }
    h->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    h->sn.resize(h->K);
    h->pred.resize(h->K); h->targ.resize(h->K);
    int64_t w1 = 0;
    for (int k = 0; k < h->K; ++k) {
        if (D[k] < 1) { delete h; return fail(DIMN_ERR_ARG, "dimn_create: D[%d] < 1", k); }
        SubnetDev& s = h->sn[k];
        s.D = D[k]; s.Dp = ceil_div(D[k], 16) * 16; s.nchunk = s.Dp / 16;
        s.kg = cfg->subnet_offset + k;
        s.xoff = 0;
        s.w1off = w1;
        s.lim1 = (float)sqrt(6.0 / ((double)s.D + h->H));
        s.lim2 = (float)sqrt(6.0 / ((double)h->H + h->O));
        w1 += (int64_t)s.Dp * dm.Hp;
    }
    h->w1_total = w1;
    build_work(h);
    if (!general) { build_mid(h); build_resident(h); }
    // One lane: every sub-net on the handle's stream.  (Two free-running lanes on two streams, a "W token" ring between them and a fixed
    // CU partition with CU-masked streams were all measured and lost to the serial step: DESIGN.md section 2, profiles/r03_cu_partition_sweep.txt.)
    const int n_lanes = 1;

    const size_t w2n = (size_t)h->K * dm.Hp * dm.Op;
#define TRY(expr) do { int rc_ = (expr); if (rc_) { dimn_destroy(h); return rc_; } } while (0)
    for (int l = 0; l < n_lanes; ++l) {
        dimn_handle_s::Lane ln;
        if (hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) != hipSuccess) {
            delete h;
            return fail(DIMN_ERR_HIP, "dimn_create: hipStreamCreate failed");
        }
        ln.k0 = (int)((int64_t)h->K * l / n_lanes);
        ln.k1 = (int)((int64_t)h->K * (l + 1) / n_lanes);
        ln.w0 = h->sn[ln.k0].slot0;
        ln.w1 = ln.k1 < h->K ? h->sn[ln.k1].slot0 : h->nslots;
        h->lanes.push_back(ln);
    }
    h->stream = h->lanes[0].stream;
    TRY(dev_alloc(&h->d_sn, (size_t)h->K));
    if (general) {                                   // the general path allocates its own state (gen_setup)
        if (hipMemcpy(h->d_sn, h->sn.data(), h->sn.size() * sizeof(SubnetDev), hipMemcpyHostToDevice) != hipSuccess) {
            dimn_destroy(h);
            return fail(DIMN_ERR_HIP, "dimn_create_general: descriptor upload failed");
        }
        *out = h;
        return DIMN_OK;
    }
    TRY(dev_alloc(&h->d_work, h->work.size()));
    TRY(dev_alloc(&h->d_W1, (size_t)w1)); TRY(dev_alloc(&h->d_M1, (size_t)w1)); TRY(dev_alloc(&h->d_V1, (size_t)w1));
    TRY(dev_alloc(&h->d_W2, w2n)); TRY(dev_alloc(&h->d_M2, w2n)); TRY(dev_alloc(&h->d_V2, w2n));
    TRY(dev_alloc(&h->d_b1, (size_t)3 * h->K * dm.Hp)); TRY(dev_alloc(&h->d_b2, (size_t)3 * h->K * dm.Op));
    TRY(dev_alloc(&h->d_P, (size_t)h->nslots * DIMN_TB * dm.Hp));
    TRY(dev_alloc(&h->d_Dd, (size_t)h->K * DIMN_TB * dm.Hp));
    TRY(dev_alloc(&h->d_dA, (size_t)h->K * DIMN_TB * dm.Hp));
    TRY(dev_alloc(&h->d_dZ, (size_t)h->K * DIMN_TB * dm.Op));
    TRY(dev_alloc(&h->d_loss_step, (size_t)h->K * dm.LS));
    TRY(dev_alloc(&h->d_loss_acc, (size_t)h->K * dm.LS));
    TRY(dev_alloc(&h->d_mask, (size_t)h->K * DIMN_TB * dm.Hp));
    TRY(dev_alloc(&h->d_rows_step, (size_t)DIMN_TB));
    if (h->prec == DIMN_PREC_BF16) {
        // inference / validation on the bf16 matrix cores unless DIMN_PREDICT_BF16=0 (then only the arena is bf16)
        h->predict_bf16 = !(getenv("DIMN_PREDICT_BF16") && atoi(getenv("DIMN_PREDICT_BF16")) == 0);
        if (h->predict_bf16) {
            const size_t w1pad = 4096;                           // k_predict_bf16 reads up to 64 hidden units past the last one of a chunk (accumulators nobody uses): mapped memory there
            TRY(dev_alloc(&h->d_W1b, (size_t)w1 + w1pad)); TRY(dev_alloc(&h->d_W2t, w2n));
            TRY(hipMemset(h->d_W1b + w1, 0, w1pad * 2) == hipSuccess ? 0 : fail(DIMN_ERR_HIP, "hipMemset failed"));
            TRY(dev_alloc(&h->d_zero1k, (size_t)512));           // the zeros k_predict_bf16 fetches for predictor chunks past a sub-net's last one
            TRY(hipMemset(h->d_zero1k, 0, 1024) == hipSuccess ? 0 : fail(DIMN_ERR_HIP, "hipMemset failed"));
        }
    }
    if (!h->predict_bf16) TRY(dev_alloc(&h->d_W2tf, w2n));
    if (h->res_G) {
        TRY(dev_alloc(&h->d_res_P, (size_t)DIMN_RES_SLOTS * h->K * h->res_G * 1024));          // forward partials (siblings -> manager)
        TRY(dev_alloc(&h->d_res_D, (size_t)DIMN_RES_SLOTS * h->K * dm.OT * 16 * 1024));        // dD partials (role 2 -> manager)
        TRY(dev_alloc(&h->d_res_T, (size_t)DIMN_RES_SLOTS * h->K * 16 * 1024));                // Dd tiles (manager -> role 2)
        TRY(dev_alloc(&h->d_res_A, (size_t)DIMN_RES_SLOTS * h->K * 16 * 1024));                // dA tiles (manager -> siblings)
        TRY(dev_alloc(&h->d_res_flags, (size_t)2 * h->K + 1));
        TRY(dev_alloc(&h->d_res_loss, (size_t)h->K * dm.OT));
    }
    if (h->mid_fused) {
        std::vector<int32_t> midk((size_t)2 * h->K);
        for (int k = 0; k < h->K; ++k) { midk[2 * k] = k * h->mid_slices; midk[2 * k + 1] = h->mid_slices; }
        TRY(dev_alloc(&h->d_midwork, h->midwork.size()));
        TRY(dev_alloc(&h->d_midk, midk.size()));
        TRY(dev_alloc(&h->d_P2, h->midwork.size() * DIMN_TB * dm.Hp));
        if (hipMemcpy(h->d_midwork, h->midwork.data(), h->midwork.size() * sizeof(MidWork), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_midk, midk.data(), midk.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            dimn_destroy(h);
            return fail(DIMN_ERR_HIP, "dimn_create: descriptor upload failed");
        }
        (void)hipFuncSetAttribute((const void*)k_mid_pipe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)k_mid_pipe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    auto zero = [&](void* p, size_t bytes) { return hipMemset(p, 0, bytes) == hipSuccess ? 0 : fail(DIMN_ERR_HIP, "hipMemset failed"); };
    TRY(zero(h->d_W1, w1 * 4)); TRY(zero(h->d_M1, w1 * 4)); TRY(zero(h->d_V1, w1 * 4));
    TRY(zero(h->d_W2, w2n * 4)); TRY(zero(h->d_M2, w2n * 4)); TRY(zero(h->d_V2, w2n * 4));
    TRY(zero(h->d_b1, (size_t)3 * h->K * dm.Hp * 4)); TRY(zero(h->d_b2, (size_t)3 * h->K * dm.Op * 4));
    TRY(zero(h->d_Dd, (size_t)h->K * DIMN_TB * dm.Hp * 4)); TRY(zero(h->d_dA, (size_t)h->K * DIMN_TB * dm.Hp * 4));
    TRY(zero(h->d_dZ, (size_t)h->K * DIMN_TB * dm.Op * 4));
    TRY(zero(h->d_loss_step, (size_t)h->K * dm.LS * 4)); TRY(zero(h->d_loss_acc, (size_t)h->K * dm.LS * 8));
    if (hipMemcpy(h->d_work, h->work.data(), h->work.size() * sizeof(Work), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(h->d_sn, h->sn.data(), h->sn.size() * sizeof(SubnetDev), hipMemcpyHostToDevice) != hipSuccess) {
        dimn_destroy(h);
        return fail(DIMN_ERR_HIP, "dimn_create: descriptor upload failed");
    }
#undef TRY
    *out = h;
    return DIMN_OK;
}

extern "C" int dimn_create(const dimn_config* cfg, const int32_t* D, dimn_handle* out) { return create_common(cfg, D, false, out); }

// build(inputdims) for ANY architecture list (multinet.py:126-167): `layers` = the hidden Dense layers in order, each with the
// rate of the Dropout layer that follows it (0: none); the softplus output layer of out_dim units is implied.
extern "C" int dimn_create_general(const dimn_config* cfg, const int32_t* D, const dimn_layer* layers, int32_t n_layers, int32_t loss, dimn_handle* out) {
    if (!cfg || !layers || n_layers < 1 || n_layers > 17) return fail(DIMN_ERR_ARG, "dimn_create_general: 1..16 hidden layers expected");
    if (loss < DIMN_LOSS_WMSE || loss > DIMN_LOSS_LAST) return fail(DIMN_ERR_UNSUP, "dimn_create_general: unknown loss id %d", loss);
    // A Dropout layer BEFORE the first Dense layer (dropout on the inputs, multinet.py:139-141 allows it) is written as a leading
    // entry with neurons == 0 and its rate; the hidden layers follow.
    float in_rate = 0.f;
    if (layers[0].neurons == 0) {
        in_rate = layers[0].dropout_rate;
        if (!(in_rate > 0.f && in_rate < 1.f) || n_layers < 2) return fail(DIMN_ERR_ARG, "dimn_create_general: an input-dropout entry needs a rate in (0,1) and a hidden layer behind it");
        ++layers; --n_layers;
    }
    if (n_layers > 16) return fail(DIMN_ERR_ARG, "dimn_create_general: 1..16 hidden layers expected");
    for (int l = 0; l < n_layers; ++l) {
        if (layers[l].neurons < 1) return fail(DIMN_ERR_ARG, "dimn_create_general: layer %d has no neurons", l);
        if (layers[l].activation < DIMN_ACT_RELU || layers[l].activation > DIMN_ACT_LAST) return fail(DIMN_ERR_UNSUP, "dimn_create_general: unknown activation id in layer %d", l);
        if (!(layers[l].dropout_rate >= 0.f && layers[l].dropout_rate < 1.f)) return fail(DIMN_ERR_ARG, "dimn_create_general: dropout rate of layer %d not in [0,1)", l);
    }
    dimn_config c = *cfg;
    c.hidden = layers[0].neurons; c.dropout_rate = 0.f;
    dimn_handle h = nullptr;
    CHK(create_common(&c, D, true, &h));
    h->cfg.loss_binary = loss == DIMN_LOSS_WMSE_BINARY;
    const int rc = gen_setup(h, layers, n_layers, loss, in_rate);
    if (rc != DIMN_OK) { dimn_destroy(h); return rc; }
    *out = h;
    return DIMN_OK;
}

extern "C" int dimn_set_layer_weights(dimn_handle h, int32_t k, int32_t layer, const float* W, const float* b) {
    if (!h || !h->gen || k < 0 || k >= h->K || !W || !b) return fail(DIMN_ERR_ARG, "dimn_set_layer_weights: bad argument (general handles only)");
    CHK(use_device(h));
    return gen_io_layer(h, k, layer, 0, (float*)W, (float*)b, true);
}
extern "C" int dimn_get_layer_weights(dimn_handle h, int32_t k, int32_t layer, int32_t which, float* W, float* b) {
    if (!h || !h->gen || k < 0 || k >= h->K || which < 0 || which > 2 || !W || !b) return fail(DIMN_ERR_ARG, "dimn_get_layer_weights: bad argument (general handles only)");
    CHK(use_device(h));
    return gen_io_layer(h, k, layer, which, W, b, false);
}

extern "C" int dimn_destroy(dimn_handle h) {
    if (!h) return DIMN_OK;
    (void)hipSetDevice(h->cfg.device_id);
    for (auto& ln : h->lanes) (void)hipStreamSynchronize(ln.stream);
    gen_free(h->gen); h->gen = nullptr;
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    for (auto e : h->ev) (void)hipEventDestroy(e);
    for (auto e : h->pred_ev) (void)hipEventDestroy(e);
    DEV_FREE(h->d_pred_iota);
    DEV_FREE(h->d_sn); DEV_FREE(h->d_work); DEV_FREE(h->d_norm); DEV_FREE(h->d_X); DEV_FREE(h->d_Y);
    DEV_FREE(h->d_pred); DEV_FREE(h->d_targ); DEV_FREE(h->d_pred_off);
    DEV_FREE(h->d_W1); DEV_FREE(h->d_M1); DEV_FREE(h->d_V1); DEV_FREE(h->d_W2); DEV_FREE(h->d_M2); DEV_FREE(h->d_V2);
    DEV_FREE(h->d_b1); DEV_FREE(h->d_b2); DEV_FREE(h->d_P); DEV_FREE(h->d_Dd); DEV_FREE(h->d_dZ); DEV_FREE(h->d_dA);
    DEV_FREE(h->d_loss_step); DEV_FREE(h->d_loss_acc); DEV_FREE(h->d_mask); DEV_FREE(h->d_rows_step);
    DEV_FREE(h->d_midwork); DEV_FREE(h->d_midk); DEV_FREE(h->d_P2); DEV_FREE(h->d_G);
    DEV_FREE(h->d_epoch_rows); DEV_FREE(h->d_val_rows); DEV_FREE(h->d_pred_rows); DEV_FREE(h->d_out);
    DEV_FREE(h->d_loss_part); DEV_FREE(h->d_full); DEV_FREE(h->d_stage); DEV_FREE(h->d_red);
    DEV_FREE(h->d_W1b); DEV_FREE(h->d_W2t); DEV_FREE(h->d_W2tf); DEV_FREE(h->d_zero1k);
    DEV_FREE(h->d_res_P); DEV_FREE(h->d_res_D); DEV_FREE(h->d_res_T); DEV_FREE(h->d_res_A); DEV_FREE(h->d_res_b1); DEV_FREE(h->d_res_alpha); DEV_FREE(h->d_res_flags); DEV_FREE(h->d_res_loss); DEV_FREE(h->d_res_snap); DEV_FREE(h->d_res_Xe); DEV_FREE(h->d_res_Ye); DEV_FREE(h->d_res_iota);
    for (auto& ln : h->lanes) (void)hipStreamDestroy(ln.stream);
    delete h;
    return DIMN_OK;
}

extern "C" int dimn_set_matrix(dimn_handle h, const float* norm, int64_t n, int64_t g) {
    if (h) h->counts = nullptr;
    if (!h || !norm || n < 1 || g < 1) return fail(DIMN_ERR_ARG, "dimn_set_matrix: bad argument");
    if (n > 0x7fffffffLL || g > 0x7fffffffLL) return fail(DIMN_ERR_UNSUP, "dimn_set_matrix: dimension exceeds int32");
    CHK(use_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n != h->n || g != h->g) {
        DEV_FREE(h->d_norm);
        CHK(dev_alloc(&h->d_norm, (size_t)n * g));
        h->gathered = false;
        if (n != h->n) { h->n_tr = 0; h->n_val = 0; h->train_rows.clear(); h->val_rows.clear(); }   // row indices of another matrix
    }
    HIPCHK(hipMemcpy(h->d_norm, norm, (size_t)n * g * sizeof(float), hipMemcpyHostToDevice));
    h->n = n; h->g = g;
    h->gathered = false; h->streamed = false;
    return DIMN_OK;
}

extern "C" int dimn_set_indices(dimn_handle h, int32_t k, const int32_t* pred_idx, int32_t D_k, const int32_t* targ_idx) {
    if (!h || k < 0 || k >= h->K || !pred_idx || !targ_idx) return fail(DIMN_ERR_ARG, "dimn_set_indices: bad argument");
    if (D_k != h->sn[k].D) return fail(DIMN_ERR_ARG, "dimn_set_indices: D_k=%d differs from create() (%d)", D_k, h->sn[k].D);
    h->pred[k].assign(pred_idx, pred_idx + D_k);
    h->targ[k].assign(targ_idx, targ_idx + h->O);
    h->gathered = false;
    return DIMN_OK;
}

// Host worker threads of the row-block pipelines (counts upload, streamed hand-over, predict()'s epilogue), made once per process.  Every
// pipeline stage used to create and join its own threads -- 24 to 64 of them per ~128 MB block, 31-62 blocks per call: ~1-2 ms of
// pthread_create / join per block beside 2-5 ms of useful work.  run(n, fn) executes fn(0 .. n-1), fn(0) on the caller; calls from
// several threads at once (a retiring block beside the next copy-in) share the workers.  The pool is never destroyed (its threads
// end with the process).
class HostPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !q.empty(); });
                job = std::move(q.front());
                q.pop_front();
            }
            job();
        }
    }
public:
    explicit HostPool(unsigned n) { for (unsigned i = 0; i < n; ++i) std::thread([this] { loop(); }).detach(); }
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 1) { if (n == 1) fn(0); return; }
        struct Ctx { std::mutex m; std::condition_variable c; int left; } ctx;
        ctx.left = n - 1;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (int t = 1; t < n; ++t)
                q.emplace_back([&ctx, &fn, t] {
                    fn(t);
                    std::lock_guard<std::mutex> l2(ctx.m);
                    if (--ctx.left == 0) ctx.c.notify_one();
                });
        }
        cv.notify_all();
        fn(0);
        std::unique_lock<std::mutex> l3(ctx.m);
        ctx.c.wait(l3, [&] { return ctx.left == 0; });
    }
};
static HostPool& host_pool() {
    static HostPool* pool = new HostPool(std::min<unsigned>(64, std::max(4u, std::thread::hardware_concurrency() / 2)));
    return *pool;
}
static void csv_parallel(int n, const std::function<void(int)>& fn) { host_pool().run(n, fn); }      // (dimn_csv.h)

// memcpy of a large block on several host threads (one pageable <-> pinned copy per pipeline stage: a single thread
// moves ~10 GB/s, the PCIe link five times that)
static void parallel_memcpy(void* dst, const void* src, size_t bytes) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(hw ? hw / 2 : 8, 24), bytes / (4u << 20)));
    if (nt <= 1) { memcpy(dst, src, bytes); return; }
    const size_t chunk = ((bytes + nt - 1) / nt + 63) & ~(size_t)63;
    host_pool().run((int)nt, [=](int i) {
        const size_t a = (size_t)i * chunk, b = std::min(bytes, a + chunk);
        if (a < b) memcpy((char*)dst + a, (const char*)src + a, b - a);
    });
}

// dst[r][j] = src[r][cols[j]], r < nr: the columns a handle needs of a row block, packed (host threads over rows; a row is walked
// front to back, so the reads stream)
static void parallel_pack_columns(float* dst, const float* src, int64_t nr, int64_t g, const int32_t* cols, int64_t gc) {
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned cap = 48;      // (16 .. 128 threads measured the same: the host reads its matrix at ~70 GB/s)
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw ? hw / 2 : 8, cap), nr * g / (1 << 20)));
    auto work = [=](int t) {
        for (int64_t r = nr * t / nt; r < nr * (t + 1) / nt; ++r) {
            const float* in = src + r * g;
            float* out = dst + r * gc;
            for (int64_t j = 0; j < gc; ++j) out[j] = in[cols[j]];
        }
    };
    host_pool().run(nt, work);
}

// Index lists and arenas of the device gather for a matrix of h->n cells (validated against h->g columns).
static int gather_prepare(dimn_handle h, int32_t with_targets) {
    for (int k = 0; k < h->K; ++k) {
        if ((int)h->pred[k].size() != h->sn[k].D) return fail(DIMN_ERR_STATE, "dimn_gather: dimn_set_indices missing for sub-net %d", k);
        for (int32_t c : h->pred[k]) if (c < 0 || c >= h->g) return fail(DIMN_ERR_ARG, "dimn_gather: predictor column %d out of range", c);
        for (int32_t c : h->targ[k]) if (c < 0 || c >= h->g) return fail(DIMN_ERR_ARG, "dimn_gather: target column %d out of range", c);
    }
    CHK(use_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    Trace tr;
    std::vector<int32_t> pflat, tflat;
    std::vector<int64_t> poff(h->K);
    for (int k = 0; k < h->K; ++k) {
        poff[k] = (int64_t)pflat.size();
        pflat.insert(pflat.end(), h->pred[k].begin(), h->pred[k].end());
        tflat.insert(tflat.end(), h->targ[k].begin(), h->targ[k].end());
    }
    if (!h->d_pred) { CHK(dev_alloc(&h->d_pred, pflat.size())); CHK(dev_alloc(&h->d_targ, tflat.size())); CHK(dev_alloc(&h->d_pred_off, (size_t)h->K)); }
    HIPCHK(hipMemcpy(h->d_pred, pflat.data(), pflat.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_targ, tflat.data(), tflat.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_pred_off, poff.data(), poff.size() * 8, hipMemcpyHostToDevice));
    tr.lap("gather: index lists");
    int64_t x = 0;
    for (int k = 0; k < h->K; ++k) {
        if ((int64_t)h->n * h->sn[k].Dp > 0xffffffffLL) return fail(DIMN_ERR_UNSUP, "dimn_gather: n*Dp exceeds 32-bit row offsets");
        h->sn[k].xoff = x;
        x += (int64_t)h->n * h->sn[k].Dp;
    }
    // the arenas are re-used across calls (19.5 GB at cfg3: a hipFree/hipMalloc pair costs up to a second)
    if (!h->d_X || h->x_total != x) {
        DEV_FREE(h->d_X); DEV_FREE(h->d_res_Xe);
        HIPCHK(dev_malloc_bytes((void**)&h->d_X, std::max<size_t>(1, (size_t)x * XBYTES(h))));
        h->x_total = x;
    }
    const int64_t y_need = (int64_t)h->K * h->n * h->dm.Op;
    if (with_targets && (!h->d_Y || h->y_total != y_need)) {
        DEV_FREE(h->d_Y); DEV_FREE(h->d_res_Ye);
        CHK(dev_alloc(&h->d_Y, (size_t)y_need));
        h->y_total = y_need;
    }
    tr.lap("gather: X / Y arenas");
    HIPCHK(hipMemcpy(h->d_sn, h->sn.data(), h->sn.size() * sizeof(SubnetDev), hipMemcpyHostToDevice));
    return DIMN_OK;
}
// X_k / Y_k rows [row0, row0 + nrows) from a device block of the matrix (the whole matrix, or one streamed block)
// (g_eff / pred / targ: a block whose rows hold only SOME columns of the matrix, with index lists that address those -- the streamed hand-over)
static int gather_block(dimn_handle h, const float* d_block, int64_t nrows, int64_t row0, int32_t with_targets, hipStream_t st,
                        int64_t g_eff = 0, const int32_t* pred = nullptr, const int32_t* targ = nullptr) {
    if (g_eff <= 0) g_eff = h->g;
    if (!pred) pred = h->d_pred;
    if (!targ) targ = h->d_targ;
    if ((size_t)g_eff * sizeof(float) <= 150 * 1024) {      // the row fits in LDS: read `norm` once, serve all sub-nets from LDS
        const size_t lds = (size_t)g_eff * sizeof(float);
        WITH_XT(h, {
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_gather_lds<XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_gather_lds<XT>, dim3((unsigned)std::min<int64_t>(nrows, 2048)), dim3(512), lds, st, h->d_sn, d_block, nrows, g_eff, pred,
                               h->d_pred_off, targ, (XT*)h->d_X, h->d_Y, h->dm, with_targets ? 1 : 0, row0, h->n);
        });
    } else {
        const dim3 grid((unsigned)h->K, (unsigned)std::min<int64_t>(nrows, 8192));
        WITH_XT(h, hipLaunchKernelGGL(k_gather<XT>, grid, dim3(256), 0, st, h->d_sn, d_block, nrows, g_eff, pred, h->d_pred_off, targ,
                                      (XT*)h->d_X, h->d_Y, h->dm, with_targets ? 1 : 0, row0, h->n));
    }
    HIPCHK(hipGetLastError());
    return DIMN_OK;
}

extern "C" int dimn_gather(dimn_handle h, int32_t with_targets) {
    if (!h) return fail(DIMN_ERR_ARG, "dimn_gather: null handle");
    if (!h->d_norm) return fail(DIMN_ERR_STATE, h->streamed ? "dimn_gather: the matrix was streamed (dimn_set_matrix_streamed gathers itself)" : "dimn_gather: call dimn_set_matrix first");
    CHK(gather_prepare(h, with_targets));
    Trace tr;
    CHK(gather_block(h, h->d_norm, h->n, 0, with_targets, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    tr.lap("gather: kernel");
    h->gathered = true;
    h->gathered_targets = with_targets != 0;
    return DIMN_OK;
}

// BASELINE configs[4]: the log1p matrix streamed from host memory in row blocks (pinned bounce buffers, the copy of one
// block overlapping the gather of the previous one); the device never holds the matrix itself, only the gathered X_k
// (fp32 or bf16) and Y_k blocks.  Replaces dimn_set_matrix + dimn_gather; needs every dimn_set_indices first.
extern "C" int dimn_set_stream_order(dimn_handle h, int32_t part, int32_t parts) {
    if (!h || parts < 1 || part < 0 || part >= parts) return fail(DIMN_ERR_ARG, "dimn_set_stream_order: need 0 <= part < parts");
    h->stream_part = part; h->stream_parts = parts;
    return DIMN_OK;
}

extern "C" int dimn_set_matrix_streamed(dimn_handle h, const float* norm, int64_t n, int64_t g, int32_t with_targets) {
    if (h) h->counts = nullptr;
    if (!h || !norm || n < 1 || g < 1) return fail(DIMN_ERR_ARG, "dimn_set_matrix_streamed: bad argument");
    if (n > 0x7fffffffLL || g > 0x7fffffffLL) return fail(DIMN_ERR_UNSUP, "dimn_set_matrix_streamed: dimension exceeds int32");
    CHK(use_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    DEV_FREE(h->d_norm);
    if (n != h->n) { h->n_tr = 0; h->n_val = 0; h->train_rows.clear(); h->val_rows.clear(); }
    h->n = n; h->g = g; h->gathered = false; h->streamed = true;
    CHK(gather_prepare(h, with_targets));
    // Only the columns this handle's sub-nets read cross PCIe: a rank of a sharded job needs the predictors and targets of ITS sub-nets
    // (configs[4], 8 of 59 sub-nets: ~55 % of the genes), so the host threads pack those columns of every row block into the bounce
    // buffer and the device gather runs on index lists that address the packed rows.  (DIMN_STREAM_PACK=0, or more than 85 % of the
    // columns needed: the rows go over as they are.)
    std::vector<int32_t> cols;
    int32_t *d_pred_c = nullptr, *d_targ_c = nullptr;
    {
        std::vector<int32_t> where((size_t)g, -1);
        for (int k = 0; k < h->K; ++k) {
            for (int32_t c : h->pred[k]) where[(size_t)c] = 0;
            if (with_targets) for (int32_t c : h->targ[k]) where[(size_t)c] = 0;
        }
        for (int64_t c = 0; c < g; ++c) if (where[(size_t)c] == 0) { where[(size_t)c] = (int32_t)cols.size(); cols.push_back((int32_t)c); }
        const char* e = getenv("DIMN_STREAM_PACK");
        if ((e && atoi(e) == 0) || (double)cols.size() > 0.85 * (double)g) cols.clear();
        if (!cols.empty()) {
            std::vector<int32_t> pflat, tflat;
            for (int k = 0; k < h->K; ++k) {
                for (int32_t c : h->pred[k]) pflat.push_back(where[(size_t)c]);
                for (int32_t c : h->targ[k]) tflat.push_back(with_targets ? where[(size_t)c] : 0);
            }
            CHK(dev_alloc(&d_pred_c, pflat.size()));
            if (dev_alloc(&d_targ_c, tflat.size()) != DIMN_OK) { (void)dev_free_any(d_pred_c); return DIMN_ERR_HIP; }
            if (hipMemcpy(d_pred_c, pflat.data(), pflat.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(d_targ_c, tflat.data(), tflat.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                (void)dev_free_any(d_pred_c); (void)dev_free_any(d_targ_c);
                return fail(DIMN_ERR_HIP, "dimn_set_matrix_streamed: index upload failed");
            }
        }
    }
    const int64_t gc = cols.empty() ? g : (int64_t)cols.size();
    const int64_t blk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(128u << 20) / (gc * 4)));
    const int NBUF = 2;          // blocks in flight (host packing | PCIe copy | device gather); 3 and 4 measured the same
    float *pin[4] = {nullptr, nullptr, nullptr, nullptr}, *dev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = DIMN_OK;
#define STR_TRY(expr) do { hipError_t e_ = (expr); if (rc == DIMN_OK && e_ != hipSuccess) rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
    for (int b = 0; b < NBUF; ++b) {
        STR_TRY(hipHostMalloc((void**)&pin[b], (size_t)blk * gc * 4, hipHostMallocDefault));
        STR_TRY(dev_malloc_bytes((void**)&dev[b], (size_t)blk * gc * 4));
        STR_TRY(hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking));
    }
    // The ranks of one node read ONE host copy of the matrix (deepimpute_amd/_shm.py): rank r of w starts at block r NB / w and wraps around, so the
    // w pack-and-copy pipelines walk different pages of it at any moment (the device gather of a block is independent of every other block)
    const int64_t NB = (n + blk - 1) / blk, first = NB * (int64_t)h->stream_part / std::max(1, h->stream_parts);
    for (int64_t bi = 0; bi < NB && rc == DIMN_OK; ++bi) {
        const int b = (int)(bi % NBUF);
        const int64_t r0 = ((bi + first) % NB) * blk;
        const int64_t nr = std::min(blk, n - r0);
        STR_TRY(hipStreamSynchronize(st[b]));               // block bi-NBUF has left these buffers
        if (rc != DIMN_OK) break;
        if (cols.empty()) parallel_memcpy(pin[b], norm + r0 * g, (size_t)nr * g * 4);
        else parallel_pack_columns(pin[b], norm + r0 * g, nr, g, cols.data(), gc);
        STR_TRY(hipMemcpyAsync(dev[b], pin[b], (size_t)nr * gc * 4, hipMemcpyHostToDevice, st[b]));
        if (rc == DIMN_OK) rc = gather_block(h, dev[b], nr, r0, with_targets, st[b], gc, d_pred_c, d_targ_c);
    }
#undef STR_TRY
    for (int b = 0; b < NBUF; ++b) {
        if (st[b]) { (void)hipStreamSynchronize(st[b]); (void)hipStreamDestroy(st[b]); }
        if (pin[b]) (void)hipHostFree(pin[b]);
        if (dev[b]) (void)dev_free_any(dev[b]);
    }
    if (d_pred_c) (void)dev_free_any(d_pred_c);
    if (d_targ_c) (void)dev_free_any(d_targ_c);
    if (rc != DIMN_OK) return rc;
    h->gathered = true;
    h->gathered_targets = with_targets != 0;
    return DIMN_OK;
}

extern "C" int dimn_set_split(dimn_handle h, const int32_t* tr, int64_t n_tr, const int32_t* va, int64_t n_val) {
    if (!h || n_tr < 0 || n_val < 0 || (n_tr > 0 && !tr) || (n_val > 0 && !va)) return fail(DIMN_ERR_ARG, "dimn_set_split: bad argument");
    if (h->n > 0) {
        for (int64_t i = 0; i < n_tr; ++i) if (tr[i] < 0 || tr[i] >= h->n) return fail(DIMN_ERR_ARG, "dimn_set_split: train row %d out of range", tr[i]);
        for (int64_t i = 0; i < n_val; ++i) if (va[i] < 0 || va[i] >= h->n) return fail(DIMN_ERR_ARG, "dimn_set_split: validation row %d out of range", va[i]);
    }
    CHK(use_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->train_rows.assign(tr, tr + n_tr);
    h->val_rows.assign(va, va + n_val);
    DEV_FREE(h->d_epoch_rows); DEV_FREE(h->d_val_rows);
    CHK(dev_alloc(&h->d_epoch_rows, (size_t)n_tr));
    CHK(dev_alloc(&h->d_val_rows, (size_t)n_val));
    if (n_val) HIPCHK(hipMemcpy(h->d_val_rows, va, (size_t)n_val * 4, hipMemcpyHostToDevice));
    h->n_tr = n_tr; h->n_val = n_val;
    return DIMN_OK;
}

static int zero_opt(dimn_handle h) {
    const Dims& dm = h->dm;
    const size_t w2n = (size_t)h->K * dm.Hp * dm.Op;
    HIPCHK(hipMemsetAsync(h->d_M1, 0, (size_t)h->w1_total * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_V1, 0, (size_t)h->w1_total * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_M2, 0, w2n * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_V2, 0, w2n * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_b1 + (size_t)h->K * dm.Hp, 0, (size_t)2 * h->K * dm.Hp * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_b2 + (size_t)h->K * dm.Op, 0, (size_t)2 * h->K * dm.Op * 4, h->stream));
    h->t = 0;
    return DIMN_OK;
}

extern "C" int dimn_set_activation(dimn_handle h, int32_t activation) {
    if (!h) return fail(DIMN_ERR_ARG, "null handle");
    if (h->gen) return fail(DIMN_ERR_ARG, "dimn_set_activation: a general handle takes its activations from dimn_create_general");
    if (activation < DIMN_ACT_RELU || activation > DIMN_ACT_LAST) return fail(DIMN_ERR_UNSUP, "dimn_set_activation: unknown activation %d", activation);
    CHK(use_device(h));
    CHK(sync_lanes_fwd(h));
    if (activation != DIMN_ACT_RELU && !h->d_G) {
        CHK(dev_alloc(&h->d_G, (size_t)h->K * DIMN_TB * h->dm.Hp));
        HIPCHK(hipMemset(h->d_G, 0, (size_t)h->K * DIMN_TB * h->dm.Hp * 4));
    }
    if (activation == DIMN_ACT_RELU) DEV_FREE(h->d_G);     // relu derives its gate from Dd > 0
    h->act = activation;
    return DIMN_OK;
}

extern "C" int dimn_reset_optimizer(dimn_handle h) {
    if (!h) return fail(DIMN_ERR_ARG, "null handle");
    CHK(use_device(h));
    if (h->gen) {
        HIPCHK(hipMemset(h->gen->d_M, 0, (size_t)h->gen->ptotal * 4));
        HIPCHK(hipMemset(h->gen->d_V, 0, (size_t)h->gen->ptotal * 4));
        h->t = 0;
        return DIMN_OK;
    }
    CHK(zero_opt(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DIMN_OK;
}

extern "C" int dimn_init_weights(dimn_handle h, uint64_t seed) {
    if (!h) return fail(DIMN_ERR_ARG, "null handle");
    CHK(use_device(h));
    if (h->gen) return gen_init_weights(h, seed);
    const Dims& dm = h->dm;
    HIPCHK(hipMemsetAsync(h->d_W1, 0, (size_t)h->w1_total * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_W2, 0, (size_t)h->K * dm.Hp * dm.Op * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_b1, 0, (size_t)h->K * dm.Hp * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->d_b2, 0, (size_t)h->K * dm.Op * 4, h->stream));
    CHK(zero_opt(h));
    hipLaunchKernelGGL(k_init_weights, dim3(256, (unsigned)h->K), dim3(256), 0, h->stream, h->d_sn, h->d_W1, h->d_W2, dm, seed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return DIMN_OK;
}

extern "C" int dimn_get_step_count(dimn_handle h, int64_t* t) {
    if (!h || !t) return fail(DIMN_ERR_ARG, "null argument");
    *t = h->t;
    return DIMN_OK;
}

// ---- Keras-layout <-> blocked-layout weight I/O (host side; replaces save/load_weights) ----
static int io_weights(dimn_handle h, int k, float* dW1, float* dB1, float* dW2, float* dB2, int bslot,
                      float* W1, float* b1, float* W2, float* b2, bool to_device) {
    const Dims& dm = h->dm;
    const SubnetDev& s = h->sn[k];
    const size_t n1 = (size_t)s.Dp * dm.Hp, n2 = (size_t)dm.Hp * dm.Op;
    std::vector<float> t1(n1, 0.f), t2(n2, 0.f), tb1(dm.Hp, 0.f), tb2(dm.Op, 0.f);
    float* p1 = dW1 + s.w1off;
    float* p2 = dW2 + (size_t)k * n2;
    float* pb1 = dB1 + ((size_t)bslot * h->K + k) * dm.Hp;
    float* pb2 = dB2 + ((size_t)bslot * h->K + k) * dm.Op;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!to_device) {
        HIPCHK(hipMemcpy(t1.data(), p1, n1 * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(t2.data(), p2, n2 * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(tb1.data(), pb1, dm.Hp * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(tb2.data(), pb2, dm.Op * 4, hipMemcpyDeviceToHost));
    }
    for (int d = 0; d < s.D; ++d)
        for (int j = 0; j < dm.H; ++j) {
            float& blk = t1[((size_t)(d >> 4) * dm.Hp + j) * 16 + (d & 15)];
            float& ker = W1[(size_t)d * dm.H + j];
            if (to_device) blk = ker; else ker = blk;
        }
    for (int j = 0; j < dm.H; ++j)
        for (int o = 0; o < dm.O; ++o) {
            float& blk = t2[((size_t)(j >> 4) * dm.OT + (o >> 4)) * 256 + (j & 15) * 16 + (o & 15)];
            float& ker = W2[(size_t)j * dm.O + o];
            if (to_device) blk = ker; else ker = blk;
        }
    for (int j = 0; j < dm.H; ++j) { if (to_device) tb1[j] = b1[j]; else b1[j] = tb1[j]; }
    for (int o = 0; o < dm.O; ++o) { if (to_device) tb2[o] = b2[o]; else b2[o] = tb2[o]; }
    if (to_device) {
        HIPCHK(hipMemcpy(p1, t1.data(), n1 * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(p2, t2.data(), n2 * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(pb1, tb1.data(), dm.Hp * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(pb2, tb2.data(), dm.Op * 4, hipMemcpyHostToDevice));
    }
    return DIMN_OK;
}

static int gen_two_layer(dimn_handle h, const char* who) {
    if (h->gen->L != 1) return fail(DIMN_ERR_ARG, "%s: the model has %d hidden layers -- use dimn_set/get_layer_weights", who, h->gen->L);
    return DIMN_OK;
}
extern "C" int dimn_set_weights(dimn_handle h, int32_t k, const float* W1, const float* b1, const float* W2, const float* b2) {
    if (!h || k < 0 || k >= h->K || !W1 || !b1 || !W2 || !b2) return fail(DIMN_ERR_ARG, "dimn_set_weights: bad argument");
    CHK(use_device(h));
    if (h->gen) {
        CHK(gen_two_layer(h, "dimn_set_weights"));
        CHK(gen_io_layer(h, k, 0, 0, (float*)W1, (float*)b1, true));
        return gen_io_layer(h, k, 1, 0, (float*)W2, (float*)b2, true);
    }
    return io_weights(h, k, h->d_W1, h->d_b1, h->d_W2, h->d_b2, 0, (float*)W1, (float*)b1, (float*)W2, (float*)b2, true);
}
extern "C" int dimn_get_weights(dimn_handle h, int32_t k, float* W1, float* b1, float* W2, float* b2) {
    if (!h || k < 0 || k >= h->K || !W1 || !b1 || !W2 || !b2) return fail(DIMN_ERR_ARG, "dimn_get_weights: bad argument");
    CHK(use_device(h));
    if (h->gen) {
        CHK(gen_two_layer(h, "dimn_get_weights"));
        CHK(gen_io_layer(h, k, 0, 0, W1, b1, false));
        return gen_io_layer(h, k, 1, 0, W2, b2, false);
    }
    return io_weights(h, k, h->d_W1, h->d_b1, h->d_W2, h->d_b2, 0, W1, b1, W2, b2, false);
}
extern "C" int dimn_get_adam_state(dimn_handle h, int32_t k, int32_t which, float* W1, float* b1, float* W2, float* b2) {
    if (!h || k < 0 || k >= h->K || which < 0 || which > 1 || !W1 || !b1 || !W2 || !b2)
        return fail(DIMN_ERR_ARG, "dimn_get_adam_state: bad argument");
    CHK(use_device(h));
    if (h->gen) {
        CHK(gen_two_layer(h, "dimn_get_adam_state"));
        CHK(gen_io_layer(h, k, 0, 1 + which, W1, b1, false));
        return gen_io_layer(h, k, 1, 1 + which, W2, b2, false);
    }
    return io_weights(h, k, which ? h->d_V1 : h->d_M1, h->d_b1, which ? h->d_V2 : h->d_M2, h->d_b2, 1 + which, W1, b1, W2, b2, false);
}

// ---- one optimiser step: [F1] -> RED -> (MFB -> RED2 | MF -> MB) -> B1F1 on the lane's stream ----
static hipEvent_t next_event(dimn_handle h) {
    if (h->ev_used == h->ev.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;   // caller falls back to an untimed launch
        h->ev.push_back(e);
    }
    return h->ev[h->ev_used++];
}

template <int NT>
static void launch_fwd1(dimn_handle h, const dimn_handle_s::Lane& ln, hipStream_t st, const int32_t* rows, int b_act) {
    WITH_XT(h, hipLaunchKernelGGL((k_fwd1<NT, XT>), dim3((unsigned)(ln.w1 - ln.w0)), dim3(256), 0, st, h->d_work + ln.w0, h->d_sn, (const XT*)h->d_X, h->d_W1,
                                  rows, b_act, h->d_P, h->dm));
}
template <int NT2>
static void launch_w1(dimn_handle h, const dimn_handle_s::Lane& ln, hipStream_t stw, const int32_t* rows, int b_act, const int32_t* rows_n, int b_next,
                      AdamP ap, hipEvent_t ev_begin, hipEvent_t ev_end) {
    const dim3 grid((unsigned)(ln.w1 - ln.w0), (unsigned)h->w1_split);
    const Work* wk = h->d_work + ln.w0;
    // ev_begin/ev_end (timed launches only): hipExtLaunchKernelGGL stamps them with the kernel's own begin and end,
    // so the elapsed time is the launch's duration without the dispatch latency an event pair around it would add
#define W1_LAUNCH(KERNEL, THREADS) hipExtLaunchKernelGGL((KERNEL), grid, dim3(THREADS), 0, stw, ev_begin, ev_end, 0, wk, h->d_sn,      \
                                                         (const XT*)h->d_X, h->d_W1, h->d_M1, h->d_V1, rows, b_act, rows_n, b_next, \
                                                         (const float*)h->d_dA, h->d_P, h->dm, ap)
    WITH_XT(h, {
        if (h->w1_waves) {                            // plenty of chunks per CU: one hidden tile per wave, FOUR-set ring, the tiles of a D-slice in w1_split halves (grid.y); H = 300: 10 x 2
            switch (h->w1_waves) {
                case 8:  W1_LAUNCH((k_w1_update_fwd_ring<8, 1, 4, 4, XT>), 512); break;      // two workgroups per CU
                case 9:  W1_LAUNCH((k_w1_update_fwd_ring<9, 1, 4, 1, XT>), 576); break;
                case 10: W1_LAUNCH((k_w1_update_fwd_ring<10, 1, 4, 1, XT>), 640); break;
                case 11: W1_LAUNCH((k_w1_update_fwd_ring<11, 1, 4, 1, XT>), 704); break;
                case 12: W1_LAUNCH((k_w1_update_fwd_ring<12, 1, 4, 1, XT>), 768); break;
                case 13: W1_LAUNCH((k_w1_update_fwd_ring<13, 1, 4, 1, XT>), 832); break;
                case 14: W1_LAUNCH((k_w1_update_fwd_ring<14, 1, 4, 1, XT>), 896); break;
                default: W1_LAUNCH((k_w1_update_fwd_ring<15, 1, 4, 1, XT>), 960); break;
            }
        } else if (h->dm.HT == 16)                      // H = 256: 16 waves x 1 hidden tile, X tiles staged once per workgroup, 3-set register ring
            W1_LAUNCH((k_w1_update_fwd_ring<16, 1, 3, 1, XT>), 1024);
        else if (h->dm.HT == 8 * NT2)
            W1_LAUNCH((k_w1_update_fwd<NT2, true, XT>), 512);
        else
            W1_LAUNCH((k_w1_update_fwd<NT2, false, XT>), 512);
    });
#undef W1_LAUNCH
}
// rows per workgroup of the fp32 forward: 64; 32 where 64-row tiles would not give every CU a workgroup, 16 where they would reach less than a quarter of the CUs
// (k_predict<.., MT = 2 / 1>: more, smaller workgroups; every workgroup reads the sub-net's whole W1 from L2, so the tiles are no smaller than they must be)
static int predict_tile_rows(dimn_handle h, int64_t n_rows, bool validation) {
    if (h->predict_bf16) return DIMN_TB;
    const int64_t wg64 = ((n_rows + DIMN_TB - 1) / DIMN_TB) * h->K;
    // hidden widths whose 64-row activation image takes more than half of a CU's LDS (20 tiles on: ONE workgroup of four waves per CU): 32-row tiles, three
    // workgroups per CU -- the forward over 50k cells 39.8 -> 34.0 ms at hidden 300, 45.4 -> 42.5 at 384 (same box, rocprofv3)
    if (((size_t)DIMN_TB * h->dm.ldp + DIMN_PRED_XS) * sizeof(float) > 80 * 1024 && wg64 >= (int64_t)h->ncu) return 32;
    // the validation pass (a few rounds of workgroups: 2 500 rows x 40 sub-nets = 3.1 rounds of 64-row tiles, the last one an eighth full): 32-row tiles halve
    // what the partial round costs -- 1.72 -> 1.52 ms at 40 sub-nets, 0.97 -> 0.85 at 20 (rocprofv3); the forward over all cells (many rounds) keeps 64 rows
    if (validation && wg64 >= 4 * (int64_t)h->ncu / 2 && wg64 < 8 * (int64_t)h->ncu) return 32;
    return wg64 >= (int64_t)h->ncu ? DIMN_TB : (4 * wg64 >= (int64_t)h->ncu ? 32 : 16);
}
template <int NT>
static void launch_predict_impl(dimn_handle h, const int32_t* rows, int64_t n_rows, float* out, float* loss_part) {
    const int tile_rows = predict_tile_rows(h, n_rows, loss_part != nullptr && out == nullptr);
    const unsigned tiles = (unsigned)((n_rows + tile_rows - 1) / tile_rows);
    if (h->predict_bf16) {                         // bf16 matrix cores (precision bf16): fresh bf16 images of the weights, then the forward
        hipLaunchKernelGGL(k_prep_bf16, dim3(256, (unsigned)h->K), dim3(256), 0, h->stream, h->d_sn, (const float*)h->d_W1, (const float*)h->d_W2,
                           h->d_W1b, h->d_W2t, h->dm);
        if (h->dm.Hp <= 256) {
            // 128 rows per workgroup, 32-deep bf16 matrix instructions, X staged through LDS (dimn_kernels.h); loss slots stay 64-row tiles
            const unsigned tiles128 = (unsigned)((n_rows + DIMN_PB_M - 1) / DIMN_PB_M);
            const int hq = (h->dm.Hp + 31) & ~31;
            const size_t ldsb = std::max<size_t>((size_t)4 * DIMN_PB_XST, (size_t)DIMN_PB_M * (hq + 8) * 2) + 64;
            const bool fast = h->act == 0 && (h->dm.O & 3) == 0;            // (the instantiation without the activation switch and the scalar stores)
#define PB_LAUNCH(F, L)                                                                                                                                          \
            {                                                                                                                                                    \
                (void)hipFuncSetAttribute((const void*)k_predict_bf16<F, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);                             \
                hipLaunchKernelGGL((k_predict_bf16<F, L>), dim3(tiles128, (unsigned)h->K), dim3(256), ldsb, h->stream, h->d_sn, (const bf16_t*)h->d_X,           \
                                   (const bf16_t*)h->d_W1b, (const float*)h->d_b1, (const bf16_t*)h->d_W2t, (const float*)h->d_b2, rows, n_rows, out,            \
                                   (const float*)h->d_Y, h->n, loss_part, (int64_t)tiles, h->dm, h->cfg.loss_binary, h->act, (const bf16_t*)h->d_zero1k);        \
            }
            if (fast) { if (loss_part) PB_LAUNCH(true, true) else PB_LAUNCH(true, false) }
            else { if (loss_part) PB_LAUNCH(false, true) else PB_LAUNCH(false, false) }
#undef PB_LAUNCH
            return;
        }
        const size_t ldsb = (size_t)DIMN_TB * (h->dm.Hp + 4) * 2 + 16;
        hipLaunchKernelGGL(k_predict_bf16_r2<NT>, dim3(tiles, (unsigned)h->K), dim3(256), ldsb, h->stream, h->d_sn, (const bf16_t*)h->d_X, (const bf16_t*)h->d_W1b,
                           (const float*)h->d_b1, (const bf16_t*)h->d_W2t, (const float*)h->d_b2, rows, n_rows, out, (const float*)h->d_Y, h->n, loss_part, h->dm,
                           h->cfg.loss_binary, h->act);
        return;
    }
    // the second-layer operand form of k_predict: a fresh W2T image (21 MB at 40 sub-nets: ~10 us per call)
    hipLaunchKernelGGL(k_prep_w2t, dim3(128, (unsigned)h->K), dim3(256), 0, h->stream, (const float*)h->d_W2, h->d_W2tf, h->dm);
    const size_t lds = ((size_t)tile_rows * h->dm.ldp + 3 * (size_t)tile_rows * 16) * sizeof(float);      // activations + the X staging ring
#define PRED_LAUNCH(MTV)                                                                                                                                     \
            {                                                                                                                                                \
                (void)hipFuncSetAttribute((const void*)k_predict<NT, XT, MTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
                hipLaunchKernelGGL((k_predict<NT, XT, MTV>), dim3(tiles, (unsigned)h->K), dim3(256), lds, h->stream, h->d_sn, (const XT*)h->d_X, h->d_W1,     \
                                   h->d_b1, h->d_W2tf, h->d_b2, rows, n_rows, out, h->d_Y, h->n, loss_part, h->dm, h->cfg.loss_binary, h->act);              \
            }
    WITH_XT(h, {
        if (tile_rows == 16) PRED_LAUNCH(1)
        else if (tile_rows == 32) PRED_LAUNCH(2)
        else {
            (void)hipFuncSetAttribute((const void*)k_predict<NT, XT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((k_predict<NT, XT>), dim3(tiles, (unsigned)h->K), dim3(256), lds, h->stream, h->d_sn, (const XT*)h->d_X, h->d_W1, h->d_b1,
                               h->d_W2tf, h->d_b2, rows, n_rows, out, h->d_Y, h->n, loss_part, h->dm, h->cfg.loss_binary, h->act);
        }
    });
#undef PRED_LAUNCH
}
template <int NT>
static void launch_predict(dimn_handle h, const int32_t* rows, int64_t n_rows, float* out, float* loss_part) {
    launch_predict_impl<NT>(h, rows, n_rows, out, loss_part);
}
#define DISPATCH_NT(fn, ...)                          \
    switch (h->NT) {                                  \
        case 1: fn<1>(__VA_ARGS__); break;            \
        case 2: fn<2>(__VA_ARGS__); break;            \
        case 3: fn<3>(__VA_ARGS__); break;            \
        case 4: fn<4>(__VA_ARGS__); break;            \
        case 5: fn<5>(__VA_ARGS__); break;            \
        default: fn<6>(__VA_ARGS__); break;           \
    }
#define DISPATCH_NT2(fn, ...)                         \
    switch (h->NT2) {                                 \
        case 1: fn<1>(__VA_ARGS__); break;            \
        case 2: fn<2>(__VA_ARGS__); break;            \
        default: fn<3>(__VA_ARGS__); break;           \
    }

// One optimiser step on the handle's stream:
//   [F1 if need_fwd]  ->  RED  ->  MFB -> RED2 (fused second layer) or MF -> MB  ->  B1F1 (W1 Adam + forward
//   partials of the NEXT batch)
// need_fwd: the split-K partials of THIS batch are not in d_P yet (first step of an epoch,
// or the single-step API); d_rows_n/b_next: the next batch (b_next = 0: none).
static int step_launch(dimn_handle h, const dimn_handle_s::Lane& ln, bool timed, const int32_t* d_rows, int b_act, bool need_fwd,
                       const int32_t* d_rows_n, int b_next, const uint8_t* d_mask, uint32_t epoch_key, uint32_t step_key,
                       double* d_loss_acc, int64_t t) {
    const Dims& dm = h->dm;
    AdamP ap;
    const double b1 = h->cfg.beta1, b2 = h->cfg.beta2;
    ap.alpha = (float)((double)h->cfg.learning_rate * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t)));
    ap.omb1 = 1.0f - h->cfg.beta1;
    ap.omb2 = 1.0f - h->cfg.beta2;
    ap.eps = h->cfg.eps;
    const float rate = h->cfg.dropout_rate;
    const float scale = 1.0f / (1.0f - rate);
    const float inv_n = (float)(1.0 / ((double)b_act * h->O));
    const size_t kh = (size_t)h->K * dm.Hp, ko = (size_t)h->K * dm.Op;
    const unsigned nk = (unsigned)(ln.k1 - ln.k0);
    hipStream_t st = ln.stream, stw = ln.stream;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    // one step in eight is timed, on every lane, with HIP events on the lane's own stream: enough samples
    // for a mean, and the event traffic stays out of the way of the other seven
    timed = h->profiling && (step_key % 8u) == 0u;
    if (timed) {
        e0 = next_event(h); e1 = next_event(h); e2 = next_event(h);
        if (!e0 || !e1 || !e2) { timed = false; e0 = e1 = e2 = nullptr; h->ev_used -= h->ev_used % 3; }
    }
    if (timed) {
        (void)hipEventRecord(e0, st);
        // ALGORITHMIC bytes of this W1 launch: 24 B per W1 parameter (read+write
        // of w, m, v) + the batch rows of X ONCE per step (the launch reads X_t and X_{t+1}; the second read is
        // the kernel's own choice, not the algorithm's) + dA, for the lane's sub-nets
        double by = 0;
        for (int k = ln.k0; k < ln.k1; ++k)
            by += 24.0 * h->sn[k].D * h->H + 4.0 * b_act * h->sn[k].D + 4.0 * b_act * h->H;
        h->ev_bytes.push_back(by);
    }

    if (need_fwd) { DISPATCH_NT(launch_fwd1, h, ln, st, d_rows, b_act); }
    hipLaunchKernelGGL(k_reduce_act, dim3((unsigned)ceil_div(DIMN_TB * dm.Hp, 1024), nk), dim3(256), 0, st, h->d_sn, h->d_P, h->d_b1, d_mask,
                       h->d_Dd, dm, b_act, rate, scale, h->cfg.seed, epoch_key, step_key, ln.k0, h->act, h->d_G);
    if (h->mid_fused) {
        // RED -> k_mid_pipe (whole second layer, W2 streamed once) -> RED2 (dD partials -> dA, Adam(b1)); with precision bf16 its GEMMs take bf16 operands
#define LAUNCH_MFP(BFV) hipLaunchKernelGGL(k_mid_pipe<BFV>, dim3(nk * (unsigned)h->mid_slices), dim3(512), (size_t)DIMN_MIDP_LDS_FLOATS * sizeof(float), st, \
                               h->d_midwork + (size_t)ln.k0 * h->mid_slices,                                                                         \
                               h->d_W2, h->d_M2, h->d_V2, h->d_b2, h->d_b2 + ko, h->d_b2 + 2 * ko, h->d_Y, h->n, d_rows, b_act, h->d_Dd, h->d_P2,     \
                               h->d_loss_step, d_loss_acc, dm, ap, inv_n, h->cfg.loss_binary)
        if (h->train_bf16) LAUNCH_MFP(true); else LAUNCH_MFP(false);
#undef LAUNCH_MFP
        hipLaunchKernelGGL(k_reduce_dd, dim3((unsigned)ceil_div(dm.Hp, 64), nk), dim3(1024), 0, st, h->d_midk, h->d_P2, h->d_Dd,
                           h->d_b1, h->d_b1 + kh, h->d_b1 + 2 * kh, h->d_dA, dm, ap, scale, ln.k0, (const float*)h->d_G);
    } else {
    {
        const dim3 grid((unsigned)dm.OS, nk);
        const size_t lds = (size_t)DIMN_TB * dm.ldd * sizeof(float);
        // Every W2 operand of a wave hoisted in front of the LDS staging (HTC = the hidden-tile count at compile time) for 8 .. 16, 18, 20, 22, 24 hidden tiles;
        // other counts take the generic form (operands requested inside the loop).  From 13 tiles on the kernel holds > 128 VGPRs, from 20 on its Dd image > 80 KB
        // of LDS -- one workgroup per CU -- and 8 slices of four output tiles per sub-net would need two rounds of workgroups at 40 sub-nets: six output tiles per workgroup then (12 waves),
        // 40 x 6 = 240 workgroups in ONE round (hidden 300: 28.2 -> 20.5 us per launch; hidden 384 on the generic form: 43.3 us)
        const bool six = dm.HT >= 13 && (int64_t)dm.OS * nk > (int64_t)h->ncu;      // (13 tiles on: > 128 VGPRs or > 80 KB of LDS, one workgroup per CU)
        const dim3 grid6((unsigned)ceil_div(dm.OT, 6), nk);
#define LAUNCH_MF(HTC, NTW, GRID) hipLaunchKernelGGL((k_mid_fwd<HTC, NTW>), GRID, dim3(128 * NTW), lds, st, h->d_W2, h->d_b2, h->d_b2 + ko, h->d_b2 + 2 * ko, h->d_Y, \
                                                     h->n, d_rows, b_act, h->d_Dd, h->d_dZ, h->d_loss_step, d_loss_acc, dm, ap, inv_n, h->cfg.loss_binary, ln.k0)
#define MF_CASE4(HTC) case HTC: LAUNCH_MF(HTC, 4, grid); break;
#define MF_CASE46(HTC) case HTC: if (six) LAUNCH_MF(HTC, 6, grid6); else LAUNCH_MF(HTC, 4, grid); break;
        switch (dm.HT) {
            MF_CASE4(8) MF_CASE4(9) MF_CASE4(10) MF_CASE4(11) MF_CASE4(12)
            MF_CASE46(13) MF_CASE46(14) MF_CASE46(15) MF_CASE46(16) MF_CASE46(18) MF_CASE46(20) MF_CASE46(22) MF_CASE46(24)
            default: LAUNCH_MF(0, 4, grid); break;                                  // generic: 78 VGPRs
        }
#undef MF_CASE46
#undef MF_CASE4
#undef LAUNCH_MF
    }
    // one hidden tile (16 rows of W2) per workgroup; 4 or 8 waves split the output tiles
#define LAUNCH_MB(FULLV, WV) hipLaunchKernelGGL((k_mid_bwd<FULLV, 1, WV>), dim3((unsigned)dm.HT, nk), dim3(WV * 64), 0, st, h->d_Dd, h->d_dZ, \
                                                h->d_W2, h->d_M2, h->d_V2, h->d_b1, h->d_b1 + kh, h->d_b1 + 2 * kh, h->d_dA, dm, ap, scale, h->OTW, ln.k0, (const float*)h->d_G)
    // TWO hidden tiles per workgroup where that still leaves a workgroup per CU (round 4): every workgroup reads the whole dZ block of its
    // sub-net (128 KB from L2), so half as many workgroups halve that traffic -- hidden 300 at 40 sub-nets: step 0.228 -> 0.221 ms, same box
    // (with the three-waves-per-SIMD register cap of the one-tile form it spills: 0.27)
    if (dm.OT == 4 * h->OTW && dm.HT % 2 == 0 && (int64_t)(dm.HT / 2) * nk >= (int64_t)h->ncu)
        hipLaunchKernelGGL((k_mid_bwd<true, 2, 4, 2>), dim3((unsigned)(dm.HT / 2), nk), dim3(256), 0, st, h->d_Dd, h->d_dZ,
                           h->d_W2, h->d_M2, h->d_V2, h->d_b1, h->d_b1 + kh, h->d_b1 + 2 * kh, h->d_dA, dm, ap, scale, h->OTW, ln.k0, (const float*)h->d_G);
    else if (dm.OT == 4 * h->OTW && (int64_t)dm.HT * nk <= 2 * (int64_t)h->ncu)   // few workgroups (a GPU that owns few sub-nets):
        hipLaunchKernelGGL((k_mid_bwd<true, 1, 4, 2>), dim3((unsigned)dm.HT, nk), dim3(256), 0, st, h->d_Dd, h->d_dZ,   // 2 per CU fit anyway -> no register cap, no spills
                           h->d_W2, h->d_M2, h->d_V2, h->d_b1, h->d_b1 + kh, h->d_b1 + 2 * kh, h->d_dA, dm, ap, scale, h->OTW, ln.k0, (const float*)h->d_G);
    else { if (dm.OT == 4 * h->OTW) LAUNCH_MB(true, 4); else LAUNCH_MB(false, 4); }
#undef LAUNCH_MB
    }
    DISPATCH_NT2(launch_w1, h, ln, stw, d_rows, b_act, d_rows_n, b_next, ap, e1, e2);   // e1/e2 (timed steps): the kernel's own begin/end
    HIPCHK(hipGetLastError());
    return DIMN_OK;
}

static int sync_lanes(dimn_handle h);
static int sync_lanes_fwd(dimn_handle h) { return sync_lanes(h); }
static int sync_lanes(dimn_handle h) {
    for (auto& ln : h->lanes) HIPCHK(hipStreamSynchronize(ln.stream));
    return DIMN_OK;
}

static int collect_timers(dimn_handle h) {
    // events are recorded as triples (step begin, before w1 update, step end)
    for (size_t i = 0; i + 3 <= h->ev_used; i += 3) {
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, h->ev[i], h->ev[i + 2]) == hipSuccess &&
            hipEventElapsedTime(&b, h->ev[i + 1], h->ev[i + 2]) == hipSuccess) {
            h->tm_step_ms += a; h->tm_steps++;
            h->tm_w1_ms += b; h->tm_w1++;
            h->tm_w1_bytes += h->ev_bytes[i / 3];
        }
    }
    h->ev_used = 0;
    h->ev_bytes.clear();
    return DIMN_OK;
}

static int ready_for_training(dimn_handle h, const char* who) {
    if (!h->gathered || !h->gathered_targets) return fail(DIMN_ERR_STATE, "%s: call dimn_set_matrix, dimn_set_indices and dimn_gather(with_targets=1) first", who);
    return DIMN_OK;
}

extern "C" int dimn_train_step(dimn_handle h, const int32_t* rows, int32_t b_act, const uint8_t* keep_mask, int32_t epoch_key,
                               int32_t step_key, float* loss_out) {
    if (!h || !rows || b_act < 1 || b_act > h->B) return fail(DIMN_ERR_ARG, "dimn_train_step: bad batch (b_act must be 1..batch_size)");
    CHK(ready_for_training(h, "dimn_train_step"));
    for (int b = 0; b < b_act; ++b) if (rows[b] < 0 || rows[b] >= h->n) return fail(DIMN_ERR_ARG, "dimn_train_step: row %d out of range", rows[b]);
    CHK(use_device(h));
    const Dims& dm = h->dm;
    if (h->gen) {
        if (keep_mask) return fail(DIMN_ERR_UNSUP, "dimn_train_step: an injected keep mask is only defined for the one-hidden-layer kernels");
        if (h->rows_step_cap < b_act) {
            HIPCHK(hipStreamSynchronize(h->stream));
            DEV_FREE(h->d_rows_step);
            CHK(dev_alloc(&h->d_rows_step, (size_t)b_act));
            h->rows_step_cap = b_act;
        }
        HIPCHK(hipMemcpyAsync(h->d_rows_step, rows, (size_t)b_act * 4, hipMemcpyHostToDevice, h->stream));
        CHK(gen_zero_loss(h));
        CHK(gen_train_step(h, h->d_rows_step, b_act, (uint32_t)epoch_key, (uint32_t)step_key, h->t + 1));
        h->t += 1;
        std::vector<double> ls;
        CHK(gen_read_loss(h, ls));
        if (loss_out) for (int k = 0; k < h->K; ++k) loss_out[k] = (float)(ls[(size_t)k] / ((double)b_act * h->O));
        return DIMN_OK;
    }
    HIPCHK(hipMemcpyAsync(h->d_rows_step, rows, (size_t)b_act * 4, hipMemcpyHostToDevice, h->stream));
    const uint8_t* dmask = nullptr;
    std::vector<uint8_t> padded;
    if (keep_mask) {
        padded.assign((size_t)h->K * DIMN_TB * dm.Hp, 0);
        for (int k = 0; k < h->K; ++k)
            for (int b = 0; b < b_act; ++b)
                memcpy(&padded[((size_t)k * DIMN_TB + b) * dm.Hp], &keep_mask[((size_t)k * b_act + b) * h->H], (size_t)h->H);
        HIPCHK(hipMemcpyAsync(h->d_mask, padded.data(), padded.size(), hipMemcpyHostToDevice, h->stream));
        dmask = h->d_mask;
    }
    HIPCHK(hipStreamSynchronize(h->stream));     // rows / mask uploads done before any lane starts
    for (size_t l = 0; l < h->lanes.size(); ++l)
        CHK(step_launch(h, h->lanes[l], l == 0, h->d_rows_step, b_act, true, nullptr, 0, dmask, (uint32_t)epoch_key, (uint32_t)step_key,
                        nullptr, h->t + 1));
    h->t += 1;
    CHK(sync_lanes(h));
    if (h->profiling) collect_timers(h);
    if (loss_out) {
        std::vector<float> ls((size_t)h->K * dm.LS);
        HIPCHK(hipMemcpy(ls.data(), h->d_loss_step, ls.size() * 4, hipMemcpyDeviceToHost));
        for (int k = 0; k < h->K; ++k) {
            double s = 0;
            for (int j = 0; j < dm.LS; ++j) s += ls[(size_t)k * dm.LS + j];
            loss_out[k] = (float)(s / ((double)b_act * h->O));
        }
    }
    return DIMN_OK;
}

extern "C" int dimn_epoch_permutation(uint64_t seed, int32_t epoch, int64_t n, int32_t* perm_out) {
    if (n < 0 || (n > 0 && !perm_out)) return fail(DIMN_ERR_ARG, "dimn_epoch_permutation: bad argument");
    dimn_fill_permutation(seed, (uint32_t)epoch, n, perm_out);
    return DIMN_OK;
}

// One epoch as ONE persistent launch with the optimiser state in registers (dimn_resident.h); d_epoch_rows is uploaded.
// The training rows of one epoch in the order the epoch visits them: Xe / Ye row `pos` of sub-net k = X / Y row rows[pos] (the arenas'
// own layout, the first n_tr rows of every sub-net used).  One wave per row and sub-net, 16-byte pieces.  For LARGE arenas: the
// resident kernel gathers 64 rows per step at random, and beyond ~16 GB those rows are beyond the TLB's reach -- every request of
// its tile loop then waits for a page walk (configs[4]'s share at 1M cells: 45 us per step against 38 at 200k cells).  Copying the
// epoch's rows once (2 x the arena at HBM rate: ~25 ms for 55 GB) makes every step read 64 CONSECUTIVE rows.
template <typename XT>
__global__ __launch_bounds__(256) void k_res_epoch_rows(const SubnetDev* __restrict__ sn, const XT* __restrict__ X, const float* __restrict__ Y, const int32_t* __restrict__ rows,
                                                        int64_t n_tr, int64_t n_cells, int Op, XT* __restrict__ Xe, float* __restrict__ Ye) {
    const int k = blockIdx.y;
    const SubnetDev s = sn[k];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int xp = (int)((int64_t)s.Dp * (int64_t)sizeof(XT) / 16), yp = Op / 4;
    for (int64_t pos = (int64_t)blockIdx.x * 4 + wave; pos < n_tr; pos += (int64_t)gridDim.x * 4) {
        const int64_t r = rows[pos];
        const uint4* xs = (const uint4*)(X + s.xoff + r * s.Dp);
        uint4* xd = (uint4*)(Xe + s.xoff + pos * s.Dp);
        for (int i = lane; i < xp; i += 64) xd[i] = xs[i];
        const uint4* ys = (const uint4*)(Y + ((int64_t)k * n_cells + r) * Op);
        uint4* yd = (uint4*)(Ye + ((int64_t)k * n_cells + pos) * Op);
        for (int i = lane; i < yp; i += 64) yd[i] = ys[i];
    }
}
__global__ __launch_bounds__(256) void k_res_iota(int32_t* __restrict__ v, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}
static int train_epoch_resident(dimn_handle h, int32_t epoch, double* train_loss) {
    const Dims& dm = h->dm;
    const int steps = (int)((h->n_tr + h->B - 1) / h->B);
    if (h->res_alpha_cap < steps) {
        HIPCHK(hipStreamSynchronize(h->stream));
        DEV_FREE(h->d_res_alpha);
        CHK(dev_alloc(&h->d_res_alpha, (size_t)steps));
        DEV_FREE(h->d_res_b1);                                   // the keep words of a whole epoch: [steps][K][512]
        CHK(dev_alloc(&h->d_res_b1, (size_t)steps * h->K * 512));
        h->res_alpha_cap = steps;
    }
    std::vector<float> alpha((size_t)steps);
    const double b1 = h->cfg.beta1, b2 = h->cfg.beta2;
    for (int t = 0; t < steps; ++t) {
        const double tt = (double)(h->t + 1 + t);
        alpha[(size_t)t] = (float)((double)h->cfg.learning_rate * sqrt(1.0 - pow(b2, tt)) / (1.0 - pow(b1, tt)));
    }
    HIPCHK(hipMemcpyAsync(h->d_res_alpha, alpha.data(), alpha.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->d_res_flags, 0, ((size_t)2 * h->K + 1) * sizeof(unsigned), h->stream));
    HIPCHK(hipMemsetAsync(h->d_res_loss, 0, (size_t)h->K * dm.OT * sizeof(double), h->stream));
    const size_t kh = (size_t)h->K * dm.Hp, ko = (size_t)h->K * dm.Op;
    ResParams p;
    p.sn = h->d_sn; p.X = h->d_X; p.Y = h->d_Y; p.n_cells = h->n;
    p.W1 = h->d_W1; p.M1 = h->d_M1; p.V1 = h->d_V1; p.W2 = h->d_W2; p.M2 = h->d_M2; p.V2 = h->d_V2;
    p.b1w = h->d_b1; p.b1m = h->d_b1 + kh; p.b1v = h->d_b1 + 2 * kh;
    p.b2w = h->d_b2; p.b2m = h->d_b2 + ko; p.b2v = h->d_b2 + 2 * ko;
    p.rows = h->d_epoch_rows; p.n_tr = (int32_t)h->n_tr; p.B = h->B; p.steps = steps;
    p.alpha = h->d_res_alpha; p.Ppart = h->d_res_P; p.Dpart = h->d_res_D; p.DdT = h->d_res_T; p.dAT = h->d_res_A; p.maskw = (unsigned*)h->d_res_b1;
    p.flags = h->d_res_flags; p.loss = h->d_res_loss; p.dm = dm;
    p.omb1 = 1.0f - h->cfg.beta1; p.omb2 = 1.0f - h->cfg.beta2; p.eps = h->cfg.eps;
    p.rate = h->cfg.dropout_rate; p.scale = 1.0f / (1.0f - h->cfg.dropout_rate);
    p.seed = h->cfg.seed; p.epoch = (uint32_t)epoch; p.G = h->res_G; p.S1 = h->res_S1; p.loss_binary = h->cfg.loss_binary;
    const size_t lds = (size_t)DIMN_RES_LDS_FLOATS * sizeof(float);
    // The workgroups of a launch wait for each other, so all of them must be resident at once.  (1) the grid is checked against
    // the kernel's occupancy on this device, and the launch is a COOPERATIVE one (the runtime refuses it unless the whole grid
    // fits the device; measured: no cost, 26.09 vs 26.00 us per step); (2) the state the launch will overwrite is snapshotted first, so that a
    // launch that still times out (a GPU shared with another process) is undone and the epoch re-run on the streaming kernels.
    const size_t w2n = (size_t)h->K * dm.Hp * dm.Op, nb1 = (size_t)3 * h->K * dm.Hp, nb2 = (size_t)3 * h->K * dm.Op;
    const size_t snap_floats = 3 * (size_t)h->w1_total + 3 * w2n + nb1 + nb2;
    if (!h->d_res_snap) CHK(dev_alloc(&h->d_res_snap, snap_floats));
    {
        float* d = h->d_res_snap;
        const float* src[8] = {h->d_W1, h->d_M1, h->d_V1, h->d_W2, h->d_M2, h->d_V2, h->d_b1, h->d_b2};
        const size_t cnt[8] = {(size_t)h->w1_total, (size_t)h->w1_total, (size_t)h->w1_total, w2n, w2n, w2n, nb1, nb2};
        for (int i = 0; i < 8; ++i) { HIPCHK(hipMemcpyAsync(d, src[i], cnt[i] * 4, hipMemcpyDeviceToDevice, h->stream)); d += cnt[i]; }
    }
    const bool coop = true;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling) { e0 = next_event(h); e1 = next_event(h); next_event(h); }
    if (e0 && e1) (void)hipEventRecord(e0, h->stream);
    // sentinel protocol: every exchange slot starts "not written" (all-ones words); the keep words of the epoch come from their own kernel
    HIPCHK(hipMemsetAsync(h->d_res_P, 0xff, (size_t)DIMN_RES_SLOTS * h->K * h->res_G * 4096, h->stream));
    HIPCHK(hipMemsetAsync(h->d_res_D, 0xff, (size_t)DIMN_RES_SLOTS * h->K * dm.OT * 65536, h->stream));
    HIPCHK(hipMemsetAsync(h->d_res_T, 0xff, (size_t)DIMN_RES_SLOTS * h->K * 65536, h->stream));
    HIPCHK(hipMemsetAsync(h->d_res_A, 0xff, (size_t)DIMN_RES_SLOTS * h->K * 65536, h->stream));
    if (h->cfg.dropout_rate > 0.f)
        hipLaunchKernelGGL(k_res_masks, dim3((unsigned)(steps * h->K)), dim3(512), 0, h->stream, h->d_sn, (unsigned*)h->d_res_b1, h->K, h->H,
                           (uint64_t)h->cfg.seed, (uint32_t)epoch, h->cfg.dropout_rate);
#define RES_LAUNCH(T, S)                                                                                                           \
    do {                                                                                                                         \
        if (h->res_bf16) { if (split) RES_LAUNCH_X(T, S, bf16_t, true, true) else RES_LAUNCH_X(T, S, bf16_t, true, false) }       \
        else if (split) WITH_XT(h, RES_LAUNCH_X(T, S, XT, false, true));                                                         \
        else WITH_XT(h, RES_LAUNCH_X(T, S, XT, false, false));                                                                   \
    } while (0)
#define RES_LAUNCH_X(T, S, XT, BFV, SPV)                                                                                           \
    {                                                                                                                            \
        const void* fn_ = (const void*)k_epoch_resident<T, S, XT, BFV, SPV>;                                                     \
        (void)hipFuncSetAttribute(fn_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
        if (!h->res_checked) {                                                                                                   \
            int per_cu_ = 0;                                                                                                     \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_, fn_, DIMN_RES_THREADS, lds) != hipSuccess) per_cu_ = 0;   \
            if ((int64_t)per_cu_ * h->ncu < (int64_t)h->res_Kg * h->res_G) not_resident = true;                                  \
        }                                                                                                                        \
        if (!not_resident) {                                                                                                     \
            void* args_[1] = {(void*)&p};                                                                                        \
            if (coop) { if (hipLaunchCooperativeKernel(fn_, grid, dim3(DIMN_RES_THREADS), args_, (unsigned)lds, h->stream) != hipSuccess) not_resident = true; } \
            else hipLaunchKernelGGL((k_epoch_resident<T, S, XT, BFV, SPV>), grid, dim3(DIMN_RES_THREADS), lds, h->stream, p);    \
        }                                                                                                                        \
    }
    // large arenas: the epoch's rows copied into visiting order first (k_res_epoch_rows), the kernel then walks rows 0 .. n_tr-1
    bool erows = false;
    {
        erows = ((double)h->x_total * XBYTES(h) + (double)h->y_total * 4.0) > 16.0 * 1073741824.0;
        if (const int v = res_test_knob("erows", -1); v >= 0) erows = v != 0;
        if (erows && h->res_erows_off) erows = false;
        if (erows) {     // (no room for the copies: the kernel gathers its rows where they are, as for small arenas)
            if (!h->d_res_Xe && dev_malloc_bytes((void**)&h->d_res_Xe, std::max<size_t>(1, (size_t)h->x_total * XBYTES(h))) != hipSuccess) { h->d_res_Xe = nullptr; erows = false; }
            if (erows && !h->d_res_Ye && dev_malloc_bytes((void**)&h->d_res_Ye, std::max<size_t>(1, (size_t)h->y_total * 4)) != hipSuccess) { h->d_res_Ye = nullptr; erows = false; }
            if (!erows) { (void)hipGetLastError(); DEV_FREE(h->d_res_Xe); DEV_FREE(h->d_res_Ye); h->res_erows_off = true; }
        }
        if (erows) {
            if (!h->d_res_iota || h->res_iota_n != h->n_tr) {
                DEV_FREE(h->d_res_iota);
                HIPCHK(dev_malloc_bytes((void**)&h->d_res_iota, std::max<size_t>(1, (size_t)h->n_tr * 4)));
                h->res_iota_n = h->n_tr;
                hipLaunchKernelGGL(k_res_iota, dim3((unsigned)((h->n_tr + 255) / 256)), dim3(256), 0, h->stream, h->d_res_iota, (int64_t)h->n_tr);
            }
            const unsigned gx = (unsigned)std::min<int64_t>((h->n_tr + 3) / 4, 4096);
            WITH_XT(h, hipLaunchKernelGGL((k_res_epoch_rows<XT>), dim3(gx, (unsigned)h->K), dim3(256), 0, h->stream, h->d_sn, (const XT*)h->d_X, h->d_Y, h->d_epoch_rows,
                                          (int64_t)h->n_tr, (int64_t)h->n, dm.Op, (XT*)h->d_res_Xe, h->d_res_Ye));
            HIPCHK(hipGetLastError());
            p.X = h->d_res_Xe; p.Y = h->d_res_Ye; p.rows = h->d_res_iota;
        }
    }
    bool not_resident = false;
    // tile order of the kernel's loop: rows of a large arena are far away (TLB reach), so their requests get two tile-times of lead
    // (only when the rows stay where they are: with the epoch-ordered copies the alternating order is the better one again, 39.1 vs 39.8 us)
    bool split = !erows && (double)h->x_total * XBYTES(h) > 16.0 * 1073741824.0;
    if (const int v = res_test_knob("split", -1); v >= 0) split = v != 0;
    if (getenv("DIMN_TRACE") && atoi(getenv("DIMN_TRACE")) && epoch == 0) fprintf(stderr, "[dimn] resident epoch: arena %.1f GB, epoch-ordered rows %d, split tile order %d\n", ((double)h->x_total * XBYTES(h) + (double)h->y_total * 4.0) / 1073741824.0, (int)erows, (int)split);
    // one launch per group of res_Kg sub-nets (all of them when they fit at once), one after the other on the stream
    for (int k0 = 0; k0 < h->K && !not_resident; k0 += h->res_Kg) {
        p.k0 = k0;
        const dim3 grid((unsigned)(std::min(h->res_Kg, h->K - k0) * h->res_G));
        // <7, 3>: one rank of the 8-GPU job (5 sub-nets of D ~ 2400 on 256 CUs); the others take the D-split count at run time
        if (h->res_T1 == 7 && h->res_S1 == 3) RES_LAUNCH(7, 3);
        else if (h->res_T1 == 2) RES_LAUNCH(2, 0);
        else if (h->res_T1 == 4) RES_LAUNCH(4, 0);
        else RES_LAUNCH(7, 0);
    }
#undef RES_LAUNCH
#undef RES_LAUNCH_X
    h->res_checked = 1;
    (void)hipGetLastError();
    if (e0 && e1) (void)hipEventRecord(e1, h->stream);
    std::vector<unsigned> flags((size_t)2 * h->K + 1);
    std::vector<double> acc((size_t)h->K * dm.OT);
    HIPCHK(hipMemcpyAsync(flags.data(), h->d_res_flags, flags.size() * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(acc.data(), h->d_res_loss, acc.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (e0 && e1) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && !not_resident && flags[(size_t)2 * h->K] == 0) { h->tm_res_ms += ms; h->tm_res_steps += steps; }
        h->ev_used = 0; h->ev_bytes.clear();
    }
    if (res_test_knob("abort", 0) == (int)epoch + 1) flags[(size_t)2 * h->K] = 1;   // tests: pretend epoch N-1 timed out
    if (not_resident || flags[(size_t)2 * h->K] != 0) {
        // not all workgroups of a launch could be resident together (or one was lost to another tenant of this GPU): the state
        // goes back to what it was before this epoch, the handle stops using the resident kernel, and the caller runs the epoch
        // on the streaming kernels -- same numbers to fp32 rounding, no error for the user
        fprintf(stderr, "libdimn: the register-resident epoch kernel %s; epoch %d re-runs on the streaming kernels and this handle keeps to them\n",
                not_resident ? "cannot have all its workgroups resident on this device" : "timed out waiting for a workgroup (another process on this GPU?)", (int)epoch);
        float* d = h->d_res_snap;
        float* dst[8] = {h->d_W1, h->d_M1, h->d_V1, h->d_W2, h->d_M2, h->d_V2, h->d_b1, h->d_b2};
        const size_t cnt[8] = {(size_t)h->w1_total, (size_t)h->w1_total, (size_t)h->w1_total, w2n, w2n, w2n, nb1, nb2};
        for (int i = 0; i < 8; ++i) { HIPCHK(hipMemcpyAsync(dst[i], d, cnt[i] * 4, hipMemcpyDeviceToDevice, h->stream)); d += cnt[i]; }
        HIPCHK(hipStreamSynchronize(h->stream));
        h->res_G = 0;
        return 1;                                              // > 0: "fell back", not an error
    }
    h->t += steps;
    if (train_loss)
        for (int k = 0; k < h->K; ++k) {
            double s = 0;
            for (int o = 0; o < dm.OT; ++o) s += acc[(size_t)k * dm.OT + o];
            train_loss[k] = s / ((double)h->O * (double)h->n_tr);
        }
    return DIMN_OK;
}

extern "C" int dimn_train_epoch(dimn_handle h, int32_t epoch, const int32_t* perm, double* train_loss) {
    if (!h) return fail(DIMN_ERR_ARG, "null handle");
    CHK(ready_for_training(h, "dimn_train_epoch"));
    if (h->n_tr < 1) return fail(DIMN_ERR_STATE, "dimn_train_epoch: call dimn_set_split first");
    CHK(use_device(h));
    const Dims& dm = h->dm;
    std::vector<int32_t> p;
    std::thread next;                      // makes epoch + 1's permutation while this epoch is enqueued and runs; joined before the function returns
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{next};
    if (!perm) {
        if (h->next_perm_epoch == (int64_t)epoch && (int64_t)h->next_perm.size() == h->n_tr) p.swap(h->next_perm);
        else {
            p.resize((size_t)h->n_tr);
            dimn_fill_permutation(h->cfg.seed, (uint32_t)epoch, h->n_tr, p.data());
        }
        perm = p.data();
        h->next_perm_epoch = -1;
        next = std::thread([h, epoch, n = h->n_tr, seed = h->cfg.seed] {
            h->next_perm.resize((size_t)n);
            dimn_fill_permutation(seed, (uint32_t)epoch + 1u, n, h->next_perm.data());
            h->next_perm_epoch = (int64_t)epoch + 1;
        });
    }
    std::vector<int32_t> rows((size_t)h->n_tr);
    for (int64_t i = 0; i < h->n_tr; ++i) {
        if (perm[i] < 0 || perm[i] >= h->n_tr) return fail(DIMN_ERR_ARG, "dimn_train_epoch: perm[%lld] out of range", (long long)i);
        rows[(size_t)i] = h->train_rows[(size_t)perm[i]];
        if (rows[(size_t)i] < 0 || rows[(size_t)i] >= h->n) return fail(DIMN_ERR_ARG, "dimn_train_epoch: train row %d outside the matrix", rows[(size_t)i]);
    }
    CHK(sync_lanes(h));
    HIPCHK(hipMemcpyAsync(h->d_epoch_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, h->stream));
    if (h->gen) {
        CHK(gen_zero_loss(h));
        int step = 0;
        for (int64_t i0 = 0; i0 < h->n_tr; i0 += h->B, ++step) {
            const int b_act = (int)std::min<int64_t>(h->B, h->n_tr - i0);
            CHK(gen_train_step(h, h->d_epoch_rows + i0, b_act, (uint32_t)epoch, (uint32_t)step, h->t + 1));
            h->t += 1;
        }
        std::vector<double> ls;
        CHK(gen_read_loss(h, ls));
        if (train_loss) for (int k = 0; k < h->K; ++k) train_loss[k] = ls[(size_t)k] / ((double)h->O * (double)h->n_tr);
        return DIMN_OK;
    }
    if (h->res_G && h->act == DIMN_ACT_RELU && ((h->n_tr + h->B - 1) / h->B) * h->K * 2048 < (1ll << 31)) {    // (keep words of the epoch: 32-bit offsets)
        const int rc = train_epoch_resident(h, epoch, train_loss);
        if (rc <= 0) return rc;                                // done, or an error; 1: the launch was undone -> the streaming kernels below
    }
    HIPCHK(hipMemsetAsync(h->d_loss_acc, 0, (size_t)h->K * dm.LS * sizeof(double), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));     // every lane reads the row list and accumulates into d_loss_acc
    // d_loss_acc accumulates sum(w e^2) per step; the per-step means are weighted by b_act,
    // i.e. sum_steps (sum/(b_act*O))*b_act / n_tr = total / (O*n_tr)
    int step = 0;
    for (int64_t i0 = 0; i0 < h->n_tr; i0 += h->B, ++step) {
        const int b_act = (int)std::min<int64_t>(h->B, h->n_tr - i0);
        const int64_t i1 = i0 + h->B;
        const int b_next = i1 < h->n_tr ? (int)std::min<int64_t>(h->B, h->n_tr - i1) : 0;
        // the lanes (disjoint sub-net groups) are independent chains: issuing them to separate streams lets
        // one group's small latency-bound kernels run under the other group's HBM-bound weight update
        for (size_t l = 0; l < h->lanes.size(); ++l)
            CHK(step_launch(h, h->lanes[l], l == 0, h->d_epoch_rows + i0, b_act, step == 0, b_next ? h->d_epoch_rows + i1 : nullptr,
                            b_next, nullptr, (uint32_t)epoch, (uint32_t)step, h->d_loss_acc, h->t + 1));
        h->t += 1;
    }
    CHK(sync_lanes(h));
    if (h->profiling) collect_timers(h);
    if (train_loss) {
        std::vector<double> acc((size_t)h->K * dm.LS);
        HIPCHK(hipMemcpy(acc.data(), h->d_loss_acc, acc.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int k = 0; k < h->K; ++k) {
            double s = 0;
            for (int j = 0; j < dm.LS; ++j) s += acc[(size_t)k * dm.LS + j];
            train_loss[k] = s / ((double)h->O * (double)h->n_tr);
        }
    }
    return DIMN_OK;
}

extern "C" int dimn_val_loss(dimn_handle h, double* val_loss) {
    if (!h || !val_loss) return fail(DIMN_ERR_ARG, "dimn_val_loss: null argument");
    CHK(ready_for_training(h, "dimn_val_loss"));
    if (h->n_val < 1) return fail(DIMN_ERR_STATE, "dimn_val_loss: no validation rows (dimn_set_split)");
    for (int32_t r : h->val_rows) if (r < 0 || r >= h->n) return fail(DIMN_ERR_ARG, "dimn_val_loss: validation row %d outside the matrix", r);
    CHK(use_device(h));
    if (h->gen) return gen_val_loss(h, val_loss);
    const int64_t tile_rows = predict_tile_rows(h, h->n_val, true);
    const int64_t tiles = (h->n_val + tile_rows - 1) / tile_rows;
    if (h->loss_part_cap < tiles * h->K) {
        HIPCHK(hipStreamSynchronize(h->stream));
        DEV_FREE(h->d_loss_part);
        CHK(dev_alloc(&h->d_loss_part, (size_t)(tiles * h->K)));
        h->loss_part_cap = tiles * h->K;
    }
    DISPATCH_NT(launch_predict, h, h->d_val_rows, h->n_val, (float*)nullptr, h->d_loss_part);
    HIPCHK(hipGetLastError());
    std::vector<float> part((size_t)(tiles * h->K));
    HIPCHK(hipMemcpyAsync(part.data(), h->d_loss_part, part.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int k = 0; k < h->K; ++k) {
        double s = 0;
        for (int64_t i = 0; i < tiles; ++i) s += part[(size_t)(k * tiles + i)];
        val_loss[k] = s / ((double)h->n_val * h->O);
    }
    return DIMN_OK;
}

extern "C" int dimn_fit(dimn_handle h, int32_t max_epochs, int32_t patience, double* loss_hist, double* val_hist, int32_t* epochs_run) {
    if (!h || max_epochs < 0) return fail(DIMN_ERR_ARG, "dimn_fit: bad argument");
    std::vector<double> tl((size_t)h->K), vl((size_t)h->K);
    double best = INFINITY;
    int wait = 0, e = 0;
    for (e = 0; e < max_epochs; ++e) {
        CHK(dimn_train_epoch(h, e, nullptr, tl.data()));
        CHK(dimn_val_loss(h, vl.data()));
        double st = 0, sv = 0;
        for (int k = 0; k < h->K; ++k) { st += tl[k]; sv += vl[k]; }
        if (loss_hist) loss_hist[e] = st;
        if (val_hist) val_hist[e] = sv;
        // EarlyStopping(monitor='val_loss', patience): strict <, min_delta 0 (multinet.py:242-243)
        if (sv < best) { best = sv; wait = 0; }
        else if (++wait >= patience) { ++e; break; }
    }
    if (epochs_run) *epochs_run = e;
    return DIMN_OK;
}

extern "C" int dimn_predict_device(dimn_handle h, const int32_t* rows, int64_t n_rows, void** dev_out) {
    if (!h || n_rows < 0) return fail(DIMN_ERR_ARG, "dimn_predict: bad argument");
    if (!h->gathered) return fail(DIMN_ERR_STATE, "dimn_predict: call dimn_set_matrix, dimn_set_indices and dimn_gather first");
    if (!rows && n_rows > h->n) return fail(DIMN_ERR_ARG, "dimn_predict: n_rows exceeds the matrix");
    CHK(use_device(h));
    // from here on the previous result is gone: no exit below may leave out_rows / pred_ev_rows describing it while d_out is a fresh,
    // unwritten block (a following dimn_impute_finish* with the old n_rows would pass its state check and read it)
    h->out_rows = 0;
    h->pred_ev_rows.clear();
    const int64_t need = n_rows * h->K * h->O;
    if (h->out_cap < need) {
        HIPCHK(hipStreamSynchronize(h->stream));
        DEV_FREE(h->d_out);
        h->out_cap = 0;
        CHK(dev_alloc(&h->d_out, (size_t)need));
        h->out_cap = need;
    }
    const int32_t* drows = nullptr;
    if (rows) {
        for (int64_t i = 0; i < n_rows; ++i) if (rows[i] < 0 || rows[i] >= h->n) return fail(DIMN_ERR_ARG, "dimn_predict: row out of range");
        if (h->pred_rows_cap < n_rows) {
            HIPCHK(hipStreamSynchronize(h->stream));
            DEV_FREE(h->d_pred_rows);
            CHK(dev_alloc(&h->d_pred_rows, (size_t)n_rows));
            h->pred_rows_cap = n_rows;
        }
        HIPCHK(hipMemcpyAsync(h->d_pred_rows, rows, (size_t)n_rows * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        drows = h->d_pred_rows;
    }
    if (n_rows > 0 && h->gen) {
        CHK(gen_predict(h, drows, n_rows));
    } else if (n_rows > 0 && !rows && n_rows >= 8192 && !h->predict_bf16) {
        // all cells, in order: eight row chunks, an event behind each, so that the epilogue (dimn_impute_finish*) starts on the first rows
        // while the last are still computed: ~23 of the fp32 forward's 26 ms at 50k cells x 40 sub-nets.  (Not for the bf16 forward: 4 ms
        // as one launch, 4.4 ms as eight -- rocprofv3, round 5 -- with nothing worth hiding.)
        if (h->pred_iota_n < n_rows) {
            HIPCHK(hipStreamSynchronize(h->stream));
            DEV_FREE(h->d_pred_iota);
            CHK(dev_alloc(&h->d_pred_iota, (size_t)n_rows));
            h->pred_iota_n = n_rows;
            hipLaunchKernelGGL(k_res_iota, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, h->stream, h->d_pred_iota, n_rows);
        }
        const int64_t per = ((n_rows + 7) / 8 + 127) / 128 * 128;
        size_t c = 0;
        for (int64_t r0 = 0; r0 < n_rows; r0 += per, ++c) {
            const int64_t nr = std::min(per, n_rows - r0);
            DISPATCH_NT(launch_predict, h, (const int32_t*)(h->d_pred_iota + r0), nr, h->d_out + r0 * h->K * h->O, (float*)nullptr);
            HIPCHK(hipGetLastError());
            if (h->pred_ev.size() <= c) {
                hipEvent_t e = nullptr;
                HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                h->pred_ev.push_back(e);
            }
            HIPCHK(hipEventRecord(h->pred_ev[c], h->stream));
            h->pred_ev_rows.push_back(r0 + nr);
        }
    } else if (n_rows > 0) {
        DISPATCH_NT(launch_predict, h, drows, n_rows, h->d_out, (float*)nullptr);
        HIPCHK(hipGetLastError());
    }
    h->out_rows = n_rows;
    if (dev_out) *dev_out = h->d_out;
    return DIMN_OK;
}

extern "C" int dimn_predict(dimn_handle h, const int32_t* rows, int64_t n_rows, float* out) {
    if (!out && n_rows > 0) return fail(DIMN_ERR_ARG, "dimn_predict: null output");
    CHK(dimn_predict_device(h, rows, n_rows, nullptr));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n_rows > 0) HIPCHK(hipMemcpy(out, h->d_out, (size_t)n_rows * h->K * h->O * 4, hipMemcpyDeviceToHost));
    return DIMN_OK;
}

// Held-out metrics of fit() (multinet.py:251-262) on the device: forward over the validation rows, then the seven sums
// over the positive truth entries.  out7 = count, Sx, Sy, Sxx, Syy, Sxy, S(x-y)^2 (x = truth, y = prediction).
extern "C" int dimn_val_metrics(dimn_handle h, double* out7) {
    if (!h || !out7) return fail(DIMN_ERR_ARG, "dimn_val_metrics: null argument");
    CHK(ready_for_training(h, "dimn_val_metrics"));
    if (h->n_val < 1) return fail(DIMN_ERR_STATE, "dimn_val_metrics: no validation rows (dimn_set_split)");
    CHK(dimn_predict_device(h, h->val_rows.data(), h->n_val, nullptr));
    double* d = nullptr;
    CHK(dev_alloc(&d, 8));
    int rc = DIMN_OK;
    if (hipMemsetAsync(d, 0, 8 * sizeof(double), h->stream) != hipSuccess) rc = fail(DIMN_ERR_HIP, "dimn_val_metrics: memset failed");
    if (rc == DIMN_OK) {
        hipLaunchKernelGGL(k_val_metrics, dim3((unsigned)std::min<int64_t>(h->n_val, 2048)), dim3(256), 0, h->stream, h->d_out, h->d_Y, h->d_pred_rows,
                           h->n_val, h->n, h->dm, d);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out7, d, 7 * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess)
            rc = fail(DIMN_ERR_HIP, "dimn_val_metrics: kernel or copy failed");
    }
    (void)dev_free_any(d);
    return rc;
}

// the event behind the forward chunk that covers rows [.., row_end) of the last dimn_predict_device (chunked form)
static hipEvent_t pred_event_for(dimn_handle h, int64_t row_end) {
    size_t c = 0;
    while (c + 1 < h->pred_ev_rows.size() && h->pred_ev_rows[c] < row_end) ++c;
    return h->pred_ev[c];
}

// ---- next row: predict() post-processing (multinet.py:282-305) as a device epilogue ----------
// Row blocks of raw stream in, the finished float64 frame streams out, both through two pinned bounce buffers per
// direction so that the PCIe copies of one block overlap the kernel and the host copies of its neighbours.
extern "C" int dimn_impute_finish(dimn_handle h, const double* raw, int64_t n_rows, int64_t g, const int32_t* gene_off,
                                  const int32_t* gene_slot, int32_t policy, double ceiling, int32_t from_gathered, double* out) {
    if (!h || !gene_off || !gene_slot || !out || n_rows < 0 || g < 1 || policy < 0 || policy > 2)
        return fail(DIMN_ERR_ARG, "dimn_impute_finish: bad argument");
    const bool resident = raw == nullptr;                  // raw == NULL: the observed counts are the resident matrix this handle was given (dimn_set_matrix_counts)
    if (resident && (!h->counts || h->counts->n != n_rows || h->counts->g != g))
        return fail(DIMN_ERR_STATE, "dimn_impute_finish: raw == NULL needs the count matrix of dimn_set_matrix_counts over the same %lld x %lld cells", (long long)n_rows, (long long)g);
    const int64_t S = gene_off[g];
    const float* pred = from_gathered ? h->d_full : h->d_out;
    if (!pred || h->out_rows != n_rows) return fail(DIMN_ERR_STATE, "dimn_impute_finish: run dimn_predict_device (and the gather) over the same %lld rows first", (long long)n_rows);
    if (!from_gathered && S != (int64_t)h->K * h->O) return fail(DIMN_ERR_ARG, "dimn_impute_finish: %lld slots listed, the prediction has %lld", (long long)S, (long long)h->K * h->O);
    if (from_gathered && (S != h->full_width || n_rows != h->full_rows))
        return fail(DIMN_ERR_ARG, "dimn_impute_finish: %lld slots over %lld rows listed, the gathered matrix is %lld x %lld", (long long)S, (long long)n_rows,
                    (long long)h->full_rows, (long long)h->full_width);
    for (int64_t j = 0; j < g; ++j) if (gene_off[j] > gene_off[j + 1]) return fail(DIMN_ERR_ARG, "dimn_impute_finish: gene_off not monotone");
    for (int64_t s = 0; s < S; ++s) if (gene_slot[s] < 0 || gene_slot[s] >= S) return fail(DIMN_ERR_ARG, "dimn_impute_finish: slot out of range");
    CHK(use_device(h));
    const bool chunked = !from_gathered && !h->pred_ev_rows.empty() && h->pred_ev_rows.back() == n_rows;   // the forward runs in row chunks: blocks wait for their rows only
    if (!chunked) HIPCHK(hipStreamSynchronize(h->stream));
    if (n_rows == 0) return DIMN_OK;
    const int64_t blk = std::max<int64_t>(1, std::min<int64_t>(n_rows, (int64_t)(128u << 20) / (g * 8)));   // ~128 MB per block
    Trace tr;
    int32_t *dOff = nullptr, *dSlot = nullptr;
    double *dRaw[2] = {nullptr, nullptr}, *dRes[2] = {nullptr, nullptr}, *pIn[2] = {nullptr, nullptr}, *pOut[2] = {nullptr, nullptr};
    hipStream_t st[2] = {nullptr, nullptr};
    hipEvent_t evOut[2] = {nullptr, nullptr};
    int rc = DIMN_OK;
    PinLease pins;
    {
        const char* why = "";
        if (!pins.take(4, std::max<size_t>((size_t)(128u << 20), (size_t)blk * g * 8), &why)) return fail(DIMN_ERR_HIP, "dimn_impute_finish: pinning the bounce buffers failed: %s", why);
    }
#define FIN_TRY(expr) do { hipError_t e_ = (expr); if (rc == DIMN_OK && e_ != hipSuccess) rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
    FIN_TRY(dev_malloc_bytes((void**)&dOff, (size_t)(g + 1) * 4));
    FIN_TRY(dev_malloc_bytes((void**)&dSlot, (size_t)std::max<int64_t>(S, 1) * 4));
    // the result blocks: the process-wide pair when this call holds the shared lease (made once, by dimn_warm_up or here), else its own
    const size_t res_bytes = (size_t)blk * g * 8;
    const bool shared_res = !pins.own && (g_fin_dev < 0 || g_fin_dev == h->cfg.device_id);      // (one process = one GPU in every supported layout)
    if (shared_res && g_fin_cap < res_bytes) {
        for (auto& p : g_fin_res) { if (p) (void)hipFree(p); p = nullptr; }
        g_fin_cap = 0;
        const size_t want = std::max<size_t>(res_bytes, (size_t)128u << 20);
        if (hipMalloc(&g_fin_res[0], want) == hipSuccess && hipMalloc(&g_fin_res[1], want) == hipSuccess) { g_fin_cap = want; g_fin_dev = h->cfg.device_id; }
        else { (void)hipGetLastError(); for (auto& p : g_fin_res) { if (p) (void)hipFree(p); p = nullptr; } }
    }
    const bool use_shared = shared_res && g_fin_cap >= res_bytes;
    for (int b = 0; b < 2; ++b) {
        if (!resident) FIN_TRY(dev_malloc_bytes((void**)&dRaw[b], (size_t)blk * g * 8));
        if (use_shared) dRes[b] = (double*)g_fin_res[b];
        else FIN_TRY(dev_malloc_bytes((void**)&dRes[b], res_bytes));
        pIn[b] = (double*)pins.buf[b]; pOut[b] = (double*)pins.buf[2 + b];
        if (use_shared) {
            if (!g_fin_st[b]) FIN_TRY(hipStreamCreateWithFlags(&g_fin_st[b], hipStreamNonBlocking));
            if (!g_fin_ev[b]) FIN_TRY(hipEventCreateWithFlags(&g_fin_ev[b], hipEventDisableTiming));
            st[b] = g_fin_st[b]; evOut[b] = g_fin_ev[b];
        } else {
            FIN_TRY(hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking));
            FIN_TRY(hipEventCreateWithFlags(&evOut[b], hipEventDisableTiming));
        }
    }
    if (rc == DIMN_OK) {
        FIN_TRY(hipMemcpy(dOff, gene_off, (size_t)(g + 1) * 4, hipMemcpyHostToDevice));
        if (S > 0) FIN_TRY(hipMemcpy(dSlot, gene_slot, (size_t)S * 4, hipMemcpyHostToDevice));
    }
    tr.lap("finish: allocations");
    const int lds_stage = (size_t)S * 4 <= 150 * 1024 ? 1 : 0;
    const size_t lds = lds_stage ? (size_t)S * 4 : 0;
    if (rc == DIMN_OK && lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)k_impute_finish<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_impute_finish<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int64_t nblk = (n_rows + blk - 1) / blk;
    // software pipeline over blocks: [host copy in | H2D | kernel | D2H] of block i on stream i%2; the host copy out of
    // block i-2 happens when its event has fired, right before its bounce buffer is re-used
    for (int64_t bi = 0; bi < nblk + 2 && rc == DIMN_OK; ++bi) {
        const int b = (int)(bi & 1);
        std::thread retire;
        if (bi >= 2) {                                   // retire block bi-2 (same buffers), beside the copy-in of block bi
            const int64_t r0 = (bi - 2) * blk, nr = std::min(blk, n_rows - r0);
            FIN_TRY(hipEventSynchronize(evOut[b]));
            if (rc == DIMN_OK) retire = std::thread([=] { parallel_memcpy(out + r0 * g, pOut[b], (size_t)nr * g * 8); });
        }
        struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_retire{retire};
        if (bi < nblk && rc == DIMN_OK) {
            const int64_t r0 = bi * blk, nr = std::min(blk, n_rows - r0);
            if (!resident) parallel_memcpy(pIn[b], raw + r0 * g, (size_t)nr * g * 8);
            if (retire.joinable()) retire.join();        // pOut[b] is free again before this block's D2H is queued
            if (chunked) FIN_TRY(hipStreamWaitEvent(st[b], pred_event_for(h, r0 + nr), 0));
            if (resident) {
                hipLaunchKernelGGL(k_impute_finish<float>, dim3((unsigned)std::min<int64_t>(nr, 4096)), dim3(512), lds, st[b], pred, S, r0,
                                   (const float*)(h->counts->d + r0 * g), nr, g, dOff, dSlot, ceiling, policy, lds_stage, dRes[b]);
            } else {
                FIN_TRY(hipMemcpyAsync(dRaw[b], pIn[b], (size_t)nr * g * 8, hipMemcpyHostToDevice, st[b]));
                hipLaunchKernelGGL(k_impute_finish<double>, dim3((unsigned)std::min<int64_t>(nr, 4096)), dim3(512), lds, st[b], pred, S, r0, (const double*)dRaw[b], nr, g,
                                   dOff, dSlot, ceiling, policy, lds_stage, dRes[b]);
            }
            FIN_TRY(hipGetLastError());
            FIN_TRY(hipMemcpyAsync(pOut[b], dRes[b], (size_t)nr * g * 8, hipMemcpyDeviceToHost, st[b]));
            FIN_TRY(hipEventRecord(evOut[b], st[b]));
        }
    }
#undef FIN_TRY
    tr.lap("finish: pipeline");
    for (int b = 0; b < 2; ++b) {
        if (st[b]) { (void)hipStreamSynchronize(st[b]); if (!use_shared) (void)hipStreamDestroy(st[b]); }
        if (evOut[b] && !use_shared) (void)hipEventDestroy(evOut[b]);
        if (dRaw[b]) (void)dev_free_any(dRaw[b]);
        if (dRes[b] && !use_shared) (void)dev_free_any(dRes[b]);
    }
    if (dOff) (void)dev_free_any(dOff);
    if (dSlot) (void)dev_free_any(dSlot);
    tr.lap("finish: frees");
    return rc;
}

template <typename ST> static uint64_t counts_row_checksum(const ST* src, int64_t g, uint64_t base);      // (defined with counts_scan below)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__AVX2__)
static inline bool counts_load4(const double* p, __m256d& x);
static inline bool counts_load4(const int64_t* p, __m256d& x);
#endif
// out = obs with its zeros replaced, in column order, by z[0 .. nz); false when obs does not hold exactly nz zeros (then it is not the
// matrix the device counted)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__AVX2__)
// lane i of a 4-double vector takes element (number of set mask bits below i) of the packed values: the "expand" AVX2 does not have
struct ExpandLut {
    alignas(32) int32_t idx[16][8];
    ExpandLut() { for (int m = 0; m < 16; ++m) { int r = 0; for (int i = 0; i < 4; ++i) { const int e = (m >> i) & 1 ? r++ : 0; idx[m][2 * i] = 2 * e; idx[m][2 * i + 1] = 2 * e + 1; } } }
};
static const ExpandLut g_expand;
#endif
template <typename ST>
static inline bool restore_row(const ST* obs, const double* z, int64_t nz, double* w, int64_t g) {
    int64_t k = 0, j = 0;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__AVX2__)
    // branch-free: the next four packed values are spread over the zero lanes by a table-driven permute and blended into the observed
    // vector; the row is written once, with streaming stores where it is 32-byte aligned (no read-for-ownership of the 8 GB result:
    // the host's copy rate, not PCIe, is what bounds this epilogue -- profiles/r05_dropin_host_side.txt)
    const bool aligned = (((uintptr_t)w) & 31) == 0;
    const __m256d zero = _mm256_setzero_pd();
    for (; j + 4 <= g && k + 4 <= nz; j += 4) {
        __m256d x;
        if (!counts_load4(obs + j, x)) {                 // (an int64 quad outside the count range: plain C++ for these four)
            for (int64_t q = j; q < j + 4; ++q) { const double xq = (double)obs[q]; w[q] = xq; if (xq == 0.0) { if (k < nz) w[q] = z[k]; ++k; } }
            continue;
        }
        const __m256d eq = _mm256_cmp_pd(x, zero, _CMP_EQ_OQ);
        const int m = _mm256_movemask_pd(eq);
        const __m256d zv = _mm256_loadu_pd(z + k);
        const __m256d spread = _mm256_castsi256_pd(_mm256_permutevar8x32_epi32(_mm256_castpd_si256(zv), _mm256_load_si256((const __m256i*)g_expand.idx[m])));
        const __m256d r = _mm256_blendv_pd(x, spread, eq);
        if (aligned) _mm256_stream_pd(w + j, r); else _mm256_storeu_pd(w + j, r);
        k += __builtin_popcount((unsigned)m);
    }
#endif
    for (; j < g; ++j) {
        const double x = (double)obs[j];
        w[j] = x;
        if (x == 0.0) { if (k < nz) w[j] = z[k]; ++k; }
    }
    return k == nz;
}

// predict()'s post-processing for policy "restore" over the RESIDENT counts (dimn_set_matrix_counts), with the caller's own float64 frame
// `observed` of those counts as the source of everything the policy leaves alone (multinet.py:296-299: observed counts win wherever they
// are positive): the device finishes and sends only the zero entries, packed per row (dimn_kernels.h: k_impute_finish_zeros), the host
// copies `observed` into `out` and drops them in.  Returns DIMN_ERR_STATE when `observed` is not the matrix the device holds (a row with a
// different number of zeros): the caller then takes dimn_impute_finish.
template <typename ST>
static int impute_finish_restore_impl(dimn_handle h, const ST* observed, int64_t n_rows, int64_t g, const int32_t* gene_off,
                                      const int32_t* gene_slot, double ceiling, int32_t from_gathered, double* out, uint64_t* observed_checksum) {
    if (!h || !observed || !gene_off || !gene_slot || !out || n_rows < 0 || g < 1) return fail(DIMN_ERR_ARG, "dimn_impute_finish_restore: bad argument");
    if (observed_checksum) *observed_checksum = 0;
    if (!h->counts || h->counts->n != n_rows || h->counts->g != g)
        return fail(DIMN_ERR_STATE, "dimn_impute_finish_restore: needs the count matrix of dimn_set_matrix_counts over the same %lld x %lld cells", (long long)n_rows, (long long)g);
    const int64_t S = gene_off[g];
    const float* pred = from_gathered ? h->d_full : h->d_out;
    if (!pred || h->out_rows != n_rows) return fail(DIMN_ERR_STATE, "dimn_impute_finish_restore: run dimn_predict_device (and the gather) over the same %lld rows first", (long long)n_rows);
    if (!from_gathered && S != (int64_t)h->K * h->O) return fail(DIMN_ERR_ARG, "dimn_impute_finish_restore: %lld slots listed, the prediction has %lld", (long long)S, (long long)h->K * h->O);
    if (from_gathered && (S != h->full_width || n_rows != h->full_rows))
        return fail(DIMN_ERR_ARG, "dimn_impute_finish_restore: %lld slots over %lld rows listed, the gathered matrix is %lld x %lld", (long long)S, (long long)n_rows,
                    (long long)h->full_rows, (long long)h->full_width);
    for (int64_t j = 0; j < g; ++j) if (gene_off[j] > gene_off[j + 1]) return fail(DIMN_ERR_ARG, "dimn_impute_finish_restore: gene_off not monotone");
    for (int64_t s = 0; s < S; ++s) if (gene_slot[s] < 0 || gene_slot[s] >= S) return fail(DIMN_ERR_ARG, "dimn_impute_finish_restore: slot out of range");
    CHK(use_device(h));
    const bool chunked = !from_gathered && !h->pred_ev_rows.empty() && h->pred_ev_rows.back() == n_rows;
    if (!chunked) HIPCHK(hipStreamSynchronize(h->stream));
    if (n_rows == 0) return DIMN_OK;
    const int64_t blk = std::max<int64_t>(1, std::min<int64_t>(n_rows, (int64_t)(128u << 20) / (g * 8)));   // rows per block: at most ~128 MB of results
    Trace tr;
    int32_t *dOff = nullptr, *dSlot = nullptr, *dZeros = nullptr;
    int64_t* dBase = nullptr;
    double *dRes[2] = {nullptr, nullptr}, *pOut[2] = {nullptr, nullptr};
    hipStream_t st[2] = {nullptr, nullptr};
    hipEvent_t evOut[2] = {nullptr, nullptr};
    int rc = DIMN_OK;
    PinLease pins;
    {
        const char* why = "";
        if (!pins.take(4, std::max<size_t>((size_t)(128u << 20), (size_t)blk * g * 8), &why)) return fail(DIMN_ERR_HIP, "dimn_impute_finish_restore: pinning the bounce buffers failed: %s", why);
    }
#define FIN_TRY(expr) do { hipError_t e_ = (expr); if (rc == DIMN_OK && e_ != hipSuccess) rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
    FIN_TRY(dev_malloc_bytes((void**)&dOff, (size_t)(g + 1) * 4));
    FIN_TRY(dev_malloc_bytes((void**)&dSlot, (size_t)std::max<int64_t>(S, 1) * 4));
    FIN_TRY(dev_malloc_bytes((void**)&dZeros, (size_t)n_rows * 4));
    FIN_TRY(dev_malloc_bytes((void**)&dBase, (size_t)(n_rows + 1) * 8));
    const size_t res_bytes = (size_t)blk * g * 8;
    const bool shared_res = !pins.own && (g_fin_dev < 0 || g_fin_dev == h->cfg.device_id);
    if (shared_res && g_fin_cap < res_bytes) {
        for (auto& p : g_fin_res) { if (p) (void)hipFree(p); p = nullptr; }
        g_fin_cap = 0;
        const size_t want = std::max<size_t>(res_bytes, (size_t)128u << 20);
        if (hipMalloc(&g_fin_res[0], want) == hipSuccess && hipMalloc(&g_fin_res[1], want) == hipSuccess) { g_fin_cap = want; g_fin_dev = h->cfg.device_id; }
        else { (void)hipGetLastError(); for (auto& p : g_fin_res) { if (p) (void)hipFree(p); p = nullptr; } }
    }
    const bool use_shared = shared_res && g_fin_cap >= res_bytes;
    for (int b = 0; b < 2; ++b) {
        if (use_shared) dRes[b] = (double*)g_fin_res[b];
        else FIN_TRY(dev_malloc_bytes((void**)&dRes[b], res_bytes));
        pOut[b] = (double*)pins.buf[2 + b];
        if (use_shared) {
            if (!g_fin_st[b]) FIN_TRY(hipStreamCreateWithFlags(&g_fin_st[b], hipStreamNonBlocking));
            if (!g_fin_ev[b]) FIN_TRY(hipEventCreateWithFlags(&g_fin_ev[b], hipEventDisableTiming));
            st[b] = g_fin_st[b]; evOut[b] = g_fin_ev[b];
        } else {
            FIN_TRY(hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking));
            FIN_TRY(hipEventCreateWithFlags(&evOut[b], hipEventDisableTiming));
        }
    }
    // how many zeros every row holds (one pass over the resident counts, beside the forward), their running sum = where a row's values go
    std::vector<int32_t> zeros((size_t)n_rows);
    std::vector<int64_t> base((size_t)n_rows + 1, 0);
    if (rc == DIMN_OK) {
        FIN_TRY(hipMemcpyAsync(dOff, gene_off, (size_t)(g + 1) * 4, hipMemcpyHostToDevice, st[0]));
        if (S > 0) FIN_TRY(hipMemcpyAsync(dSlot, gene_slot, (size_t)S * 4, hipMemcpyHostToDevice, st[0]));
        hipLaunchKernelGGL(k_row_zeros, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, st[0], (const float*)h->counts->d, n_rows, g, dZeros);
        FIN_TRY(hipGetLastError());
        FIN_TRY(hipMemcpyAsync(zeros.data(), dZeros, (size_t)n_rows * 4, hipMemcpyDeviceToHost, st[0]));
        FIN_TRY(hipStreamSynchronize(st[0]));
        for (int64_t i = 0; i < n_rows; ++i) base[(size_t)i + 1] = base[(size_t)i] + zeros[(size_t)i];
        FIN_TRY(hipMemcpy(dBase, base.data(), (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice));      // (synchronous: st[1] may start at once)
    }
    tr.lap("finish (restore): allocations + zero counts");
    const int lds_stage = (size_t)S * 4 <= 150 * 1024 ? 1 : 0;
    const size_t lds = lds_stage ? (size_t)S * 4 : 0;
    if (rc == DIMN_OK && lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_impute_finish_zeros, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int64_t nblk = (n_rows + blk - 1) / blk;
    std::atomic<int> mismatch{0};
    std::atomic<uint64_t> checksum{0};
    const bool want_sum = observed_checksum != nullptr;
    for (int64_t bi = 0; bi < nblk + 2 && rc == DIMN_OK; ++bi) {
        const int b = (int)(bi & 1);
        std::thread retire;
        if (bi >= 2) {                                   // retire block bi-2: observed -> out with the block's zeros filled in, on the host pool
            const int64_t r0 = (bi - 2) * blk, nr = std::min(blk, n_rows - r0);
            FIN_TRY(hipEventSynchronize(evOut[b]));
            if (rc == DIMN_OK) retire = std::thread([=, &base, &mismatch, &checksum] {
                const unsigned hw = std::thread::hardware_concurrency();
                // (same box, tools/finish_ab.py, round 5: 16 threads 0.134-0.146 s per epilogue, 32: 0.114-0.126, 64: 0.119-0.137)
                const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw ? hw / 2 : 8, 32), nr * g / (1 << 20)));
                const double* z0 = pOut[b];
                const int64_t b0 = base[(size_t)r0];
                host_pool().run(nt, [=, &base, &mismatch, &checksum](int t) {
                    bool good = true;
                    uint64_t sum = 0;
                    for (int64_t i = r0 + nr * t / nt; i < r0 + nr * (t + 1) / nt; ++i) {
                        // the row is read twice, the second time from the core's cache: its checksum (the one dimn_counts_create took of the frame
                        // it uploaded -- equal sums: `observed` IS that frame, bit for bit), then the merge
                        if (want_sum) sum += counts_row_checksum(observed + i * g, g, (uint64_t)i * (uint64_t)g);
                        good &= restore_row(observed + i * g, z0 + (base[(size_t)i] - b0), base[(size_t)i + 1] - base[(size_t)i], out + i * g, g);
                    }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__AVX2__)
                    _mm_sfence();                            // (the streaming stores of restore_row are visible before the block is reported done)
#endif
                    if (!good) mismatch.store(1);
                    if (want_sum) checksum.fetch_add(sum);
                });
            });
        }
        struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_retire{retire};
        if (bi < nblk && rc == DIMN_OK) {
            const int64_t r0 = bi * blk, nr = std::min(blk, n_rows - r0);
            const int64_t nz = base[(size_t)(r0 + nr)] - base[(size_t)r0];
            if (retire.joinable()) retire.join();        // pOut[b] is free again before this block's D2H is queued
            if (chunked) FIN_TRY(hipStreamWaitEvent(st[b], pred_event_for(h, r0 + nr), 0));
            hipLaunchKernelGGL(k_impute_finish_zeros, dim3((unsigned)std::min<int64_t>(nr, 4096)), dim3(512), lds, st[b], pred, S, r0,
                               (const float*)(h->counts->d + r0 * g), nr, g, dOff, dSlot, (const int64_t*)(dBase + r0), ceiling, lds_stage, dRes[b]);
            FIN_TRY(hipGetLastError());
            if (nz > 0) FIN_TRY(hipMemcpyAsync(pOut[b], dRes[b], (size_t)nz * 8, hipMemcpyDeviceToHost, st[b]));
            FIN_TRY(hipEventRecord(evOut[b], st[b]));
        }
    }
#undef FIN_TRY
    tr.lap("finish (restore): pipeline");
    for (int b = 0; b < 2; ++b) {
        if (st[b]) { (void)hipStreamSynchronize(st[b]); if (!use_shared) (void)hipStreamDestroy(st[b]); }
        if (evOut[b] && !use_shared) (void)hipEventDestroy(evOut[b]);
        if (dRes[b] && !use_shared) (void)dev_free_any(dRes[b]);
    }
    if (dOff) (void)dev_free_any(dOff);
    if (dSlot) (void)dev_free_any(dSlot);
    if (dZeros) (void)dev_free_any(dZeros);
    if (dBase) (void)dev_free_any(dBase);
    if (rc == DIMN_OK && chunked) { hipError_t e_ = hipStreamSynchronize(h->stream); if (e_ != hipSuccess) rc = fail(DIMN_ERR_HIP, "dimn_impute_finish_restore: %s", hipGetErrorString(e_)); }
    if (rc == DIMN_OK && mismatch.load()) rc = fail(DIMN_ERR_STATE, "dimn_impute_finish_restore: `observed` is not the count matrix the device holds (a row has a different number of zeros)");
    if (rc == DIMN_OK && observed_checksum) *observed_checksum = checksum.load();
    tr.lap("finish (restore): frees");
    return rc;
}

extern "C" int dimn_impute_finish_restore(dimn_handle h, const void* observed, int32_t observed_dtype, int64_t n_rows, int64_t g, const int32_t* gene_off,
                                          const int32_t* gene_slot, double ceiling, int32_t from_gathered, double* out, uint64_t* observed_checksum) {
    if (observed_dtype == DIMN_DTYPE_F64) return impute_finish_restore_impl(h, (const double*)observed, n_rows, g, gene_off, gene_slot, ceiling, from_gathered, out, observed_checksum);
    if (observed_dtype == DIMN_DTYPE_I64) return impute_finish_restore_impl(h, (const int64_t*)observed, n_rows, g, gene_off, gene_slot, ceiling, from_gathered, out, observed_checksum);
    return fail(DIMN_ERR_ARG, "dimn_impute_finish_restore: observed_dtype must be DIMN_DTYPE_F64 or DIMN_DTYPE_I64");
}

extern "C" int dimn_synchronize(dimn_handle h) {
    if (!h) return fail(DIMN_ERR_ARG, "null handle");
    CHK(use_device(h));
    CHK(sync_lanes(h));
    return DIMN_OK;
}

extern "C" int dimn_set_profiling(dimn_handle h, int32_t on) {
    if (!h) return fail(DIMN_ERR_ARG, "null handle");
    h->profiling = on != 0;
    return DIMN_OK;
}
extern "C" int dimn_training_precision(dimn_handle h) {
    if (!h) return fail(DIMN_ERR_ARG, "dimn_training_precision: null handle");
    return (h->res_G ? h->res_bf16 : h->train_bf16) ? DIMN_PREC_BF16 : DIMN_PREC_F32;
}

extern "C" int dimn_path_info(dimn_handle h, int32_t* out8) {
    if (!h || !out8) return fail(DIMN_ERR_ARG, "dimn_path_info: null argument");
    out8[0] = h->gen ? 2 : (h->res_G ? 1 : 0);                 // 0 streaming kernels, 1 register-resident epoch kernel, 2 general path
    out8[1] = h->res_G ? ceil_div(h->K, h->res_Kg) : 0;        // resident: epoch launches (groups of sub-nets) per epoch
    out8[2] = h->res_S1;                                        // resident: D-splits per hidden tile
    out8[3] = h->mid_fused;                                     // streaming: 1 fused second layer (RED -> MFB -> RED2), 0 two kernels (MF + MB)
    out8[4] = h->mid_fused ? h->mid_slices : 0;                 // ... output slices per sub-net
    out8[5] = h->mid_fused ? 2 : 0;                             // ... its form: 2 = the tile pipeline k_mid_pipe (1 / 0 were the three-phase kernel of rounds 2-4, retired)
    out8[6] = h->res_G ? 2 * h->res_bf16 : h->train_bf16;      // training GEMMs on the bf16 matrix cores: 1 the second layer's (fused kernel), 2 all (resident kernel)
    out8[7] = h->dm.HT == 16 ? 1 : (h->w1_waves ? 3 : 0);       // first layer: 1 ring B1F1 (H = 256), 3 four-set ring with one hidden tile per wave (8 .. 24 tiles other than 16), 0 generic (2 was the shared-staging kernel of hidden 300, retired)
    return DIMN_OK;
}

extern "C" int dimn_get_timers(dimn_handle h, double* out4, int32_t reset) {
    if (!h || !out4) return fail(DIMN_ERR_ARG, "null argument");
    out4[0] = h->tm_step_ms; out4[1] = (double)h->tm_steps; out4[2] = h->tm_w1_ms; out4[3] = (double)h->tm_w1;
    out4[4] = h->tm_w1_bytes; out4[5] = (double)h->lanes.size();
    out4[6] = h->tm_res_ms; out4[7] = (double)h->tm_res_steps;
    if (reset) { h->tm_step_ms = h->tm_w1_ms = h->tm_w1_bytes = h->tm_res_ms = 0; h->tm_steps = h->tm_w1 = h->tm_res_steps = 0; }
    return DIMN_OK;
}

// ---- RCCL over xGMI ------------------------------------------------------------------------
extern "C" int dimn_comm_unique_id(uint8_t* id) {
    if (!id) return fail(DIMN_ERR_ARG, "null id");
    CHK(rccl_bind());
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    static_assert(sizeof(u) == DIMN_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return DIMN_OK;
}
extern "C" int dimn_comm_init(dimn_handle h, const uint8_t* id, int32_t n_ranks, int32_t rank) {
    if (!h || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(DIMN_ERR_ARG, "dimn_comm_init: bad argument");
    CHK(rccl_bind());
    CHK(use_device(h));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    NCCLCHK(g_rccl.CommInitRank(&h->comm, n_ranks, u, rank));
    h->n_ranks = n_ranks; h->rank = rank;
    return DIMN_OK;
}
extern "C" int dimn_comm_info(dimn_handle h, int32_t* out2) {
    if (!h || !out2) return fail(DIMN_ERR_ARG, "dimn_comm_info: null argument");
    if (!h->comm) return fail(DIMN_ERR_STATE, "dimn_comm_info: dimn_comm_init first");
    int n = 0, r = -1;
    NCCLCHK(g_rccl.CommCount(h->comm, &n));
    NCCLCHK(g_rccl.CommUserRank(h->comm, &r));
    out2[0] = n; out2[1] = r;
    return DIMN_OK;
}
extern "C" int dimn_comm_allreduce_sum(dimn_handle h, double* v, int32_t n) {
    if (!h || !v || n < 1) return fail(DIMN_ERR_ARG, "dimn_comm_allreduce_sum: bad argument");
    if (!h->comm) return fail(DIMN_ERR_STATE, "dimn_comm_allreduce_sum: dimn_comm_init first");
    CHK(use_device(h));
    if (h->red_cap < n) {                                   // scratch kept across calls (one all-reduce per epoch)
        HIPCHK(hipStreamSynchronize(h->stream));
        DEV_FREE(h->d_red);
        CHK(dev_alloc(&h->d_red, (size_t)std::max(n, 16)));
        h->red_cap = std::max(n, 16);
    }
    double* d = h->d_red;
    HIPCHK(hipMemcpyAsync(d, v, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    NCCLCHK(g_rccl.AllReduce(d, d, (size_t)n, kNcclFloat64, kNcclSum, h->comm, h->stream));
    HIPCHK(hipMemcpyAsync(v, d, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DIMN_OK;
}
// Root's side of the gather, after the peers' blocks have landed contiguously in the staging arena (block of rank r at
// n_rows * koff[r] * O floats): every [n_rows][K_r*O] block is placed into its column range of the full [n_rows][K_global*O]
// matrix in HBM by a strided D2D copy (root's own block straight from d_out); host copy only if out != NULL.
static int gather_arenas(dimn_handle h, int64_t n_rows, const int32_t* counts, int n_ranks, std::vector<int64_t>& koff, int64_t& ktot) {
    ktot = 0;
    koff.assign((size_t)n_ranks, 0);
    for (int r = 0; r < n_ranks; ++r) {
        if (counts[r] < 0) return fail(DIMN_ERR_ARG, "gather: counts[%d] < 0", r);
        koff[(size_t)r] = ktot; ktot += counts[r];
    }
    const int64_t need = n_rows * ktot * h->O;
    if (h->full_cap < need) {
        HIPCHK(hipStreamSynchronize(h->stream));
        DEV_FREE(h->d_full); DEV_FREE(h->d_stage);
        CHK(dev_alloc(&h->d_full, (size_t)need));
        CHK(dev_alloc(&h->d_stage, (size_t)need));
        h->full_cap = need;
    }
    return DIMN_OK;
}
static int gather_place(dimn_handle h, int64_t n_rows, const int32_t* counts, int n_ranks, int root, const std::vector<int64_t>& koff, int64_t ktot, float* out) {
    const int O = h->O;
    for (int r = 0; r < n_ranks; ++r) {
        const float* src = (r == root) ? h->d_out : h->d_stage + (size_t)n_rows * koff[(size_t)r] * O;
        if (n_rows > 0 && counts[r] > 0)
            HIPCHK(hipMemcpy2DAsync(h->d_full + koff[(size_t)r] * O, (size_t)ktot * O * 4, src, (size_t)counts[r] * O * 4,
                                    (size_t)counts[r] * O * 4, (size_t)n_rows, hipMemcpyDeviceToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->full_rows = n_rows; h->full_width = ktot * O;
    const int64_t need = n_rows * ktot * O;
    if (out && need > 0) HIPCHK(hipMemcpy(out, h->d_full, (size_t)need * 4, hipMemcpyDeviceToHost));
    return DIMN_OK;
}
extern "C" int dimn_comm_gather_predictions(dimn_handle h, int64_t n_rows, const int32_t* counts, int32_t root, float* out) {
    if (!h || !counts || n_rows < 0) return fail(DIMN_ERR_ARG, "dimn_comm_gather_predictions: bad argument");
    if (!h->comm) return fail(DIMN_ERR_STATE, "dimn_comm_gather_predictions: dimn_comm_init first");
    if (root < 0 || root >= h->n_ranks) return fail(DIMN_ERR_ARG, "dimn_comm_gather_predictions: root out of range");
    if (h->out_rows != n_rows) return fail(DIMN_ERR_STATE, "dimn_comm_gather_predictions: last dimn_predict_device had %lld rows", (long long)h->out_rows);
    if (counts[h->rank] != h->K) return fail(DIMN_ERR_ARG, "dimn_comm_gather_predictions: counts[rank] != n_subnets");
    CHK(use_device(h));
    const int O = h->O;
    if (h->rank != root) {
        NCCLCHK(g_rccl.GroupStart());
        NCCLCHK(g_rccl.Send(h->d_out, (size_t)n_rows * h->K * O, kNcclFloat32, root, h->comm, h->stream));
        NCCLCHK(g_rccl.GroupEnd());
        HIPCHK(hipStreamSynchronize(h->stream));
        return DIMN_OK;
    }
    // root: every peer sends over its own xGMI link; the blocks land contiguously in the staging arena, then gather_place
    int64_t ktot = 0;
    std::vector<int64_t> koff;
    CHK(gather_arenas(h, n_rows, counts, h->n_ranks, koff, ktot));
    NCCLCHK(g_rccl.GroupStart());
    for (int r = 0; r < h->n_ranks; ++r) {
        if (r == root) continue;
        NCCLCHK(g_rccl.Recv(h->d_stage + (size_t)n_rows * koff[(size_t)r] * O, (size_t)n_rows * counts[r] * O, kNcclFloat32, r, h->comm, h->stream));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return gather_place(h, n_rows, counts, h->n_ranks, root, koff, ktot, out);
}
// The same gather with the transport replaced by device-to-device copies on ONE GPU: handles[r] plays rank r (handles[root] is the
// root; every handle on the root's device, each with a dimn_predict_device result over the same n_rows).  Everything behind the
// ncclRecv -- arena sizing, block offsets, the strided placement, full_rows / full_width for dimn_impute_finish(from_gathered) -- is
// the code of dimn_comm_gather_predictions; only RCCL itself is not exercised.  For the single-GPU boxes the tests run on.
extern "C" int dimn_comm_gather_loopback(const dimn_handle* handles, int32_t n_ranks, int64_t n_rows, const int32_t* counts, int32_t root, float* out) {
    if (!handles || !counts || n_ranks < 1 || n_rows < 0 || root < 0 || root >= n_ranks) return fail(DIMN_ERR_ARG, "dimn_comm_gather_loopback: bad argument");
    dimn_handle h = handles[root];
    if (!h) return fail(DIMN_ERR_ARG, "dimn_comm_gather_loopback: null root handle");
    for (int r = 0; r < n_ranks; ++r) {
        dimn_handle p = handles[r];
        if (!p || p->cfg.device_id != h->cfg.device_id || p->O != h->O) return fail(DIMN_ERR_ARG, "dimn_comm_gather_loopback: handle %d is null, on another device or of another out_dim", r);
        if (p->out_rows != n_rows || !p->d_out) return fail(DIMN_ERR_STATE, "dimn_comm_gather_loopback: handle %d has no dimn_predict_device result over %lld rows", r, (long long)n_rows);
        if (counts[r] != p->K) return fail(DIMN_ERR_ARG, "dimn_comm_gather_loopback: counts[%d] != n_subnets of handle %d", r, r);
    }
    CHK(use_device(h));
    int64_t ktot = 0;
    std::vector<int64_t> koff;
    CHK(gather_arenas(h, n_rows, counts, n_ranks, koff, ktot));
    for (int r = 0; r < n_ranks; ++r) {
        if (r == root) continue;
        HIPCHK(hipStreamSynchronize(handles[r]->stream));        // (the "send": the peer's forward has finished)
        if (n_rows > 0)
            HIPCHK(hipMemcpyAsync(h->d_stage + (size_t)n_rows * koff[(size_t)r] * h->O, handles[r]->d_out, (size_t)n_rows * counts[r] * h->O * 4, hipMemcpyDeviceToDevice, h->stream));
    }
    return gather_place(h, n_rows, counts, n_ranks, root, koff, ktot, out);
}
extern "C" int dimn_comm_destroy(dimn_handle h) {
    if (!h) return fail(DIMN_ERR_ARG, "null handle");
    if (h->comm) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    return DIMN_OK;
}

// ---- next row: the CSV edges of the CLI (deepImpute.py:13, :35), host code, no GPU needed ----------
extern "C" int dimn_csv_scan(const char* path, int64_t* n_rows, int64_t* n_cols, int64_t* label_bytes) {
    if (!path || !n_rows || !n_cols || !label_bytes) return fail(DIMN_ERR_ARG, "dimn_csv_scan: null argument");
    std::string err;
    const int rc = csv_scan(path, n_rows, n_cols, label_bytes, err);
    return rc ? fail(rc == -4 ? DIMN_ERR_UNSUP : DIMN_ERR_ARG, "dimn_csv_scan(%s): %s", path, err.c_str()) : DIMN_OK;
}
extern "C" int dimn_csv_read(const char* path, int64_t n_rows, int64_t n_cols, int64_t* values, char* labels, int64_t label_bytes) {
    if (!path || !values || !labels || n_rows < 1 || n_cols < 1) return fail(DIMN_ERR_ARG, "dimn_csv_read: bad argument");
    std::string err;
    const int rc = csv_read(path, n_rows, n_cols, values, labels, label_bytes, err);
    return rc ? fail(rc == -4 ? DIMN_ERR_UNSUP : DIMN_ERR_ARG, "dimn_csv_read(%s): %s", path, err.c_str()) : DIMN_OK;
}
extern "C" int dimn_csv_write(const char* path, const double* values, int64_t n_rows, int64_t n_cols, const char* index_name, const char* col_labels,
                              const char* row_labels) {
    if (!path || !values || !col_labels || !row_labels || n_rows < 0 || n_cols < 0) return fail(DIMN_ERR_ARG, "dimn_csv_write: bad argument");
    std::string err;
    const int rc = csv_write(path, values, n_rows, n_cols, index_name, col_labels, row_labels, err);
    return rc ? fail(DIMN_ERR_ARG, "dimn_csv_write(%s): %s", path, err.c_str()) : DIMN_OK;
}

// ---- per-gene statistics of fit()'s planning (multinet.py:191), host code, no GPU needed ----------
extern "C" int dimn_col_stats(const double* a, int64_t n, int64_t g, int64_t ld, double* mean, double* var, double* vmax, int32_t* has_nan, int32_t threads) {
    if (!a || !mean || !vmax || !has_nan || n < 1 || g < 1 || ld < g) return fail(DIMN_ERR_ARG, "dimn_col_stats: bad argument");
    if (var && n < 2) return fail(DIMN_ERR_ARG, "dimn_col_stats: the variance needs two rows");
    int hn = 0;
    hoststats_run(a, n, g, ld, mean, var, vmax, &hn, threads > 0 ? threads : (int)std::min<unsigned>(64u, std::max(1u, std::thread::hardware_concurrency())));
    *has_nan = hn;
    return DIMN_OK;
}

// the same statistics as two calls, so that other work can run between the sweeps: dimn_col_stats_first (mean, nanvar's own
// average, per-column minimum and maximum, matrix maximum, NaN), then dimn_col_stats_var (var from those averages)
extern "C" int dimn_col_stats_first(const double* a, int64_t n, int64_t g, int64_t ld, double* mean, double* avg, double* cmin, double* cmax, double* vmax,
                                    int32_t* has_nan, int32_t threads) {
    if (!a || !mean || !avg || !cmin || !cmax || !vmax || !has_nan || n < 1 || g < 1 || ld < g) return fail(DIMN_ERR_ARG, "dimn_col_stats_first: bad argument");
    int hn = 0;
    hoststats_run(a, n, g, ld, mean, nullptr, vmax, &hn, threads > 0 ? threads : (int)std::min<unsigned>(64u, std::max(1u, std::thread::hardware_concurrency())), 1, avg, cmin, cmax);
    *has_nan = hn;
    return DIMN_OK;
}
extern "C" int dimn_col_stats_var(const double* a, int64_t n, int64_t g, int64_t ld, const double* avg, double* var, int32_t threads) {
    if (!a || !avg || !var || n < 2 || g < 1 || ld < g) return fail(DIMN_ERR_ARG, "dimn_col_stats_var: bad argument");
    double vmax = 0; int hn = 0;
    std::vector<double> mean_unused((size_t)g);
    hoststats_run(a, n, g, ld, mean_unused.data(), var, &vmax, &hn, threads > 0 ? threads : (int)std::min<unsigned>(64u, std::max(1u, std::thread::hardware_concurrency())), 2,
                  const_cast<double*>(avg));
    return DIMN_OK;
}

#ifdef DIMN_PRED_TL
// diagnostic build only (tools/predict_timeline.py): phase clocks of k_predict, summed over waves since the last call
extern "C" int dimn_debug_pred_timeline(unsigned long long* out) {
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pred_tl), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_pred_tl), z, sizeof(z)));
    return DIMN_OK;
}
#endif
#ifdef DIMN_RES_TL
// diagnostic build only (tools/res_timeline.py): per-workgroup phase clocks of the last resident epoch launch
extern "C" int dimn_debug_res_timeline(unsigned long long* out, int n_words) {
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_res_tl), (size_t)n_words * sizeof(unsigned long long)));
    return DIMN_OK;
}
#endif

#ifdef DIMN_RES_TL2
// diagnostic build only (tools/res_trace.py): absolute time stamps of four steps of the last resident epoch launch, per workgroup
extern "C" int dimn_debug_res_trace(unsigned long long* out, int n_words) {
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_res_tl2), (size_t)n_words * sizeof(unsigned long long)));
    return DIMN_OK;
}
#endif

// ---- get_distance_matrix on the GPU (reference multinet.py:20-34) -------------------
// The same for a matrix that does not fit the GPU beside its g x g result (BASELINE configs[4]: 1M x 30k = 240 GB of float64):
// two streamed passes over row blocks of X through pinned bounce buffers -- column sums, then centre each block and
// accumulate C += Zb^T Zb on the fp64 matrix cores (2 GB per block: 0.2 s of GEMM behind 50 ms of copy).
static int corr_on_device_streamed(const double* X, int64_t n, int64_t g, hipStream_t st, double** dOutp) {
    const int64_t gp = (g + CORR_BT - 1) / CORR_BT * CORR_BT;
    const int nb = (int)(gp / CORR_BT);
    int64_t blk = std::max<int64_t>(CORR_KC, (int64_t)(2048ll << 20) / (gp * 8) / CORR_KC * CORR_KC);      // rows per block, a multiple of 16
    if (const char* e = getenv("DIMN_CORR_BUDGET_GB")) if (const char* c = strchr(e, ':')) blk = std::max<int64_t>(CORR_KC, atoll(c + 1) / CORR_KC * CORR_KC);   // "B:rows" (tests)
    double *dZ[2] = {nullptr, nullptr}, *pin[2] = {nullptr, nullptr}, *dC = nullptr, *dOut = nullptr, *dMean = nullptr;
    int2* dPairs = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = DIMN_OK;
    const std::vector<int2> pairs = xcd_tiled_pairs(nb);
#define CORR_TRY(expr) do { hipError_t e_ = (expr); if (rc == DIMN_OK && e_ != hipSuccess) rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
    for (int b = 0; b < 2; ++b) {
        CORR_TRY(dev_malloc_bytes((void**)&dZ[b], (size_t)blk * gp * 8));
        CORR_TRY(hipHostMalloc((void**)&pin[b], (size_t)blk * g * 8, hipHostMallocDefault));
        CORR_TRY(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
    }
    CORR_TRY(dev_malloc_bytes((void**)&dC, (size_t)gp * gp * 8));
    CORR_TRY(dev_malloc_bytes((void**)&dOut, (size_t)g * g * 8));
    CORR_TRY(dev_malloc_bytes((void**)&dMean, (size_t)gp * 8));
    CORR_TRY(dev_malloc_bytes((void**)&dPairs, pairs.size() * sizeof(int2)));
    if (rc == DIMN_OK) {
        CORR_TRY(hipMemsetAsync(dMean, 0, (size_t)gp * 8, st));
        CORR_TRY(hipMemcpyAsync(dPairs, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    }
    bool recorded[2] = {false, false};
    for (int pass = 0; pass < 2 && rc == DIMN_OK; ++pass) {
        int64_t bi = 0;
        for (int64_t r0 = 0; r0 < n && rc == DIMN_OK; r0 += blk, ++bi) {
            const int b = (int)(bi & 1);
            const int64_t nr = std::min(blk, n - r0), nrp = (nr + CORR_KC - 1) / CORR_KC * CORR_KC;
            if (recorded[b]) CORR_TRY(hipEventSynchronize(ev[b]));      // the previous user of this buffer pair (also across the two passes)
            if (rc != DIMN_OK) break;
            parallel_memcpy(pin[b], X + r0 * g, (size_t)nr * g * 8);
            if (nrp > nr || gp > g) CORR_TRY(hipMemsetAsync(dZ[b], 0, (size_t)nrp * gp * 8, st));      // zero padding rows / columns
            CORR_TRY(hipMemcpy2DAsync(dZ[b], (size_t)gp * 8, pin[b], (size_t)g * 8, (size_t)g * 8, (size_t)nr, hipMemcpyHostToDevice, st));
            if (pass == 0) {
                hipLaunchKernelGGL(k_corr_colsum_acc, dim3((unsigned)((gp + 255) / 256)), dim3(256), 0, st, dZ[b], nr, gp, dMean);
            } else {
                hipLaunchKernelGGL(k_corr_center, dim3((unsigned)((g + 255) / 256), (unsigned)std::min<int64_t>(nr, 1024)), dim3(256), 0, st, dZ[b], nr, g, gp, dMean);
                hipLaunchKernelGGL(k_corr_gemm, dim3((unsigned)pairs.size()), dim3(256), 0, st, dZ[b], nrp, gp, dPairs, dC, bi > 0 ? 1 : 0);
            }
            CORR_TRY(hipGetLastError());
            CORR_TRY(hipEventRecord(ev[b], st));
            recorded[b] = true;
        }
        if (pass == 0 && rc == DIMN_OK) hipLaunchKernelGGL(k_corr_scale, dim3((unsigned)((gp + 255) / 256)), dim3(256), 0, st, dMean, gp, 1.0 / (double)n);
    }
    if (rc == DIMN_OK) {
        hipLaunchKernelGGL(k_corr_finish, dim3((unsigned)((g + 255) / 256), (unsigned)g), dim3(256), 0, st, dC, g, gp, 1.0 / (double)(n - 1), dOut);
        CORR_TRY(hipGetLastError());
    }
    CORR_TRY(hipStreamSynchronize(st));
#undef CORR_TRY
    for (int b = 0; b < 2; ++b) {
        if (dZ[b]) (void)dev_free_any(dZ[b]);
        if (pin[b]) (void)hipHostFree(pin[b]);
        if (ev[b]) (void)hipEventDestroy(ev[b]);
    }
    if (dC) (void)dev_free_any(dC);
    if (dMean) (void)dev_free_any(dMean);
    if (dPairs) (void)dev_free_any(dPairs);
    if (rc != DIMN_OK && dOut) { (void)dev_free_any(dOut); dOut = nullptr; }
    *dOutp = dOut;
    return rc;
}

// |corr| of the columns of host X[n][g] (fp64) into a fresh device matrix *dOutp [g][g]; the caller frees it.
// device-resident source of the candidate columns: counts[n][ld] float32 (exact integers), column j of the pool = cols[j]
struct CorrDevSrc { const float* counts; int64_t ld; const int32_t* d_cols; };
__global__ __launch_bounds__(256) void k_counts_to_z(const float* __restrict__ counts, int64_t ld, const int32_t* __restrict__ cols, int64_t n, int64_t g,
                                                     int64_t gp, double* __restrict__ Z) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= g) return;
    const int32_t c = cols[j];
    for (int64_t i = blockIdx.y; i < n; i += gridDim.y) Z[i * gp + j] = (double)counts[i * ld + c];
}
static int corr_on_device(const double* X, int64_t n, int64_t g, hipStream_t st, double** dOutp, const CorrDevSrc* src = nullptr) {
    const int64_t gp = (g + CORR_BT - 1) / CORR_BT * CORR_BT, np_ = (n + CORR_KC - 1) / CORR_KC * CORR_KC;
    {   // the resident form needs np*gp + 2 g^2 doubles; above the budget (default 64 GB) the matrix is streamed in row blocks
        double budget = 64.0;
        if (const char* e = getenv("DIMN_CORR_BUDGET_GB")) budget = atof(e);
        if (((double)np_ * gp + 2.0 * gp * gp) * 8.0 > budget * 1073741824.0) {
            if (src) return fail(DIMN_ERR_UNSUP, "corr: the resident-counts form does not stream (matrix beyond DIMN_CORR_BUDGET_GB)");
            return corr_on_device_streamed(X, n, g, st, dOutp);
        }
    }
    const int nb = (int)(gp / CORR_BT);
    double *dZ = nullptr, *dC = nullptr, *dOut = nullptr, *dMean = nullptr, *dPart = nullptr;
    int2* dPairs = nullptr;
    int rc = DIMN_OK;
    const std::vector<int2> pairs = xcd_tiled_pairs(nb);
    const int nparts = (int)std::min<int64_t>(64, (n + 255) / 256);
    const int64_t rows_per_block = (n + nparts - 1) / nparts;
#define CORR_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } } while (0)
    Trace tr;
    CORR_TRY(dev_malloc_bytes((void**)&dZ, (size_t)np_ * gp * 8));
    CORR_TRY(dev_malloc_bytes((void**)&dC, (size_t)gp * gp * 8));
    CORR_TRY(dev_malloc_bytes((void**)&dOut, (size_t)g * g * 8));
    CORR_TRY(dev_malloc_bytes((void**)&dMean, (size_t)gp * 8));
    CORR_TRY(dev_malloc_bytes((void**)&dPart, (size_t)nparts * gp * 8));
    CORR_TRY(dev_malloc_bytes((void**)&dPairs, pairs.size() * sizeof(int2)));
    CORR_TRY(hipMemsetAsync(dZ, 0, (size_t)np_ * gp * 8, st));
    tr.lap("corr: device allocations");
    if (src) {      // the candidate columns are already on the device (dimn_counts): one conversion kernel instead of an 8 GB upload
        hipLaunchKernelGGL(k_counts_to_z, dim3((unsigned)((g + 255) / 256), (unsigned)std::min<int64_t>(n, 2048)), dim3(256), 0, st, src->counts, src->ld, src->d_cols,
                           n, g, gp, dZ);
        CORR_TRY(hipGetLastError());
    } else {   // X (pageable) -> pinned bounce buffers on several host threads -> device rows of pitch gp, double-buffered
        const int64_t blk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(128u << 20) / (g * 8)));
        double* pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        for (int b = 0; b < 2 && rc == DIMN_OK; ++b) {
            if (hipHostMalloc((void**)&pin[b], (size_t)blk * g * 8, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess)
                rc = fail(DIMN_ERR_HIP, "corr: pinned staging allocation failed");
        }
        int64_t bi = 0;
        for (int64_t r0 = 0; r0 < n && rc == DIMN_OK; r0 += blk, ++bi) {
            const int b = (int)(bi & 1);
            const int64_t nr = std::min(blk, n - r0);
            if (bi >= 2 && hipEventSynchronize(ev[b]) != hipSuccess) rc = fail(DIMN_ERR_HIP, "corr: event wait failed");
            if (rc != DIMN_OK) break;
            parallel_memcpy(pin[b], X + r0 * g, (size_t)nr * g * 8);
            if (hipMemcpy2DAsync(dZ + r0 * gp, (size_t)gp * 8, pin[b], (size_t)g * 8, (size_t)g * 8, (size_t)nr, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipEventRecord(ev[b], st) != hipSuccess)
                rc = fail(DIMN_ERR_HIP, "corr: host-to-device copy failed");
        }
        (void)hipStreamSynchronize(st);
        for (int b = 0; b < 2; ++b) { if (pin[b]) (void)hipHostFree(pin[b]); if (ev[b]) (void)hipEventDestroy(ev[b]); }
        if (rc != DIMN_OK) goto done;
    }
    tr.lap("corr: upload X");
    CORR_TRY(hipMemcpyAsync(dPairs, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_corr_colsum, dim3((unsigned)((gp + 255) / 256), (unsigned)nparts), dim3(256), 0, st, dZ, n, gp, rows_per_block, dPart);
    hipLaunchKernelGGL(k_corr_mean, dim3((unsigned)((gp + 255) / 256)), dim3(256), 0, st, dPart, nparts, n, gp, dMean);
    hipLaunchKernelGGL(k_corr_center, dim3((unsigned)((g + 255) / 256), (unsigned)std::min<int64_t>(n, 1024)), dim3(256), 0, st, dZ, n, g, gp, dMean);
    hipLaunchKernelGGL(k_corr_gemm, dim3((unsigned)pairs.size()), dim3(256), 0, st, dZ, np_, gp, dPairs, dC, 0);
    hipLaunchKernelGGL(k_corr_finish, dim3((unsigned)((g + 255) / 256), (unsigned)g), dim3(256), 0, st, dC, g, gp, 1.0 / (double)(n - 1), dOut);
    CORR_TRY(hipGetLastError());
    CORR_TRY(hipStreamSynchronize(st));
    tr.lap("corr: kernels");
#undef CORR_TRY
done:
    if (dZ) (void)dev_free_any(dZ);
    if (dC) (void)dev_free_any(dC);
    if (dMean) (void)dev_free_any(dMean);
    if (dPart) (void)dev_free_any(dPart);
    if (dPairs) (void)dev_free_any(dPairs);
    if (rc != DIMN_OK && dOut) { (void)dev_free_any(dOut); dOut = nullptr; }
    tr.lap("corr: free temporaries");
    *dOutp = dOut;
    return rc;
}
static int corr_device_ok(const char* who, int32_t device_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(DIMN_ERR_HIP, "%s: no HIP device visible", who);
    if (device_id < 0 || device_id >= ndev) return fail(DIMN_ERR_ARG, "%s: device_id out of range", who);
    HIPCHK(hipSetDevice(device_id));
    return DIMN_OK;
}

extern "C" int dimn_abs_corrcoef(int32_t device_id, const double* X, int64_t n, int64_t g, double* out) {
    if (!X || !out || n < 2 || g < 1) return fail(DIMN_ERR_ARG, "dimn_abs_corrcoef: bad argument");
    CHK(corr_device_ok("dimn_abs_corrcoef", device_id));
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double* dOut = nullptr;
    int rc = corr_on_device(X, n, g, st, &dOut);
    if (rc == DIMN_OK && hipMemcpy(out, dOut, (size_t)g * g * 8, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(DIMN_ERR_HIP, "dimn_abs_corrcoef: device-to-host copy failed");
    if (dOut) (void)dev_free_any(dOut);
    (void)hipStreamDestroy(st);
    return rc;
}

// ---- next row: setPredictors on the device (multinet.py:344-365) ---------------------------
// top-`ntop` predictors of every target over a resident |corr| matrix dCorr[g][g]
static int topk_core(const char* who, const double* dCorr, int64_t g, const int32_t* targ_pos, int32_t K, int32_t O, const int32_t* col_rank, int32_t ntop,
                     int32_t* out_idx, hipStream_t st) {
    if (ntop > 16) return fail(DIMN_ERR_UNSUP, "%s: ntop %d > 16 (use the host selection)", who, ntop);
    for (int64_t i = 0; i < (int64_t)K * O; ++i)
        if (targ_pos[i] < 0 || targ_pos[i] >= g) return fail(DIMN_ERR_ARG, "%s: target position out of range", who);
    const int NT = ntop <= 5 ? 5 : (ntop <= 8 ? 8 : 16);
    const size_t lds = ((((size_t)(g + 31) / 32) * 4 + 15) & ~(size_t)15) + (size_t)256 * NT * 16 + 64;
    if (lds > 160 * 1024) return fail(DIMN_ERR_UNSUP, "%s: %lld candidate genes exceed the LDS bitmap", who, (long long)g);
    int32_t *dT = nullptr, *dR = nullptr, *dI = nullptr;
    int rc = DIMN_OK;
#define SEL_TRY(expr) do { hipError_t e_ = (expr); if (rc == DIMN_OK && e_ != hipSuccess) rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
    SEL_TRY(dev_malloc_bytes((void**)&dT, (size_t)K * O * 4));
    SEL_TRY(dev_malloc_bytes((void**)&dR, (size_t)g * 4));
    SEL_TRY(dev_malloc_bytes((void**)&dI, (size_t)K * O * ntop * 4));
    if (rc == DIMN_OK) {
        SEL_TRY(hipMemcpyAsync(dT, targ_pos, (size_t)K * O * 4, hipMemcpyHostToDevice, st));
        SEL_TRY(hipMemcpyAsync(dR, col_rank, (size_t)g * 4, hipMemcpyHostToDevice, st));
        const dim3 grid((unsigned)O, (unsigned)K);
        if (NT == 5) {
            (void)hipFuncSetAttribute((const void*)k_corr_topk<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_corr_topk<5>, grid, dim3(256), lds, st, dCorr, g, dT, O, dR, dI, ntop);
        } else if (NT == 8) {
            (void)hipFuncSetAttribute((const void*)k_corr_topk<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_corr_topk<8>, grid, dim3(256), lds, st, dCorr, g, dT, O, dR, dI, ntop);
        } else {
            (void)hipFuncSetAttribute((const void*)k_corr_topk<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_corr_topk<16>, grid, dim3(256), lds, st, dCorr, g, dT, O, dR, dI, ntop);
        }
        SEL_TRY(hipGetLastError());
        SEL_TRY(hipMemcpyAsync(out_idx, dI, (size_t)K * O * ntop * 4, hipMemcpyDeviceToHost, st));
        SEL_TRY(hipStreamSynchronize(st));
    }
#undef SEL_TRY
    if (dT) (void)dev_free_any(dT);
    if (dR) (void)dev_free_any(dR);
    if (dI) (void)dev_free_any(dI);
    return rc;
}
static int select_predictors_core(const char* who, int32_t device_id, const double* X, const CorrDevSrc* src, int64_t n, int64_t g, const int32_t* targ_pos,
                                  int32_t K, int32_t O, const int32_t* col_rank, int32_t ntop, int32_t* out_idx, hipStream_t st) {
    if (ntop > 16) return fail(DIMN_ERR_UNSUP, "%s: ntop %d > 16 (use the host selection)", who, ntop);
    double* dOut = nullptr;
    int rc = corr_on_device(X, n, g, st, &dOut, src);
    if (rc == DIMN_OK) rc = topk_core(who, dOut, g, targ_pos, K, O, col_rank, ntop, out_idx, st);
    if (dOut) (void)dev_free_any(dOut);
    return rc;
}

extern "C" int dimn_select_predictors(int32_t device_id, const double* X, int64_t n, int64_t g, const int32_t* targ_pos, int32_t K, int32_t O,
                                      const int32_t* col_rank, int32_t ntop, int32_t* out_idx) {
    if (!X || !targ_pos || !col_rank || !out_idx || n < 2 || g < 1 || K < 1 || O < 1 || ntop < 1)
        return fail(DIMN_ERR_ARG, "dimn_select_predictors: bad argument");
    CHK(corr_device_ok("dimn_select_predictors", device_id));
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int rc = select_predictors_core("dimn_select_predictors", device_id, X, nullptr, n, g, targ_pos, K, O, col_rank, ntop, out_idx, st);
    (void)hipStreamDestroy(st);
    return rc;
}

// ---- the raw counts resident on the device (extension of the drop-in; the reference passes the same 8 GB frame through numpy four
// times: multinet.py:191 var/mean, :20-34 corrcoef, :216 log1p, :292-303 restore).  dimn_counts_create uploads the count matrix ONCE
// as float32 -- host threads convert the float64 frame row block by row block into pinned buffers and verify on the way that every
// value is a non-negative integer <= 2^22 (exact in float32; anything else: DIMN_ERR_UNSUP, the caller keeps the host path) -- and
// every later stage reads it there: the correlation (converted to float64 on the device), log1p through a table the caller
// computed with numpy (bit-identical to np.log1p(raw).astype(float32)), and predict()'s restore / max against the observed counts.
static inline uint64_t counts_mix(uint64_t x) {      // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}
// one host pass over rows [r0, r1): optional float32 copy, maximum, position-dependent checksum of the float64 bit patterns,
// and whether every value is a count (non-negative integer <= 2^22).
struct RowScan { double m; uint64_t h; bool fine; };
// One row, plain C++: the definition of the pass (and the tail of the vector form below).  ST = double, or int64_t -- what pd.read_csv
// makes of a count matrix: every quantity is that of the float64 frame holding the same numbers ((double)v: its bit pattern is hashed).
template <typename ST>
static inline void counts_scan_scalar(const ST* src, float* out, int64_t j0, int64_t j1, uint64_t base, RowScan& rs) {
    double m = rs.m; uint64_t h = rs.h; bool fine = rs.fine;
    for (int64_t j = j0; j < j1; ++j) {
        const double x = (double)src[j];
        uint64_t bits;
        memcpy(&bits, &x, 8);
        h += counts_mix(bits + 0x9e3779b97f4a7c15ull * (base + (uint64_t)j + 1));
        m = x > m ? x : m;
        // a count: in [0, 2^22], integral, not -0.0.  The range test comes first, so the conversion below only ever sees values it is
        // defined for (NaN / Inf / huge values take the 0.5 and fail); no libm call per element (trunc() was one on plain x86-64)
        const bool in_range = x >= 0.0 && x <= 4194304.0;
        const double xr = in_range ? x : 0.5;
        fine &= in_range & ((double)(int32_t)xr == xr) & ((bits >> 63) == 0);
        if (out) out[j] = (float)x;
    }
    rs.m = m; rs.h = h; rs.fine = fine;
}
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__AVX2__)
// four source elements as doubles; false when the quad cannot take the vector path (int64 values outside [0, 2^22]: AVX2 has no
// int64 -> double conversion, in-range values convert exactly through their low 32 bits; the caller handles such a quad in plain C++)
static inline bool counts_load4(const double* p, __m256d& x) { x = _mm256_loadu_pd(p); return true; }
static inline bool counts_load4(const int64_t* p, __m256d& x) {
    const __m256i v = _mm256_loadu_si256((const __m256i*)p);
    const __m256i bad = _mm256_or_si256(_mm256_cmpgt_epi64(v, _mm256_set1_epi64x(4194304)), _mm256_cmpgt_epi64(_mm256_setzero_si256(), v));
    if (!_mm256_testz_si256(bad, bad)) return false;
    x = _mm256_cvtepi32_pd(_mm256_castsi256_si128(_mm256_permutevar8x32_epi32(v, _mm256_setr_epi32(0, 2, 4, 6, 0, 0, 0, 0))));
    return true;
}
// The same row four elements at a time (round 5): the scalar loop is bound by its arithmetic -- two 64-bit multiplies of the splitmix round
// per element, which AVX2 has no instruction for and the compiler therefore leaves scalar: 1.45 ns per element and core on the GPU boxes'
// hosts against 0.66 here (profiles/r05_dropin_host_side.txt).  The multiplies are three 32 x 32 -> 64 products each (`vpmuludq`), sums are
// per lane (addition mod 2^64 commutes: the same checksum to the bit), the range / integrality tests are compares and one truncating
// conversion, the sign test is an OR over all bit patterns.  tests/test_abi.py checks the checksum against a numpy restatement of its definition.
static inline __m256i counts_mul64(__m256i v, __m256i clo, __m256i chi) {
    const __m256i lo = _mm256_mul_epu32(v, clo);
    const __m256i cross = _mm256_add_epi64(_mm256_mul_epu32(_mm256_srli_epi64(v, 32), clo), _mm256_mul_epu32(v, chi));
    return _mm256_add_epi64(lo, _mm256_slli_epi64(cross, 32));
}
template <bool OUT, typename ST>
static inline void counts_scan_row(const ST* src, float* out, int64_t g, uint64_t base, RowScan& rs) {
    const uint64_t GOLD = 0x9e3779b97f4a7c15ull, C1 = 0xbf58476d1ce4e5b9ull, C2 = 0x94d049bb133111ebull;
    const __m256i c1lo = _mm256_set1_epi64x((long long)(C1 & 0xffffffffull)), c1hi = _mm256_set1_epi64x((long long)(C1 >> 32));
    const __m256i c2lo = _mm256_set1_epi64x((long long)(C2 & 0xffffffffull)), c2hi = _mm256_set1_epi64x((long long)(C2 >> 32));
    __m256i kv = _mm256_set_epi64x((long long)(GOLD * (base + 4)), (long long)(GOLD * (base + 3)), (long long)(GOLD * (base + 2)), (long long)(GOLD * (base + 1)));
    const __m256i kstep = _mm256_set1_epi64x((long long)(GOLD * 4));
    __m256i hv = _mm256_setzero_si256(), orv = _mm256_setzero_si256();
    __m256d mv = _mm256_set1_pd(-INFINITY), goodv = _mm256_castsi256_pd(_mm256_set1_epi64x(-1));
    const __m256d zero = _mm256_setzero_pd(), top = _mm256_set1_pd(4194304.0), half = _mm256_set1_pd(0.5);
    const bool nt_store = OUT && (((uintptr_t)out) & 15) == 0;      // the float32 copy goes to a pinned bounce buffer the DMA engine reads next: streaming stores (no read-for-ownership)
    int64_t j = 0;
    for (; j + 4 <= g; j += 4) {
        __m256d x;
        const bool quad = counts_load4(src + j, x);
        const __m256i key = kv;
        kv = _mm256_add_epi64(kv, kstep);
        if (!quad) { counts_scan_scalar(src, OUT ? out : nullptr, j, j + 4, base, rs); continue; }
        const __m256i bits = _mm256_castpd_si256(x);
        __m256i v = _mm256_add_epi64(bits, key);
        v = _mm256_xor_si256(v, _mm256_srli_epi64(v, 30)); v = counts_mul64(v, c1lo, c1hi);
        v = _mm256_xor_si256(v, _mm256_srli_epi64(v, 27)); v = counts_mul64(v, c2lo, c2hi);
        v = _mm256_xor_si256(v, _mm256_srli_epi64(v, 31));
        hv = _mm256_add_epi64(hv, v);
        mv = _mm256_max_pd(x, mv);                                  // (x NaN: mv stays, like `x > m ? x : m`)
        const __m256d in = _mm256_and_pd(_mm256_cmp_pd(x, zero, _CMP_GE_OQ), _mm256_cmp_pd(x, top, _CMP_LE_OQ));
        const __m256d xr = _mm256_blendv_pd(half, x, in);
        const __m256d back = _mm256_cvtepi32_pd(_mm256_cvttpd_epi32(xr));
        goodv = _mm256_and_pd(goodv, _mm256_and_pd(in, _mm256_cmp_pd(back, xr, _CMP_EQ_OQ)));
        orv = _mm256_or_si256(orv, bits);
        if (OUT) { if (nt_store) _mm_stream_ps(out + j, _mm256_cvtpd_ps(x)); else _mm_storeu_ps(out + j, _mm256_cvtpd_ps(x)); }
    }
    alignas(32) uint64_t hl[4], ol[4];
    alignas(32) double ml[4];
    _mm256_store_si256((__m256i*)hl, hv); _mm256_store_si256((__m256i*)ol, orv); _mm256_store_pd(ml, mv);
    rs.h += hl[0] + hl[1] + hl[2] + hl[3];
    for (int i = 0; i < 4; ++i) rs.m = ml[i] > rs.m ? ml[i] : rs.m;
    rs.fine &= _mm256_movemask_pd(goodv) == 0xf && (((ol[0] | ol[1] | ol[2] | ol[3]) >> 63) == 0);
    counts_scan_scalar(src, OUT ? out : nullptr, j, g, base, rs);
}
#else
template <bool OUT, typename ST>
static inline void counts_scan_row(const ST* src, float* out, int64_t g, uint64_t base, RowScan& rs) { counts_scan_scalar(src, OUT ? out : nullptr, 0, g, base, rs); }
#endif
template <typename ST>
static uint64_t counts_row_checksum(const ST* src, int64_t g, uint64_t base) {
    RowScan rs{-INFINITY, 0, true};
    counts_scan_row<false>(src, nullptr, g, base, rs);
    return rs.h;
}
template <typename ST>
static void counts_scan(const ST* raw, int64_t g, int64_t r0, int64_t r1, float* dst, double* vmax, uint64_t* sum, int* ok) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int64_t rows = r1 - r0;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw ? hw / 2 : 8, 64), rows * g / (1 << 20)));
    std::vector<double> mx((size_t)nt, -INFINITY);
    std::vector<uint64_t> cs((size_t)nt, 0);
    std::vector<int> good((size_t)nt, 1);
    auto work = [&](int t) {
        const int64_t a = r0 + rows * t / nt, b = r0 + rows * (t + 1) / nt;
        RowScan rs{-INFINITY, 0, true};
        for (int64_t i = a; i < b; ++i) {
            const ST* src = raw + i * g;
            const uint64_t base = (uint64_t)i * (uint64_t)g;
            if (dst) counts_scan_row<true>(src, dst + (i - r0) * g, g, base, rs);
            else counts_scan_row<false>(src, nullptr, g, base, rs);
        }
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__AVX2__)
        _mm_sfence();
#endif
        mx[(size_t)t] = rs.m; cs[(size_t)t] = rs.h; good[(size_t)t] = rs.fine ? 1 : 0;
    };
    host_pool().run(nt, work);
    for (int t = 0; t < nt; ++t) { *vmax = mx[(size_t)t] > *vmax ? mx[(size_t)t] : *vmax; *sum += cs[(size_t)t]; *ok &= good[(size_t)t]; }
}
template <typename ST>
static int counts_checksum_impl(const ST* raw, int64_t n, int64_t g, uint64_t* checksum) {
    if (!raw || !checksum || n < 1 || g < 1) return fail(DIMN_ERR_ARG, "dimn_counts_checksum: bad argument");
    double vmax = -INFINITY; uint64_t sum = 0; int ok = 1;
    counts_scan(raw, g, 0, n, nullptr, &vmax, &sum, &ok);
    *checksum = sum;
    return DIMN_OK;
}
extern "C" int dimn_counts_checksum(const double* raw, int64_t n, int64_t g, uint64_t* checksum) { return counts_checksum_impl(raw, n, g, checksum); }
extern "C" int dimn_counts_checksum_typed(const void* raw, int32_t dtype, int64_t n, int64_t g, uint64_t* checksum) {
    if (dtype == DIMN_DTYPE_F64) return counts_checksum_impl((const double*)raw, n, g, checksum);
    if (dtype == DIMN_DTYPE_I64) return counts_checksum_impl((const int64_t*)raw, n, g, checksum);
    return fail(DIMN_ERR_ARG, "dimn_counts_checksum_typed: dtype must be DIMN_DTYPE_F64 or DIMN_DTYPE_I64");
}
extern "C" int dimn_counts_destroy(dimn_counts c) {
    if (!c) return DIMN_OK;
    (void)hipSetDevice(c->device);
    if (c->d) (void)dev_free_any(c->d);
    if (c->d_corr) (void)dev_free_any(c->d_corr);
    delete c;
    return DIMN_OK;
}
template <typename ST>
static int counts_create_impl(int32_t device_id, const ST* raw, int64_t n, int64_t g, double* vmax_out, uint64_t* checksum_out, dimn_counts* out) {
    if (!raw || !out || n < 1 || g < 1) return fail(DIMN_ERR_ARG, "dimn_counts_create: bad argument");
    CHK(corr_device_ok("dimn_counts_create", device_id));
    dimn_counts c = new dimn_counts_s();
    c->device = device_id; c->n = n; c->g = g; c->vmax = -INFINITY;
    float* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
    int rc = DIMN_OK, ok = 1;
    PinLease pins;
    const int64_t blk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(128u << 20) / (g * 4)));
#define CNT_TRY(expr) do { hipError_t e_ = (expr); if (rc == DIMN_OK && e_ != hipSuccess) rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)
    CNT_TRY(dev_malloc_bytes((void**)&c->d, (size_t)n * g * 4));
    CNT_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    {
        const char* why = "";
        if (rc == DIMN_OK && !pins.take(2, std::max<size_t>((size_t)(128u << 20), (size_t)blk * g * 4), &why)) rc = fail(DIMN_ERR_HIP, "dimn_counts_create: pinning the bounce buffers failed: %s", why);
    }
    for (int b = 0; b < 2; ++b) {
        pin[b] = (float*)pins.buf[b];
        CNT_TRY(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
    }
    int64_t bi = 0;
    for (int64_t r0 = 0; r0 < n && rc == DIMN_OK && ok; r0 += blk, ++bi) {
        const int b = (int)(bi & 1);
        const int64_t nr = std::min(blk, n - r0);
        if (bi >= 2) CNT_TRY(hipEventSynchronize(ev[b]));
        if (rc != DIMN_OK) break;
        counts_scan(raw, g, r0, r0 + nr, pin[b], &c->vmax, &c->checksum, &ok);
        CNT_TRY(hipMemcpyAsync(c->d + r0 * g, pin[b], (size_t)nr * g * 4, hipMemcpyHostToDevice, st));
        CNT_TRY(hipEventRecord(ev[b], st));
    }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
#undef CNT_TRY
    for (int b = 0; b < 2; ++b) if (ev[b]) (void)hipEventDestroy(ev[b]);
    if (rc == DIMN_OK && !ok) rc = fail(DIMN_ERR_UNSUP, "dimn_counts_create: the matrix holds values that are not counts (non-negative integers <= 2^22)");
    if (rc != DIMN_OK) { dimn_counts_destroy(c); return rc; }
    if (vmax_out) *vmax_out = c->vmax;
    if (checksum_out) *checksum_out = c->checksum;
    *out = c;
    return DIMN_OK;
}
extern "C" int dimn_counts_create(int32_t device_id, const double* raw, int64_t n, int64_t g, double* vmax_out, uint64_t* checksum_out, dimn_counts* out) {
    return counts_create_impl(device_id, raw, n, g, vmax_out, checksum_out, out);
}
extern "C" int dimn_counts_create_typed(int32_t device_id, const void* raw, int32_t dtype, int64_t n, int64_t g, double* vmax_out, uint64_t* checksum_out, dimn_counts* out) {
    if (dtype == DIMN_DTYPE_F64) return counts_create_impl(device_id, (const double*)raw, n, g, vmax_out, checksum_out, out);
    if (dtype == DIMN_DTYPE_I64) return counts_create_impl(device_id, (const int64_t*)raw, n, g, vmax_out, checksum_out, out);
    return fail(DIMN_ERR_ARG, "dimn_counts_create_typed: dtype must be DIMN_DTYPE_F64 or DIMN_DTYPE_I64");
}
// |corr| of the pool columns from the resident counts on the int8 matrix cores (dimn_counts_dev.h part 2); *dOutp: [pool_n][pool_n] float64
static int corr_counts_i8(dimn_counts c, const int32_t* dCols, int64_t pool_n, hipStream_t st, double** dOutp) {
    const int P = c->vmax < 256.0 ? 1 : 2;
    const int64_t n = c->n, gp = (pool_n + CI8_BT - 1) / CI8_BT * CI8_BT, KC = (n + CI8_KS - 1) / CI8_KS, plane_bytes = gp * KC * CI8_KS;
    const int nb = (int)(gp / CI8_BT);
    int8_t* dPlanes = nullptr;
    long long *dSums = nullptr, *dC = nullptr;
    double* dRoot = nullptr;
    int2* dPairs = nullptr;
    int rc = DIMN_OK;
    const std::vector<int2> pairs = xcd_tiled_pairs(nb);
    const int nblk = (int)std::min<int64_t>(64, (n + 255) / 256);
    const size_t lds = (size_t)CI8_NBUF * 16 * P * 1024;
#define CI8_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } } while (0)
    Trace tr;
    CI8_TRY(dev_malloc_bytes((void**)&dPlanes, (size_t)P * plane_bytes));
    CI8_TRY(dev_malloc_bytes((void**)&dC, (size_t)pool_n * pool_n * 8));
    CI8_TRY(dev_malloc_bytes((void**)&dSums, (size_t)gp * 8));
    CI8_TRY(dev_malloc_bytes((void**)&dRoot, (size_t)gp * 8));
    CI8_TRY(dev_malloc_bytes((void**)&dPairs, pairs.size() * sizeof(int2)));
    CI8_TRY(hipMemsetAsync(dSums, 0, (size_t)gp * 8, st));
    CI8_TRY(hipMemcpyAsync(dPairs, pairs.data(), pairs.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    tr.lap("corr i8: device allocations");
    if (P == 1) {
        CI8_TRY(hipFuncSetAttribute((const void*)k_ci8_gemm<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_ci8_planes<1>, dim3((unsigned)(gp / 64), (unsigned)KC), dim3(256), 0, st, c->d, c->g, dCols, n, pool_n, KC, dPlanes, plane_bytes);
        hipLaunchKernelGGL(k_ci8_colsum, dim3((unsigned)((pool_n + 255) / 256), (unsigned)nblk), dim3(256), 0, st, c->d, c->g, dCols, n, pool_n, (n + nblk - 1) / nblk, 128ll, dSums);
        hipLaunchKernelGGL(k_ci8_gemm<1>, dim3((unsigned)pairs.size()), dim3(256), lds, st, dPlanes, plane_bytes, KC, dPairs, dC, pool_n, pool_n);
    } else {
        CI8_TRY(hipFuncSetAttribute((const void*)k_ci8_gemm<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_ci8_planes<2>, dim3((unsigned)(gp / 64), (unsigned)KC), dim3(256), 0, st, c->d, c->g, dCols, n, pool_n, KC, dPlanes, plane_bytes);
        hipLaunchKernelGGL(k_ci8_colsum, dim3((unsigned)((pool_n + 255) / 256), (unsigned)nblk), dim3(256), 0, st, c->d, c->g, dCols, n, pool_n, (n + nblk - 1) / nblk, 32896ll, dSums);
        hipLaunchKernelGGL(k_ci8_gemm<2>, dim3((unsigned)pairs.size()), dim3(256), lds, st, dPlanes, plane_bytes, KC, dPairs, dC, pool_n, pool_n);
    }
    hipLaunchKernelGGL(k_ci8_diag, dim3((unsigned)((pool_n + 255) / 256)), dim3(256), 0, st, dC, pool_n, dSums, n, pool_n, dRoot);
    hipLaunchKernelGGL(k_ci8_finish, dim3((unsigned)((pool_n + 255) / 256), (unsigned)pool_n), dim3(256), 0, st, dC, pool_n, dSums, dRoot, n, pool_n);
    CI8_TRY(hipGetLastError());
    CI8_TRY(hipStreamSynchronize(st));
    tr.lap("corr i8: kernels");
#undef CI8_TRY
done:
    if (dPlanes) (void)dev_free_any(dPlanes);
    if (dSums) (void)dev_free_any(dSums);
    if (dRoot) (void)dev_free_any(dRoot);
    if (dPairs) (void)dev_free_any(dPairs);
    if (rc != DIMN_OK) { if (dC) (void)dev_free_any(dC); return rc; }
    *dOutp = (double*)dC;
    return DIMN_OK;
}
// DataFrame.mean() / .var() / column extremes of the resident counts, to the bit (dimn_counts_dev.h part 1); each output [g] or NULL
extern "C" int dimn_counts_gene_stats(dimn_counts c, double* mean, double* var, double* cmin, double* cmax) {
    if (!c || c->n < 2) return fail(DIMN_ERR_ARG, "dimn_counts_gene_stats: bad argument");
    CHK(corr_device_ok("dimn_counts_gene_stats", c->device));
    const int64_t n = c->n, g = c->g;
    const int chunks = (int)((n + 8191) / 8192);
    double *dSum = nullptr, *dMin = nullptr, *dMax = nullptr, *dPart = nullptr, *dAvg = nullptr, *dVar = nullptr;
    hipStream_t st = nullptr;
    int rc = DIMN_OK;
#define GS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(DIMN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); goto done; } } while (0)
    GS_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    GS_TRY(dev_malloc_bytes((void**)&dSum, (size_t)g * 8 * 5));
    dMin = dSum + g; dMax = dMin + g; dAvg = dMax + g; dVar = dAvg + g;
    GS_TRY(dev_malloc_bytes((void**)&dPart, (size_t)chunks * g * 8));
    hipLaunchKernelGGL(k_cnt_seqsum, dim3((unsigned)((g + 63) / 64)), dim3(64), 0, st, c->d, n, g, dSum, dMin, dMax);
    hipLaunchKernelGGL(k_cnt_div, dim3((unsigned)((g + 255) / 256)), dim3(256), 0, st, dSum, g, (double)n);
    if (var) {
        hipLaunchKernelGGL(k_cnt_pairwise<false>, dim3((unsigned)((g + 63) / 64), (unsigned)chunks), dim3(64), 0, st, c->d, n, g, (const double*)nullptr, dPart);
        hipLaunchKernelGGL(k_cnt_chunks, dim3((unsigned)((g + 255) / 256)), dim3(256), 0, st, dPart, chunks, g, (double)n, dAvg);
        hipLaunchKernelGGL(k_cnt_pairwise<true>, dim3((unsigned)((g + 63) / 64), (unsigned)chunks), dim3(64), 0, st, c->d, n, g, (const double*)dAvg, dPart);
        hipLaunchKernelGGL(k_cnt_chunks, dim3((unsigned)((g + 255) / 256)), dim3(256), 0, st, dPart, chunks, g, (double)(n - 1), dVar);
    }
    GS_TRY(hipGetLastError());
    if (mean) GS_TRY(hipMemcpyAsync(mean, dSum, (size_t)g * 8, hipMemcpyDeviceToHost, st));
    if (var) GS_TRY(hipMemcpyAsync(var, dVar, (size_t)g * 8, hipMemcpyDeviceToHost, st));
    if (cmin) GS_TRY(hipMemcpyAsync(cmin, dMin, (size_t)g * 8, hipMemcpyDeviceToHost, st));
    if (cmax) GS_TRY(hipMemcpyAsync(cmax, dMax, (size_t)g * 8, hipMemcpyDeviceToHost, st));
    GS_TRY(hipStreamSynchronize(st));
#undef GS_TRY
done:
    if (dSum) (void)dev_free_any(dSum);
    if (dPart) (void)dev_free_any(dPart);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}
// The same selection as two calls, so that the matrix product (which needs only the candidate pool) can run while the host is
// still ranking genes: dimn_counts_corr leaves |corr| of the pool on the device, dimn_counts_topk selects from it and frees it.
extern "C" int dimn_counts_corr(dimn_counts c, const int32_t* pool_cols, int64_t pool_n) {
    if (!c || !pool_cols || pool_n < 1 || c->n < 2) return fail(DIMN_ERR_ARG, "dimn_counts_corr: bad argument");
    if (pool_n > 65535) return fail(DIMN_ERR_UNSUP, "dimn_counts_corr: more than 65535 candidate genes");
    for (int64_t j = 0; j < pool_n; ++j) if (pool_cols[j] < 0 || pool_cols[j] >= c->g) return fail(DIMN_ERR_ARG, "dimn_counts_corr: pool column out of range");
    CHK(corr_device_ok("dimn_counts_corr", c->device));
    if (c->d_corr) { (void)dev_free_any(c->d_corr); c->d_corr = nullptr; c->corr_g = 0; }
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int32_t* dCols = nullptr;
    int rc = DIMN_OK;
    if (dev_malloc_bytes((void**)&dCols, (size_t)pool_n * 4) != hipSuccess || hipMemcpyAsync(dCols, pool_cols, (size_t)pool_n * 4, hipMemcpyHostToDevice, st) != hipSuccess)
        rc = fail(DIMN_ERR_HIP, "dimn_counts_corr: pool upload failed");
    if (rc == DIMN_OK) {
        // integer counts below 65536: exactly, on the int8 matrix cores (dimn_counts_dev.h); anything else in float64 (dimn_corr.h)
        const char* e = getenv("DIMN_CORR_I8");
        if (c->vmax <= 65535.0 && !(e && atoi(e) == 0)) rc = corr_counts_i8(c, dCols, pool_n, st, &c->d_corr);
        else {
            const CorrDevSrc src{c->d, c->g, dCols};
            rc = corr_on_device(nullptr, c->n, pool_n, st, &c->d_corr, &src);
        }
        if (rc == DIMN_OK) c->corr_g = pool_n;
    }
    if (dCols) (void)dev_free_any(dCols);
    (void)hipStreamDestroy(st);
    return rc;
}
extern "C" int dimn_counts_topk(dimn_counts c, const int32_t* targ_pos, int32_t K, int32_t O, const int32_t* col_rank, int32_t ntop, int32_t* out_idx) {
    if (!c || !targ_pos || !col_rank || !out_idx || K < 1 || O < 1 || ntop < 1) return fail(DIMN_ERR_ARG, "dimn_counts_topk: bad argument");
    if (!c->d_corr) return fail(DIMN_ERR_STATE, "dimn_counts_topk: dimn_counts_corr first");
    CHK(corr_device_ok("dimn_counts_topk", c->device));
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int rc = topk_core("dimn_counts_topk", c->d_corr, c->corr_g, targ_pos, K, O, col_rank, ntop, out_idx, st);
    (void)hipStreamDestroy(st);
    (void)dev_free_any(c->d_corr); c->d_corr = nullptr; c->corr_g = 0;
    return rc;
}
// Give the |corr| matrix of dimn_counts_corr back without selecting from it (the caller's selection took another path).
extern "C" int dimn_counts_corr_drop(dimn_counts c) {
    if (!c) return fail(DIMN_ERR_ARG, "dimn_counts_corr_drop: null argument");
    if (c->d_corr) {
        (void)hipSetDevice(c->device);
        dev_free_any(c->d_corr); c->d_corr = nullptr; c->corr_g = 0;
    }
    return DIMN_OK;
}
// setPredictors over the resident counts: the candidate pool = columns pool_cols[pool_n] of the count matrix
extern "C" int dimn_counts_select_predictors(dimn_counts c, const int32_t* pool_cols, int64_t pool_n, const int32_t* targ_pos, int32_t K, int32_t O,
                                             const int32_t* col_rank, int32_t ntop, int32_t* out_idx) {
    if (!c || !pool_cols || !targ_pos || !col_rank || !out_idx || pool_n < 1 || K < 1 || O < 1 || ntop < 1 || c->n < 2)
        return fail(DIMN_ERR_ARG, "dimn_counts_select_predictors: bad argument");
    for (int64_t j = 0; j < pool_n; ++j) if (pool_cols[j] < 0 || pool_cols[j] >= c->g) return fail(DIMN_ERR_ARG, "dimn_counts_select_predictors: pool column out of range");
    CHK(corr_device_ok("dimn_counts_select_predictors", c->device));
    const int rc = dimn_counts_corr(c, pool_cols, pool_n);
    return rc != DIMN_OK ? rc : dimn_counts_topk(c, targ_pos, K, O, col_rank, ntop, out_idx);
}
// (tests / diagnostics) the |corr| matrix dimn_counts_corr left on the device: out[corr_g][corr_g]
extern "C" int dimn_counts_corr_read(dimn_counts c, double* out, int64_t pool_n) {
    if (!c || !out) return fail(DIMN_ERR_ARG, "dimn_counts_corr_read: bad argument");
    if (!c->d_corr || c->corr_g != pool_n) return fail(DIMN_ERR_STATE, "dimn_counts_corr_read: dimn_counts_corr of %lld columns first", (long long)pool_n);
    CHK(corr_device_ok("dimn_counts_corr_read", c->device));
    HIPCHK(hipMemcpy(out, c->d_corr, (size_t)pool_n * pool_n * 8, hipMemcpyDeviceToHost));
    return DIMN_OK;
}
// the log1p matrix of the engine from the resident counts: norm[i][j] = lut[(int)counts[i][j]], lut = float32(log1p(0..vmax)) as numpy computes it
__global__ __launch_bounds__(256) void k_counts_lut(const float* __restrict__ counts, const float* __restrict__ lut, int64_t lut_n, int64_t total, float* __restrict__ norm) {
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < total; e += (int64_t)gridDim.x * 1024) {
        if (e + 4 <= total) {
            const f32x4 v = *(const f32x4*)(counts + e);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int64_t i = (int64_t)v[r]; o[r] = lut[i < 0 ? 0 : (i < lut_n ? i : lut_n - 1)]; }
            *(f32x4*)(norm + e) = o;
        } else {
            for (int64_t q = e; q < total; ++q) { const int64_t i = (int64_t)counts[q]; norm[q] = lut[i < 0 ? 0 : (i < lut_n ? i : lut_n - 1)]; }
        }
    }
}
extern "C" int dimn_set_matrix_counts(dimn_handle h, dimn_counts c, const float* lut, int64_t lut_n) {
    if (h) h->counts = nullptr;                    // (a failed rebind must not leave the handle pointing at the previous counts object)
    if (!h || !c || !lut || lut_n < 1) return fail(DIMN_ERR_ARG, "dimn_set_matrix_counts: bad argument");
    if (c->device != h->cfg.device_id) return fail(DIMN_ERR_ARG, "dimn_set_matrix_counts: the counts live on another device");
    if ((double)lut_n <= c->vmax) return fail(DIMN_ERR_ARG, "dimn_set_matrix_counts: the table has %lld entries, the largest count is %.0f", (long long)lut_n, c->vmax);
    if (c->n > 0x7fffffffLL || c->g > 0x7fffffffLL) return fail(DIMN_ERR_UNSUP, "dimn_set_matrix_counts: dimension exceeds int32");
    CHK(use_device(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    Trace tr;
    if (!h->d_norm || h->n != c->n || h->g != c->g) {
        DEV_FREE(h->d_norm);
        CHK(dev_alloc(&h->d_norm, (size_t)c->n * c->g));
    }
    tr.lap("set_matrix_counts: matrix allocation");
    if (c->n != h->n) { h->n_tr = 0; h->n_val = 0; h->train_rows.clear(); h->val_rows.clear(); }
    h->n = c->n; h->g = c->g; h->gathered = false; h->streamed = false;
    float* dLut = nullptr;
    CHK(dev_alloc(&dLut, (size_t)lut_n));
    int rc = DIMN_OK;
    if (hipMemcpyAsync(dLut, lut, (size_t)lut_n * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = fail(DIMN_ERR_HIP, "dimn_set_matrix_counts: table upload failed");
    if (rc == DIMN_OK) {
        hipLaunchKernelGGL(k_counts_lut, dim3(4096), dim3(256), 0, h->stream, (const float*)c->d, (const float*)dLut, lut_n, c->n * c->g, h->d_norm);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(DIMN_ERR_HIP, "dimn_set_matrix_counts: table kernel failed");
    }
    (void)dev_free_any(dLut);
    tr.lap("set_matrix_counts: log1p table kernel");
    if (rc == DIMN_OK) h->counts = c;
    return rc;
}
