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
#include "dimn_host_build.inc"
#include "dimn_host_data.inc"
#include "dimn_host_train.inc"
#include "dimn_host_predict.inc"
#include "dimn_host_comm.inc"
#include "dimn_host_edges.inc"
#include "dimn_host_planning.inc"
