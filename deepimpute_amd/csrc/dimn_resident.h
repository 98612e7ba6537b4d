// dimn_resident.h -- the REGISTER-RESIDENT epoch kernel of libdimn (gfx950).
//
// When a GPU owns only a few sub-networks (8-GPU sharding of BASELINE configs[3]: K_local = 5), the whole
// optimiser state of the rank -- W, m, v of both layers, ~45 MB -- fits the chip's 128 MB of vector registers.
// One persistent launch then runs a WHOLE EPOCH of model.fit (reference deepimpute/multinet.py:238-244):
// every workgroup keeps its slice of W1/m/v (and of W2/m/v) in registers for all ~743 optimiser steps, so
// no optimiser state moves through HBM at all; per step the workgroups of one sub-net exchange only small
// activation tiles through the memory-side cache (write-through stores validated by the readers, no grid-wide barrier:
// sub-nets share nothing, multinet.py:132-146).  The four launches per step of the streaming path (RED, MF, MB, B1F1 --
// each latency-bound at this size) disappear.
//
// Decomposition of sub-net k over G = 16*S1 workgroups (512 threads = 8 waves x 256 VGPRs, one per CU):
//   role 1 (all G):      workgroup (ht, s) owns hidden tile ht (16 units) of the D-split s: the W1 tiles
//                        [chunk c in split s][ht], T1 tiles per wave.  Per step: dA tile, W1 gradient + Adam in
//                        registers, forward partial P[64][16] of the NEXT batch (k-slot trick, dimn_kernels.h).
//   role 2 (first OT):   workgroup ot owns the W2 column block [all 16 hidden tiles][output tile ot], two tiles
//                        per wave.  Per step: Dd = dropout(relu(sum_s P + b1)), Z tile, softplus, wMSE, dZ,
//                        Adam(b2), dD partial [64][256] over its 16 outputs (published), then the W2 gradient + Adam on the
//                        column block, which lives in LDS.
// Exchange per step and sub-net (the only inter-workgroup traffic):
//   P partials   G x [64][16]  (role 1 -> role 2, and to the S1 siblings of a hidden tile for the relu gate);
//                the split-0 workgroup of a hidden tile adds b1 to its partial, so A = sum_s P_s everywhere
//   dD partials  OT x 16 x [64][16]  (role 2 -> role 1: workgroup (ht, s) sums tile ht over the OT producers)
// Both through 16-byte sc0 sc1 (write-through) stores and sc0 sc1 loads (L1 is never refreshed by other CUs' stores, the L2s
// of different XCDs are not coherent: MI355X guide, Guideline 16).  Hand-off protocol (DIMN_RES_SENT, the default): a slot
// is filled with all-ones words before its producer writes it, so a consumer sees in the DATA whether a piece has arrived
// (res_poll on one piece per producer tile, then res_fix on every piece where it is consumed): no store drain, no barrier,
// no arrival counter on the producer side.  Slots of step t: P in t % 3, dD in t % 2; a producer re-arms the slot whose
// readers have provably finished (see the step loop).  DIMN_RES_SENT=0 keeps the first protocol -- stores, vmcnt(0) drain,
// one relaxed agent-scope counter per sub-net and direction, one-lane polls -- which costs ~8 us more per step.
// Every wait is bounded in wall-clock time; a timeout raises an abort word, the host then restores the pre-epoch state and
// re-runs the epoch on the streaming kernels, so a lost workgroup can neither hang the GPU nor fail a fit.  A handle with more sub-nets than fit at once runs one launch per GROUP of sub-nets (ResParams.k0).
//
// All arithmetic is the exact-fp32 path of the streaming kernels (v_mfma_f32_16x16x4_f32, adam4, the Philox
// dropout streams, softplus_sigmoid_fast): only summation orders differ.
#pragma once
#include "dimn_kernels.h"

#define DIMN_RES_THREADS 512
#define DIMN_RES_LDD 260          // LDS row stride of Dd (as k_mid_fused)
#define DIMN_RES_SPIN_LIMIT (1u << 22)
// every wait is bounded in WALL-CLOCK time (s_memrealtime: the 100 MHz constant clock), not in polls: a workgroup that does not
// hear from a producer for this long raises the abort word, every other wait sees it and leaves, the host undoes the launch
#ifndef DIMN_RES_WAIT_TICKS
#define DIMN_RES_WAIT_TICKS 150000000ull      // 1.5 s
#endif
#ifndef DIMN_RES_AUX
#define DIMN_RES_AUX 17           // sc0 sc1 on every exchanged 16-byte access
#endif
#define DIMN_RES_W2S 26976        // LDS float offset of the W2 state
#define DIMN_RES_LDS_FLOATS (DIMN_RES_W2S + 3 * 16 * 256)
#ifndef DIMN_RES_SENT
#define DIMN_RES_SENT 1           // hand-offs validated by a sentinel in the data (no drain, no arrival counter); 0: counters, as first built
#endif
#define DIMN_RES_PSLOTS (DIMN_RES_SENT ? 3 : 2)
#define DIMN_RES_DSLOTS (DIMN_RES_SENT ? 2 : 1)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef DIMN_RES_TL   // tools/res_timeline.py: per-workgroup phase times (shader clock, thread 0), summed over the epoch
__device__ unsigned long long g_res_tl[1024 * 16];
#define RES_TL_DECL unsigned long long tl_acc[16] = {0}; unsigned long long tl_t = __builtin_amdgcn_s_memtime();
#define RES_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[i] += t_ - tl_t; tl_t = t_; }
#define RES_TL_FLUSH if (threadIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) g_res_tl[blockIdx.x * 16 + i_] = tl_acc[i_];
#else
#define RES_TL_DECL
#define RES_STAMP(i)
#define RES_TL_FLUSH
#endif

// The D-chunks [cb, ce) of sub-net chunks 0..nchunk-1 that D-split `sp` of S1 owns.  Splits whose workgroups also carry role 2
// (the first n2 = ceil(OT/16)) are on the critical path of a step for the whole second layer, the others idle meanwhile: the
// role-1-only splits take what the others give up (at most cap = 8 * T1 chunks each: what the registers of a workgroup hold).
__host__ __device__ static inline void res_chunk_range(int nchunk, int S1, int n2, int cap, int sp, int& cb, int& ce) {
    const int n1 = S1 - n2;
    if (n1 <= 0 || n2 <= 0 || nchunk < 8 * S1) {             // one role everywhere, or a tile loop too short to matter: the even split
        cb = (int)((int64_t)nchunk * sp / S1); ce = (int)((int64_t)nchunk * (sp + 1) / S1);
        return;
    }
    const int lo_t = nchunk * 15 / (16 * S1);                // a role-2 split gives up ~6 % of the even share ...
    int hi = (nchunk - n2 * lo_t + n1 - 1) / n1;             // ... which the role-1-only splits take,
    hi = hi < cap ? hi : cap;                                // as far as their registers go
    const int rest = nchunk - n1 * hi;
    if (sp < n2) { cb = (int)((int64_t)rest * sp / n2); ce = (int)((int64_t)rest * (sp + 1) / n2); }
    else { cb = rest + (sp - n2) * hi; ce = cb + hi; }
}

struct ResParams {
    const SubnetDev* sn;
    const void* X;                  // gathered predictors (arena): float, or bfloat16 for handles of precision bf16 (template XT)
    const float* Y; int64_t n_cells;
    float *W1, *M1, *V1, *W2, *M2, *V2;
    float *b1w, *b1m, *b1v, *b2w, *b2m, *b2v;
    const int32_t* rows;            // [n_tr] the epoch's row order (train_rows[perm])
    int32_t n_tr, B, steps;
    const float* alpha;             // [steps] lr*sqrt(1-b2^t)/(1-b1^t) of every step of the epoch
#if DIMN_RES_SENT
    float* Ppart;                   // [K][3][G][64][16]   forward partials of step t in slot t % 3 (all-ones = "not written yet")
    float* Dpart;                   // [K][2][OT][16][64][16] dD partials of step t in slot t % 2
    unsigned* maskw;                // [steps][K][64 rows][8 words] dropout keep bits of the whole epoch (k_res_masks, before the launch)
#else
    float* Ppart;                   // [K][2][G][64][16]   forward partials, double-buffered by step parity
    float* Dpart;                   // [K][OT][16][64][16] dD partials
    unsigned* maskw;                // [3][K][64 rows][8 words] dropout keep bits of step t in buffer t % 3 (written one step
                                    // ahead, before the writer has passed that step's flagD: two buffers would race with slow readers)
#endif
    unsigned* flags;                // [2K+1]: flagP[k], flagD[k] (counter protocol only), abort
    double* loss;                   // [K][OT] sum over the epoch of sum(w e^2) per output tile
    Dims dm;
    float omb1, omb2, eps, rate, scale;
    uint64_t seed; uint32_t epoch;
    int32_t G, S1, loss_binary;
    int32_t k0;                     // first sub-net of this launch (a handle may train its sub-nets in groups, one launch each)
};

__device__ __forceinline__ f32x4 res_ld(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, DIMN_RES_AUX);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void res_st(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, f32x4 x) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x), r, byte_off, 0, DIMN_RES_AUX);
}

// One lane waits until *flag >= target (relaxed agent-scope polls, s_sleep between them); false on abort.
__device__ __forceinline__ bool res_wait(unsigned* flag, unsigned target, unsigned* abort_w) {
    unsigned spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > DIMN_RES_WAIT_TICKS) {
                __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

#if DIMN_RES_SENT
// The sentinel protocol.  An exchange slot is filled with all-ones words before its producer writes it (a NaN pattern no
// arithmetic produces), so a consumer sees in the DATA whether a 16-byte piece has arrived: no store drain, no barrier, no
// arrival counter on the producer side -- the hand-off costs one store latency plus one load latency.
#define DIMN_RES_SENTW 0xffffffffu
__device__ __forceinline__ bool res_unwritten(f32x4 x) {
    const u32x4 v = __builtin_bit_cast(u32x4, x);
    return (v[0] == DIMN_RES_SENTW) | (v[1] == DIMN_RES_SENTW) | (v[2] == DIMN_RES_SENTW) | (v[3] == DIMN_RES_SENTW);
}
// One whole wave waits until piece 0 of each of `n` tiles (base + i * stride bytes) is written; false on abort / timeout.
// The bulk loads that follow validate every piece again (res_fix): the canary only keeps the polling traffic small.
__device__ __forceinline__ bool res_poll(__amdgpu_buffer_rsrc_t r, uint32_t base, int n, uint32_t stride, unsigned* abort_w) {
#ifdef DIMN_RES_NOCANARY   // test build (tests/test_gpu_configs.py): no canary, so the bulk requests leave early and res_fix's retry path runs all the time
    return true;
#endif
    const int lane = threadIdx.x & 63;
    unsigned spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        bool missing = false;
        for (int i = lane; i < n; i += 64) missing |= res_unwritten(res_ld(r, base + (uint32_t)i * stride));
        if (__builtin_amdgcn_ballot_w64(missing) == 0) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0u) {
            if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > DIMN_RES_WAIT_TICKS) {
                __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}
// A piece requested before it was written (rare: its tile's canary was) is requested again; wave-uniform loop, bounded.
__device__ __forceinline__ void res_fix(f32x4& x, __amdgpu_buffer_rsrc_t r, uint32_t off, unsigned* abort_w) {
    unsigned spins = 0;
    while (__builtin_amdgcn_ballot_w64(res_unwritten(x)) != 0) {
        asm volatile("" ::: "memory");                       // the reload is a new observation of memory
        x = res_ld(r, off);
        if (++spins > DIMN_RES_SPIN_LIMIT) { __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
}
// The dropout keep words of a whole epoch (they depend on no data): maskw[t][k][row b][word h/32], bit h%32 = keep(b, h);
// Philox block (b*H)/4 + h/4 of step t gives four units.  grid (steps * K), 512 threads: one word per thread.
__global__ __launch_bounds__(512) void k_res_masks(const SubnetDev* __restrict__ sn, unsigned* __restrict__ maskw, int K, int H,
                                                   uint64_t seed, uint32_t epoch, float rate) {
    const int t = blockIdx.x / K, k = blockIdx.x - t * K;
    const int ww = threadIdx.x, b = ww >> 3, q = ww & 7;
    unsigned word = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const dimn_u32x4 rnd = dimn_dropout_block(seed, (uint32_t)sn[k].kg, epoch, (uint32_t)t, (uint32_t)((b * H) >> 2) + (uint32_t)(8 * q + i));
#pragma unroll
        for (int r = 0; r < 4; ++r) word |= (dimn_u01(rnd.v[r]) >= rate ? 1u : 0u) << (4 * i + r);
    }
    maskw[(size_t)blockIdx.x * 512 + ww] = word;
}
#endif

template <int T1, int S1C, typename XT = float>   // W1 tiles per wave; D-splits (0: run-time p.S1); element type of the X arena
__global__ __launch_bounds__(DIMN_RES_THREADS, 2) void k_epoch_resident(ResParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int ldd = DIMN_RES_LDD;
    // LDS map (floats).  Phase A and phase B regions alias (a workgroup runs them one after the other).
    float* ddl = lds;                         // A: Dd [64][ldd]                       16640
    float* zred = ddl + DIMN_TB * ldd;        // A: Z partials [8 waves][64][16]        8192
    float* xst = lds;                         // B: X staging [8 waves][XT 1024 | XN 1024] 16384 (aliases ddl)
    float* pred = lds + 16384;                // B: forward partials [8 waves][64][16]  8192 (aliases ddl/zred)
    float* dzl = lds + 24832;                 // dZ tile [64][16]  / B: dA tile         1024
    float* yl = dzl + 1024;                   // A: targets tile [64][16] / B: dD half sums 1024
    float* b1l = yl + 1024;                   // b1 of own hidden tile [16]
    float* smallf = b1l + 16;                 // [8] loss partials, [32..47] b2 tile     64
    int* flagl = (int*)(smallf + 64);         // [4] broadcast of the poll result
    float* w2s = lds + DIMN_RES_W2S;          // role 2 state: W2, m, v column block [3][16 hidden tiles][16 h][16 o]  12288
    // (W1/m/v live in registers; the W2 column block lives in LDS -- its tiles are needed in two operand forms anyway,
    //  and 24 more registers of state made the compiler spill)

    const Dims dm = p.dm;
    const int G = p.G, S1 = S1C > 0 ? S1C : p.S1;
    const int kl = blockIdx.x / G, wi = blockIdx.x - kl * G;
    const int k = p.k0 + kl;                                 // sub-net of the handle (every array below is indexed by it)
    const int ht = wi & 15, sp = wi >> 4;
    const bool is_o = wi < dm.OT;
    const int ot = is_o ? wi : 0;
    const SubnetDev s = p.sn[k];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int Hp = dm.Hp, Op = dm.Op, OT = dm.OT;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int K = dm.K;
    unsigned* flagP = p.flags + k;
    unsigned* flagD = p.flags + K + k;
    unsigned* abort_w = p.flags + 2 * K;
    (void)flagP; (void)flagD;

    // exchange buffers of this sub-net as buffer resources (wave-uniform descriptors); slots are byte offsets inside them
    const uint32_t pslot = (uint32_t)G * 4096u, dslot = (uint32_t)OT * 65536u;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(p.Ppart + (size_t)k * DIMN_RES_PSLOTS * G * 1024, 0, (int)(DIMN_RES_PSLOTS * pslot), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(p.Dpart + (size_t)k * DIMN_RES_DSLOTS * OT * 16 * 1024, 0, (int)(DIMN_RES_DSLOTS * dslot), 0x00020000);
#if DIMN_RES_SENT
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(p.maskw, 0, (int)((size_t)p.steps * K * 2048), 0x00020000);
    auto moff = [&](int t) -> uint32_t { return (uint32_t)((t * K + k) * 2048); };             // byte offset of step t's words
    const int maux = 0;                                                                         // written by an earlier kernel: plain loads
    const f32x4 sent4 = __builtin_bit_cast(f32x4, (u32x4){DIMN_RES_SENTW, DIMN_RES_SENTW, DIMN_RES_SENTW, DIMN_RES_SENTW});
#else
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(p.maskw, 0, (int)((size_t)3 * K * 2048), 0x00020000);
    auto moff = [&](int t) -> uint32_t { return (uint32_t)(((t % 3) * K + k) * 2048); };     // byte offset of step t's words
    const int maux = DIMN_RES_AUX;
#endif

    // ---- role 1 state: W1 tiles (chunk cb + wave + 8j, hidden tile ht) in registers ----
    int cb, ce;
    res_chunk_range(s.nchunk, S1, (OT + 15) >> 4, 8 * T1, sp, cb, ce);
    const int64_t cstride = (int64_t)Hp * 16;
    const int64_t wbase = s.w1off + (int64_t)(16 * ht + li) * 16 + 4 * lj;
    f32x4 w1[T1], m1[T1], v1[T1];
    bool tv[T1];
    int tc[T1];
#pragma unroll
    for (int j = 0; j < T1; ++j) {
        const int c = cb + wave + 8 * j;
        tv[j] = c < ce;
        tc[j] = tv[j] ? c : cb;                                  // clamped: loads stay in bounds, results unused
        const int64_t idx = wbase + tc[j] * cstride;
        w1[j] = *(const f32x4*)(p.W1 + idx); m1[j] = *(const f32x4*)(p.M1 + idx); v1[j] = *(const f32x4*)(p.V1 + idx);
    }
    float b1w0 = 0.f, b1m0 = 0.f, b1v0 = 0.f;
    const int64_t b1i = (int64_t)k * Hp + 16 * ht + (tid & 15);
    if (tid < 16) { b1w0 = p.b1w[b1i]; b1m0 = p.b1m[b1i]; b1v0 = p.b1v[b1i]; b1l[tid] = b1w0; }
    // ---- role 2 state: W2 tiles (hidden tiles 2*wave, 2*wave+1; output tile ot) ----
    const int64_t t2base = (int64_t)k * Hp * Op + li * 16 + 4 * lj;
    const int w2o = li * 16 + 4 * lj;                            // this lane's float4 of a [16 h][16 o] tile
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const int tile = 2 * wave + h2;
        const int64_t idx = t2base + ((int64_t)tile * OT + ot) * 256;
        *(f32x4*)(w2s + tile * 256 + w2o) = *(const f32x4*)(p.W2 + idx);
        *(f32x4*)(w2s + 4096 + tile * 256 + w2o) = *(const f32x4*)(p.M2 + idx);
        *(f32x4*)(w2s + 8192 + tile * 256 + w2o) = *(const f32x4*)(p.V2 + idx);
    }
    float b2w0 = 0.f, b2m0 = 0.f, b2v0 = 0.f;
    const int64_t b2i = (int64_t)k * Op + 16 * ot + (tid & 15);
    if (is_o && tid < 16) { b2w0 = p.b2w[b2i]; b2m0 = p.b2m[b2i]; b2v0 = p.b2v[b2i]; smallf[32 + tid] = b2w0; }
    double loss_total = 0.0;
    RES_TL_DECL

    const int B = p.B;

    // Forward partial of batch `rows_n` with the CURRENT W1 registers (+ optional gradient/Adam of batch rows_t):
    // the body of role 1.  do_grad: bfr = dA[b = 4kb+lj][h = li] is valid.
    // Per-lane addressing of the X tiles of a batch: pass i of a lane moves row 16i + lane/4, 16-byte quarter lane%4.
    // Row indices of the batch that starts at position pos0 of the epoch's row order (b_cnt rows; lanes past the batch --
    // and a batch past the epoch -- read a valid position: every load unconditional, no branch for hipcc to drain at).
    auto xrows_raw = [&](const int tid, int pos0, int b_cnt, int32_t (&rr)[4]) {
        const int lane = tid & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = 16 * i + (lane >> 2);
            int pos = pos0 + (b < b_cnt ? b : 0);
            pos = pos < p.n_tr ? pos : p.n_tr - 1;
            rr[i] = p.rows[pos];
        }
    };
    auto xrows = [&](const int tid, int pos0, int b_cnt, uint32_t (&xo)[4]) {
        int32_t rr[4];
        xrows_raw(tid, pos0, b_cnt, rr);
#pragma unroll
        for (int i = 0; i < 4; ++i) xo[i] = (uint32_t)rr[i] * (uint32_t)s.Dp;
    };
    // The body of role 1: W1 gradient of batch t + Adam in registers (do_grad; bfr = dA[b = 4kb+lj][h = li]), then the
    // forward partial of batch t+1 with the fresh W1 (do_fwd).  xa/xb: the X_t / X_{t+1} tiles of the wave's first
    // chunk, requested by the caller (before its wait); xot/xon from xrows().
    auto role1 = [&](const int tid, const uint32_t (&xot)[4], const uint32_t (&xon)[4], XRaw<XT> (&xa)[4], XRaw<XT> (&xb)[4], bool do_grad, bool do_fwd,
                     const float (&bfr)[16], const AdamP ap, uint32_t slot_out) {
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lj = lane >> 4;
        const XT* xk = (const XT*)p.X + s.xoff + 4 * (lane & 3);
        float* xt = xst + wave * 2048;
        float* xn = xt + 1024;
        f32x4 pT[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
        for (int j = 0; j < T1; ++j) {
            if (j > 0 && !tv[j]) break;                          // wave-uniform: a wave's tiles are its first ones (tile 0 always runs: it may be a clamped one)
#pragma unroll
            for (int i = 0; i < 4; ++i) {                        // wave-private staging (in-order LDS, no barrier)
                *(f32x4*)(xt + 256 * i + 4 * lane) = xa[i].get();
                *(f32x4*)(xn + 256 * i + 4 * lane) = xb[i].get();
            }
            if (j + 1 < T1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { xa[i].load(xk + xot[i] + 16 * tc[j + 1]); xb[i].load(xk + xon[i] + 16 * tc[j + 1]); }
            }
            __builtin_amdgcn_sched_barrier(0);                   // the requests of the next tile leave before this tile's MFMAs
            if (do_grad) {
                f32x4 g = zero4;
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) g = MFMA16(xt[64 * kb + lane], bfr[kb], g);     // A = X_t^T[d = li][b = 4kb+lj]
                if (tv[j]) adam4(w1[j], m1[j], v1[j], g, ap);
            }
            if (do_fwd && tv[j]) {                               // wave-uniform
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const f32x4 x4 = *(const f32x4*)(xn + (16 * n + li) * 16 + 4 * lj);         // X_next[b = 16n+li][d = 4lj+r]
#pragma unroll
                    for (int r = 0; r < 4; ++r) pT[n] = MFMA16(w1[j][r], x4[r], pT[n]);          // P^T[h][b] += W1^T X^T
                }
            }
        }
        RES_STAMP(8)
        if (do_fwd) {
#pragma unroll
            for (int n = 0; n < 4; ++n) *(f32x4*)(pred + wave * 1024 + (16 * n + li) * 16 + 4 * lj) = pT[n];   // P[b = 16n+li][h = 4lj..]
            __syncthreads();
            if (tid < 256) {
                f32x4 a = *(const f32x4*)(pred + 4 * tid);
#pragma unroll
                for (int wv = 1; wv < 8; ++wv) a += *(const f32x4*)(pred + wv * 1024 + 4 * tid);
                if (sp == 0) a += *(const f32x4*)(b1l + 4 * (tid & 3));     // split 0 carries the bias: A = sum_s P_s
#if DIMN_RES_SENT
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the reset of the slot after this one (issued a phase ago) is in place
                res_st(rP, slot_out + (uint32_t)(wi * 4096 + 16 * tid), a);
            }
#else
                res_st(rP, slot_out + (uint32_t)(wi * 4096 + 16 * tid), a);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave drains before the arrival
            }
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(flagP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        }
        RES_STAMP(9)
    };

    // Dropout keep bits of the [64][256] activation of a step: 512 words (row b, word h/32; 4 bits per Philox block).  They
    // depend on no data, so the workgroups that have no role 2 (idle while the second layer runs) -- or all of them when
    // there are none -- compute them one step ahead, each its share of the words, and publish them with their partials.
    const int nprod = G > OT ? G - OT : G, pidx = wi - (G - nprod);
    const int mw0 = pidx >= 0 ? 512 * pidx / nprod : 0, mw1 = pidx >= 0 ? 512 * (pidx + 1) / nprod : 0;
    unsigned* mscr = (unsigned*)(smallf + 48);                   // [<= 16] staging words of one pass (LDS)
    auto publish_mask = [&](const int tid, int t) {
        if (DIMN_RES_SENT || !(p.rate > 0.f)) return;            // sentinel protocol: k_res_masks made them; rate 0: consumers do not read the mask
        for (int w0 = mw0; w0 < mw1; w0 += 16) {                 // 16 words = 128 blocks per pass, one block per thread
            const int nw = (mw1 - w0) < 16 ? (mw1 - w0) : 16;
            if (tid < 16) mscr[tid] = 0u;
            __syncthreads();
            if (tid < 8 * nw) {
                const int ww = w0 + (tid >> 3), b = ww >> 3, hq = 8 * (ww & 7) + (tid & 7);
                const dimn_u32x4 rnd = dimn_dropout_block(p.seed, (uint32_t)s.kg, p.epoch, (uint32_t)t, (uint32_t)((b * dm.H) >> 2) + (uint32_t)hq);
                unsigned nib = 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) nib |= (dimn_u01(rnd.v[r]) >= p.rate ? 1u : 0u) << r;
                atomicOr(&mscr[tid >> 3], nib << (4 * (tid & 7)));
            }
            __syncthreads();
            if (tid < nw) __builtin_amdgcn_raw_buffer_store_b32(mscr[tid], rM, moff(t) + (uint32_t)(4 * (w0 + tid)), 0, DIMN_RES_AUX);
            __syncthreads();
        }
    };
    auto batch_size_of = [&](int t) -> int { const int rem = p.n_tr - t * B; return rem <= 0 ? 0 : (rem < B ? rem : B); };
    auto target_row = [&](const int tid, int t) -> int32_t {     // matrix row of this thread's piece of the targets tile of step t
        const int ub = (tid & 255) >> 2, b_cnt = batch_size_of(t);
        int pos = t * B + (ub < b_cnt ? ub : 0);
        pos = pos < p.n_tr ? pos : p.n_tr - 1;
        return p.rows[pos];
    };
    auto targets = [&](const int tid, int32_t row) -> f32x4 {
        return *(const f32x4*)(p.Y + ((int64_t)k * p.n_cells + row) * Op + 16 * ot + 4 * (tid & 3));
    };
    f32x4 y_a = zero4;
    uint32_t xo0[4];                                             // X row offsets of the CURRENT batch (kept from the step before)
    {   // prologue: forward partials of step 0
        const float nob[16] = {0.f};
        const int b0 = p.n_tr < B ? p.n_tr : B;
        AdamP ap0; ap0.alpha = 0.f; ap0.omb1 = p.omb1; ap0.omb2 = p.omb2; ap0.eps = p.eps;
        xrows(tid, 0, b0, xo0);
        XRaw<XT> xa[4], xb[4];
        const XT* xk = (const XT*)p.X + s.xoff + 4 * (lane & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xa[i].load(xk + xo0[i] + 16 * tc[0]); xb[i] = xa[i]; }
        __syncthreads();                                         // b1l written
        publish_mask(tid, 0);
        role1(tid, xo0, xo0, xa, xb, false, true, nob, ap0, 0u);
        y_a = targets(tid, target_row(tid, 0));
    }

    for (int t = 0; t < p.steps; ++t) {
        // Every per-thread index below derives from a copy of threadIdx.x the optimiser cannot see through: otherwise
        // LICM hoists ~60 registers of address arithmetic out of the step loop and keeps them live next to the state
        // (measured: 210 VGPRs at two state tiles per wave), which turns into scratch spills at seven tiles.
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int tid = tl, lane = tid & 63, wave = tid >> 6, li = lane & 15, lj = lane >> 4;
        const int w2o = li * 16 + 4 * lj;
        const int par = t & 1;
        const int b_act = (p.n_tr - t * B) < B ? (p.n_tr - t * B) : B;
        const int rem = p.n_tr - (t + 1) * B;
        const int b_next = rem <= 0 ? 0 : (rem < B ? rem : B);
        const float inv_n = (float)(1.0 / ((double)b_act * dm.O));
        AdamP ap; ap.alpha = p.alpha[t]; ap.omb1 = p.omb1; ap.omb2 = p.omb2; ap.eps = p.eps;
#if DIMN_RES_SENT
        const uint32_t pcur = (uint32_t)(t % 3) * pslot, pnext = (uint32_t)((t + 1) % 3) * pslot, pfree = (uint32_t)((t + 2) % 3) * pslot;
        const uint32_t dcur = (uint32_t)par * dslot, dfree = (uint32_t)(par ^ 1) * dslot;
#else
        const uint32_t pcur = (uint32_t)par * pslot, pnext = (uint32_t)(par ^ 1) * pslot, dcur = 0u;
#endif
        // row indices of the NEXT batch, requested a whole phase before their use (they head two dependent loads)
        int32_t rn[4];
        xrows_raw(tid, (t + 1) * B, b_next, rn);
        const int32_t yrow_n = target_row(tid, t + 1);          // unconditional: a load under a divergent branch makes hipcc drain vmcnt at the join

        // =============================== phase A (role 2) ===============================
        RES_STAMP(10)
        if (is_o) {
            const int ub = (tid & 255) >> 2, uq = tid & 3;
            if (tid < 256) *(f32x4*)(yl + 4 * tid) = y_a;
            RES_STAMP(0)
#if DIMN_RES_SENT
            if (wave == 0) { const bool ok = res_poll(rP, pcur, G, 4096u, abort_w); if (lane == 0) flagl[0] = ok ? 1 : 0; }
            __syncthreads();
            if (!flagl[0]) return;
#else
            if (tid == 0) flagl[0] = res_wait(flagP, (unsigned)(G * (t + 1)), abort_w) ? 1 : 0;
            __syncthreads();
            if (!flagl[0]) return;
#endif
            RES_STAMP(1)
            u32x4 km0 = (u32x4){0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, km1 = km0;
            if (p.rate > 0.f) {                                  // keep bits of row ub: 8 words
                km0 = __builtin_amdgcn_raw_buffer_load_b128(rM, moff(t) + (uint32_t)(32 * ub), 0, maux);
                km1 = __builtin_amdgcn_raw_buffer_load_b128(rM, moff(t) + (uint32_t)(32 * ub + 16), 0, maux);
            }
            const int ksh = 16 * (tid >> 8) + 4 * uq;            // unit q: hidden units 32q + ksh .. +3 = word q, bits ksh..
            const unsigned rowmask = ub < b_act ? 0xffffffffu : 0u;
            // Dd = dropout(relu(sum_s P_s)) -> LDS (split 0 carries b1); every partial is requested before the first is used
            auto put_dd = [&](int q, f32x4 a) {
                const int tile = 2 * q + (tid >> 8);
                f32x4 dd;
                const unsigned kw = (q < 4 ? km0[q & 3] : km1[q & 3]) & rowmask;
#pragma unroll
                for (int r = 0; r < 4; ++r) dd[r] = ((kw >> (ksh + r)) & 1u) ? fmaxf(a[r], 0.f) * p.scale : 0.f;      // one select, no branch
                *(f32x4*)(ddl + ub * ldd + 16 * tile + 4 * uq) = dd;
            };
            auto poff = [&](int q, int ss) -> uint32_t { return pcur + (uint32_t)((ss * 16 + 2 * q + (tid >> 8)) * 4096 + 16 * (tid & 255)); };
#if DIMN_RES_SENT
#define RES_FIX(x, r, off) res_fix((x), (r), (off), abort_w)
#else
#define RES_FIX(x, r, off)
#endif
            if (S1C > 0) {
                // a rolling window of 4 x S1 requests: the slot a tile pair has just left takes the requests of the pair four
                // places on, so the second half of the partials travels while the first is summed (one round trip, not two)
                f32x4 pv[4][S1C > 0 ? S1C : 1];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int ss = 0; ss < S1C; ++ss) pv[q][ss] = res_ld(rP, poff(q, ss));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
#pragma unroll
                    for (int ss = 0; ss < S1C; ++ss) RES_FIX(pv[q & 3][ss], rP, poff(q, ss));
                    f32x4 a = pv[q & 3][0];
#pragma unroll
                    for (int ss = 1; ss < S1C; ++ss) a += pv[q & 3][ss];
                    if (q < 4) {
#pragma unroll
                        for (int ss = 0; ss < S1C; ++ss) pv[q][ss] = res_ld(rP, poff(q + 4, ss));
                    }
                    put_dd(q, a);
                }
            } else {
#pragma unroll 2
                for (int q = 0; q < 8; ++q) {
                    f32x4 a = res_ld(rP, poff(q, 0));
                    RES_FIX(a, rP, poff(q, 0));
                    for (int ss = 1; ss < S1; ++ss) {
                        f32x4 a2 = res_ld(rP, poff(q, ss));
                        RES_FIX(a2, rP, poff(q, ss));
                        a += a2;
                    }
                    put_dd(q, a);
                }
            }
            const float* ws = w2s + 2 * wave * 256;              // this wave's two W2 tiles [h][o]
#if DIMN_RES_SENT
            // every partial of this step was out, so every workgroup has finished the step before: the dD slot of that step is
            // free -- mark it "not written" for the step after this one.  Behind the partial requests (in front of them the
            // 64 KB of stores delayed the loads), and acknowledged before this step's dD stores leave (vmcnt(0) there).
#pragma unroll
            for (int i = 0; i < 8; ++i) res_st(rD, dfree + (uint32_t)(ot * 65536 + (i * 512 + tid) * 16), sent4);
#endif
            __syncthreads();
            RES_STAMP(2)
            {   // Z partial over this wave's 32 hidden units
                f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    float bq[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) bq[r] = ws[h2 * 256 + (4 * lj + r) * 16 + li];   // W2[h = 4lj+r][o = li]
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const f32x4 a4 = *(const f32x4*)(ddl + (16 * m + li) * ldd + 16 * (2 * wave + h2) + 4 * lj);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[m] = MFMA16(a4[r], bq[r], acc[m]);
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zred[wave * 1024 + (16 * m + 4 * lj + r) * 16 + li] = acc[m][r];
            }
            __syncthreads();
            {   // epilogue of the forward: two elements per thread of the [64][16] tile
                float lsum = 0.f;
                const bool col_ok = (16 * ot + (tid & 15)) < dm.O;
                const float bias = smallf[32 + (tid & 15)];      // b2 tile (written at the end of the previous step / prologue)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int e = tid + 512 * half, b = e >> 4;
                    float z = bias;
#pragma unroll
                    for (int wv = 0; wv < 8; ++wv) z += zred[wv * 1024 + e];
                    float dz = 0.f;
                    if (b < b_act && col_ok) {
                        const float y = yl[e];
                        const float w = p.loss_binary ? (y > 0.f ? 1.f : 0.f) : y;   // multinet.py:37-40
                        float spv, sg;
                        softplus_sigmoid_fast(z, spv, sg);
                        const float er = y - spv;
                        lsum += w * er * er;
                        dz = -2.f * w * er * inv_n * sg;
                    }
                    dzl[e] = dz;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
                if (lane == 0) smallf[wave] = lsum;
            }
            __syncthreads();
            if (tid == 0) {
                float tot = 0.f;
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) tot += smallf[wv];
                loss_total += (double)tot;
            }
            RES_STAMP(3)
            if (tid < 16) {                                      // gb2 = column sums of dZ -> Adam(b2)
                float gb = 0.f;
                for (int b = 0; b < DIMN_TB; ++b) gb += dzl[b * 16 + tid];
                adam1(b2w0, b2m0, b2v0, gb, ap);
            }
            {   // dD^T partial with the OLD W2: published BEFORE the W2 gradient, which then runs while the role-1 workgroups
                // already pick the partials up
                f32x4 zf[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) zf[n] = *(const f32x4*)(dzl + (16 * n + li) * 16 + 4 * lj);      // dZ[b = 16n+li][o = 4lj+r]
#if DIMN_RES_SENT
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the reset of the other dD slot (issued before the Dd build) is in place
#endif
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int tile = 2 * wave + h2;
                    const f32x4 wq = *(const f32x4*)(w2s + tile * 256 + w2o);                                   // OLD W2 (h = li, o = 4lj..)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        f32x4 d = zero4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) d = MFMA16(wq[r], zf[n][r], d);                              // dD^T[h][b] = W2 dZ^T
                        res_st(rD, dcur + (uint32_t)(((ot * 16 + tile) * 1024 + (16 * n + li) * 16 + 4 * lj) * 4), d);   // [b = 16n+li][h = 4lj..]
                    }
                }
            }
#if !DIMN_RES_SENT
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(flagD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            RES_STAMP(4)
            {   // W2 gradient + Adam on the LDS-resident state
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int tile = 2 * wave + h2;
                    f32x4 g = zero4;
#pragma unroll
                    for (int kb = 0; kb < 16; ++kb)
                        g = MFMA16(dzl[64 * kb + lane], ddl[(4 * kb + lj) * ldd + 16 * tile + li], g);          // dZ^T Dd
                    f32x4 wq = *(const f32x4*)(w2s + tile * 256 + w2o);
                    f32x4 mq = *(const f32x4*)(w2s + 4096 + tile * 256 + w2o), vq = *(const f32x4*)(w2s + 8192 + tile * 256 + w2o);
                    adam4(wq, mq, vq, g, ap);
                    *(f32x4*)(w2s + tile * 256 + w2o) = wq; *(f32x4*)(w2s + 4096 + tile * 256 + w2o) = mq; *(f32x4*)(w2s + 8192 + tile * 256 + w2o) = vq;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                                     // everybody is done reading ddl/dzl/zred (phase B re-uses them)
            if (tid < 16) smallf[32 + tid] = b2w0;
            RES_STAMP(11)
        }

        // =============================== phase B (role 1) ===============================
        {
            const int ub = (tid & 255) >> 2, uq = tid & 3, half = tid >> 8;
            // Idle time before the dD partials arrive: keep bits of the own tile, the first X tiles of this step's tile
            // loop, and (role 2) the mask and targets of the NEXT step
            uint32_t xon[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xon[i] = b_next > 0 ? (uint32_t)rn[i] * (uint32_t)s.Dp : xo0[i];
            XRaw<XT> xa[4], xb[4];
            {   // first X tiles of the tile loop, requested before the wait
                const XT* xk = (const XT*)p.X + s.xoff + 4 * (lane & 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) { xa[i].load(xk + xo0[i] + 16 * tc[0]); xb[i].load(xk + xon[i] + 16 * tc[0]); }
            }
            // (tried: touching the next batch's X rows here / before the flagP wait, by asm loads or LDS-DMA, to move
            //  their HBM latency out of the tile loop: +3..5 us per step -- 16 hidden-tile workgroups fetch the same rows
            //  and the extra requests queue in front of the hand-off traffic)
            RES_STAMP(12)
            if (t + 1 < p.steps) {
                publish_mask(tid, t + 1);                        // drained with the partials below, before the flagP arrival
                y_a = targets(tid, yrow_n);                      // every thread (unconditional load); role 2 uses the first 256
            }
            RES_STAMP(5)
#if DIMN_RES_SENT
            if (wave == 0) { const bool ok = res_poll(rD, dcur + (uint32_t)(ht * 4096), OT, 65536u, abort_w); if (lane == 0) flagl[1] = ok ? 1 : 0; }
#else
            if (tid == 0) flagl[1] = res_wait(flagD, (unsigned)(OT * (t + 1)), abort_w) ? 1 : 0;
#endif
            __syncthreads();
            if (!flagl[1]) return;
            RES_STAMP(6)
            // requests first, in the order of need: the relu gate (A of the own tile from the S1 siblings' partials, its keep
            // bits), then the dD partials of the OT producers (their two halves on the two thread halves)
            unsigned keep = 0xfu;
            if (p.rate > 0.f)
                keep = __builtin_amdgcn_raw_buffer_load_b32(rM, moff(t) + (uint32_t)(32 * ub + 4 * (ht >> 1)), 0, maux) >> (16 * (ht & 1) + 4 * uq);
            auto goff = [&](int ss) -> uint32_t { return pcur + (uint32_t)((ss * 16 + ht) * 4096 + 16 * (tid & 255)); };
            f32x4 a;
            if (S1C > 0) {
                f32x4 gp[S1C > 0 ? S1C : 1];
#pragma unroll
                for (int ss = 0; ss < S1C; ++ss) gp[ss] = res_ld(rP, goff(ss));
#pragma unroll
                for (int ss = 0; ss < S1C; ++ss) RES_FIX(gp[ss], rP, goff(ss));
                a = gp[0];
#pragma unroll
                for (int ss = 1; ss < S1C; ++ss) a += gp[ss];
            } else {
                a = res_ld(rP, goff(0));
                RES_FIX(a, rP, goff(0));
                for (int ss = 1; ss < S1; ++ss) { f32x4 a2 = res_ld(rP, goff(ss)); RES_FIX(a2, rP, goff(ss)); a += a2; }
            }
            f32x4 d = zero4;
            {
                const int o0 = half * (OT >> 1), o1 = half ? OT : (OT >> 1);
                const uint32_t base = dcur + (uint32_t)((ht * 1024 + 4 * (tid & 255)) * 4);
                int o = o0;
                if (o + 16 <= o1) {                              // rolling window of 8 requests over 16 producers
                    f32x4 tq[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) tq[i] = res_ld(rD, base + (uint32_t)((o + i) * 65536));
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        RES_FIX(tq[i], rD, base + (uint32_t)((o + i) * 65536));
                        d += tq[i];
                        tq[i] = res_ld(rD, base + (uint32_t)((o + 8 + i) * 65536));
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) { RES_FIX(tq[i], rD, base + (uint32_t)((o + 8 + i) * 65536)); d += tq[i]; }
                    o += 16;
                }
                for (; o + 4 <= o1; o += 4) {
                    f32x4 tq[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) tq[i] = res_ld(rD, base + (uint32_t)((o + i) * 65536));
#pragma unroll
                    for (int i = 0; i < 4; ++i) { RES_FIX(tq[i], rD, base + (uint32_t)((o + i) * 65536)); d += tq[i]; }
                }
                for (; o < o1; ++o) { f32x4 t1 = res_ld(rD, base + (uint32_t)(o * 65536)); RES_FIX(t1, rD, base + (uint32_t)(o * 65536)); d += t1; }
            }
#if DIMN_RES_SENT
            // every dD partial of this step was out, so every workgroup has read the partials of the step before: that slot is
            // free -- mark it "not written" for the step after the next (acknowledged before P of the next step leaves: vmcnt(0) there)
            if (tid < 256) res_st(rP, pfree + (uint32_t)(wi * 4096 + 16 * tid), sent4);
#endif
            if (half) *(f32x4*)(yl + 4 * (tid & 255)) = d;
            __syncthreads();
            if (half == 0) {
                d += *(const f32x4*)(yl + 4 * tid);
                if (ub >= b_act) keep = 0u;
                f32x4 da;
#pragma unroll
                for (int r = 0; r < 4; ++r) da[r] = (((keep >> r) & 1u) & (a[r] > 0.f ? 1u : 0u)) ? d[r] * p.scale : 0.f;
                *(f32x4*)(dzl + 4 * tid) = da;                   // dA tile [64][16]
            }
            __syncthreads();
            if (tid < 16) {                                      // gb1 = column sums of dA -> Adam(b1); identical on the S1 siblings
                float gb = 0.f;
                for (int b = 0; b < DIMN_TB; ++b) gb += dzl[b * 16 + tid];
                adam1(b1w0, b1m0, b1v0, gb, ap);
                b1l[tid] = b1w0;
            }
            float bfr[16];
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) bfr[kb] = dzl[64 * kb + lane];                                        // dA[b = 4kb+lj][h = li]
            RES_STAMP(7)
            role1(tid, xo0, xon, xa, xb, true, b_next > 0, bfr, ap, pnext);     // its first barrier orders b1l / dzl
            if (b_next == 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) xo0[i] = xon[i];         // the next batch becomes the current one
        }
    }

    RES_TL_FLUSH
    // ---- epilogue: the state goes back to its tile-native place in HBM ----
#pragma unroll
    for (int j = 0; j < T1; ++j)
        if (tv[j]) {
            const int64_t idx = wbase + tc[j] * cstride;
            *(f32x4*)(p.W1 + idx) = w1[j]; *(f32x4*)(p.M1 + idx) = m1[j]; *(f32x4*)(p.V1 + idx) = v1[j];
        }
    if (sp == 0 && tid < 16) { p.b1w[b1i] = b1w0; p.b1m[b1i] = b1m0; p.b1v[b1i] = b1v0; }
    if (is_o) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int tile = 2 * wave + h2;
            const int64_t idx = t2base + ((int64_t)tile * OT + ot) * 256;
            *(f32x4*)(p.W2 + idx) = *(const f32x4*)(w2s + tile * 256 + w2o);
            *(f32x4*)(p.M2 + idx) = *(const f32x4*)(w2s + 4096 + tile * 256 + w2o);
            *(f32x4*)(p.V2 + idx) = *(const f32x4*)(w2s + 8192 + tile * 256 + w2o);
        }
        if (tid < 16) { p.b2w[b2i] = b2w0; p.b2m[b2i] = b2m0; p.b2v[b2i] = b2v0; }
        if (tid == 0) p.loss[(int64_t)k * OT + ot] = loss_total;
    }
}
