// dimn_resident.h -- the REGISTER-RESIDENT epoch kernel of libdimn (gfx950).
//
// When a GPU owns only a few sub-networks (8-GPU sharding of BASELINE configs[3]: K_local = 5), the whole
// optimiser state of the rank -- W, m, v of both layers, ~45 MB -- fits the chip's 128 MB of vector registers.
// One persistent launch then runs a WHOLE EPOCH of model.fit (reference deepimpute/multinet.py:238-244):
// every workgroup keeps its slice of W1/m/v (and of W2/m/v) in registers / LDS for all ~743 optimiser steps, so
// no optimiser state moves through HBM at all; per step the workgroups of one sub-net exchange only small
// activation tiles through the memory-side cache (write-through stores validated by the readers, no grid-wide barrier:
// sub-nets share nothing, multinet.py:132-146).  The four launches per step of the streaming path (RED, MF, MB, B1F1 --
// each latency-bound at this size) disappear.
//
// Decomposition of sub-net k over G = 16*S1 workgroups (512 threads = 8 waves x 256 VGPRs, one per CU):
//   role 1 (all G):      workgroup (ht, s) owns hidden tile ht (16 units) of the D-split s: the W1 tiles
//                        [chunk c in split s][ht], T1 tiles per wave.  Per step: W1 gradient + Adam in registers from the
//                        dA tile of ht, then the forward partial P_s[64][16] of the NEXT batch (k-slot trick, dimn_kernels.h).
//   manager (s = S1-1):  one workgroup per hidden tile collects what belongs to the tile and publishes it ONCE:
//                        M1: A = sum_s P_s + b1, Dd = dropout(relu(A)) -> the Dd tile;  M2: dD = sum over the OT output
//                        tiles' partials, dA = gate * dD * scale -> the dA tile.  (Round 2 let every consumer reduce for
//                        itself: every role-2 workgroup read all G partials, every role-1 workgroup all OT dD tiles -- 12.6 MB
//                        of hand-off reads per sub-net and step, 20k of 62k clocks; now 4.3 MB.  PMC, 5 sub-nets: 104 -> 92 MB per
//                        step in all -- the rest is the batch rows of X, which every XCD fetches for itself: profiles/r03_traffic_k5.json)
//   role 2 (first OT):   workgroup ot owns the W2 column block [all 16 hidden tiles][output tile ot] in LDS.  Per step:
//                        the 16 Dd tiles -> Z tile, softplus, wMSE, dZ, Adam(b2), the dD^T partial [64][256] over its 16
//                        outputs with the OLD W2 (published), then the W2 gradient + Adam on the column block.
// Exchange per step and sub-net (the only inter-workgroup traffic), hop by hop:
//   P partials   (S1-1) x 16 tiles [64][16]   siblings -> manager of their hidden tile
//   Dd tiles     16 x [64][16]                manager -> every role-2 workgroup
//   dD partials  OT x 16 x [64][16]           role 2 -> manager (tile ht of every producer)
//   dA tiles     16 x [64][16]                manager -> its S1-1 siblings
// All through 16-byte sc0 sc1 (write-through) stores and sc0 sc1 loads (L1 is never refreshed by other CUs' stores, the L2s
// of different XCDs are not coherent: MI355X guide, Guideline 16).  Hand-off protocol: a slot is filled with all-ones words
// before its producer writes it, so a consumer sees in the DATA whether a piece has arrived (res_poll on one piece per
// producer tile, then res_fix on every piece where it is consumed): no store drain, no barrier, no arrival counter on the
// producer side.  Every buffer has two slots (step parity); the producer of a slot re-arms it when its readers have PROVABLY
// finished, and drains that store before its next data store leaves:
//   P slot t%2      re-armed by the sibling when it has seen dA(t)        (the manager summed P(t) before it made Dd(t))
//   dA slot (t+1)%2 re-armed by the manager when it has gathered P(t)     (its siblings ran the tile loop of step t-1 on dA(t-1))
//   dD slot (t+1)%2 re-armed by role 2 when it has gathered Dd(t)         (every manager summed dD(t-1) before its tile loop)
//   Dd slot t%2     re-armed by the manager when it has gathered dD(t)    (every role-2 workgroup read Dd(t) before its dD(t))
// Every wait is bounded in wall-clock time; a timeout raises an abort word, the host then restores the pre-epoch state and
// re-runs the epoch on the streaming kernels, so a lost workgroup can neither hang the GPU nor fail a fit.  A handle with more
// sub-nets than fit at once runs one launch per GROUP of sub-nets (ResParams.k0).
//
// Arithmetic: the exact-fp32 path of the streaming kernels (v_mfma_f32_16x16x4_f32, adam4, the Philox dropout streams,
// softplus_sigmoid_fast) -- only summation orders differ.  Template BF (handles of precision bf16): every GEMM of the step on
// v_mfma_f32_16x16x16_bf16 -- the four k-slot operands of four consecutive fp32 instructions ARE the four-element operand of
// the bf16 one, rounded to nearest even in registers; fp32 accumulation, fp32 master weights and Adam state.
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
#define DIMN_RES_W2S 27104        // LDS float offset of the W2 state
#define DIMN_RES_LDS_FLOATS (DIMN_RES_W2S + 3 * 16 * 256)
#define DIMN_RES_SLOTS 2
// (round 3's experiment switches -- DIMN_RES_EVEN / _DIRECT / _XCD / _GDIRECT / _M2WIN / _ABL -- are gone from the source; what each
//  measured is in DESIGN.md section 2b and profiles/r03_resident_*.txt)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef DIMN_RES_TL   // tools/res_timeline.py: per-workgroup phase times (shader clock, thread 0), summed over the epoch
__device__ unsigned long long g_res_tl[1024 * 16];
#define RES_TL_DECL unsigned long long tl_acc[16] = {0}; unsigned long long tl_t = __builtin_amdgcn_s_memtime();
#define RES_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tl_acc[i] += t_ - tl_t; tl_t = t_; }
#define RES_TL_FLUSH if (threadIdx.x == 0) for (int i_ = 0; i_ < 16; ++i_) g_res_tl[(kl * G + wi) * 16 + i_] = tl_acc[i_];
#else
#define RES_TL_DECL
#define RES_STAMP(i)
#define RES_TL_FLUSH
#endif

#ifdef DIMN_RES_TL2  // tools/res_trace.py: ABSOLUTE time stamps (s_memrealtime: the chip-wide 100 MHz clock) of 4 chosen steps, thread 0 of every workgroup: the
                     // steady state is not perturbed (one compare per mark outside those steps), and the stamps of different workgroups share one time base
__device__ unsigned long long g_res_tl2[1024 * 4 * 32];
#define DIMN_RES_TL2_T0 300
#define RES_MARK(i) { if (threadIdx.x == 0 && tl2_t >= DIMN_RES_TL2_T0 && tl2_t < DIMN_RES_TL2_T0 + 4) g_res_tl2[(((size_t)kl * G + wi) * 4 + (tl2_t - DIMN_RES_TL2_T0)) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); }
#define RES_MARK_WAVE(i) { if ((threadIdx.x & 63) == 0 && tl2_t >= DIMN_RES_TL2_T0 && tl2_t < DIMN_RES_TL2_T0 + 4) g_res_tl2[(((size_t)kl * G + wi) * 4 + (tl2_t - DIMN_RES_TL2_T0)) * 32 + (i) + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime(); }
#define RES_MARK_DECL int tl2_t = -1;
#define RES_MARK_STEP(t) tl2_t = (t);
#else
#define RES_MARK(i)
#define RES_MARK_WAVE(i)
#define RES_MARK_DECL
#define RES_MARK_STEP(t)
#endif

// The D-chunks [cb, ce) of sub-net chunks 0..nchunk-1 that D-split `sp` of S1 owns: the even split.  A workgroup's tile loop
// is bound by the matrix pipe of its busiest SIMD (waves w and w + 4 share one): ceil(c / 4) tile-times for c chunks, so the
// even split is also the fastest one for the manager, whose loop every step waits for.  (Measured in round 3: giving the
// siblings whole tiles and the manager the rest -- 150 chunks: 48 / 48 / 54, so that both hand-offs around the manager travel
// under its last tile -- puts 14 instead of 13 tiles on two of the manager's SIMDs: 24.3 vs 23.7 us per step.)
__host__ __device__ static inline void res_chunk_range(int nchunk, int S1, int sp, int& cb, int& ce) {
    cb = (int)((int64_t)nchunk * sp / S1); ce = (int)((int64_t)nchunk * (sp + 1) / S1);
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
    float* Ppart;                   // [K][2][G][64][16]      forward partials of step t in slot t % 2 (all-ones = "not written yet")
    float* Dpart;                   // [K][2][OT][16][64][16] dD partials of step t in slot t % 2
    float* DdT;                     // [K][2][16][64][16]     Dd tiles (manager -> role 2)
    float* dAT;                     // [K][2][16][64][16]     dA tiles (manager -> siblings)
    unsigned* maskw;                // [steps][K][64 rows][8 words] dropout keep bits of the whole epoch (k_res_masks, before the launch)
    unsigned* flags;                // [2K+1]: the last word is the abort word
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

// The sentinel protocol.  An exchange slot is filled with all-ones words before its producer writes it (a NaN pattern no
// arithmetic produces), so a consumer sees in the DATA whether a 16-byte piece has arrived: no store drain, no barrier, no
// arrival counter on the producer side -- the hand-off costs one store latency plus one load latency.  (Round 2's first
// protocol -- stores, vmcnt(0) drain, one relaxed agent-scope counter per sub-net and direction, one-lane polls -- cost ~8 us
// more per step; it is in the history of this file.)
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
        // (no look at the abort word in here: measured, round 6 -- one relaxed load every 256 spins in this loop, inlined at every consumer, cost 2.6 us per step)
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

// One X tile of a virtual tile of role1, as raw register words: four row-major 16-byte pieces (row16).  (Round 3 also tried taking
// a gradient tile as SIXTEEN single elements X[b = 4kb+lj][d = li] -- exactly the A operand of X_t^T dA, no transposition through
// LDS: 27.7 vs 23.5 us per step, sixteen vector-memory instructions per tile cost more than four plus the LDS round trip.)
template <typename XT> struct XSlot;
template <> struct XSlot<float> {
    uint32_t w[16];
    __device__ __forceinline__ void load_row16(int i, const float* q) { const f32x4 t = *(const f32x4*)q; for (int r = 0; r < 4; ++r) w[4 * i + r] = __float_as_uint(t[r]); }
    __device__ __forceinline__ f32x4 row16(int i) const { return (f32x4){__uint_as_float(w[4 * i]), __uint_as_float(w[4 * i + 1]), __uint_as_float(w[4 * i + 2]), __uint_as_float(w[4 * i + 3])}; }
};
template <> struct XSlot<bf16_t> {
    uint32_t w[16];
    __device__ __forceinline__ void load_row16(int i, const bf16_t* q) { const uint2 t = *(const uint2*)q; w[2 * i] = t.x; w[2 * i + 1] = t.y; }
    __device__ __forceinline__ f32x4 row16(int i) const {
        return (f32x4){__uint_as_float(w[2 * i] << 16), __uint_as_float(w[2 * i] & 0xffff0000u), __uint_as_float(w[2 * i + 1] << 16), __uint_as_float(w[2 * i + 1] & 0xffff0000u)};
    }
};
// Order of the tile loop's "virtual tiles" (role1).  Alternating (g0 f0 g1 f1 ..: gradient tile j, forward tile j): every row
// request has ONE tile-time of lead.  SPLIT (all gradient tiles, then the forward tiles: g0 g1 .. f0 f1 ..): TWO tile-times from
// the same two register sets.  Which one wins depends on how far away the rows are: with a small X arena the alternating order
// (K = 5 of configs[3], 1.2 GB: 19.7 vs 20.0 us per step with bf16 operands, 23.8 vs 24.1 with fp32; configs[4]'s 8 sub-nets at
// 200k cells, 7.8 GB: 38.0 vs 38.6), with a large one the split order (the same 8 sub-nets at the full 1M cells, 39 GB of rows
// gathered at random: 44.8 vs 52.6 us per step) -- the host picks SPLIT above 16 GB of arena (DIMN_RES_SPLIT=0/1 forces).
// Without GRAD (the epoch's first forward) there are forward tiles only.
template <bool GRAD, bool SPLIT, int T1> __device__ __forceinline__ constexpr bool res_vfwd(int v) { return !GRAD ? true : (SPLIT ? v >= T1 : (v & 1) != 0); }
template <bool GRAD, bool SPLIT, int T1> __device__ __forceinline__ constexpr int res_vtile(int v) { return !GRAD ? v : (SPLIT ? (v >= T1 ? v - T1 : v) : (v >> 1)); }
// c += sum_r a[r] (x) b[r] over the four k-slots a lane owns: four exact-fp32 matrix instructions, or (BF) ONE bf16 instruction
// whose four-element operands are those k-slots rounded to nearest even
template <bool BF>
__device__ __forceinline__ f32x4 res_mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
    if (BF) return MFMA_BF16(pk4(a), pk4(b), c);
#pragma unroll
    for (int r = 0; r < 4; ++r) c = MFMA16(a[r], b[r], c);
    return c;
}

// Four independent accumulations c[n] += sum_r a[n][r] (x) b[n][r] with their matrix instructions INTERLEAVED (r outer, n inner): a dependent
// pair (same accumulator) is three instructions apart instead of back to back -- round 6: the sixteen-deep chain of a gradient tile in two
// chains was 0.35 us per step faster, so the exact-fp32 instruction does not forward its result into the next one for free.
template <bool BF>
__device__ __forceinline__ void res_mfma4x4(const f32x4 (&a)[4], const f32x4 (&b)[4], f32x4 (&c)[4]) {
    if (BF) {
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = MFMA_BF16(pk4(a[n]), pk4(b[n]), c[n]);
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = MFMA16(a[n][r], b[n][r], c[n]);
}

template <int T1, int S1C, typename XT = float, bool BF = false, bool SPLIT = false>   // W1 tiles per wave; D-splits (0: run-time p.S1); element type of the X arena; bf16 matrix cores; tile order (res_vfwd)
__global__ __launch_bounds__(DIMN_RES_THREADS, 2) void k_epoch_resident(ResParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int ldd = DIMN_RES_LDD;
    // LDS map (floats).  Phase A and the tile-loop regions alias (a workgroup runs them one after the other).
    float* ddl = lds;                         // A: Dd [64][ldd]                       16640
    float* zred = ddl + DIMN_TB * ldd;        // A: Z partials [8 waves][64][16]        8192
    float* xst = lds;                         // tile loop: X staging [8 waves][XT 1024 | XN 1024] 16384 (aliases ddl)
    float* pred = lds + 16384;                // tile loop: forward partials [8 waves][64][16]  8192 (aliases ddl/zred)
    float* dzl = lds + 24832;                 // A: dZ tile [64][16]  / M2, role 1: dA tile 1024
    float* yl = dzl + 1024;                   // A: targets tile [64][16] / M2: dD half sums / manager: own forward partial between steps 1024
    float* b1l = yl + 1024;                   // b1 of own hidden tile [16]
    float* smallf = b1l + 16;                 // [8] loss partials, [32..47] b2 tile     64
    int* flagl = (int*)(smallf + 64);         // [4] broadcast of the poll results
    float* csum = smallf + 80;                // [8 waves][16] column-sum partials of the dZ / dA tile (bias gradients)   128
    float* w2s = lds + DIMN_RES_W2S;          // role 2 state: W2, m, v column block [3][16 hidden tiles][16 h][16 o]  12288
    // (W1/m/v live in registers; the W2 column block lives in LDS -- its tiles are needed in two operand forms anyway,
    //  and 24 more registers of state made the compiler spill)

    const Dims dm = p.dm;
    const int G = p.G, S1 = S1C > 0 ? S1C : p.S1;
    // Workgroup -> (sub-net, D-split, hidden tile): the plain order.  Observed (MI355X guide: "for speed only"): block b runs on XCD
    // b % 8, so the sixteen hidden-tile workgroups of one (sub-net, D-split) -- which read the SAME batch rows of X -- sit on all
    // eight XCDs and X reaches every L2 separately (PMC: ~49 of the 92 MB per step at 5 sub-nets).  Round 3 tried blocks of one XCD
    // for each such group: the tile loop got ~9 % shorter, but a sub-net's 32 role-2 workgroups then read their 64 KB of Dd tiles
    // through two XCDs' fabric ports instead of eight at the same moment: 24.7 vs 23.8 us per step.  The plain order ships.
    const int slot = (int)blockIdx.x;
    const int kl = slot / G, wi = slot - kl * G;
    const int k = p.k0 + kl;                                 // sub-net of the handle (every array below is indexed by it)
    const int ht = wi & 15, sp = wi >> 4;
    const bool is_o = wi < dm.OT;                            // role 2: owns output tile wi
    const bool is_m = sp == S1 - 1;                          // manager of hidden tile ht
    const int ot = is_o ? wi : 0;
    const bool pub_da = S1 > 1;
    const SubnetDev s = p.sn[k];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int Hp = dm.Hp, Op = dm.Op, OT = dm.OT;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int K = dm.K;
    unsigned* abort_w = p.flags + 2 * K;

    // exchange buffers of this sub-net as buffer resources (wave-uniform descriptors); slots are byte offsets inside them
    const uint32_t pslot = (uint32_t)G * 4096u, dslot = (uint32_t)OT * 65536u, tslot = 65536u;
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(p.Ppart + (size_t)k * DIMN_RES_SLOTS * G * 1024, 0, (int)(DIMN_RES_SLOTS * pslot), 0x00020000);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc(p.Dpart + (size_t)k * DIMN_RES_SLOTS * OT * 16 * 1024, 0, (int)(DIMN_RES_SLOTS * dslot), 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(p.DdT + (size_t)k * DIMN_RES_SLOTS * 16 * 1024, 0, (int)(DIMN_RES_SLOTS * tslot), 0x00020000);
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(p.dAT + (size_t)k * DIMN_RES_SLOTS * 16 * 1024, 0, (int)(DIMN_RES_SLOTS * tslot), 0x00020000);
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(p.maskw, 0, (int)((size_t)p.steps * K * 2048), 0x00020000);
    auto moff = [&](int t) -> uint32_t { return (uint32_t)((t * K + k) * 2048); };             // byte offset of step t's keep words
    const f32x4 sent4 = __builtin_bit_cast(f32x4, (u32x4){DIMN_RES_SENTW, DIMN_RES_SENTW, DIMN_RES_SENTW, DIMN_RES_SENTW});

    // ---- role 1 state: W1 tiles (chunk cb + wave + 8j, hidden tile ht) in registers ----
    int cb, ce;
    res_chunk_range(s.nchunk, S1, sp, cb, ce);
    const int64_t cstride = (int64_t)Hp * 16;
    const int64_t wbase = s.w1off + (int64_t)(16 * ht + li) * 16 + 4 * lj;
    f32x4 w1[T1], m1[T1], v1[T1];
    bool tv[T1];
    int tc[T1];
#pragma unroll
    for (int j = 0; j < T1; ++j) {
        const int c = cb + wave + 8 * j;
        tv[j] = c < ce;
        tc[j] = tv[j] ? c : (j > 0 ? tc[j - 1] : cb);            // clamped to the wave's last real tile: the request a wave makes past its
                                                                 // tiles hits lines it has just loaded (a far-away chunk cost an HBM round trip
                                                                 // that the drain before the P store then waited for)
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
    RES_MARK_DECL

    const int B = p.B;

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
    // The body of role 1: W1 gradient of batch t + Adam in registers (GRAD; bfr = dA[b = 4kb+lj][h = li]) and the forward
    // partial of batch t+1 with the fresh W1 (do_fwd) -- as ONE sequence of "virtual tiles" (res_vfwd / res_vtile say which
    // tile a position is): a gradient tile reads the rows of X_t, a forward tile those of X_{t+1}, each needs ONE X tile
    // (16 bytes per lane and row group), and two register sets hold the X tiles of positions v+1 and v+2.
    // xr[0] / xr[1]: virtual tiles 0 and 1, requested by the caller (before its wait);
    // xot / xon from xrows().  The partial goes to the manager of the hidden tile: a sibling publishes it (slot_out), the manager
    // keeps its own in LDS (yl) until it sums the tile (M1).
    auto role1 = [&](auto grad_c, const int tid, const uint32_t (&xot)[4], const uint32_t (&xon)[4], XSlot<XT> (&xr)[2], bool do_fwd,
                     const float (&bfr)[16], const AdamP ap, uint32_t slot_out) {
        constexpr bool GRAD = decltype(grad_c)::value;
        constexpr int NV = GRAD ? 2 * T1 : T1;
        const int lane = tid & 63, wave = tid >> 6, li = lane & 15, lj = lane >> 4;
        const XT* xk = (const XT*)p.X + s.xoff + 4 * (lane & 3);
        float* xs = xst + wave * 2048;                           // two wave-private staging tiles, alternating
        f32x4 pT[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool fwd = res_vfwd<GRAD, SPLIT, T1>(v);
            const int j = res_vtile<GRAD, SPLIT, T1>(v);
            const bool live = tv[j] || j == 0;                   // wave-uniform: a wave's tiles are its first ones (tile 0 always runs: it may be a clamped one)
            // Two waves share a SIMD, and the arbiter favours the older one: rounds 2-5 had waves 0..3 out of the loop ~2 us before waves 4..7, which
            // then ran their last tiles alone -- a lone wave is bound by its own latencies (~1 us per tile pair), not by the matrix pipe
            // (profiles/r06_resident_trace.txt).  Priority by progress: whoever is behind goes first, so the pair finishes together.
            if (GRAD && (v & 3) == 0)
                switch ((v * 4) / NV) {                          // (the builtin wants a literal; the loop is fully unrolled)
                    case 0: __builtin_amdgcn_s_setprio(3); break;
                    case 1: __builtin_amdgcn_s_setprio(2); break;
                    case 2: __builtin_amdgcn_s_setprio(1); break;
                    default: __builtin_amdgcn_s_setprio(0); break;
                }
            float* xv = xs + (v & 1) * 1024;
            if (live) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(f32x4*)(xv + 256 * i + 4 * lane) = xr[v & 1].row16(i);   // wave-private staging (in-order LDS, no barrier)
            }
            if (v + 2 < NV) {                                    // the X tile of position v + 2 into the register set of position v
                const bool f2 = res_vfwd<GRAD, SPLIT, T1>(v + 2);
                const int j2 = res_vtile<GRAD, SPLIT, T1>(v + 2);
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[v & 1].load_row16(i, xk + (f2 ? xon[i] : xot[i]) + 16 * tc[j2]);
            }
            __builtin_amdgcn_sched_barrier(0);                   // the requests leave before this tile's MFMAs
            if (!fwd) {
                // A = X_t^T[d = li][b = 4kb+lj], B = dA[b = 4kb+lj][h = li], kb = 4q + r: four chains over q, summed at the end
                f32x4 xq[4], bq[4], gq[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xq[q] = (f32x4){xv[64 * (4 * q) + lane], xv[64 * (4 * q + 1) + lane], xv[64 * (4 * q + 2) + lane], xv[64 * (4 * q + 3) + lane]};
                    bq[q] = (f32x4){bfr[4 * q], bfr[4 * q + 1], bfr[4 * q + 2], bfr[4 * q + 3]};
                }
                res_mfma4x4<BF>(xq, bq, gq);
                const f32x4 g = (gq[0] + gq[1]) + (gq[2] + gq[3]);
                if (tv[j]) adam4(w1[j], m1[j], v1[j], g, ap);
            } else if (do_fwd && tv[j]) {                        // wave-uniform
                // (Round 6 also requested the forward tiles in the matrix instruction's own operand order -- pass n: row 16n + lane % 16, quarter
                //  lane / 16 IS X_next[b = 16n + li][d = 4 lj ..] -- with no LDS staging: same numbers, 24.4-24.7 vs 24.0 us per step; the register
                //  set is then busy until the instructions have issued, and its next request leaves a tile-time later.)
                f32x4 x4[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) x4[n] = *(const f32x4*)(xv + (16 * n + li) * 16 + 4 * lj);         // X_next[b = 16n+li][d = 4lj+r]
                const f32x4 wj[4] = {w1[j], w1[j], w1[j], w1[j]};
                res_mfma4x4<BF>(wj, x4, pT);                                                    // P^T[h][b] += W1^T X^T
            }
        }
        if (GRAD) __builtin_amdgcn_s_setprio(0);
        RES_STAMP(8)
        RES_MARK(12)
        RES_MARK_WAVE(16)
        // every wave: its stores of this step (re-arms of hand-off slots among them) are acknowledged before the next step's data stores into the
        // slots they re-armed -- here, where nothing else is in flight (every row request of the loop has been consumed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (do_fwd) {
#pragma unroll
            for (int n = 0; n < 4; ++n) *(f32x4*)(pred + wave * 1024 + (16 * n + li) * 16 + 4 * lj) = pT[n];   // P[b = 16n+li][h = 4lj..]
            __syncthreads();
            if (tid < 256) {
                f32x4 a = *(const f32x4*)(pred + 4 * tid);
#pragma unroll
                for (int wv = 1; wv < 8; ++wv) a += *(const f32x4*)(pred + wv * 1024 + 4 * tid);
                if (is_m) *(f32x4*)(yl + 4 * tid) = a;           // the manager's own partial waits in LDS for M1 (the same thread reads it)
                else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the re-arm of this workgroup's other P slot (issued before the tile loop) is in place
                    res_st(rP, slot_out + (uint32_t)(wi * 4096 + 16 * tid), a);
                }
            }
            // role 2 of the next step writes LDS wave by wave with no barrier in front (its Dd tiles and Z partials alias `pred`): the
            // waves that do not sum wait here for those that do -- behind the publishing store, off the step's critical path
            if (is_o) __syncthreads();
        }
        RES_STAMP(9)
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
    // column sums of a [64][16] tile held as one float4 per thread of waves 0..3 (row tid/4, columns 4(tid%4)..): the 16 rows of
    // a wave through four shuffles, the waves' partials to LDS (csum[wave][16]); the caller's next barrier publishes them
    auto col4 = [&](f32x4 v) {
#pragma unroll
        for (int off = 4; off < 64; off <<= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], off);
        if ((threadIdx.x & 63) < 4) *(f32x4*)(csum + (threadIdx.x >> 6) * 16 + 4 * (threadIdx.x & 3)) = v;
    };
    f32x4 y_a = zero4;
    uint32_t xo0[4];                                             // X row offsets of the CURRENT batch (kept from the step before)
    {   // prologue: forward partials of step 0
        const float nob[16] = {0.f};
        const int b0 = p.n_tr < B ? p.n_tr : B;
        AdamP ap0; ap0.alpha = 0.f; ap0.omb1 = p.omb1; ap0.omb2 = p.omb2; ap0.eps = p.eps;
        xrows(tid, 0, b0, xo0);
        XSlot<XT> xr[2];
        const XT* xk = (const XT*)p.X + s.xoff + 4 * (lane & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) { xr[0].load_row16(i, xk + xo0[i] + 16 * tc[0]); xr[1].load_row16(i, xk + xo0[i] + 16 * tc[T1 > 1 ? 1 : 0]); }
        __syncthreads();                                         // b1l written
        role1(std::false_type{}, tid, xo0, xo0, xr, true, nob, ap0, 0u);
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
        const uint32_t pcur = (uint32_t)par * pslot, pnext = (uint32_t)(par ^ 1) * pslot;
        const uint32_t dcur = (uint32_t)par * dslot, dfree = (uint32_t)(par ^ 1) * dslot;
        const uint32_t tcur = (uint32_t)par * tslot, tnext = (uint32_t)(par ^ 1) * tslot;
        const int ub = (tid & 255) >> 2, uq = tid & 3, half = tid >> 8;
        // row indices of the NEXT batch, requested a whole phase before their use (they head two dependent loads)
        int32_t rn[4];
        xrows_raw(tid, (t + 1) * B, b_next, rn);
        const int32_t yrow_n = target_row(tid, t + 1);          // unconditional: a load under a divergent branch makes hipcc drain vmcnt at the join
        RES_MARK_STEP(t)
        RES_MARK(0)
        // the dropout keep word of this thread's four units (M1, manager): requested HERE, a whole P hand-off before its use -- rounds 2-5
        // requested it behind the gather of the siblings' partials, a dependent round trip to HBM (the epoch's keep words are read once)
        // in front of the Dd tile every role-2 workgroup of the sub-net waits for
        unsigned keep_w = 0xffffffffu;
        if (p.rate > 0.f) keep_w = __builtin_amdgcn_raw_buffer_load_b32(rM, moff(t) + (uint32_t)(32 * ((tid & 255) >> 2) + 4 * (ht >> 1)), 0, 0);
        RES_STAMP(10)

        // =============================== M1 (manager): A = sum_s P_s + b1 -> the Dd tile ===============================
        unsigned gate = 0u;                                      // bit r: keep & (A > 0) of this thread's four units (tid < 256), used again in M2
        if (is_m && tid < 256) {
            // waves 0..3, each on its own (no barrier): canaries of the sibling tiles, their pieces, the own partial from LDS
            constexpr int NS = S1C > 0 ? S1C - 1 : 7;            // siblings (run-time S1: up to 7, clamped requests)
            f32x4 a = *(const f32x4*)(yl + 4 * tid);
            if (S1 > 1) {
                const bool okp = res_poll(rP, pcur + (uint32_t)(ht * 4096), S1 - 1, 65536u, abort_w);      // (abort: no piece is waited for below; the next workgroup-wide wait leaves)
                RES_MARK(1)
                f32x4 pv[NS > 0 ? NS : 1];
#pragma unroll
                for (int ss = 0; ss < NS; ++ss) pv[ss] = res_ld(rP, pcur + (uint32_t)(((ss < S1 - 1 ? ss : 0) * 16 + ht) * 4096 + 16 * tid));
#pragma unroll
                for (int ss = 0; ss < NS; ++ss) {
                    if (okp) res_fix(pv[ss], rP, pcur + (uint32_t)(((ss < S1 - 1 ? ss : 0) * 16 + ht) * 4096 + 16 * tid), abort_w);
                    if (ss < S1 - 1) a += pv[ss];
                }
            }
            a += *(const f32x4*)(b1l + 4 * uq);
            unsigned keep = 0xfu;
            if (p.rate > 0.f) keep = keep_w >> (16 * (ht & 1) + 4 * uq);
            if (ub >= b_act) keep = 0u;
            f32x4 dd;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned g1 = ((keep >> r) & 1u) & (a[r] > 0.f ? 1u : 0u);
                gate |= g1 << r;
                dd[r] = g1 ? a[r] * p.scale : 0.f;
            }
            // Round 4: no store round trip on the critical path.  The drain below waits for the re-arm of the Dd slot that M2 of the
            // step before issued BEHIND its dA store, i.e. a whole tile loop ago: it is in place long since, so the Dd tile leaves at
            // once (round 3 issued a re-arm right here and waited ~1 us for its acknowledgement before the tile everybody waits for).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            res_st(rT, tcur + (uint32_t)(ht * 4096 + 16 * tid), dd);
            RES_MARK(2)
            // P(t) of every sibling is in, so each of them has run the tile loop of step t-1, i.e. consumed dA(t-1): its slot is
            // free -- marked "not written" for dA(t+1) BEHIND the Dd tile; M2's drain (a role-2 phase later) acknowledges it before
            // dA(t) -- which is what lets a sibling get as far as polling that slot -- leaves
            if (pub_da) res_st(rA, tnext + (uint32_t)(ht * 4096 + 16 * tid), sent4);
        }
        RES_STAMP(0)

        // Idle time before the dA tile arrives: the first two X tiles of this step's tile loop and (role 2) the targets of the NEXT step.  (These rows
        // come from anywhere in the arena -- 2-3 us -- and vector-memory results return in order: whatever is requested behind them waits for them.
        // Round 6 tried them earlier -- behind the dZ tile, under the dD partial and the W2 gradient -- and lost 0.9 us per step: a CU reads past
        // its L2 at one rate, and the dD stores and the W2 gradient queue behind them.)
        uint32_t xon[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xon[i] = b_next > 0 ? (uint32_t)rn[i] * (uint32_t)s.Dp : xo0[i];
        XSlot<XT> xr[2];
        auto prefetch_x = [&]() {
            const XT* xk = (const XT*)p.X + s.xoff + 4 * (lane & 3);
#pragma unroll
            for (int vv = 0; vv < 2; ++vv) {                 // virtual tiles 0 and 1 of the step's tile loop
                const bool f = res_vfwd<true, SPLIT, T1>(vv);
                const int jj = res_vtile<true, SPLIT, T1>(vv);
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[vv].load_row16(i, xk + (f ? xon[i] : xo0[i]) + 16 * tc[jj]);
            }
            if (t + 1 < p.steps) y_a = targets(tid, yrow_n); // every thread (unconditional load); role 2 uses the first 256
        };
        // =============================== phase A (role 2) ===============================
        if (is_o) {
            if (tid < 256) *(f32x4*)(yl + 4 * tid) = y_a;
            // Round 6: every wave takes ITS two Dd tiles (hidden tiles 2 wave, 2 wave + 1: the 32 hidden units of its Z partial) by itself,
            // straight into the A-operand registers of the matrix instructions -- piece 64 m + 4 li + lj of a [64][16] tile IS
            // Dd[b = 16 m + li][h = 4 lj ..]: no workgroup-wide poll, no staging of all sixteen tiles through LDS, no barrier in front of
            // the first matrix instruction (rounds 3-5: one wave polled all sixteen managers, barrier, 64 KB -> LDS, barrier).  The tiles go
            // to LDS afterwards, for the W2 gradient of the same wave (wave-private columns: no barrier either).
            const uint32_t tb = tcur + (uint32_t)(2 * wave * 4096);
            const bool okd = res_poll(rT, tb, 2, 4096u, abort_w);   // (abort: no piece is waited for below -- an abandoned launch must not spin on pieces that never come; the next workgroup-wide wait leaves)
            RES_MARK(3)
            f32x4 a4[2][4];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int m = 0; m < 4; ++m) a4[h2][m] = res_ld(rT, tb + (uint32_t)(h2 * 4096 + 16 * (64 * m + 4 * li + lj)));
            // Dd(t) of managers 2 wave, 2 wave + 1 is out, so both -- and their siblings, whose P(t) they summed first -- have finished
            // gathering the dD partials of the step before: tiles (ot, 2 wave), (ot, 2 wave + 1) of that dD slot are free -- marked "not
            // written" for the step after this one; acknowledged by the drain at the end of this step's tile loop (role1), a step before
            // this wave stores into them again.
#pragma unroll
            for (int i = 0; i < 8; ++i) res_st(rD, dfree + (uint32_t)(ot * 65536 + 2 * wave * 4096 + (i * 64 + lane) * 16), sent4);
            const float* ws = w2s + 2 * wave * 256;              // this wave's two W2 tiles [h][o]
            RES_STAMP(1)
            {   // Z partial over this wave's 32 hidden units
                f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    f32x4 bq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) bq[r] = ws[h2 * 256 + (4 * lj + r) * 16 + li];   // W2[h = 4lj+r][o = li]
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (okd) res_fix(a4[h2][m], rT, tb + (uint32_t)(h2 * 4096 + 16 * (64 * m + 4 * li + lj)), abort_w);
                        acc[m] = res_mfma4<BF>(a4[h2][m], bq, acc[m]);
                    }
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int m = 0; m < 4; ++m) *(f32x4*)(ddl + (16 * m + li) * ldd + 16 * (2 * wave + h2) + 4 * lj) = a4[h2][m];
                RES_STAMP(2)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) zred[wave * 1024 + (16 * m + 4 * lj + r) * 16 + li] = acc[m][r];
            }
            RES_MARK(4)
            __syncthreads();
            RES_MARK(5)
            {   // epilogue of the forward: two elements per thread of the [64][16] tile
                float lsum = 0.f, dzc = 0.f;
                const bool col_ok = (16 * ot + (tid & 15)) < dm.O;
                const float bias = smallf[32 + (tid & 15)];      // b2 tile (written at the end of the previous step / prologue)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int e = tid + 512 * hh, b = e >> 4;
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
                    dzc += dz;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
                if (lane == 0) smallf[wave] = lsum;
                // column sums of dZ (the b2 gradient): a thread holds rows tid/16 and 32 + tid/16 of column tid%16 -- the four
                // rows of a wave through two shuffles, the eight waves through LDS (a serial loop over 64 rows cost ~4k clocks)
                dzc += __shfl_xor(dzc, 16);
                dzc += __shfl_xor(dzc, 32);
                if (lane < 16) csum[wave * 16 + lane] = dzc;
            }
            __syncthreads();
            RES_MARK(6)
            if (tid == 0) {
                float tot = 0.f;
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) tot += smallf[wv];
                loss_total += (double)tot;
            }
            RES_STAMP(3)
            if (tid < 16) {                                      // gb2 = column sums of dZ -> Adam(b2)
                float gb = 0.f;
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) gb += csum[wv * 16 + tid];
                adam1(b2w0, b2m0, b2v0, gb, ap);
            }
            {   // dD^T partial with the OLD W2: published BEFORE the W2 gradient, which then runs while the managers
                // already pick the partials up
                f32x4 zf[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) zf[n] = *(const f32x4*)(dzl + (16 * n + li) * 16 + 4 * lj);      // dZ[b = 16n+li][o = 4lj+r]
                // (the re-arm of the other dD slot, issued by this wave behind its Dd requests of the step BEFORE, was acknowledged by the drain at
                //  the end of that step's tile loop: no drain here -- it would wait for the rows requested above)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int tile = 2 * wave + h2;
                    const f32x4 wq = *(const f32x4*)(w2s + tile * 256 + w2o);                                   // OLD W2 (h = li, o = 4lj..)
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const f32x4 d = res_mfma4<BF>(wq, zf[n], zero4);                                         // dD^T[h][b] = W2 dZ^T
                        res_st(rD, dcur + (uint32_t)(((ot * 16 + tile) * 1024 + (16 * n + li) * 16 + 4 * lj) * 4), d);   // [b = 16n+li][h = 4lj..]
                    }
                }
            }
            RES_STAMP(4)
            RES_MARK(7)
            // the wave that will poll for the dA tile requests ITS rows now (they are back under the W2 gradient): vector-memory results return in
            // order, and its poll was seen ~0.7 us after the tile had landed while it queued behind them (22.24 -> 21.83 us per step; all eight
            // waves requesting here lose 0.7: the dD stores and the W2 gradient queue behind 64 KB of rows)
            if (wave == 0 && !is_m) prefetch_x();
            {   // W2 gradient + Adam on the LDS-resident state
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int tile = 2 * wave + h2;
                    f32x4 g = zero4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                // dZ^T Dd: A = dZ[b = 4kb+lj][o = li], B = Dd[b = 4kb+lj][h = li], kb = 4q + r
                        f32x4 zq, dq;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { zq[r] = dzl[64 * (4 * q + r) + lane]; dq[r] = ddl[(4 * (4 * q + r) + lj) * ldd + 16 * tile + li]; }
                        g = res_mfma4<BF>(zq, dq, g);
                    }
                    f32x4 wq = *(const f32x4*)(w2s + tile * 256 + w2o);
                    f32x4 mq = *(const f32x4*)(w2s + 4096 + tile * 256 + w2o), vq = *(const f32x4*)(w2s + 8192 + tile * 256 + w2o);
                    adam4(wq, mq, vq, g, ap);
                    *(f32x4*)(w2s + tile * 256 + w2o) = wq; *(f32x4*)(w2s + 4096 + tile * 256 + w2o) = mq; *(f32x4*)(w2s + 8192 + tile * 256 + w2o) = vq;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();                                     // everybody is done reading ddl/dzl/zred (what follows re-uses them)
            if (tid < 16) smallf[32 + tid] = b2w0;
            RES_STAMP(11)
            RES_MARK(8)
        }

        // =============================== the dA tile: M2 (manager) or the hand-off from it (siblings) ===============================
        {
            RES_STAMP(5)
            if (!(is_o && !is_m && wave == 0)) prefetch_x();
            if (is_m) {
                if (wave == 0) { const bool ok = res_poll(rD, dcur + (uint32_t)(ht * 4096), OT, 65536u, abort_w); if (lane == 0) flagl[1] = ok ? 1 : 0; }
                __syncthreads();
                if (!flagl[1]) return;
                RES_STAMP(6)
                RES_MARK(9)
                // tile ht of the OT producers' dD partials, their two halves on the two thread halves
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
                            res_fix(tq[i], rD, base + (uint32_t)((o + i) * 65536), abort_w);
                            d += tq[i];
                            tq[i] = res_ld(rD, base + (uint32_t)((o + 8 + i) * 65536));
                        }
#pragma unroll
                        for (int i = 0; i < 8; ++i) { res_fix(tq[i], rD, base + (uint32_t)((o + 8 + i) * 65536), abort_w); d += tq[i]; }
                        o += 16;
                    }
                    for (; o + 4 <= o1; o += 4) {
                        f32x4 tq[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) tq[i] = res_ld(rD, base + (uint32_t)((o + i) * 65536));
#pragma unroll
                        for (int i = 0; i < 4; ++i) { res_fix(tq[i], rD, base + (uint32_t)((o + i) * 65536), abort_w); d += tq[i]; }
                    }
                    for (; o < o1; ++o) { f32x4 t1 = res_ld(rD, base + (uint32_t)(o * 65536)); res_fix(t1, rD, base + (uint32_t)(o * 65536), abort_w); d += t1; }
                }
                if (half) *(f32x4*)(yl + 4 * (tid & 255)) = d;
                __syncthreads();
                if (half == 0) {
                    d += *(const f32x4*)(yl + 4 * tid);
                    f32x4 da;
#pragma unroll
                    for (int r = 0; r < 4; ++r) da[r] = ((gate >> r) & 1u) ? d[r] * p.scale : 0.f;
                    *(f32x4*)(dzl + 4 * tid) = da;               // dA tile [64][16]
                    if (pub_da) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the re-arm of this dA slot's sibling, issued behind the Dd tile in M1: long in place)
                        res_st(rA, tcur + (uint32_t)(ht * 4096 + 16 * tid), da);
                    }
                    RES_MARK(10)
                    // every dD partial of this step was out, so every role-2 workgroup has read the Dd tiles of this step: that slot is
                    // free -- marked "not written" for step t+2, BEHIND the dA tile; acknowledged by M1's drain of the next step, a tile
                    // loop from here, before Dd(t+1) leaves
                    res_st(rT, tcur + (uint32_t)(ht * 4096 + 16 * tid), sent4);
                    col4(da);
                }
            } else {
                if (wave == 0) { const bool ok = res_poll(rA, tcur + (uint32_t)(ht * 4096), 1, 4096u, abort_w); if (lane == 0) flagl[1] = ok ? 1 : 0; }
                __syncthreads();
                if (!flagl[1]) return;
                RES_STAMP(6)
                RES_MARK(9)
                if (tid < 256) {
                    f32x4 da = res_ld(rA, tcur + (uint32_t)(ht * 4096 + 16 * tid));
                    res_fix(da, rA, tcur + (uint32_t)(ht * 4096 + 16 * tid), abort_w);
                    RES_MARK(10)
                    *(f32x4*)(dzl + 4 * tid) = da;
                    // dA(t) exists, so the manager has summed P(t): this workgroup's P slot of step t is free -- marked "not
                    // written" for P(t+2); acknowledged before P(t+1) leaves (the drain at the end of the tile loop)
                    res_st(rP, pcur + (uint32_t)(wi * 4096 + 16 * tid), sent4);
                    // (no column sums here: a sibling keeps no b1 -- only the manager's M1 reads it, and the manager saves it)
                }
            }
            __syncthreads();
            if (is_m && tid < 16) {                              // gb1 = column sums of dA -> Adam(b1): the manager's copy is the one M1 reads and the epilogue saves
                                                                 // (rounds 3-5: every sibling too, on the wave that the tile loop then waited for: 21.89 -> 21.56 us per step without)
                const float gb = (csum[tid] + csum[16 + tid]) + (csum[32 + tid] + csum[48 + tid]);
                adam1(b1w0, b1m0, b1v0, gb, ap);
                b1l[tid] = b1w0;
            }
            float bfr[16];
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) bfr[kb] = dzl[64 * kb + lane];                                        // dA[b = 4kb+lj][h = li]
            RES_STAMP(7)
            RES_MARK(11)
            RES_MARK_WAVE(24)
            role1(std::true_type{}, tid, xo0, xon, xr, b_next > 0, bfr, ap, pnext);     // its first barrier orders b1l / dzl
            RES_MARK(13)
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
    if (is_m && tid < 16) { p.b1w[b1i] = b1w0; p.b1m[b1i] = b1m0; p.b1v[b1i] = b1v0; }
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
