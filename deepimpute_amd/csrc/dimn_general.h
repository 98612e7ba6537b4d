// dimn_general.h -- the GENERAL path of libdimn: every architecture MultiNet.build() accepts that the tuned kernels of
// dimn_kernels.h do not take (reference deepimpute/multinet.py:135-143: any sequence of Dense / Dropout layers;
// :150-162: other losses; parser.py:50-66: any batch size, any hidden width).  Same arithmetic definitions (Keras-form
// Adam, Philox dropout streams, softplus output, wMSE), plain structure: per layer one batched fp32-MFMA GEMM over the
// sub-nets with the layer's epilogue fused -- k_gen_rowgemm (round 6: operands straight from global memory in the matrix
// instructions' lane layout, aligned shapes) for the forward and the hidden backward, k_gen_gemm (64 x 64 output tile, operands
// staged through LDS, transposes resolved while staging) for the weight gradients with Keras-Adam and the bias gradient behind
// them and for whatever k_gen_rowgemm does not take --, the output layer's loss in the last forward's epilogue (training) or
// k_gen_output (validation, prediction).  The default architecture (one hidden layer <= 384, batch <= 64) never comes here.
#pragma once
#include "dimn_kernels.h"

// loss ids: DIMN_LOSS_* of include/dimn.h

struct GDesc {                 // one GEMM of one sub-net:  C[M][N] = op(A)[M][K] * op(B)[K][N]
    const float* A; const float* B; float* C;
    const float* bias;         // epilogue 1, 2: [N]
    float* G;                  // epilogue 1: gate out [M][N] (ldc); epilogue 3: gate in
    int32_t N, K, lda, ldb, ldc, kg;
    int32_t agather;           // 1: A is not a dense matrix but the batch rows of this sub-net's block of the X arena (GEpi.xbase / arows): the
                               //    first layer's forward and weight-gradient GEMMs read the gathered predictors in place (fp32 arenas)
};

struct GEpi {
    int32_t mode;              // 0 store; 1 hidden forward (bias, activation, dropout -> C = H, G = gate); 2 bias only; 3 C = acc * G;
                               // 4 (weight gradients): Keras-Adam on the parameters the tile belongs to -- C addresses the tile inside the flat
                               //   PARAMETER array P, whose offsets the m / v arrays share: the gradient itself never goes to memory
    int32_t act, train;        // mode 1
    float rate, scale;
    uint64_t seed; uint32_t epoch, step;   // step already carries the dropout layer in its top byte
    float *P, *Mo, *Vo;                    // mode 4: flat parameter array and Adam moments (same offsets)
    AdamP ap;                              // mode 4
    const float* xbase; const SubnetDev* sn;   // agather descriptors: the X arena and the sub-nets' blocks in it (xoff, Dp), descriptor i <-> sub-net i
    const int32_t* arows; int64_t arow0;       //   batch row b = arows[b] (device row list) or arow0 + b
    const float* Y; int64_t n_cells; int32_t Op, loss; float inv_n; double* loss_sum;   // k_gen_rowgemm EPI 4: the output layer's loss and dZ fused behind its forward (as k_gen_output, train)
    int32_t ksplit;                        // > 1: split-K -- grid.z = descriptors x ksplit, every workgroup multiplies one k-range and stores its raw
    float* part; int64_t part_stride;      //   partial tile into part[(desc * ksplit + s) * part_stride + row * N + col]; k_gen_splitk_fin applies `mode`
};

// Operand pointers come out of a descriptor in memory, so the compiler cannot tell they are global: it emits flat_load, whose results may
// come back out of order with LDS traffic -- every wait becomes vmcnt(0) lgkmcnt(0) and nothing stays in flight across the matrix
// instructions.  These say "global".
typedef __attribute__((address_space(1))) const f32x4* gen_gp4;
typedef __attribute__((address_space(1))) const float* gen_gp1;
__device__ __forceinline__ f32x4 gen_gld4(const float* p) { return *(gen_gp4)p; }
__device__ __forceinline__ float gen_gld1(const float* p) { return *(gen_gp1)p; }
__device__ __forceinline__ void gen_gst4(float* p, f32x4 v) { *(__attribute__((address_space(1))) f32x4*)p = v; }
__device__ __forceinline__ void gen_gst1(float* p, float v) { *(__attribute__((address_space(1))) float*)p = v; }

// element r of a four-vector / one Philox block with r a run-time index (rolled loops over the four elements keep ONE copy of every activation
// in the code: see the note on code size at k_gen_rowgemm)
__device__ __forceinline__ float gen_sel4(const f32x4 v, int r) { return r == 0 ? v[0] : (r == 1 ? v[1] : (r == 2 ? v[2] : v[3])); }
__device__ __forceinline__ void gen_put4(f32x4& v, int r, float x) { v[0] = r == 0 ? x : v[0]; v[1] = r == 1 ? x : v[1]; v[2] = r == 2 ? x : v[2]; v[3] = r == 3 ? x : v[3]; }
// bias, activation and dropout of four consecutive units of one row (hidden forward): C = H, G = gate
__device__ __forceinline__ void gen_hidden4(const GEpi& ep, const f32x4 x, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, bool drop, f32x4& c, f32x4& g) {
    c = g = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        float f, df;
        hidden_act(ep.act, gen_sel4(x, r), f, df);
        if (ep.train) {
            const uint32_t w = r == 0 ? r0 : (r == 1 ? r1 : (r == 2 ? r2 : r3));
            const bool keep = !drop || dimn_u01(w) >= ep.rate;
            gen_put4(c, r, keep ? f * ep.scale : 0.f);
            gen_put4(g, r, keep ? df * ep.scale : 0.f);
        } else gen_put4(c, r, f);
    }
}

// the per-element epilogue of modes 0 .. 3 (k_gen_gemm without split-K, k_gen_splitk_fin with it)
__device__ __forceinline__ void gen_epilogue(const GEpi& ep, const GDesc& d, int row, int col, float v) {
    const int64_t o = (int64_t)row * d.ldc + col;
    if (ep.mode == 1) {
        v += gen_gld1(d.bias + col);
        float f, df;
        hidden_act(ep.act, v, f, df);
        if (ep.train) {
            const bool keep = !(ep.rate > 0.f) || dimn_dropout_keep(ep.seed, (uint32_t)d.kg, ep.epoch, ep.step, (uint32_t)(row * d.N + col), ep.rate);
            gen_gst1(d.C + o, keep ? f * ep.scale : 0.f);
            gen_gst1(d.G + o, keep ? df * ep.scale : 0.f);
        } else {
            gen_gst1(d.C + o, f);
        }
    } else if (ep.mode == 2) {
        gen_gst1(d.C + o, v + gen_gld1(d.bias + col));
    } else if (ep.mode == 3) {
        gen_gst1(d.C + o, v * gen_gld1(d.G + o));
    } else {
        gen_gst1(d.C + o, v);
    }
}

// dropout key of layer `dl` (0 = first Dropout layer, the stream of the tuned kernels): the layer ordinal rides in the
// top byte of the step word, so dl = 0 reproduces dimn_dropout_block(seed, kg, epoch, step, ...) exactly
__host__ __device__ static inline uint32_t gen_step_key(uint32_t step, int dl) { return (step & 0xFFFFFFu) | ((uint32_t)dl << 24); }

// Mo >= 0: the row count of every sub-net (batch rows); Mo < 0: the descriptor's K field is the row count (weight
// gradients: rows = inputs of the layer, which differ per sub-net) and Ko is the inner dimension (the batch).
// Round 4: 64-deep k-steps (round 3: 16-deep -- the first layer's forward, K = D ~ 2 400 on 4 workgroups per sub-net, spent 150
// dependent load -> LDS -> barrier rounds of ~1.3 us each: 0.2 ms of a 0.46 ms step, rocprofv3) and a 64 x 32 output tile (BN = 32)
// for the batch-row GEMMs, whose grids were 160 workgroups on 256 CUs.
// The weight-gradient instance (TA: K = the batch, one or two steps) takes 32-deep steps and lets its output tile alias the
// operand tiles: 17 KB of LDS, eight workgroups per CU -- with 64-deep steps and a tile of its own (52 KB, three per CU) the
// 6 080 workgroups of the first layer's gradient ran 139 instead of 120 us.
template <bool TA, bool TB, int BN, int BK = 64>
__global__ __launch_bounds__(256) void k_gen_gemm(const GDesc* __restrict__ descs, int Mo, int Ko, GEpi ep) {
    constexpr int NJ = BN / 32;                                  // 16-column tiles per wave
    constexpr int LDB = BN + 4;
    static_assert(!TA || BK * 68 + BK * LDB >= 64 * 68, "the output tile of mode 4 aliases the operand tiles");
    __shared__ __attribute__((aligned(16))) float smem[BK * 68 + BK * LDB];
    float (*As)[68] = (float (*)[68])smem;                       // [k][m]; rows 16-byte aligned, stride 4 mod 32 words
    float (*Bs)[LDB] = (float (*)[LDB])(smem + BK * 68);         // [k][n]
    float (*Cs)[68] = (float (*)[68])smem;                       // mode 4 (the weight-gradient instance): the output tile, re-read row-wise
    // split-K (ep.ksplit > 1; batch-row GEMMs with a long inner dimension -- the first layer's forward: K = D ~ 2 400 against 64 rows):
    // workgroup z multiplies k-range z % ksplit of descriptor z / ksplit and stores its raw partial tile; k_gen_splitk_fin adds the
    // ranges in order and applies the epilogue.  Without it that GEMM is 4 workgroups per sub-net walking 38 dependent
    // load -> LDS -> barrier steps: 146 us for 98 MB of weights (rocprofv3, 40 sub-nets).
    const int S = ep.ksplit > 1 ? ep.ksplit : 1;
    const int zi = (int)blockIdx.z / S, sp = (int)blockIdx.z - zi * S;
    GDesc d = descs[zi];
    const int M = Mo >= 0 ? Mo : d.K;
    if (Ko >= 0) d.K = Ko;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * BN;
    if (n0 >= d.N || m0 >= M) return;
    int kbeg = 0;
    if (S > 1) {
        const int kc = ((d.K + S - 1) / S + BK - 1) / BK * BK;   // whole 64-deep steps per range (a late range may be empty: it stores zeros)
        kbeg = sp * kc;
        d.K = d.K < kbeg + kc ? d.K : kbeg + kc;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * (BN / 2);
    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Staging: every thread moves 16-byte pieces -- four elements along the operand's memory-contiguous index (k for a row-major A /
    // a transposed B, m or n otherwise) -- requested for the NEXT step before the current one is multiplied.  Pieces that cross an
    // edge, or operands whose rows are not 16-byte aligned, fall back to guarded scalar loads.
    // Round 4: the first layer reads its batch rows straight from the gathered X arena (row b of the operand = arena row arows[b]):
    // the dense copy k_gen_gather_batch made of them every step -- 24.6 MB out and in again, 18.5 us -- is gone for fp32 arenas.
    const float* Abase = d.A;
    int64_t a_ld = d.lda;
    if (d.agather) { const SubnetDev sd = ep.sn[zi]; Abase = ep.xbase + sd.xoff; a_ld = sd.Dp; }
    constexpr bool a_k = !TA, b_k = TB;                          // the operand's contiguous index is k
    constexpr int NA = BK / 16, NB = BK * BN / 1024;             // 16-byte pieces per thread: A 64 x BK, B BK x BN
    constexpr int KQ = BK / 4, NQ = BN / 4;                      // pieces along k / along n
    const bool a_vec = (a_ld & 3) == 0 && ((uintptr_t)Abase & 15) == 0, b_vec = (d.ldb & 3) == 0 && ((uintptr_t)d.B & 15) == 0;
    auto piece = [&](const float* base, int64_t ld, int gmaj, int gmin, int lim_maj, int lim_min, bool vec, bool rows = false) -> f32x4 {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (gmaj < lim_maj) {
            const int64_t r = rows ? (ep.arows ? (int64_t)ep.arows[gmaj] : ep.arow0 + gmaj) : (int64_t)gmaj;    // (the major index of A is the batch row in both forms)
            const float* p = base + r * ld + gmin;
            if (vec && gmin + 3 < lim_min) v = gen_gld4(p);
            else
                for (int r = 0; r < 4; ++r) if (gmin + r < lim_min) v[r] = gen_gld1(p + r);
        }
        return v;
    };
    // piece q of this thread: (major, minor) inside the tile
    auto load_a = [&](int k0, f32x4 (&ra)[NA]) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int p = tid + 256 * q;
            if (a_k) { const int maj = p / KQ, mn = (p % KQ) * 4; ra[q] = piece(Abase, a_ld, m0 + maj, k0 + mn, M, d.K, a_vec, d.agather != 0); }
            else { const int maj = p >> 4, mn = (p & 15) * 4; ra[q] = piece(Abase, a_ld, k0 + maj, m0 + mn, d.K, M, a_vec, d.agather != 0); }
        }
    };
    auto load_b = [&](int k0, f32x4 (&rb)[NB]) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int p = tid + 256 * q;
            if (b_k) { const int maj = p / KQ, mn = (p % KQ) * 4; rb[q] = piece(d.B, d.ldb, n0 + maj, k0 + mn, d.N, d.K, b_vec); }
            else { const int maj = p / NQ, mn = (p % NQ) * 4; rb[q] = piece(d.B, d.ldb, k0 + maj, n0 + mn, d.K, d.N, b_vec); }
        }
    };
    f32x4 ra[NA], rb[NB];
    load_a(kbeg, ra); load_b(kbeg, rb);
    const bool colsum = TA && ep.mode == 4 && d.bias != nullptr && blockIdx.y == 0;
    float cs = 0.f;
    for (int k0 = kbeg; k0 < d.K; k0 += BK) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int p = tid + 256 * q;
            if (a_k) { const int maj = p / KQ, mn = (p % KQ) * 4; for (int r = 0; r < 4; ++r) As[mn + r][maj] = ra[q][r]; }
            else { const int maj = p >> 4, mn = (p & 15) * 4; *(f32x4*)&As[maj][mn] = ra[q]; }
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int p = tid + 256 * q;
            if (b_k) { const int maj = p / KQ, mn = (p % KQ) * 4; for (int r = 0; r < 4; ++r) Bs[mn + r][maj] = rb[q][r]; }
            else { const int maj = p / NQ, mn = (p % NQ) * 4; *(f32x4*)&Bs[maj][mn] = rb[q]; }
        }
        __syncthreads();
        if (k0 + BK < d.K) { load_a(k0 + BK, ra); load_b(k0 + BK, rb); }
        if constexpr (TA && BN == 64) {
            // Round 6: the bias gradient rides here (it was a launch of its own, k_gen_colsum_adam, 7 us per layer): the workgroups of the
            // first row block see every dZ tile of their 64 columns pass through LDS.  Same order of additions as that kernel: batch rows
            // b = rg, rg + 4, ... per row group (k0 is a multiple of 4), then (p0 + p1) + (p2 + p3).
            if (colsum) for (int kk = tid >> 6; kk < BK; kk += 4) cs += Bs[kk][tid & 63];
        }
        const int kend = d.K - k0 < BK ? ((d.K - k0 + 3) & ~3) : BK;      // (the tile beyond K is zero-filled: whole 4-deep instructions only)
        for (int kk = 0; kk < kend; kk += 4) {
            float a[2], b[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[kk + lj][wm + 16 * i + li];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = Bs[kk + lj][wn + 16 * j + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = MFMA16(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    if constexpr (TA && BN == 64) {
        if (ep.mode == 4) {
            // Round 4: Adam fused behind the weight-gradient GEMM (round 3 wrote the gradient -- 4 B per parameter -- and a separate
            // pass over the flat arrays read it back with w, m, v: 36 B per parameter and step, now 24).  The tile goes through LDS
            // so that every thread owns float4 pieces of a ROW: 256-byte runs per row and wave on each of the six streams, instead
            // of the accumulator layout's 64-byte ones.  Same gradient bits, same adam1 per element as the separate pass.
            __shared__ float csp[4][64];
            if (colsum) csp[tid >> 6][tid & 63] = cs;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Cs[wm + 16 * i + 4 * lj + r][wn + 16 * j + li] = acc[i][j][r];
            __syncthreads();
            if (colsum && tid < 64 && n0 + tid < d.N) {
                const float g = (csp[0][tid] + csp[1][tid]) + (csp[2][tid] + csp[3][tid]);
                const int64_t o = (d.bias - ep.P) + n0 + tid;
                float w = ep.P[o], m = ep.Mo[o], v = ep.Vo[o];
                adam1(w, m, v, g, ep.ap);
                ep.P[o] = w; ep.Mo[o] = m; ep.Vo[o] = v;
            }
            const int64_t base = d.C - ep.P;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pc = tid + 256 * q, rr = pc >> 4, c4 = (pc & 15) * 4;
                const int row = m0 + rr, col = n0 + c4;
                if (row >= M || col >= d.N) continue;
                const int64_t o = base + (int64_t)row * d.ldc + col;
                if (col + 3 < d.N && ((o | d.ldc) & 3) == 0) {
                    f32x4 w = *(const f32x4*)(ep.P + o), m = *(const f32x4*)(ep.Mo + o), v = *(const f32x4*)(ep.Vo + o);
                    const f32x4 g = *(const f32x4*)&Cs[rr][c4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { float w1 = w[r], m1 = m[r], v1 = v[r]; adam1(w1, m1, v1, g[r], ep.ap); w[r] = w1; m[r] = m1; v[r] = v1; }
                    *(f32x4*)(ep.P + o) = w; *(f32x4*)(ep.Mo + o) = m; *(f32x4*)(ep.Vo + o) = v;
                } else {
                    for (int r = 0; r < 4 && col + r < d.N; ++r) {
                        float w = ep.P[o + r], m = ep.Mo[o + r], v = ep.Vo[o + r];
                        adam1(w, m, v, Cs[rr][c4 + r], ep.ap);
                        ep.P[o + r] = w; ep.Mo[o + r] = m; ep.Vo[o + r] = v;
                    }
                }
            }
            return;
        }
    }
    if constexpr (!TA) {                                         // (the weight-gradient instance only ever runs mode 4: sixteen copies of every activation less to fetch)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = n0 + wn + 16 * j + li;
                if (col >= d.N) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm + 16 * i + 4 * lj + r;
                    if (row >= M) continue;
                    if (S > 1) ep.part[((int64_t)zi * S + sp) * ep.part_stride + (int64_t)row * d.N + col] = acc[i][j][r];
                    else gen_epilogue(ep, d, row, col, acc[i][j][r]);
                }
            }
    }
}

// second half of a split-K GEMM: element (row, col) = sum over the k-ranges of the partial tiles, in range order, then the epilogue.
// grid (ceil(M * N / 1024), descriptors), 256 threads x FOUR CONSECUTIVE elements: one 16-byte load per partial tile, and (hidden
// layers in training) ONE Philox block for the four keep decisions -- element e draws word e & 3 of block e >> 2, so four aligned
// elements share a block (the first version took one element per thread and computed the block four times: 25 us per launch).
__global__ __launch_bounds__(256) void k_gen_splitk_fin(const GDesc* __restrict__ descs, int M, GEpi ep) {
    const GDesc d = descs[blockIdx.y];
    const int S = ep.ksplit;
    const float* p0 = ep.part + (int64_t)blockIdx.y * S * ep.part_stride;
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4, total = (int64_t)M * d.N;
    if (e0 >= total) return;
    const bool fast = (d.N & 3) == 0 && (ep.part_stride & 3) == 0 && (d.ldc & 3) == 0 && ep.mode == 1 && e0 + 3 < total;
    if (!fast) {
        for (int q = 0; q < 4 && e0 + q < total; ++q) {
            const int64_t e = e0 + q;
            const int row = (int)(e / d.N), col = (int)(e - (int64_t)row * d.N);
            float v = 0.f;
            for (int s2 = 0; s2 < S; ++s2) v += p0[(int64_t)s2 * ep.part_stride + e];
            gen_epilogue(ep, d, row, col, v);
        }
        return;
    }
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < S; ++s2) v += *(const f32x4*)(p0 + (int64_t)s2 * ep.part_stride + e0);
    const int row = (int)(e0 / d.N), col = (int)(e0 - (int64_t)row * d.N);      // N % 4 == 0: the four elements share the row
    const int64_t o = (int64_t)row * d.ldc + col;
    const f32x4 bias = gen_gld4(d.bias + col);
    dimn_u32x4 rnd = {{0u, 0u, 0u, 0u}};
    const bool drop = ep.train && ep.rate > 0.f;
    if (drop) rnd = dimn_dropout_block(ep.seed, (uint32_t)d.kg, ep.epoch, ep.step, (uint32_t)(e0 >> 2));     // (row * N + col == e0)
    f32x4 c, g;
    gen_hidden4(ep, v + bias, rnd.v[0], rnd.v[1], rnd.v[2], rnd.v[3], drop, c, g);
    gen_gst4(d.C + o, c);
    if (ep.train) gen_gst4(d.G + o, g);
}

#define GEN_OUT_CH 64         // loss slots per sub-net: one owner workgroup per slot and launch (k_gen_output, k_gen_rowgemm's output epilogue)
// One element of build()'s `loss` (multinet.py:150-162): the term that is summed and dL/dyhat * N, from the target y, yhat = sp and er = y - sp.
__device__ __forceinline__ void gen_loss_term(int loss, float y, float sp, float er, float& term, float& dy) {
    if (loss == DIMN_LOSS_MAE) { term = fabsf(er); dy = er > 0.f ? -1.f : (er < 0.f ? 1.f : 0.f); }
    else if (loss == DIMN_LOSS_MSLE) {                       // keras: first_log = log(max(yhat, eps) + 1), second_log = log(max(y, eps) + 1)
        const float a = fmaxf(sp, 1e-7f), d = log1pf(fmaxf(y, 1e-7f)) - log1pf(a);
        term = d * d; dy = sp > 1e-7f ? -2.f * d / (a + 1.f) : 0.f;
    } else if (loss == DIMN_LOSS_LOGCOSH) {                  // x + softplus(-2x) - log 2, x = yhat - y; d/dx = tanh x
        const float x = -er;
        term = x + softplus_f(-2.f * x) - 0.69314718055994531f; dy = tanhf(x);
    } else if (loss == DIMN_LOSS_HUBER) {                    // delta = 1
        const float ae = fabsf(er);
        term = ae <= 1.f ? 0.5f * er * er : ae - 0.5f; dy = ae <= 1.f ? -er : (er > 0.f ? -1.f : 1.f);
    } else if (loss == DIMN_LOSS_POISSON) {
        term = sp - y * logf(sp + 1e-7f); dy = 1.f - y / (sp + 1e-7f);
    } else {
        const float w = loss == DIMN_LOSS_WMSE ? y : (loss == DIMN_LOSS_WMSE_BINARY ? (y > 0.f ? 1.f : 0.f) : 1.f);
        term = w * er * er; dy = -2.f * w * er;
    }
}

// ---- round 6: the batch-row GEMMs (forward of every layer, hidden backward) with NO operand staged through LDS ----
// C[M][N] = A[M][K] * B;  TB = false: B is [K][N] row-major (forward: the layer's kernel);  TB = true: B is given as [N][K] (hidden backward:
// dZ W^T reads the kernel as stored).  k_gen_gemm moves both operands global -> registers -> LDS (the row-major A transposed by scalar
// ds_writes, eight-way bank conflicts) -> registers, one barrier pair per 64-deep step.  Here the operands of the matrix instructions come
// straight from global memory in the instructions' own lane layout:
//   A: lane (li, lj) loads 16 bytes of row 16 m + li at k = 16 c + 4 lj .. + 3: element r feeds instruction r, whose four k's are then
//      {16 c + 4 lj + r}: any assignment of k's to instructions gives the same sum as long as B follows it;
//   B (TB = false): 16 bytes of row k = 16 c + 4 lj + r at columns n0 + 4 li .. + 3: element j feeds the instruction of column set j, so
//      accumulator j holds columns n0 + 4 li + j -- a permutation of the block's columns that costs nothing and turns the epilogue into
//      16-byte pieces of a row; every B instruction reads 4 rows x 256 contiguous bytes;
//   B (TB = true): 16 bytes of row n0 + 16 nt + li at k = 16 c + 4 lj .. + 3, as A.
// 4 + 4 requests of 16 bytes per 64 matrix instructions, three chunks of 16 k's in flight per wave, no barrier and no branch in
// the steady-state loop.  Two lessons of the first versions, both measured with tools/probe/gemm_probe.hip:
//   * operand pointers that come out of a descriptor in memory are "flat" to the compiler (see gen_gld4 above): every wait was vmcnt(0);
//   * CODE SIZE is launch time.  With the guarded forms of every request (unaligned operands, ragged edges) and every epilogue of the path
//     compiled into one kernel, a launch of 320 workgroups with ONE chunk each took 32 us against 2.7 us for an empty kernel of the same
//     footprint and 9.7 us without the unused code: 256 CUs miss their instruction caches on the same lines at the same moment.  So
//     this kernel is the ALIGNED form only (gen_rowgemm_ok on the host: 16-byte aligned operands and outputs, N % 4 == 0; everything
//     else stays on k_gen_gemm), the epilogue is a template parameter, the ragged end of K is one masked chunk without a branch.
// A workgroup owns 64 x 64 of C over one k-range; wave w takes every fourth chunk and accumulates the whole block, the four blocks are
// added through LDS in the fixed order (w0 + w2) + (w1 + w3).  (Tried and dropped, same probe: waves along N with the workgroup walking B in
// whole 1 KB rows -- no better at any number of k-ranges; blocks of 128 rows for batches > 64 -- 396 registers, one wave per SIMD, 217
// against 178 us per 256-row validation block: the kernel of a layer read again per 64-row block comes from L2.)
// EPI: 1 hidden forward (bias, activation, dropout -> C = H, G = gate), 2 bias, 3 C = acc * G, 9 the raw block into the split-K scratch
//      (ep.ksplit k-ranges; k_gen_splitk_fin adds them and applies the layer's epilogue), 4 the output layer of a training step: bias,
//      softplus, the loss term and dZ in place of Z (what k_gen_output does in a launch of its own: 13 us per step), the workgroup's loss
//      sum into its slot of loss_sum[descriptor][GEN_OUT_CH] (the host launches this form only while a sub-net has <= GEN_OUT_CH blocks).
// Workgroups of one descriptor share their A rows: with xcd_map they sit on one XCD (block b runs on XCD b % 8).
template <bool TB, int EPI>
__global__ __launch_bounds__(256) void k_gen_rowgemm(const GDesc* __restrict__ descs, int nd, int M, int nblk, int mblk, int xcd_map, GEpi ep) {
    constexpr int MT = 4, ROWS = 16 * MT, KW = 4;                // 64 x 64 of C per workgroup; wave w takes every fourth chunk
    static_assert(TB ? EPI == 3 : (EPI == 1 || EPI == 2 || EPI == 4 || EPI == 9), "epilogues of the forward / the hidden backward");
    __shared__ __attribute__((aligned(16))) float red[2][64][68];
    const int S = EPI == 9 ? ep.ksplit : 1;
    const int per = nblk * mblk * S;
    int zi, sub;
    if (xcd_map) { const int q = (int)blockIdx.x >> 3; zi = (q / per) * 8 + ((int)blockIdx.x & 7); sub = q % per; }
    else { zi = (int)blockIdx.x / per; sub = (int)blockIdx.x - zi * per; }
    if (zi >= nd) return;
    const GDesc d = descs[zi];
    const int sp = sub % S, nb = (sub / S) % nblk, mb = sub / (S * nblk);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lj = lane >> 4;
    const int m0 = mb * ROWS, n0 = nb * 64;
    if (m0 >= M || n0 >= d.N) return;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int TC = (d.K + 15) >> 4, kc = (TC + S - 1) / S;       // 16-deep chunks: all of them, per range
    const int cbeg = sp * kc, cend = TC < cbeg + kc ? TC : cbeg + kc;
    const int kw = wave;                                         // this wave takes chunks cbeg + kw, + KW, ...
    const float* Abase = d.A;
    int64_t a_ld = d.lda;
    if (d.agather) { const SubnetDev sd = ep.sn[zi]; Abase = ep.xbase + sd.xoff; a_ld = sd.Dp; }
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = zero4;
    // EPI 4: the targets of this thread's four pieces of the block (piece q: row m0 + (tid + 256 q) / 16, columns n0 + ((tid + 256 q) % 16) * 4 ..)
    // are requested HERE, before the k-loop: they are 64 gathered rows of a [cells][Op] arena -- requested inside the rolled epilogue loop
    // they were four dependent round trips behind TLB misses (the fused launch ran 24.7 us against 16-19 for the plain forward)
    // the bias of this thread's four pieces is ONE 16-byte piece (their column is n0 + (tid % 16) * 4 in every piece): requested here, once
    f32x4 biasv = zero4;
    if constexpr (EPI == 1 || EPI == 2 || EPI == 4) { if (n0 + (tid & 15) * 4 < d.N) biasv = gen_gld4(d.bias + n0 + (tid & 15) * 4); }
    f32x4 yq[4];
    if constexpr (EPI == 3) {                                    // ... and so are the gates of the hidden backward (ldc = N, a multiple of 4: gen_rowgemm_ok)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = tid + 256 * q, row = m0 + (p >> 4), col = n0 + (p & 15) * 4;
            yq[q] = zero4;
            if (row < M && col < d.N) yq[q] = gen_gld4(d.G + (int64_t)row * d.ldc + col);
        }
    }
    if constexpr (EPI == 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = tid + 256 * q, row = m0 + (p >> 4), col = n0 + (p & 15) * 4;
            yq[q] = zero4;
            if (row < M && col < d.N) {
                const int64_t arow = ep.arows ? (int64_t)ep.arows[row] : ep.arow0 + row;
                yq[q] = gen_gld4(ep.Y + ((int64_t)zi * ep.n_cells + arow) * ep.Op + col);
            }
        }
    }
    auto mma = [&](const f32x4 (&a)[MT], const f32x4 (&b)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[m][j] = MFMA16(a[m][r], TB ? b[j][r] : b[r][j], acc[m][j]);
    };
    // Every request is unconditional: rows past M and columns past N are clamped onto valid ones (their results are never stored).
    const float* arc[MT];
    const float* brc[4];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int row = m0 + 16 * m + li;
        row = row < M ? row : M - 1;
        const int64_t r = d.agather ? (ep.arows ? (int64_t)ep.arows[row] : ep.arow0 + row) : (int64_t)row;
        arc[m] = Abase + r * a_ld + 4 * lj;
    }
    if constexpr (!TB) {
        int col = n0 + 4 * li;
        col = col + 4 <= d.N ? col : d.N - 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) brc[r] = d.B + (int64_t)(4 * lj + r) * d.ldb + col;
    } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int n = n0 + 16 * nt + li;
            n = n < d.N ? n : d.N - 1;
            brc[nt] = d.B + (int64_t)n * d.ldb + 4 * lj;
        }
    }
    const int64_t bstep = TB ? 16 : (int64_t)16 * d.ldb;
    auto loadf = [&](int c, f32x4 (&a)[MT], f32x4 (&b)[4]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = gen_gld4(arc[m] + 16 * c);
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = gen_gld4(brc[r] + bstep * c);
    };
    const int nfull = d.K >> 4;
    const int cf = cend < nfull ? cend : nfull;                  // chunks [cbeg, cf): whole; chunk nfull (if this range holds it): masked
    {
        int c = cbeg + kw;
        const int nw = c < cf ? (cf - c + KW - 1) / KW : 0;      // whole chunks of this wave: c, c + KW, ...
        if (nw > 0) {
            const int clast = c + KW * (nw - 1);
            auto cl = [&](int x) { return x < clast ? x : clast; };   // (past the wave's last chunk: the last one again, an L1 hit nobody uses)
            f32x4 a0[MT], b0[4], a1[MT], b1[4], a2[MT], b2[4];      // three chunks in flight per wave
            loadf(c, a0, b0);
            loadf(cl(c + KW), a1, b1);
            const int n3 = nw / 3;
            for (int i = 0; i < n3; ++i, c += 3 * KW) {
                loadf(cl(c + 2 * KW), a2, b2);
                mma(a0, b0);
                loadf(cl(c + 3 * KW), a0, b0);
                mma(a1, b1);
                loadf(cl(c + 4 * KW), a1, b1);
                mma(a2, b2);
            }
            const int rem = nw - 3 * n3;
            if (rem >= 1) mma(a0, b0);
            if (rem >= 2) mma(a1, b1);
        }
    }
    if ((d.K & 15) != 0 && nfull >= cbeg && nfull < cend && ((nfull - cbeg) % KW) == kw) {
        // the chunk that crosses K: requests at clamped k's (inside the row / the matrix), elements at k >= K replaced by zeros
        const int kl = d.K - 1;
        f32x4 a0[MT], b0[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 16 * nfull + 4 * lj + e, kq = (k < kl ? k : kl) - 4 * lj;     // (arc / brc already carry the lane's 4 lj)
#pragma unroll
            for (int m = 0; m < MT; ++m) { const float x = gen_gld1(arc[m] + kq); a0[m][e] = k < d.K ? x : 0.f; }
            if constexpr (!TB) {
                const f32x4 x = gen_gld4(brc[0] + (int64_t)kq * d.ldb);              // brc[0]: row 4 lj of the matrix
                b0[e] = k < d.K ? x : zero4;
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) { const float x = gen_gld1(brc[nt] + kq); b0[nt][e] = k < d.K ? x : 0.f; }
            }
        }
        mma(a0, b0);
    }
    // accumulator (m, j), element rr: row 16 m + 4 lj + rr;  TB = false: column 4 li + j;  TB = true: column 16 j + li
    double ls = 0.0;                                             // EPI 4: this thread's loss terms
    auto finish4 = [&](int row, int col, f32x4 v, f32x4 y) __attribute__((always_inline)) {   // TB = false: four consecutive columns of one row (y: EPI 4's targets)
        if (row >= M || col >= d.N) return;
        const int64_t o = (int64_t)row * d.ldc + col;
        if constexpr (EPI == 9) {
            gen_gst4(ep.part + ((int64_t)zi * S + sp) * ep.part_stride + (int64_t)row * d.N + col, v);
        } else if constexpr (EPI == 2) {
            gen_gst4(d.C + o, v + biasv);
        } else if constexpr (EPI == 4) {
            const f32x4 z = v + biasv;
            f32x4 dz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sp, sg, term, dy;
                softplus_sigmoid_fast(z[r], sp, sg);
                gen_loss_term(ep.loss, y[r], sp, y[r] - sp, term, dy);
                ls += (double)term;
                dz[r] = dy * ep.inv_n * sg;
            }
            gen_gst4(d.C + o, dz);
        } else {                                                 // EPI 1: as k_gen_splitk_fin's fast path
            const bool drop = ep.train && ep.rate > 0.f;
            dimn_u32x4 rnd = {{0u, 0u, 0u, 0u}};
            if (drop) rnd = dimn_dropout_block(ep.seed, (uint32_t)d.kg, ep.epoch, ep.step, (uint32_t)(row * d.N + col) >> 2);
            f32x4 c, g;
            gen_hidden4(ep, v + biasv, rnd.v[0], rnd.v[1], rnd.v[2], rnd.v[3], drop, c, g);
            gen_gst4(d.C + o, c);
            if (ep.train) gen_gst4(d.G + o, g);
        }
    };
    // the four waves' blocks: (w0 + w2) + (w1 + w3)
    auto put = [&](int slot) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if constexpr (!TB) *(f32x4*)&red[slot][16 * m + 4 * lj + rr][4 * li] = (f32x4){acc[m][0][rr], acc[m][1][rr], acc[m][2][rr], acc[m][3][rr]};
                else {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) red[slot][16 * m + 4 * lj + rr][16 * nt + li] = acc[m][nt][rr];
                }
            }
    };
    if (wave >= 2) put(wave - 2);
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if constexpr (!TB) {
                    const f32x4 t = *(const f32x4*)&red[wave][16 * m + 4 * lj + rr][4 * li];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][j][rr] += t[j];
                } else {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[m][nt][rr] += red[wave][16 * m + 4 * lj + rr][16 * nt + li];
                }
            }
    }
    __syncthreads();
    if (wave < 2) put(wave);
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {                        // (rolled: one copy of the epilogue's code)
        const int p = tid + 256 * q, rr = p >> 4, c4 = (p & 15) * 4;
        const f32x4 v = *(const f32x4*)&red[0][rr][c4] + *(const f32x4*)&red[1][rr][c4];
        const int row = m0 + rr, col = n0 + c4;
        if constexpr (!TB) finish4(row, col, v, EPI == 4 ? (q == 0 ? yq[0] : (q == 1 ? yq[1] : (q == 2 ? yq[2] : yq[3]))) : v);
        else if (row < M && col < d.N) gen_gst4(d.C + (int64_t)row * d.ldc + col, v * (q == 0 ? yq[0] : (q == 1 ? yq[1] : (q == 2 ? yq[2] : yq[3]))));
    }    if constexpr (EPI == 4) {
        __shared__ double lred[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off);
        if (lane == 0) lred[wave] = ls;
        __syncthreads();
        // one owner per slot and launch, so the sum is still deterministic; the atomic form returns nothing -- a load + add + store here kept the
        // workgroup alive for a memory round trip
        if (tid == 0) unsafeAtomicAdd(ep.loss_sum + (int64_t)zi * GEN_OUT_CH + nb + nblk * mb, (lred[0] + lred[1]) + (lred[2] + lred[3]));
    }
}

// Xb[k][b][Dp_k] = X_k[rows[b]][:]  (the batch rows of every sub-net, dense, so that every GEMM operand is a plain matrix)
template <typename XT>
__global__ __launch_bounds__(256) void k_gen_gather_batch(const SubnetDev* __restrict__ sn, const XT* __restrict__ X, const int32_t* __restrict__ rows,
                                                          int64_t row0, int b_cnt, float* __restrict__ Xb, int64_t xb_stride, int ldx,
                                                          float rate, float scale, uint64_t seed, uint32_t epoch, uint32_t step) {
    // rate > 0: a Dropout layer in front of the first Dense layer (training only): element (b, d) of sub-net kg keeps its value, scaled, iff the
    // Philox stream of dropout ordinal 0 says so (element index b * D + d, as every dropout layer counts its elements)
    const int k = blockIdx.y;
    const SubnetDev s = sn[k];
    for (int b = blockIdx.x; b < b_cnt; b += gridDim.x) {
        const int64_t row = rows ? rows[b] : row0 + b;
        const XT* src = X + s.xoff + row * s.Dp;
        float* dst = Xb + (int64_t)k * xb_stride + (int64_t)b * ldx;
        for (int d = threadIdx.x; d < s.Dp; d += 256) {
            float v = sizeof(XT) == 2 ? bf16_to_f32((bf16_t)src[d]) : (float)src[d];
            if (rate > 0.f) v = (d < s.D && dimn_dropout_keep(seed, (uint32_t)s.kg, epoch, step, (uint32_t)(b * s.D + d), rate)) ? v * scale : 0.f;
            dst[d] = v;
        }
    }
}

// Output layer: yhat = softplus(Z); the loss of build()'s `loss` (multinet.py:150-162) and dZ = dL/dZ in place.
// grid (K, GEN_OUT_CH), block 256: workgroup (k, c) takes every GEN_OUT_CH-th stripe of 256 elements of sub-net k (round 2
// launched ONE workgroup per sub-net: 40 of 256 CUs busy, 28 % of a general-path step; round 3 sixteen, each walking eight dependent
// load -> softplus -> store rounds: 20 us; now 64).  out != NULL (predict):
// out[(row0 + b)][k*O + o] = yhat, no loss.  loss_sum[k][c] += sum of the per-element loss terms of the workgroup's stripes (one
// owner per slot: deterministic; the host adds the slots and divides by the element count); train != 0 writes dZ over Z.
__global__ __launch_bounds__(256) void k_gen_output(float* __restrict__ Z, int ldz, int64_t z_stride, const float* __restrict__ Y, int64_t n_cells,
                                                    const int32_t* __restrict__ rows, int64_t row0, int b_cnt, Dims dm, int loss, int train,
                                                    float inv_n, double* __restrict__ loss_sum, float* __restrict__ out, int64_t out_row0, int k_off) {
    __shared__ double red[4];
    const int k = blockIdx.x;
    float* z = Z + (int64_t)k * z_stride;
    double ls = 0.0;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < b_cnt * dm.O; e += 256 * GEN_OUT_CH) {
        const int b = e / dm.O, o = e - b * dm.O;
        const float zz = z[(int64_t)b * ldz + o];
        if (out) {
            out[((out_row0 + b) * dm.K + k) * dm.O + o] = softplus_f(zz);
            continue;
        }
        const int64_t row = rows ? rows[b] : row0 + b;
        const float y = Y[((int64_t)(k + k_off) * n_cells + row) * dm.Op + o];
        float sp, sg;
        if (train) softplus_sigmoid_fast(zz, sp, sg); else { sp = softplus_f(zz); sg = 0.f; }
        const float er = y - sp;
        float term, dy;                                          // loss term and dL/dyhat * N
        gen_loss_term(loss, y, sp, er, term, dy);
        ls += (double)term;
        if (train) z[(int64_t)b * ldz + o] = dy * inv_n * sg;
    }
    if (out) return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ls;
    __syncthreads();
    if (threadIdx.x == 0) loss_sum[(int64_t)k * GEN_OUT_CH + blockIdx.y] += red[0] + red[1] + red[2] + red[3];
}

// Glorot-uniform kernels (Keras Dense default), Philox stream keyed (seed, global sub-net, layer, element) as k_init_weights
__global__ __launch_bounds__(256) void k_gen_init(float* __restrict__ W, int64_t n, uint64_t seed, uint32_t kg, uint32_t layer, float limit) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
        W[e] = dimn_init_value(seed, kg, layer, (uint32_t)e, limit);
}
