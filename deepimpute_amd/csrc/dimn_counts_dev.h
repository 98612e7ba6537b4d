// dimn_counts_dev.h -- what the planning of a drop-in fit() computes from the count matrix once it is RESIDENT on the device
// (dimn_counts, float32 [n][g], every value a non-negative integer, exact): the per-gene statistics and the gene-gene correlation.
//
// (1) Gene statistics (reference deepimpute/multinet.py:191: `raw.var() / (1 + raw.mean())` ranks the genes, so the numbers must
//     be pandas' TO THE BIT; dimn_hoststats.h states the two summation orders and is the host form of the same arithmetic).  The
//     float32 counts convert to float64 exactly, so one thread per column replays the same additions in the same order:
//     k_cnt_seqsum (DataFrame.mean(): a running sum down the column), k_cnt_pairwise (nanvar: numpy's pairwise sum per 8192-row
//     chunk) + k_cnt_chunks (the chunks added in order).  fp contraction is OFF for this file: (avg - x)^2 must round twice.
//
// (2) |Pearson correlation| of the candidate genes on the INTEGER matrix cores, exactly (reference multinet.py:20-34,
//     np.abs(np.corrcoef(raw.T.loc[pool])); dimn_corr.h is the general float64 form of the same row).  Counts are integers, so
//         corr_ij = (n S_ij - s_i s_j) / sqrt((n S_ii - s_i^2)(n S_jj - s_j^2)),   S_ij = sum_c x_ci x_cj,  s_i = sum_c x_ci
//     has an integer numerator and integer radicands.  A correlation does not change when a constant is subtracted from a column,
//     so with x - (128 + 256*128) = 256 a1 + a0, a0 / a1 the two bytes of the count each shifted into int8 range,
//         S''_ij = 65536 (a1_i . a1_j) + 256 (a1_i . a0_j + a0_i . a1_j) + (a0_i . a0_j)
//     is four int8 dot products per gene pair: v_mfma_i32_16x16x64_i8, int32 accumulation (exact: |a| <= 128, slabs of 32768
//     cells), int64 between slabs, __int128 for n S - s s, ONE rounding into float64 per numerator / radicand.  Counts below
//     256 need one plane (one product), counts up to 65535 two (four products); anything larger takes the float64 kernel.
//     The result is closer to the true correlation than numpy's float64 evaluation (whose error is ~1e-15); exactly tied pairs
//     come out exactly equal.  2 n g^2 integer operations per product: 50k cells x 20k genes, two planes: 8e13.
//
//     Layout: the int8 planes are stored the way the matrix instruction wants its operands -- per 16-gene tile T and 64-cell
//     chunk kc one 1 KB block [q = cell/16 % 4][r = gene % 16][16 consecutive cells]: 16-byte piece number q*16 + r belongs to
//     lane q*16 + r, so a fragment is ONE lane-linear 1 KB read, from global memory (global_load_lds: no staging registers,
//     the LDS image is lane-linear by construction) and from LDS (ds_read_b128 at lane*16: conflict-free).  Which 16 of the 64
//     k's a lane holds is immaterial as long as A and B agree (a dot product).
//     k_ci8_gemm: one workgroup (4 waves as 2 x 2) per upper-triangular 128 x 128 block pair; per 64-cell step the workgroup
//     brings 16 x P fragment blocks into a ring of LDS stages while every wave runs 16 P^2 matrix instructions from the previous one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------------------------------------
// (1) gene statistics
// ---------------------------------------------------------------------------------------------------------------------------
// sum[j] = ((x[0][j] + x[1][j]) + x[2][j]) + ... in float64 (numpy reduces the transposed block row by row), cmin / cmax
__global__ __launch_bounds__(64) void k_cnt_seqsum(const float* __restrict__ x, int64_t n, int64_t g, double* __restrict__ sum, double* __restrict__ cmin,
                                                   double* __restrict__ cmax) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= g) return;
    const float* p = x + j;
    double s = 0.0;
    float lo = INFINITY, hi = -INFINITY;
    int64_t i = 0;
    for (; i + 16 <= n; i += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = p[(i + u) * g];
#pragma unroll
        for (int u = 0; u < 16; ++u) { s += (double)v[u]; lo = fminf(lo, v[u]); hi = fmaxf(hi, v[u]); }
    }
    for (; i < n; ++i) { const float v = p[i * g]; s += (double)v; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    sum[j] = s; cmin[j] = (double)lo; cmax[j] = (double)hi;
}

// numpy's pairwise sum (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum) over rows [i0, i0 + n) of column p (stride g):
// n < 8 a plain loop; n <= 128 eight partial sums over i = k mod 8, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail;
// larger n split at n/2 rounded down to a multiple of 8, left half first.  Recursion unrolled onto a small explicit stack
// (the shape depends on n only, so the threads of a wave never diverge).
template <bool SQDEV>
__device__ __forceinline__ double cnt_term(const float* __restrict__ p, int64_t g, int64_t i, double avg) {
    const double xv = (double)p[i * g];
    if (!SQDEV) return xv;
    const double d = avg - xv;
    return d * d;
}
template <bool SQDEV>
__device__ double cnt_leaf(const float* __restrict__ p, int64_t g, int64_t i0, int64_t n, double avg) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += cnt_term<SQDEV>(p, g, i0 + i, avg);
        return res;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = cnt_term<SQDEV>(p, g, i0 + k, avg);
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = cnt_term<SQDEV>(p, g, i0 + i + k, avg);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += t[k];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += cnt_term<SQDEV>(p, g, i0 + i, avg);
    return res;
}
template <bool SQDEV>
__device__ double cnt_pairwise(const float* __restrict__ p, int64_t g, int64_t i0, int64_t n, double avg) {
    int64_t off[24], len[24];
    double left[24];
    int state[24];                     // 0: nothing done, 1: waiting for the left half, 2: waiting for the right half
    int sp = 0;
    off[0] = i0; len[0] = n; state[0] = 0;
    double ret = 0.0;
    bool have = false;                 // a finished value travelling up
    for (;;) {
        if (!have) {
            if (len[sp] <= 128) { ret = cnt_leaf<SQDEV>(p, g, off[sp], len[sp], avg); have = true; --sp; }
            else {
                int64_t n2 = len[sp] / 2;
                n2 -= n2 % 8;
                state[sp] = 1;
                off[sp + 1] = off[sp]; len[sp + 1] = n2; state[sp + 1] = 0;
                ++sp;
            }
        } else {
            if (sp < 0) return ret;
            if (state[sp] == 1) {
                int64_t n2 = len[sp] / 2;
                n2 -= n2 % 8;
                left[sp] = ret; state[sp] = 2; have = false;
                off[sp + 1] = off[sp] + n2; len[sp + 1] = len[sp] - n2; state[sp + 1] = 0;
                ++sp;
            } else { ret = left[sp] + ret; --sp; }
        }
    }
}
// part[c][j] = pairwise sum of chunk c (8192 rows, numpy's reduction buffer) of column j; grid (ceil(g/64), chunks)
template <bool SQDEV>
__global__ __launch_bounds__(64) void k_cnt_pairwise(const float* __restrict__ x, int64_t n, int64_t g, const double* __restrict__ avg, double* __restrict__ part) {
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= g) return;
    const int64_t c0 = (int64_t)blockIdx.y * 8192;
    part[(int64_t)blockIdx.y * g + j] = cnt_pairwise<SQDEV>(x + j, g, c0, (n - c0 < 8192 ? n - c0 : 8192), SQDEV ? avg[j] : 0.0);
}
// out[j] = (0 + part[0][j] + part[1][j] + ...) / div   (numpy: out = 0; out += pairwise(chunk) per chunk)
__global__ __launch_bounds__(256) void k_cnt_chunks(const double* __restrict__ part, int chunks, int64_t g, double div, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= g) return;
    double q = 0.0;
    for (int c = 0; c < chunks; ++c) q += part[(int64_t)c * g + j];
    out[j] = q / div;
}
__global__ __launch_bounds__(256) void k_cnt_div(double* __restrict__ v, int64_t g, double div) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < g) v[j] = v[j] / div;
}

// ---------------------------------------------------------------------------------------------------------------------------
// (2) the correlation on the int8 matrix cores
// ---------------------------------------------------------------------------------------------------------------------------
typedef int ci8_v4i __attribute__((ext_vector_type(4)));
#define CI8_BT 128          // genes per block side
#define CI8_KS 64           // cells per step (one matrix instruction deep)
#define CI8_SLAB 512        // steps per int32 slab: 32768 cells, |sum| <= 2 * 128 * 128 * 32768 = 2^30
#define CI8_NBUF 3          // LDS stages (the counted wait in k_ci8_gemm is written for three)

// planes[p] block (T, kc): 1 KB at ((T * KC + kc) * 1024); 64 genes x 64 cells per workgroup through LDS.  Entries beyond the
// matrix (padding genes / cells) are 0 in every plane: they add nothing to any dot product.
template <int P>
__global__ __launch_bounds__(256) void k_ci8_planes(const float* __restrict__ counts, int64_t ld, const int32_t* __restrict__ cols, int64_t n, int64_t pool_n,
                                                    int64_t KC, int8_t* __restrict__ planes, int64_t plane_bytes) {
    __shared__ int tile[64][65];
    const int t = threadIdx.x;
    const int64_t g0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    {
        const int gl = t & 63;
        const int64_t gene = g0 + gl;
        const int32_t col = gene < pool_n ? cols[gene] : -1;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int cl = (t >> 6) + 4 * it;
            const int64_t cell = c0 + cl;
            tile[cl][gl] = (col >= 0 && cell < n) ? (int)counts[cell * ld + col] : -1;
        }
    }
    __syncthreads();
    const int tt = t >> 6, lane = t & 63, q = lane >> 4, r = lane & 15;
    const int64_t T = (int64_t)blockIdx.x * 4 + tt;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        ci8_v4i piece;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int v = tile[q * 16 + w * 4 + b][tt * 16 + r];
                const int byte = v < 0 ? 0 : (((v >> (8 * p)) & 255) - 128);
                word |= (unsigned)(byte & 255) << (8 * b);
            }
            piece[w] = (int)word;
        }
        *(ci8_v4i*)(planes + (int64_t)p * plane_bytes + (T * KC + blockIdx.y) * 1024 + lane * 16) = piece;
    }
}

// sums[j] += sum over this block's cells of (x - shift): exact integers, so the order of the atomic additions is immaterial
__global__ __launch_bounds__(256) void k_ci8_colsum(const float* __restrict__ counts, int64_t ld, const int32_t* __restrict__ cols, int64_t n, int64_t pool_n,
                                                    int64_t rows_per_block, long long shift, long long* __restrict__ sums) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= pool_n) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const float* p = counts + cols[j];
    long long s = 0;
    for (int64_t i = r0; i < r1; ++i) s += (long long)p[i * ld] - shift;
    atomicAdd((unsigned long long*)&sums[j], (unsigned long long)s);
}

// C[i][j] (int64, leading dimension ldc) = S''_ij for the block pair pairs[blockIdx.x] = (I, J), I <= J.
template <int P>
__global__ __launch_bounds__(256, 1) void k_ci8_gemm(const int8_t* __restrict__ planes, int64_t plane_bytes, int64_t KC, const int2* __restrict__ pairs,
                                                     long long* __restrict__ C, int64_t ldc, int64_t pool_n) {
#if defined(__HIP_DEVICE_COMPILE__)      // (device-only constructs below -- the LDS address space, register constraints -- make the host pass drop the kernel's stub silently)
    constexpr int NB = 16 * P;                       // fragment blocks per stage: [side][plane][8 tiles]
    constexpr int PER_WAVE = NB / 4;
    extern __shared__ __attribute__((aligned(1024))) int8_t ci8_lds[];
    const int2 pr = pairs[blockIdx.x];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, wi = w >> 1, wj = w & 1;
    // the blocks this wave brings in per stage: b = w * PER_WAVE + u -> (side, plane, tile)
    const int8_t* src[PER_WAVE];
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
        const int b = w * PER_WAVE + u, side = b / (8 * P), p = (b / 8) % P, t8 = b & 7;
        const int64_t T = (int64_t)(side ? pr.y : pr.x) * 8 + t8;
        src[u] = planes + (int64_t)p * plane_bytes + T * KC * 1024 + lane * 16;
    }
    auto issue = [&](int64_t kc, int stage) {
#pragma unroll
        for (int u = 0; u < PER_WAVE; ++u)
            __builtin_amdgcn_global_load_lds(src[u] + kc * 1024, (__attribute__((address_space(3))) void*)(ci8_lds + (stage * NB + w * PER_WAVE + u) * 1024), 16, 0, 0);
    };
    ci8_v4i acc_ll[4][4], acc_mid[P == 2 ? 4 : 1][P == 2 ? 4 : 1], acc_hh[P == 2 ? 4 : 1][P == 2 ? 4 : 1];
    auto zero = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                acc_ll[a][b] = ci8_v4i{0, 0, 0, 0};
                if constexpr (P == 2) { acc_mid[a][b] = ci8_v4i{0, 0, 0, 0}; acc_hh[a][b] = ci8_v4i{0, 0, 0, 0}; }
            }
    };
    // D layout of the 16 x 16 instruction: element r of a lane is row (lane / 16) * 4 + r (A's gene), column lane % 16 (B's gene).
    // The owner of a block pair adds slab after slab into its own int64 elements (the first one stores).
    auto flush = [&](bool first) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));                 // (keeps the 64 element addresses out of the registers of the step loop)
        const int ln = tid & 63, fi = tid >> 7, fj = (tid >> 6) & 1;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {          // (and the widening of the sums out of the step loop: hipcc otherwise keeps int64 copies alive in it)
                asm volatile("" : "+v"(acc_ll[ti][tj]));
                if constexpr (P == 2) { asm volatile("" : "+v"(acc_mid[ti][tj])); asm volatile("" : "+v"(acc_hh[ti][tj])); }
            }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t i = (int64_t)pr.x * CI8_BT + fi * 64 + ti * 16 + (ln >> 4) * 4 + r;
                    const int64_t j = (int64_t)pr.y * CI8_BT + fj * 64 + tj * 16 + (ln & 15);
                    long long v = (long long)acc_ll[ti][tj][r];
                    if constexpr (P == 2) v += 256ll * (long long)acc_mid[ti][tj][r] + 65536ll * (long long)acc_hh[ti][tj][r];
                    if (i < pool_n && j < pool_n) C[i * ldc + j] = first ? v : C[i * ldc + j] + v;
                }
    };
    zero();
    // prologue: stages 0 .. NBUF-2 in flight
#pragma unroll
    for (int s = 0; s < CI8_NBUF - 1; ++s)
        if (s < KC) issue(s, s);
    for (int64_t k0 = 0; k0 < KC; k0 += CI8_SLAB) {
    const int64_t k1 = k0 + CI8_SLAB < KC ? k0 + CI8_SLAB : KC;
    for (int64_t kc = k0; kc < k1; ++kc) {
        // stage kc has landed for every wave once each wave has seen its own loads of it arrive and the workgroup has met:
        // loads complete in order, so "only the younger stage outstanding" is a counted wait
        if (kc + 1 < KC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // (every wave has also finished reading stage kc-1: its buffer is the one stage kc+2 goes into)
        const int stage = (int)(kc % CI8_NBUF);
        const int8_t* base = ci8_lds + stage * NB * 1024 + lane * 16;
        ci8_v4i a[P][4], b[P][4];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[p][t] = *(const ci8_v4i*)(base + ((0 * P + p) * 8 + wi * 4 + t) * 1024);
                b[p][t] = *(const ci8_v4i*)(base + ((1 * P + p) * 8 + wj * 4 + t) * 1024);
            }
        if (kc + CI8_NBUF - 1 < KC) issue(kc + CI8_NBUF - 1, (int)((kc + CI8_NBUF - 1) % CI8_NBUF));
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                acc_ll[ti][tj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0][ti], b[0][tj], acc_ll[ti][tj], 0, 0, 0);
                if constexpr (P == 2) {
                    acc_mid[ti][tj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1][ti], b[0][tj], acc_mid[ti][tj], 0, 0, 0);
                    acc_mid[ti][tj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[0][ti], b[1][tj], acc_mid[ti][tj], 0, 0, 0);
                    acc_hh[ti][tj] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[1][ti], b[1][tj], acc_hh[ti][tj], 0, 0, 0);
                }
            }
    }
    flush(k0 == 0);
    zero();
    }
#endif
}

__device__ __forceinline__ double ci8_to_double(__int128 v) {
    const bool neg = v < 0;
    const unsigned __int128 u = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    const double d = (double)(unsigned long long)(u >> 64) * 18446744073709551616.0 + (double)(unsigned long long)u;
    return neg ? -d : d;
}
// root[i] = n S_ii - s_i^2 rounded to float64 (0 for a constant gene).  The quotient is taken as num / sqrt(root_i * root_j):
// for i = j, and for two identical genes, sqrt(a * a) == a exactly, so those correlations are exactly 1
__global__ __launch_bounds__(256) void k_ci8_diag(const long long* __restrict__ C, int64_t ldc, const long long* __restrict__ sums, int64_t n, int64_t pool_n,
                                                  double* __restrict__ root) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pool_n) return;
    const __int128 v = (__int128)n * (__int128)C[i * ldc + i] - (__int128)sums[i] * (__int128)sums[i];
    root[i] = v > 0 ? ci8_to_double(v) : 0.0;
}
// in place over the upper triangle (the int64 S'' becomes the float64 |corr|), mirrored into the lower one; a constant gene's
// row / column is 0 (numpy: NaN, then fillna(0)), the diagonal of every other gene 1
__global__ __launch_bounds__(256) void k_ci8_finish(long long* __restrict__ C, int64_t ldc, const long long* __restrict__ sums, const double* __restrict__ root, int64_t n,
                                                    int64_t pool_n) {
    const int64_t i = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= pool_n || j < i) return;
    double out;
    const double ri = root[i], rj = root[j];
    if (ri == 0.0 || rj == 0.0) out = 0.0;
    else {
        const __int128 num = (__int128)n * (__int128)C[i * ldc + j] - (__int128)sums[i] * (__int128)sums[j];
        double c = ci8_to_double(num) / sqrt(ri * rj);
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        out = fabs(c);
    }
    double* D = (double*)C;
    D[i * ldc + j] = out;
    if (j != i) D[j * ldc + i] = out;
}
