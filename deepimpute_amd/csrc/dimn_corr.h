// dimn_corr.h -- |Pearson correlation| of the columns of a [n][g] matrix on gfx950, in fp64.
//
// Replaces the host computation of get_distance_matrix (reference deepimpute/multinet.py:20-34:
// np.abs(np.corrcoef(raw.T.loc[potential_pred])) followed by fillna(0)) -- SURVEY.md section 8(f)
// rank 1: 2*n*g^2 = 4e13 flop in float64 at 50k x 20k, ~10 minutes of numpy, the largest cost of a
// drop-in fit once the training itself runs on the GPU.
//
// The reference computes in float64 and the predictor selection (top-5 |corr| per target) consumes the
// result, so the kernel uses the fp64 matrix instruction v_mfma_f64_16x16x4_f64 (A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15], C/D: col = l&15, row = (l>>4) + 4*reg -- NOT the fp32 row map).  numpy's order
// of operations is kept: centre the columns, C = Z^T Z, c = C * (1/(n-1)), c /= s_i, c /= s_j with
// s = sqrt(diag(c)), clip to [-1,1], abs; NaN (constant columns) -> 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

#define CORR_BT 128     // output tile (columns x columns) per workgroup
#define CORR_KC 16      // rows staged per LDS chunk
#define CORR_LD 144     // LDS row stride in doubles: rows k and k+1 land on disjoint bank halves

// column means of X[n][gp] (gp = padded column count); one thread per column, coalesced rows
__global__ __launch_bounds__(256) void k_corr_colsum(const double* __restrict__ X, int64_t n, int64_t gp, int64_t rows_per_block,
                                                     double* __restrict__ partial) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    if (j >= gp) return;
    double s = 0;
    for (int64_t i = r0; i < r1; ++i) s += X[i * gp + j];
    partial[(int64_t)blockIdx.y * gp + j] = s;
}
__global__ __launch_bounds__(256) void k_corr_mean(const double* __restrict__ partial, int nparts, int64_t n, int64_t gp, double* __restrict__ mean) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= gp) return;
    double s = 0;
    for (int p = 0; p < nparts; ++p) s += partial[(int64_t)p * gp + j];
    mean[j] = s / (double)n;
}
// partial += column sums of one row block (streamed matrix): one thread per column, sequential over the block's rows
__global__ __launch_bounds__(256) void k_corr_colsum_acc(const double* __restrict__ X, int64_t n, int64_t gp, double* __restrict__ acc) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= gp) return;
    double s = 0;
    for (int64_t i = 0; i < n; ++i) s += X[i * gp + j];
    acc[j] += s;
}
__global__ __launch_bounds__(256) void k_corr_scale(double* __restrict__ v, int64_t gp, double f) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < gp) v[j] *= f;
}
// X -= mean (rows < n, columns < g); padding rows/columns are zero and stay zero
__global__ __launch_bounds__(256) void k_corr_center(double* __restrict__ X, int64_t n, int64_t g, int64_t gp, const double* __restrict__ mean) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= g) return;
    const double m = mean[j];
    for (int64_t i = blockIdx.y; i < n; i += gridDim.y) X[i * gp + j] -= m;
}

// C[I-block][J-block] = Z[:, I]^T Z[:, J] for the upper-triangular block pairs (pairs[] lists them);
// 4 waves as 2 x 2, each wave a 64 x 64 sub-tile = 4 x 4 MFMA tiles.  Z is [np][gp], np % 16 == 0.
// accumulate != 0: C += (the matrix is streamed in row blocks, corr_on_device_streamed).
__global__ __launch_bounds__(256) void k_corr_gemm(const double* __restrict__ Z, int64_t np_, int64_t gp, const int2* __restrict__ pairs,
                                                   double* __restrict__ C, int accumulate) {
    __shared__ __attribute__((aligned(16))) double sm[2][2][CORR_KC * CORR_LD];   // [buffer][A|B][k][col]
    const int2 pr = pairs[blockIdx.x];
    const int64_t I0 = (int64_t)pr.x * CORR_BT, J0 = (int64_t)pr.y * CORR_BT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;

    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f64x4){0, 0, 0, 0};

    // staging: 16 rows x 128 doubles per operand = 2048 doubles; 256 threads x 2 x double4 each per operand
    const int srow = tid >> 4, scol = (tid & 15) * 8;         // row 0..15, 8 consecutive doubles
    const double* za = Z + (int64_t)srow * gp + I0 + scol;
    const double* zb = Z + (int64_t)srow * gp + J0 + scol;
    f64x4 ra0, ra1, rb0, rb1;
    auto gload = [&](int64_t k0) {
        const double* pa = za + k0 * gp;
        const double* pb = zb + k0 * gp;
        ra0 = *(const f64x4*)pa; ra1 = *(const f64x4*)(pa + 4);
        rb0 = *(const f64x4*)pb; rb1 = *(const f64x4*)(pb + 4);
    };
    auto lstore = [&](int buf) {
        double* da = &sm[buf][0][srow * CORR_LD + scol];
        double* db = &sm[buf][1][srow * CORR_LD + scol];
        *(f64x4*)da = ra0; *(f64x4*)(da + 4) = ra1;
        *(f64x4*)db = rb0; *(f64x4*)(db + 4) = rb1;
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int64_t nchunk = np_ / CORR_KC;
    for (int64_t c = 0; c < nchunk; ++c) {
        const int buf = (int)(c & 1);
        if (c + 1 < nchunk) gload((c + 1) * CORR_KC);
        const double* A = &sm[buf][0][0];
        const double* B = &sm[buf][1][0];
#pragma unroll
        for (int ks = 0; ks < CORR_KC / 4; ++ks) {
            double af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = A[(4 * ks + lj) * CORR_LD + wi + 16 * t + li];    // Z^T[i][k]
                bf[t] = B[(4 * ks + lj) * CORR_LD + wj + 16 * t + li];    // Z[k][j]
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = MFMA64(af[a], bf[b], acc[a][b]);
        }
        if (c + 1 < nchunk) lstore(buf ^ 1);
        __syncthreads();
    }
    // C/D of the fp64 MFMA: col = li, row = lj + 4*reg
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = I0 + wi + 16 * a + lj + 4 * r, j = J0 + wj + 16 * b + li;
                const double v = accumulate ? C[i * gp + j] + acc[a][b][r] : acc[a][b][r];
                C[i * gp + j] = v;
                if (pr.x != pr.y) C[j * gp + i] = v;
            }
}

// numpy's normalisation order (np.corrcoef): c = C*(1/(n-1)); c /= s_i; c /= s_j; clip; abs; NaN -> 0
__global__ __launch_bounds__(256) void k_corr_finish(const double* __restrict__ C, int64_t g, int64_t gp, double inv_fact,
                                                     double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= g) return;
    const double si = sqrt(C[i * gp + i] * inv_fact), sj = sqrt(C[j * gp + j] * inv_fact);
    double c = C[i * gp + j] * inv_fact;
    c /= si;
    c /= sj;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);      // np.clip(c.real, -1, 1); NaN stays NaN
    c = fabs(c);
    out[i * g + j] = (c == c) ? c : 0.0;            // DataFrame.fillna(0)
}


// ---------------------------------------------------------------------------------------------------------------
// setPredictors on the device (reference deepimpute/multinet.py:344-365): for every target gene of every sub-net the
// NTOP most correlated pool genes outside the sub-net's own target set.  One workgroup per (target, sub-net) scans
// the target's row of the resident |corr| matrix once (coalesced), each thread keeping a sorted private top-NTOP;
// the 256 lists are merged through LDS in NTOP rounds of a block-wide arg-max.
// Order: |corr| descending; ties by the column's position in label-sorted order (col_rank) -- the order in which the
// reference's np.setdiff1d lists the candidate columns (its argsort is not stable, so on exact ties the reference
// itself is arbitrary; the host implementation in multinet.py uses the same rule).  col_rank < 0: column not a
// candidate (a repeated label).  out[k][t][i] = pool position of the i-th pick, -1 when fewer candidates exist.
// ---------------------------------------------------------------------------------------------------------------
template <int NTOP>
__global__ __launch_bounds__(256) void k_corr_topk(const double* __restrict__ corr, int64_t g, const int32_t* __restrict__ targ_pos, int O,
                                                   const int32_t* __restrict__ col_rank, int32_t* __restrict__ out, int ntop) {
    extern __shared__ __attribute__((aligned(16))) unsigned char topk_lds[];
    const int words = (int)((g + 31) >> 5);
    unsigned* bitmap = (unsigned*)topk_lds;                                      // excluded columns (the sub-net's targets)
    double* lval = (double*)(topk_lds + (((size_t)words * 4 + 15) & ~(size_t)15));   // [256][NTOP]
    int32_t* lrank = (int32_t*)(lval + 256 * NTOP);
    int32_t* lidx = lrank + 256 * NTOP;
    double* wval = (double*)(lidx + 256 * NTOP);                                 // per-wave winners [4]
    int32_t* wrank = (int32_t*)(wval + 4);
    int32_t* wthr = wrank + 4;
    const int t = blockIdx.x, k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int w = tid; w < words; w += 256) bitmap[w] = 0u;
    __syncthreads();
    for (int i = tid; i < O; i += 256) {
        const int c = targ_pos[(int64_t)k * O + i];
        atomicOr(&bitmap[c >> 5], 1u << (c & 31));
    }
    __syncthreads();
    const int64_t row = targ_pos[(int64_t)k * O + t];
    const double* cr = corr + row * g;
    double val[NTOP];
    int32_t rk[NTOP], ix[NTOP];
#pragma unroll
    for (int i = 0; i < NTOP; ++i) { val[i] = -1.0; rk[i] = 0x7fffffff; ix[i] = -1; }
    for (int64_t j = tid; j < g; j += 256) {
        const int32_t r = col_rank[j];
        const double v = cr[j];
        const bool ok = r >= 0 && !((bitmap[j >> 5] >> (j & 31)) & 1u);
        if (ok && (v > val[NTOP - 1] || (v == val[NTOP - 1] && r < rk[NTOP - 1]))) {
            // insertion into the sorted private list (fully unrolled: the list stays in registers)
            double cv = v; int32_t cr_ = r, ci = (int32_t)j;
#pragma unroll
            for (int i = 0; i < NTOP; ++i) {
                const bool before = cv > val[i] || (cv == val[i] && cr_ < rk[i]);
                const double tv = val[i]; const int32_t tr = rk[i], ti = ix[i];
                val[i] = before ? cv : tv; rk[i] = before ? cr_ : tr; ix[i] = before ? ci : ti;
                cv = before ? tv : cv; cr_ = before ? tr : cr_; ci = before ? ti : ci;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NTOP; ++i) { lval[tid * NTOP + i] = val[i]; lrank[tid * NTOP + i] = rk[i]; lidx[tid * NTOP + i] = ix[i]; }
    __syncthreads();
    int head = 0;
    for (int round = 0; round < ntop; ++round) {
        double bv = head < NTOP ? lval[tid * NTOP + head] : -1.0;
        int32_t br = head < NTOP ? lrank[tid * NTOP + head] : 0x7fffffff;
        int32_t bt = tid;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(bv, off); const int32_t orr = __shfl_xor(br, off), ot = __shfl_xor(bt, off);
            const bool take = ov > bv || (ov == bv && (orr < br || (orr == br && ot < bt)));
            bv = take ? ov : bv; br = take ? orr : br; bt = take ? ot : bt;
        }
        if (lane == 0) { wval[wave] = bv; wrank[wave] = br; wthr[wave] = bt; }
        __syncthreads();
        double fv = wval[0]; int32_t fr = wrank[0], ft = wthr[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const bool take = wval[w] > fv || (wval[w] == fv && (wrank[w] < fr || (wrank[w] == fr && wthr[w] < ft)));
            fv = take ? wval[w] : fv; fr = take ? wrank[w] : fr; ft = take ? wthr[w] : ft;
        }
        if (tid == ft) {
            out[((int64_t)k * O + t) * ntop + round] = (fv >= 0.0 && head < NTOP) ? lidx[tid * NTOP + head] : -1;
            ++head;
        }
        __syncthreads();
    }
}
