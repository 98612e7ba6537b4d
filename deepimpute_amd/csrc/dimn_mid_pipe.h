// dimn_mid_pipe.h -- MFP: the second layer of one optimiser step as a TILE PIPELINE (round 4; fp32, H = 256).
//
// Same work, same work table (MidWork: sub-net k, output tiles [ot0, ot1)), same inputs and outputs as k_mid_fused
// (dimn_kernels.h): Z = Dd W2 + b2 ; softplus ; wMSE (multinet.py:36-41) ; dZ ; Adam on W2 / b2 (multinet.py:164) ;
// the slice's dD partial -> P2.  What differs is the schedule.  k_mid_fused runs three workgroup-wide phases (load
// burst -> forward of all tiles, one wave per tile -> backward of all tiles, one wave per pair of hidden tiles): its
// reads all happen in the first half, its writes in the second, and the matrix pipe idles under both -- 152 MB that
// need 27 us at the rate B1F1 streams and 15 us of fp32 MFMA add up to 40 us.  Here every wave owns hidden rows
// [32w, 32w + 32) of W2 for the WHOLE kernel and the slice's output tiles go through it one after the other:
//
//   block i:   request w / m / v of tile i+2 (16-byte loads, each wave its own rows: W2 is read ONCE, in one layout)
//              forward(i+1): partial Z over the wave's 32 hidden rows (32 MFMAs) -> LDS                      } barrier
//              softplus / wMSE / dZ(i+1): all 512 threads, two elements each, summing the 8 partials
//              backward(i): gW2^T = dZ^T Dd (32 MFMAs), dD += dZ W2old^T (32), Adam in registers, 16-byte stores
//
// so tile i's stores, tile i+2's loads and the MFMAs of tiles i, i+1 are in flight together, from the first
// microsecond to the last; one barrier per tile; no phase in which the memory system waits for the matrix pipe.
// Summation orders differ from k_mid_fused (Z over 8 partial sums of 32, b2's gradient over 8 x 8 rows); both are
// within the parity tolerance of the oracle (oracle/dimo.c sums in its own order).
#pragma once

#define DIMN_MIDP_LDD 260                                 // as DIMN_MID_LDD
#define DIMN_MIDP_LDS_FLOATS (64 * DIMN_MIDP_LDD + 2 * 8192 + 2 * 1024 + 8 * 512 + 8 * 128 + 8)

#ifdef DIMN_MIDP_TL   // tools/k_probe_mid.hip: per-wave stamps
__device__ unsigned long long g_midp_tl[512 * 8 * 12];
#define MIDP_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_midp_tl[(blockIdx.x * 8 + wave) * 12 + (i)] = t_; }
#else
#define MIDP_STAMP(i)
#endif

__global__ __launch_bounds__(512) void k_mid_pipe(const MidWork* __restrict__ mwork,
                                                  float* __restrict__ W2, float* __restrict__ M2, float* __restrict__ V2,
                                                  float* __restrict__ b2w, float* __restrict__ b2m, float* __restrict__ b2v,
                                                  const float* __restrict__ Y, int64_t n_cells,
                                                  const int32_t* __restrict__ rows, int b_act,
                                                  const float* __restrict__ Dd, float* __restrict__ P2,
                                                  float* __restrict__ loss_step, double* __restrict__ loss_acc,
                                                  Dims dm, AdamP ap, float inv_n, int loss_binary) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MidWork mw = mwork[blockIdx.x];
    const int k = mw.k, ot0 = mw.ot0, ot_last = mw.ot1 - 1, T = mw.ot1 - mw.ot0;
    const int Hp = dm.Hp, OT = dm.OT, Op = dm.Op;
    constexpr int ldd = DIMN_MIDP_LDD;
    float* ddl = lds;                                        // Dd [64][ldd]
    float* zpl = ddl + 64 * ldd;                             // partial Z [2][wave][mt][half][64 lanes][2]
    float* dzl = zpl + 2 * 8192;                             // dZ [2][64 b][16 o]
    float* wsl = dzl + 2 * 1024;                             // per-wave W2 transposes [8][2 tiles][256]
    float* gbl = wsl + 8 * 512;                              // b2 gradient parts [tile <= 8][wave][16]
    float* lsl = gbl + 8 * 128;                              // loss partials [8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    MIDP_STAMP(0)

    // ---- requests, in the order they are needed ----
    const float* ddk = Dd + (int64_t)k * DIMN_TB * Hp;
    f32x4 ddv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ddv[i] = *(const f32x4*)(ddk + tid * 4 + i * 2048);
    struct Set { f32x4 w[2], m[2], v[2]; };
    Set s[3];
    const int64_t tbase = (int64_t)k * Hp * Op + li * 16 + 4 * lj;                     // lane <-> W2[h = 16 ht' + li][o = 16 ot + 4 lj ..]
    auto tidx = [&](int ht, int ot) { return tbase + ((int64_t)(2 * wave + ht) * OT + ot) * 256; };
    auto fetch = [&](Set& st, int t) {
        const int o2 = ot0 + t < ot_last ? ot0 + t : ot_last;                          // clamped: requests beyond the slice stay in bounds
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) st.w[ht] = *(const f32x4*)(W2 + tidx(ht, o2));
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) { st.m[ht] = DIMN_LD_MV(M2 + tidx(ht, o2)); st.v[ht] = DIMN_LD_MV(V2 + tidx(ht, o2)); }
    };
    // softplus stage: thread <-> two elements of the [64 b][16 o] tile: b = 16 mt + 4 lj + 2 hf + {0, 1}, o = li
    const int hf = wave >> 2, smt = wave & 3;
    const int sb0 = 16 * smt + 4 * lj + 2 * hf;
    const int64_t yrow0 = ((int64_t)k * n_cells + rows[sb0 < b_act ? sb0 : 0]) * Op + li;
    const int64_t yrow1 = ((int64_t)k * n_cells + rows[sb0 + 1 < b_act ? sb0 + 1 : 0]) * Op + li;
    float yn[2][2], bn[2];
    auto fetch_y = [&](int g, int t) {
        const int o2 = ot0 + t < ot_last ? ot0 + t : ot_last;
        yn[g][0] = Y[yrow0 + 16 * o2]; yn[g][1] = Y[yrow1 + 16 * o2];
        bn[g] = b2w[(int64_t)k * Op + 16 * o2 + li];
    };
    fetch(s[0], 0);
    fetch_y(0, 0);
    fetch(s[1], 1);
    fetch_y(1, 1);
    // Adam(b2) happens once, after the last tile: thread tid < 16 T <-> (tile tid >> 4, column tid & 15)
    const bool b2_owner = tid < 16 * T;
    const int64_t b2i = (int64_t)k * Op + 16 * (ot0 + (b2_owner ? tid >> 4 : 0)) + (tid & 15);
    float b2w0 = 0.f, b2m0 = 0.f, b2v0 = 0.f;
    if (b2_owner) { b2w0 = b2w[b2i]; b2m0 = b2m[b2i]; b2v0 = b2v[b2i]; }

#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid * 4 + i * 2048, b = e >> 8, h = e & 255;
        *(f32x4*)(ddl + b * ldd + h) = ddv[i];
    }
    __syncthreads();
    MIDP_STAMP(1)
    float ddf[16][2];    // B operand of gW2^T: Dd[b = 4kb+lj][h = 16(2w+ht)+li]
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) ddf[kb][ht] = ddl[(4 * kb + lj) * ldd + 16 * (2 * wave + ht) + li];
    f32x4 dacc[4][2];
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) dacc[m4][ht] = zero4;
    float* ws = wsl + wave * 512;
    float lsum = 0.f;

    // forward(t): partial Z[64 b][16 o] over this wave's 32 hidden rows -> zp[t & 1]
    auto forward = [&](const Set& st, int t) {
        // k-slot form: MFMA r of a hidden tile takes k = 4 lj + r, so one 16-byte LDS read of a Dd row feeds four MFMAs (A)
        // and the W2 operand is the transposed tile read at row 4 lj + r (B)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) *(f32x4*)(ws + ht * 256 + li * 16 + 4 * lj) = st.w[ht];
        float bq[2][4];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r) bq[ht][r] = ws[ht * 256 + (4 * lj + r) * 16 + li];       // W2[h = 16 ht' + 4lj + r][o = li]
        f32x4 acc[4] = {zero4, zero4, zero4, zero4};
        const float* arow = ddl + li * ldd + 32 * wave + 4 * lj;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            f32x4 a4[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a4[mt] = *(const f32x4*)(arow + 16 * mt * ldd + 16 * ht);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA16(a4[mt][r], bq[ht][r], acc[mt]);
        }
        float* zp = zpl + (t & 1) * 8192 + wave * 1024;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {                     // halves apart: the readers' 8-byte reads are lane-contiguous
            *(float2*)(zp + (mt * 2 + 0) * 128 + lane * 2) = make_float2(acc[mt][0], acc[mt][1]);
            *(float2*)(zp + (mt * 2 + 1) * 128 + lane * 2) = make_float2(acc[mt][2], acc[mt][3]);
        }
    };
    // softplus(t): Z = sum of the partials + b2 ; loss ; dZ -> dz[t & 1] ; column sums of dZ -> gbl[t]
    auto softplus = [&](int g, int t) {
        const float* zp = zpl + (t & 1) * 8192 + (smt * 2 + hf) * 128 + lane * 2;
        float2 z2 = *(const float2*)zp;
#pragma unroll
        for (int wv = 1; wv < 8; ++wv) { const float2 p = *(const float2*)(zp + wv * 1024); z2.x += p.x; z2.y += p.y; }
        const bool col_ok = 16 * (ot0 + t) + li < dm.O;
        float* dz = dzl + (t & 1) * 1024;
        float gb = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int b = sb0 + e;
            const bool ok = b < b_act && col_ok;
            const float z = (e ? z2.y : z2.x) + bn[g];
            const float y = yn[g][e];
            const float w = loss_binary ? (y > 0.f ? 1.f : 0.f) : y;       // multinet.py:37-40
            float sp, sg;
            softplus_sigmoid_fast(z, sp, sg);
            const float er = y - sp;
            lsum += ok ? w * er * er : 0.f;
            const float d = ok ? -2.f * w * er * inv_n * sg : 0.f;
            dz[b * 16 + li] = d;
            gb += d;
        }
        gb += __shfl_xor(gb, 16);
        gb += __shfl_xor(gb, 32);
        if (lj == 0) gbl[t * 128 + wave * 16 + li] = gb;
    };
    // backward(t): as phase 2 of k_mid_fused, the old W2 from the set's registers
    auto backward = [&](Set& cur, int t) {
        const float* zb = dzl + (t & 1) * 1024;
        f32x4 g[2] = {zero4, zero4};
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float az = zb[64 * kb + lane];                             // dZ^T[o = li][b = 4kb+lj]
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) g[ht] = MFMA16(az, ddf[kb][ht], g[ht]);
        }
        f32x4 zf[4];
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) zf[m4] = *(const f32x4*)(zb + (16 * m4 + li) * 16 + 4 * lj);   // dZ[b][o = 4lj+r]
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4) dacc[m4][ht] = MFMA16(zf[m4][r], cur.w[ht][r], dacc[m4][ht]);    // OLD W2
            adam4(cur.w[ht], cur.m[ht], cur.v[ht], g[ht], ap);
            const int64_t i = tidx(ht, ot0 + t);
            DIMN_ST_STATE(W2 + i, cur.w[ht]); DIMN_ST_STATE(M2 + i, cur.m[ht]); DIMN_ST_STATE(V2 + i, cur.v[ht]);
        }
    };

    forward(s[0], 0);
    __syncthreads();
    softplus(0, 0);
    MIDP_STAMP(2)
    // straight-line code, one block per tile, left at the slice's last tile: the compiler counts the exact vmcnt of every wait
#define DIMN_MIDP_BLOCK(I)                                                              \
    {                                                                                   \
        const bool more = (I) + 1 < T;                                                  \
        fetch(s[((I) + 2) % 3], (I) + 2);                                               \
        fetch_y((I) & 1, (I) + 2);                                                      \
        if (more) forward(s[((I) + 1) % 3], (I) + 1);                                   \
        __syncthreads();                                                                \
        if (more) softplus(((I) + 1) & 1, (I) + 1);                                     \
        backward(s[(I) % 3], (I));                                                      \
        if (!more) break;                                                               \
    }
    do {
        DIMN_MIDP_BLOCK(0) DIMN_MIDP_BLOCK(1) DIMN_MIDP_BLOCK(2) DIMN_MIDP_BLOCK(3)
        DIMN_MIDP_BLOCK(4) DIMN_MIDP_BLOCK(5) DIMN_MIDP_BLOCK(6) DIMN_MIDP_BLOCK(7)
    } while (0);
#undef DIMN_MIDP_BLOCK
    MIDP_STAMP(3)

    float* p2 = P2 + (int64_t)mw.slot * DIMN_TB * Hp;
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r) DIMN_ST_P2(&p2[(16 * m4 + 4 * lj + r) * Hp + 16 * (2 * wave + ht) + li], dacc[m4][ht][r]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    if (lane == 0) lsl[wave] = lsum;
    __syncthreads();                                         // every tile's b2 parts and the loss partials are in LDS
    if (b2_owner) {
        const float* gp = gbl + (tid >> 4) * 128 + (tid & 15);
        float gb = gp[0];
#pragma unroll
        for (int wv = 1; wv < 8; ++wv) gb += gp[wv * 16];
        adam1(b2w0, b2m0, b2v0, gb, ap);
        b2w[b2i] = b2w0; b2m[b2i] = b2m0; b2v[b2i] = b2v0;
    }
    if (tid == 0) {
        float tot = 0.f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) tot += lsl[wv];
        loss_step[k * dm.LS + mw.sidx] = tot;
        if (loss_acc) loss_acc[k * dm.LS + mw.sidx] += (double)tot;
    }
    MIDP_STAMP(4)
}
