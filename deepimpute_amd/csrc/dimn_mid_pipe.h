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
//              forward(i+1): partial Z over the wave's 32 hidden rows (32 MFMAs) -> LDS                      | one barrier
//              softplus / wMSE / dZ(i+1): all 512 threads, two elements each, summing the 8 partials    }  one scheduling region: the
//              backward(i): gW2^T = dZ^T Dd (32 MFMAs), dD += dZ W2old^T (32), Adam, 16-byte stores      }  VALU work issues under the MFMAs
//
// so tile i's stores, tile i+2's loads and the MFMAs of tiles i, i+1 are in flight together; one barrier per tile.  Dd lives in
// wave-private LDS slabs (all a wave ever reads of Dd are ITS 32 hidden columns: no staging barrier); the slices of one sub-net run on
// one XCD.  In the cfg3 step: 36.1 us against k_mid_fused's 39.9 (rocprofv3, same box); DESIGN.md section 2 has the history and what
// bounds it now.  Summation orders differ from k_mid_fused (Z over 8 partial sums of 32, b2's gradient over 8 x 8 rows); both are
// within the parity tolerance of the oracle (oracle/dimo.c sums in its own order).  LDS: 8 slabs (73.7 KB) + two partial-Z buffers
// (64 KB; a wave's part doubles as its W2 transpose scratch) + two dZ tiles (8 KB) + b2 gradient parts (4.5 KB) = 152 096 bytes.
#pragma once

#define DIMN_MIDP_LDW 36                                  // row stride of a wave's Dd slab [64 b][32 h]: 16-byte aligned rows, 4 mod 32 words
#define DIMN_MIDP_LDS_FLOATS (8 * 64 * DIMN_MIDP_LDW + 2 * 8192 + 2 * 1024 + 9 * 128 + 8)     // 152 096 bytes


// BF (handles of precision bf16): the three GEMMs take bf16 operands -- Dd, W2, dZ rounded to nearest even in registers, one v_mfma_f32_16x16x16_bf16 where
// four fp32 MFMAs were (the k-slot register groups ARE its operands), fp32 accumulation, fp32 master weights and Adam state; as k_mid_fused<KEEP, BF>, which
// oracle/dimo.c restates (dimo_set_training_bf16).
template <bool BF>
__global__ __launch_bounds__(512) void k_mid_pipe(const MidWork* __restrict__ mwork,
                                                  float* __restrict__ W2, float* __restrict__ M2, float* __restrict__ V2,
                                                  float* __restrict__ b2w, float* __restrict__ b2m, float* __restrict__ b2v,
                                                  const float* __restrict__ Y, int64_t n_cells,
                                                  const int32_t* __restrict__ rows, int b_act,
                                                  const float* __restrict__ Dd, float* __restrict__ P2,
                                                  float* __restrict__ loss_step, double* __restrict__ loss_acc,
                                                  Dims dm, AdamP ap, float inv_n, int loss_binary) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // slices of one sub-net on ONE XCD (workgroup b runs on XCD b % 8, every XCD has its own L2): the sub-net's Dd block and batch rows of Y
    // come from memory once per sub-net instead of once per slice
    const int nwg = gridDim.x, xq = nwg >> 3, xr = nwg & 7, xcd = blockIdx.x & 7;
    const MidWork mw = mwork[xcd * xq + (xcd < xr ? xcd : xr) + (blockIdx.x >> 3)];
    const int k = mw.k, ot0 = mw.ot0, ot_last = mw.ot1 - 1, T = mw.ot1 - mw.ot0;
    const int Hp = dm.Hp, OT = dm.OT, Op = dm.Op;
    constexpr int ldw = DIMN_MIDP_LDW;
    float* ddl = lds;                                        // Dd[:, 32w .. 32w+31] of every wave [8][64][ldw]: all a wave ever reads of Dd are ITS hidden columns
    float* zpl = ddl + 8 * 64 * ldw;                         // partial Z [2][wave][mt][half][64 lanes][2]
    float* dzl = zpl + 2 * 8192;                             // dZ [2][64 b][16 o]
    float* gbl = dzl + 2 * 1024;                             // b2 gradient parts [tile <= 8][wave][16] (+ one block for the pass behind the last tile)
    float* lsl = gbl + 9 * 128;                              // loss partials [8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- requests, in the order they are needed: the wave's Dd slab and the first tile's w, then everything else ----
    const float* ddk = Dd + (int64_t)k * DIMN_TB * Hp + 32 * wave;
    f32x4 ddv[8];                                            // 8 lanes per row of 128 bytes, 8 rows per request
#pragma unroll
    for (int i = 0; i < 8; ++i) ddv[i] = *(const f32x4*)(ddk + (8 * i + (lane >> 3)) * Hp + 4 * (lane & 7));
    struct Set { f32x4 w[2], m[2], v[2]; };
    Set sA, sB, sC;                                          // (named, not an array: blocks that differ only in the set they use must not be merged into one with a runtime index)
    // addresses: a wave-uniform base (sub-net, output tile: scalar registers) + a 32-bit lane offset that never changes (hidden tile, lane)
    const size_t kbase = (size_t)k * Hp * Op;
    const char *w2k = (const char*)(W2 + kbase), *m2k = (const char*)(M2 + kbase), *v2k = (const char*)(V2 + kbase);
    unsigned voff[2];                                        // lane <-> W2[h = 16 ht' + li][o = 16 ot + 4 lj ..], bytes
#pragma unroll
    for (int ht = 0; ht < 2; ++ht) voff[ht] = 4u * (unsigned)(((2 * wave + ht) * OT) * 256 + li * 16 + 4 * lj);
    // Every block issues the same requests (so the compiler's vmcnt arithmetic is exact on every path); those of a tile beyond the
    // slice's last one ask all 64 lanes for the SAME 16 bytes of the slice's first tile -- one cache line per instruction, no traffic.
    auto fetch_w = [&](Set& st, int t) {
        const size_t tb = (size_t)(t < T ? ot0 + t : ot0) * 1024;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) st.w[ht] = *(const f32x4*)(w2k + tb + (t < T ? voff[ht] : 0u));
    };
    auto fetch_mv = [&](Set& st, int t) {
        const size_t tb = (size_t)(t < T ? ot0 + t : ot0) * 1024;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            const unsigned vo = t < T ? voff[ht] : 0u;
            st.m[ht] = DIMN_LD_MV(m2k + tb + vo); st.v[ht] = DIMN_LD_MV(v2k + tb + vo);
        }
    };
    // softplus stage: thread <-> two elements of the [64 b][16 o] tile: b = 16 mt + 4 lj + 2 hf + {0, 1}, o = li
    const int hf = wave >> 2, smt = wave & 3;
    const int sb0 = 16 * smt + 4 * lj + 2 * hf;
    const int64_t yrow0 = ((int64_t)k * n_cells + rows[sb0 < b_act ? sb0 : 0]) * Op + li;
    const int64_t yrow1 = ((int64_t)k * n_cells + rows[sb0 + 1 < b_act ? sb0 + 1 : 0]) * Op + li;
    struct YGen { float y0, y1, b; };
    YGen gE, gO;                                             // targets and bias of the next even / odd tile
    auto fetch_y = [&](YGen& g, int t) {
        const int o2 = t < T ? ot0 + t : ot_last;
        g.y0 = Y[yrow0 + 16 * o2]; g.y1 = Y[yrow1 + 16 * o2];
        g.b = b2w[(int64_t)k * Op + 16 * o2 + li];
    };
    fetch_w(sA, 0);
    __builtin_amdgcn_sched_barrier(0);                       // (what forward(0) waits for goes first, chip-wide; the rest follows when the slab has arrived)
    float* slab = ddl + wave * 64 * ldw;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(f32x4*)(slab + (8 * i + (lane >> 3)) * ldw + 4 * (lane & 7)) = ddv[i];
    __builtin_amdgcn_sched_barrier(0);
    fetch_y(gE, 0);
    fetch_mv(sA, 0);
    fetch_w(sB, 1);                                          // (its m and v: in block 0 -- the less the start burst carries, the sooner forward(0) has its operands)
    fetch_y(gO, 1);
    // Adam(b2) happens once, after the last tile: thread tid < 16 T <-> (tile tid >> 4, column tid & 15)
    const bool b2_owner = tid < 16 * T;
    const int64_t b2i = (int64_t)k * Op + 16 * (ot0 + (b2_owner ? tid >> 4 : 0)) + (tid & 15);
    float b2w0 = 0.f, b2m0 = 0.f, b2v0 = 0.f;
    if (b2_owner) { b2w0 = b2w[b2i]; b2m0 = b2m[b2i]; b2v0 = b2v[b2i]; }
    float ddf[16][2];    // B operand of gW2^T, kept for every tile: Dd[b = 4kb+lj][h = 16(2w+ht)+li]   (the slab is private to the wave: no barrier)
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) ddf[kb][ht] = slab[(4 * kb + lj) * ldw + 16 * ht + li];
    f32x4 dacc[4][2];
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) dacc[m4][ht] = zero4;
    float lsum = 0.f;

    // forward(t): partial Z[64 b][16 o] over this wave's 32 hidden rows -> zp[t & 1]
    auto forward = [&](const Set& st, int t) {
        // k-slot form: MFMA r of a hidden tile takes k = 4 lj + r, so one 16-byte LDS read of a Dd row feeds four MFMAs (A)
        // and the W2 operand is the transposed tile read at row 4 lj + r (B)
        float* zp = zpl + (t & 1) * 8192 + wave * 1024;      // this wave's part of the tile's partial-Z buffer; until the partial is written, its transpose scratch
        float* ws = zp;                                      // (the buffer's last readers passed a barrier since; LDS operations of one wave complete in order)
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) *(f32x4*)(ws + ht * 256 + li * 16 + 4 * lj) = st.w[ht];
        float bq[2][4];
#pragma unroll
        for (int ht = 0; ht < 2; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r) bq[ht][r] = ws[ht * 256 + (4 * lj + r) * 16 + li];       // W2[h = 16 ht' + 4lj + r][o = li]
        f32x4 acc[4] = {zero4, zero4, zero4, zero4};
        const float* arow = slab + li * ldw + 4 * lj;
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            f32x4 a4[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) a4[mt] = *(const f32x4*)(arow + 16 * mt * ldw + 16 * ht);
            if constexpr (BF) {
                const bf16x4 bp = pk4((f32x4){bq[ht][0], bq[ht][1], bq[ht][2], bq[ht][3]});
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA_BF16(pk4(a4[mt]), bp, acc[mt]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[mt] = MFMA16(a4[mt][r], bq[ht][r], acc[mt]);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {                     // halves apart: the readers' 8-byte reads are lane-contiguous
            *(float2*)(zp + (mt * 2 + 0) * 128 + lane * 2) = make_float2(acc[mt][0], acc[mt][1]);
            *(float2*)(zp + (mt * 2 + 1) * 128 + lane * 2) = make_float2(acc[mt][2], acc[mt][3]);
        }
    };
    // softplus(t): Z = sum of the partials + b2 ; loss ; dZ -> dz[t & 1] ; column sums of dZ -> gbl[t].  Branch-free (selects only), so that it
    // shares a scheduling region with backward(t - 1) and its arithmetic issues under that tile's matrix instructions; the block of the slice's last
    // tile runs it once more on stale data (live = false: nothing of it is kept).
    auto softplus = [&](const YGen& g, int t, bool live) {
        const float* zp = zpl + (t & 1) * 8192 + (smt * 2 + hf) * 128 + lane * 2;
        float2 z2 = *(const float2*)zp;
#pragma unroll
        for (int wv = 1; wv < 8; ++wv) { const float2 p = *(const float2*)(zp + wv * 1024); z2.x += p.x; z2.y += p.y; }
        const bool col_ok = live && 16 * (ot0 + t) + li < dm.O;
        float* dz = dzl + (t & 1) * 1024;
        float gb = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int b = sb0 + e;
            const bool ok = b < b_act && col_ok;
            const float z = (e ? z2.y : z2.x) + g.b;
            const float y = e ? g.y1 : g.y0;
            const float w = loss_binary ? (y > 0.f ? 1.f : 0.f) : y;       // multinet.py:37-40
            float sp, sg;
            softplus_sigmoid_fast(z, sp, sg);
            const float er = y - sp;
            const float le = w * er * er, de = -2.f * w * er * inv_n * sg;
            lsum += ok ? le : 0.f;
            const float d = ok ? de : 0.f;
            dz[b * 16 + li] = d;
            gb += d;
        }
        gb += __shfl_xor(gb, 16);
        gb += __shfl_xor(gb, 32);
        gbl[(live ? t : 8) * 128 + wave * 16 + li] = gb;     // (every lj writes the same sum to the same word)
    };
    // backward(t): as phase 2 of k_mid_fused, the old W2 from the set's registers
    auto backward = [&](Set& cur, int t) {
        const float* zb = dzl + (t & 1) * 1024;
        f32x4 g[2] = {zero4, zero4};
        if constexpr (BF) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                    // four batch rows per lane and instruction: k-slot i of lane (.., lj) <-> b = 16q + 4i + lj
                const bf16x4 ap = pk4((f32x4){zb[64 * (4 * q) + lane], zb[64 * (4 * q + 1) + lane], zb[64 * (4 * q + 2) + lane], zb[64 * (4 * q + 3) + lane]});
#pragma unroll
                for (int ht = 0; ht < 2; ++ht)
                    g[ht] = MFMA_BF16(ap, pk4((f32x4){ddf[4 * q][ht], ddf[4 * q + 1][ht], ddf[4 * q + 2][ht], ddf[4 * q + 3][ht]}), g[ht]);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                const float az = zb[64 * kb + lane];                         // dZ^T[o = li][b = 4kb+lj]
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) g[ht] = MFMA16(az, ddf[kb][ht], g[ht]);
            }
        }
        f32x4 zf[4];
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) zf[m4] = *(const f32x4*)(zb + (16 * m4 + li) * 16 + 4 * lj);   // dZ[b][o = 4lj+r]
#pragma unroll
        for (int ht = 0; ht < 2; ++ht) {
            if constexpr (BF) {
                const bf16x4 wp = pk4(cur.w[ht]);
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4) dacc[m4][ht] = MFMA_BF16(pk4(zf[m4]), wp, dacc[m4][ht]);      // OLD W2
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m4 = 0; m4 < 4; ++m4) dacc[m4][ht] = MFMA16(zf[m4][r], cur.w[ht][r], dacc[m4][ht]);   // OLD W2
            }
            adam4(cur.w[ht], cur.m[ht], cur.v[ht], g[ht], ap);
            const size_t tb = (size_t)(ot0 + t) * 1024;
            { DIMN_ST_STATE(const_cast<char*>(w2k) + tb + voff[ht], cur.w[ht]); DIMN_ST_STATE(const_cast<char*>(m2k) + tb + voff[ht], cur.m[ht]); DIMN_ST_STATE(const_cast<char*>(v2k) + tb + voff[ht], cur.v[ht]); }
        }
    };

    forward(sA, 0);
    __syncthreads();
    softplus(gE, 0, true);
    // Straight-line code, one block per tile, left at the slice's last tile (CUR / NXT / FRE: the sets of tiles I, I+1, I+2;
    // GN / GF: the targets of tiles I+1, I+2).  No path joins another with a different number of requests in flight, so every wait is
    // an exact vmcnt and none of them covers the stores of the tile before.
#define DIMN_MIDP_BLOCK(I, CUR, NXT, FRE, GN, GF)                                       \
    {                                                                                   \
        const bool more = (I) + 1 < T;                                                  \
        if ((I) == 0) fetch_mv(NXT, 1);                                                 \
        fetch_w(FRE, (I) + 2); fetch_mv(FRE, (I) + 2);                                  \
        fetch_y(GF, (I) + 2);                                                           \
        if (more) forward(NXT, (I) + 1);                                                \
        __syncthreads();                                                                \
        softplus(GN, (I) + 1, more);                                                    \
        backward(CUR, (I));                                                             \
        if (!more) break;                                                               \
    }
    do {
        DIMN_MIDP_BLOCK(0, sA, sB, sC, gO, gE)
        DIMN_MIDP_BLOCK(1, sB, sC, sA, gE, gO)
        DIMN_MIDP_BLOCK(2, sC, sA, sB, gO, gE)
        DIMN_MIDP_BLOCK(3, sA, sB, sC, gE, gO)
        DIMN_MIDP_BLOCK(4, sB, sC, sA, gO, gE)
        DIMN_MIDP_BLOCK(5, sC, sA, sB, gE, gO)
        DIMN_MIDP_BLOCK(6, sA, sB, sC, gO, gE)
        DIMN_MIDP_BLOCK(7, sB, sC, sA, gE, gO)
    } while (0);
#undef DIMN_MIDP_BLOCK

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
}
