// dimn_kernels.h -- hand-written CDNA4 (gfx950) kernels of libdimn.
//
// The hot path of DeepImpute's MultiNet (reference deepimpute/multinet.py:126-167 build,
// :238-244 model.fit, :253/:278 model.predict) as a batched pipeline over K independent
// sub-networks  Dense(H,relu) -> Dropout(p) -> Dense(O,softplus)  with wMSE and Keras-form
// Adam.  All matmuls run on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32, 64-lane
// wavefronts); nothing here is a translation of the reference, which has no kernels.
//
// MFMA 16x16x4 f32 fragment maps (lane l, li = l&15, lj = l>>4):
//   A[i=li][k=lj]   B[k=lj][j=li]   C/D: col = li, row = 4*lj + reg
// "k-slot trick": the hardware sums over the 4 k-slots in any order we like, so for a
// 16-deep K chunk we let MFMA r (r=0..3) use k = 4*lj + r.  A lane then needs 4
// CONSECUTIVE k's of one row: one 16-byte load feeds 4 MFMAs, and a C/D register block
// (row = 4*lj + r) can be re-used directly as a B operand.
//
// HBM layouts (all fp32):
//   X_k  [n][Dp_k]            gathered predictors of sub-net k, Dp = ceil16(D), zero padded
//   Y_k  [n][Op]              gathered targets, Op = ceil16(O)
//   W1_k [Dp/16][Hp][16]      "chunk-blocked": chunk c, hidden unit h, 16 consecutive d
//                             -> a 16x16 MFMA tile is one contiguous 1 KiB; same for m, v
//   W2_k [Hp/16][Op/16][16h][16o]  tile-blocked, 1 KiB per tile; same for m, v
//   b1 [Hp], b2 [Op]; workspaces P (split-K partials), Dd, dZ, dA: [64][Hp|Op] per sub-net
// Padded rows/columns are zero and provably stay zero under Adam (g=0, m=v=0 -> dw=0).
//
// Kernels (one optimiser step = RED -> MF -> MB -> B1F1; see DESIGN.md section 2 for measurements):
//   k_gather_lds / k_gather   device gather of X_k, Y_k from the shared log1p matrix
//   k_init_weights            Glorot-uniform Philox init into the blocked layouts
//   k_fwd1                    split-K first layer (first step of an epoch, single-step API)
//   k_reduce_act     RED      sum of split-K partials + bias + ReLU + Philox dropout -> Dd
//   k_mid_fwd        MF       second layer, softplus, wMSE, dZ, Adam(b2)
//   k_mid_bwd        MB       W2 gradient + Adam in registers, dD with the old W2, dA, Adam(b1)
//   k_w1_update_fwd_ring      B1F1 for 8 .. 24 hidden tiles (H = 113 .. 384): W1 gradient + Adam in registers + next step's split-K
//                             forward; one hidden tile per wave, LDS-staged X tiles, three- (H = 256) or four-set register ring
//   k_w1_update_fwd           B1F1 for any other H (8 independent waves, wave-private staging, no barrier)
//   k_mid_pipe (dimn_mid_pipe.h)  the whole second layer at H = 256 as a tile pipeline (RED -> k_mid_pipe -> RED2 -> B1F1)
//   k_predict                 fused forward for model.predict and the validation loss
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

#include "../../include/dimn_rng.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

// ---- bf16 X arena (precision DIMN_PREC_BF16: BASELINE configs[4]) -------------------------------------------------
// The gathered predictor blocks X_k may be stored in bfloat16 (half the arena and half the X traffic; the weights, the
// Adam state and every accumulation stay fp32).  A kernel templated on the element type XT keeps a tile piece in its
// RAW form (what the load instruction returns: 16 bytes of fp32 or 8 bytes of bf16) across the software pipeline and
// expands it to four floats only where the fp32 path would use the registers -- so the load stays asynchronous.
typedef unsigned short bf16_t;
typedef short bf16x4 __attribute__((ext_vector_type(4)));
template <typename XT> struct XRaw;
template <> struct XRaw<float> {
    f32x4 v;
    __device__ __forceinline__ void load(const float* p) { v = *(const f32x4*)p; }
    __device__ __forceinline__ f32x4 get() const { return v; }
};
template <> struct XRaw<bf16_t> {
    uint2 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *(const uint2*)p; }
    __device__ __forceinline__ f32x4 get() const {
        return (f32x4){__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
    }
};
__host__ __device__ static inline bf16_t f32_to_bf16(float f) {       // round to nearest even (finite inputs)
    union { float f; uint32_t u; } c; c.f = f;
    return (bf16_t)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}
__host__ __device__ static inline float bf16_to_f32(bf16_t b) { union { float f; uint32_t u; } c; c.u = (uint32_t)b << 16; return c.f; }
template <typename XT> __device__ __forceinline__ XT x_store(float f);
template <> __device__ __forceinline__ float x_store<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t x_store<bf16_t>(float f) { return f32_to_bf16(f); }
// four floats -> the bf16 operand of one v_mfma_f32_16x16x16_bf16, rounded to nearest even: the compiler turns the vector cast
// into two v_cvt_pk_bf16_f32 and knows their hazards (hand-written as inline asm the MFMA that followed read the operand too early)
typedef __bf16 bf16x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 pk4(f32x4 v) { return __builtin_bit_cast(bf16x4, __builtin_convertvector(v, bf16x4n)); }

// Streamed optimiser state (read once, written once per step).  DIMN_NT bit 0 marks all of B1F1's state loads, bit 4
// only its m and v loads, bit 1 the state stores non-temporal; bit 2 the split-K partial stores of B1F1, bit 3 the dD
// partial stores of MFB.  Measured (tools/ab_nt.sh / ab_def.sh, one box, cfg3): non-temporal STATE STORES keep the X
// tiles / dA / partials in L2: step 0.1864 -> 0.1731 ms, B1F1 121 -> 113 us; non-temporal loads of m and v (not of w):
// B1F1 120.8 -> 114.1 us, step 0.1776 -> 0.1709 ms (four A/B pairs); all three loads non-temporal: no better than none.
// Default: stores + m, v loads.
#ifndef DIMN_NT
#define DIMN_NT 18
#endif
#if DIMN_NT & 1
#define DIMN_LD_STATE(p) __builtin_nontemporal_load((const f32x4*)(p))
#else
#define DIMN_LD_STATE(p) (*(const f32x4*)(p))
#endif
#if DIMN_NT & 16
#define DIMN_LD_STATE_MV(p) __builtin_nontemporal_load((const f32x4*)(p))
#else
#define DIMN_LD_STATE_MV(p) DIMN_LD_STATE(p)
#endif
#if DIMN_NT & 2
#define DIMN_ST_STATE(p, v) __builtin_nontemporal_store((v), (f32x4*)(p))
#else
#define DIMN_ST_STATE(p, v) (*(f32x4*)(p) = (v))
#endif
#if DIMN_NT & 4
#define DIMN_ST_P(p, v) __builtin_nontemporal_store((v), (p))
#else
#define DIMN_ST_P(p, v) (*(p) = (v))
#endif
#if DIMN_NT & 8
#define DIMN_ST_P2(p, v) __builtin_nontemporal_store((v), (p))
#else
#define DIMN_ST_P2(p, v) (*(p) = (v))
#endif
// m and v of the second layer are read once per step by the fused kernel (W2 itself twice: phases 1 and 2): non-temporal
// loads keep them out of the caches' way -- step 0.1693 -> 0.1656 ms, both weight kernels gain (tools/ab_def.sh)
#ifndef DIMN_MFB_NT_MV
#define DIMN_MFB_NT_MV 1
#endif
#if DIMN_MFB_NT_MV
#define DIMN_LD_MV(p) __builtin_nontemporal_load((const f32x4*)(p))
#else
#define DIMN_LD_MV(p) (*(const f32x4*)(p))
#endif
#ifndef DIMN_W_PEEL
#define DIMN_W_PEEL 1   // B1F1 ring: first chunk triple peeled out of the loop (0: wait for two chunks before the loop)
#endif
#define DIMN_TB 64  // batch rows per optimiser step tile (4 MFMA M-tiles)

struct SubnetDev {
    int32_t D, Dp, nchunk;   // predictors, padded, Dp/16
    int32_t slot0, nslice;   // split-K partial slots of this sub-net
    int32_t kg;              // global sub-net index (RNG key)
    int64_t xoff;            // float offset of X_k in the X arena (row stride Dp)
    int64_t w1off;           // float offset of W1_k in the W1/M1/V1 arenas
    float lim1, lim2;        // Glorot limits
};

struct Work {               // one workgroup of the split-K / weight-update kernels
    int32_t k, c0, c1, slot;
};

struct Dims {
    int32_t K, H, O, Hp, Op, HT, OT, ldd, OS, LS, ldp;
    // Hp = ceil16(H), HT = Hp/16, Op = ceil16(O), OT = Op/16, ldd = LDS row stride of Dd (2 mod 32 words: b32 column reads),
    // ldp = LDS row stride of k_predict's activations (4 mod 32 words, 16-byte aligned rows: b128 row reads),
    // OS = ceil(OT/4) output slices of 64 columns; LS = loss slots per sub-net (>= OS, >= slices of the fused kernel)
};

struct AdamP {
    float alpha, omb1, omb2, eps;   // alpha = lr*sqrt(1-b2^t)/(1-b1^t); 1-beta1; 1-beta2
};

__device__ __forceinline__ void adam4(f32x4& w, f32x4& m, f32x4& v, const f32x4 g, const AdamP a) {
    // TF ResourceApplyAdam (Keras Adam, multinet.py:164): eps outside the bias correction
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m[r] += (g[r] - m[r]) * a.omb1;
        v[r] += (g[r] * g[r] - v[r]) * a.omb2;
        // 1-ulp hardware sqrt/rcp instead of the IEEE expansions (~3x fewer VALU ops); the update
        // is <= lr in magnitude, so the extra ~2e-7 relative error is ~1e-11 absolute per step
        w[r] -= (m[r] * a.alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v[r]) + a.eps);
    }
}
__device__ __forceinline__ void adam1(float& w, float& m, float& v, const float g, const AdamP a) {
    m += (g - m) * a.omb1;
    v += (g * g - v) * a.omb2;
    w -= (m * a.alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + a.eps);
}

__device__ __forceinline__ float softplus_f(float x) {
    // TensorFlow's fp32 softplus thresholds (log(eps)+2 ~ -13.94), S4
    const float thr = 13.942385f;
    if (x > thr) return x;
    if (x < -thr) return expf(x);
    return log1pf(expf(x));
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
// The output layer of the inference kernels: the same function within ~3 ulp of libm's (log1pf(expf(x)) is ~100 VALU
// instructions per element -- 40 % of k_predict's wave time went there), no branch:  t = exp(-|x|) on v_exp_f32 with the
// exponent product x*log2(e) carried in two pieces;  log1p(t) = log(u) * t / (u - 1), u = 1 + t, on v_log_f32 / v_rcp_f32
// (the classic correction; its series t - t^2/2 + t^3/3 below 2^-12);  TensorFlow's thresholds as selects.
__device__ __forceinline__ float softplus_out(float x) {
    const float thr = 13.942385f;
    const float ax = fabsf(x);
    const float p = -ax * 1.44269502f;
    float pe = fmaf(-ax, 1.44269502f, -p);
    pe = fmaf(-ax, 1.92596299e-8f, pe);
    float t = __builtin_amdgcn_exp2f(p);
    t = fmaf(t, pe * 0.693147182f, t);                       // exp(-|x|)
    const float u = 1.0f + t, d = u - 1.0f;
    const float lg = __builtin_amdgcn_logf(u) * 0.693147182f;
    // (both forms computed, then selected: written as a conditional expression over the two computations the compiler emits a BRANCH per
    //  element -- exec-mask juggling and a scheduling barrier 128 times per thread in the unrolled output epilogues)
    const float l_series = t * fmaf(t, fmaf(t, 0.333333343f, -0.5f), 1.0f);                   // below 2^-12: no cancellation in u - 1
    const float l_log = lg * (t * __builtin_amdgcn_rcpf(d));
    const float l = t < 2.44140625e-4f ? l_series : l_log;
    const float sp = fmaxf(x, 0.f) + l;
    float r = x < -thr ? t : sp;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(r));                              // (or the compiler turns the select below into a branch around everything above)
#endif
    return x > thr ? x : r;
}

// The same function on two elements at a time: every multiply / add / fma becomes ONE packed instruction (v_pk_mul_f32, v_pk_add_f32,
// v_pk_fma_f32: IEEE results, element for element those of softplus_out), only the three transcendentals and the selects stay per element --
// 18 instead of 27 instructions per element in k_predict_bf16's output epilogue.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 softplus_out2(f32x2 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float thr = 13.942385f;
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 p = -ax * 1.44269502f;
    f32x2 pe = __builtin_elementwise_fma(-ax, (f32x2)1.44269502f, -p);
    pe = __builtin_elementwise_fma(-ax, (f32x2)1.92596299e-8f, pe);
    f32x2 t = (f32x2){__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
    t = __builtin_elementwise_fma(t, pe * 0.693147182f, t);                                   // exp(-|x|)
    const f32x2 u = 1.0f + t, d = u - 1.0f;
    const f32x2 lg = (f32x2){__builtin_amdgcn_logf(u.x), __builtin_amdgcn_logf(u.y)} * 0.693147182f;
    const f32x2 l_series = t * __builtin_elementwise_fma(t, __builtin_elementwise_fma(t, (f32x2)0.333333343f, (f32x2)-0.5f), (f32x2)1.0f);
    const f32x2 l_log = lg * (t * (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)});
    const f32x2 l = t < 2.44140625e-4f ? l_series : l_log;
    const f32x2 sp = __builtin_elementwise_max(x, (f32x2)0.f) + l;
    f32x2 r = x < -thr ? t : sp;
    asm volatile("" : "+v"(r));                              // (as in softplus_out)
    return x > thr ? x : r;
#else
    return (f32x2){softplus_out(x.x), softplus_out(x.y)};
#endif
}

// Hidden activation f and derivative f' at pre-activation a (multinet.py:137; ids = DIMN_ACT_* of dimn.h, elu alpha = 1).
// relu is handled inline by the kernels (bit-identical to the path that has no activation switch).
__device__ __forceinline__ void hidden_act(int act, float a, float& f, float& df) {
    switch (act) {
        case 1: f = a; df = 1.f; break;
        case 2: { const float s = sigmoid_f(a); f = s; df = s * (1.f - s); break; }
        case 3: { const float t = tanhf(a); f = t; df = 1.f - t * t; break; }
        case 4: { const float e = expm1f(a); f = a > 0.f ? a : e; df = a > 0.f ? 1.f : e + 1.f; break; }
        case 5: f = softplus_f(a); df = sigmoid_f(a); break;
        case 6: { const float e = expm1f(a); f = 1.0507009873554805f * (a > 0.f ? a : 1.6732632423543772f * e);          // selu
                  df = 1.0507009873554805f * (a > 0.f ? 1.f : 1.6732632423543772f * (e + 1.f)); break; }
        case 7: { const float r = 1.f / (1.f + fabsf(a)); f = a * r; df = r * r; break; }                                  // softsign
        case 8: { const float s = sigmoid_f(a); f = a * s; df = s + a * s * (1.f - s); break; }                           // swish
        case 9: { const float c = 0.5f * (1.f + erff(a * 0.70710678118654752f)); f = a * c;                                // gelu (erf form)
                  df = c + a * 0.3989422804014327f * expf(-0.5f * a * a); break; }
        case 10: { const float e = expf(a); f = e; df = e; break; }                                                        // exponential
        case 11: { const float y = 0.2f * a + 0.5f; f = y < 0.f ? 0.f : (y > 1.f ? 1.f : y); df = (a > -2.5f && a < 2.5f) ? 0.2f : 0.f; break; }   // hard_sigmoid
        default: f = a > 0.f ? a : 0.f; df = a > 0.f ? 1.f : 0.f; break;
    }
}

// One out-of-line copy for kernels that unroll an activation many times (see k_predict_bf16).
__device__ __attribute__((noinline)) float hidden_act_call(int act, float a) {
    float f, df;
    hidden_act(act, a, f, df);
    return f;
}

// Training-path versions on the hardware exp2/log2/rcp units (v_exp_f32, v_log_f32, v_rcp_f32):
// t = exp(-|x|) in (0,1];  softplus = max(x,0) + log1p(t);  sigmoid = x>=0 ? 1/(1+t) : t/(1+t).
// log1p(t) switches to its series below 2^-12 so tiny outputs keep their relative accuracy.  They feed
// only the loss and dZ (absolute error ~1e-7); the inference kernels use softplus_out (within ~3 ulp of libm's).
__device__ __forceinline__ void softplus_sigmoid_fast(float x, float& sp, float& sg) {
    const float t = __expf(-fabsf(x));
    const float l_series = t * (1.0f - 0.5f * t), l_log = __logf(1.0f + t);    // (both, then a select: as a conditional expression over the
    const float l = t < 2.44140625e-4f ? l_series : l_log;                        //  computations it compiles to a branch per element)
    sp = fmaxf(x, 0.f) + l;
    const float r = __builtin_amdgcn_rcpf(1.0f + t);
    sg = x >= 0.f ? r : t * r;
}

// ---------------------------------------------------------------------------------------
// gather: X_k[i][d] = norm[i][pred_k[d]], Y_k[i][o] = norm[i][targ_k[o]]
// (replaces the K pandas .loc gathers, multinet.py:231-235 / 273-274).  grid (K, rows)
// ---------------------------------------------------------------------------------------
// `norm` holds rows [row0, row0 + n) of the matrix of n_all cells (row0 = 0, n = n_all: the whole matrix resident;
// otherwise one streamed row block, dimn_set_matrix_streamed).
template <typename XT>
__global__ __launch_bounds__(256) void k_gather(const SubnetDev* __restrict__ sn, const float* __restrict__ norm,
                                                int64_t n, int64_t g, const int32_t* __restrict__ pred,
                                                const int64_t* __restrict__ pred_off, const int32_t* __restrict__ targ,
                                                XT* __restrict__ X, float* __restrict__ Y, Dims dm, int with_targets, int64_t row0, int64_t n_all) {
    const int k = blockIdx.x;
    const SubnetDev s = sn[k];
    const int32_t* pk = pred + pred_off[k];
    for (int64_t i = blockIdx.y; i < n; i += gridDim.y) {
        const float* row = norm + i * g;
        XT* xr = X + s.xoff + (row0 + i) * s.Dp;
        for (int d = threadIdx.x; d < s.Dp; d += 256) xr[d] = x_store<XT>(d < s.D ? row[pk[d]] : 0.f);
        if (with_targets) {
            float* yr = Y + ((int64_t)k * n_all + row0 + i) * dm.Op;
            const int32_t* tk = targ + (int64_t)k * dm.O;
            for (int o = threadIdx.x; o < dm.Op; o += 256) yr[o] = o < dm.O ? row[tk[o]] : 0.f;
        }
    }
}

// Same gather with the matrix row staged ONCE in LDS (g floats <= 160 KB): every sub-net of the
// workgroup's row is served from LDS, so `norm` is read from HBM once instead of once per sub-net
// (measured on cfg3: 174 GB fetched by k_gather vs 4 GB needed).  grid = rows (grid-stride).
template <typename XT> __device__ __forceinline__ void x_store4(XT* p, const f32x4 v);      // four consecutive arena elements, one non-temporal store
template <> __device__ __forceinline__ void x_store4<float>(float* p, const f32x4 v) { __builtin_nontemporal_store(v, (f32x4*)p); }
template <> __device__ __forceinline__ void x_store4<bf16_t>(bf16_t* p, const f32x4 v) {
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((u16x4){f32_to_bf16(v[0]), f32_to_bf16(v[1]), f32_to_bf16(v[2]), f32_to_bf16(v[3])}, (u16x4*)p);
}
// Round 5: a thread takes FOUR consecutive columns -- four independent index loads in flight, four LDS gathers, one 16-byte (bf16: 8-byte) store; Dp and Op are
// multiples of 16, the arena rows 64-byte aligned; the row itself is staged with 16-byte loads: 9.49 -> 8.57 ms at cfg3 (28 GB: 3.3 TB/s).  Measured and not
// kept: sub-net descriptors in LDS + the next sub-net's index loads requested before the current stores (9.07 ms) -- the kernel's time hardly depends on its bytes
// (bf16 arena, 17.8 GB: 7.7 ms) or on its instruction count; it runs once per fit.
template <typename XT>
__global__ __launch_bounds__(512) void k_gather_lds(const SubnetDev* __restrict__ sn, const float* __restrict__ norm,
                                                    int64_t n, int64_t g, const int32_t* __restrict__ pred,
                                                    const int64_t* __restrict__ pred_off, const int32_t* __restrict__ targ,
                                                    XT* __restrict__ X, float* __restrict__ Y, Dims dm, int with_targets, int64_t row0, int64_t n_all) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const float* row = norm + i * g;
        __syncthreads();                                   // previous row fully consumed
        if ((g & 3) == 0 && (((uintptr_t)norm) & 15) == 0) {
            for (int64_t c = 4 * threadIdx.x; c < g; c += 2048) *(f32x4*)(rowbuf + c) = *(const f32x4*)(row + c);
        } else {
            for (int64_t c = threadIdx.x; c < g; c += 512) rowbuf[c] = row[c];
        }
        __syncthreads();
        for (int k = 0; k < dm.K; ++k) {
            const SubnetDev s = sn[k];
            const int32_t* pk = pred + pred_off[k];
            XT* xr = X + s.xoff + (row0 + i) * s.Dp;
            for (int d0 = 4 * threadIdx.x; d0 < s.Dp; d0 += 2048) {
                int32_t ix[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) ix[j] = pk[d0 + j < s.D ? d0 + j : s.D - 1];           // (the list has D entries: the padding columns re-read the last one)
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = d0 + j < s.D ? rowbuf[ix[j]] : 0.f;
                x_store4<XT>(xr + d0, v);                                                          // written once, read much later
            }
            if (with_targets) {
                float* yr = Y + ((int64_t)k * n_all + row0 + i) * dm.Op;
                const int32_t* tk = targ + (int64_t)k * dm.O;
                for (int o0 = 4 * threadIdx.x; o0 < dm.Op; o0 += 2048) {
                    int32_t ix[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) ix[j] = tk[o0 + j < dm.O ? o0 + j : dm.O - 1];
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = o0 + j < dm.O ? rowbuf[ix[j]] : 0.f;
                    __builtin_nontemporal_store(v, (f32x4*)(yr + o0));
                }
            }
        }
    }
}

// Glorot-uniform init straight into the blocked layouts; element index = Keras row-major.
__global__ __launch_bounds__(256) void k_init_weights(const SubnetDev* __restrict__ sn, float* __restrict__ W1,
                                                      float* __restrict__ W2, Dims dm, uint64_t seed) {
    const int k = blockIdx.y;
    const SubnetDev s = sn[k];
    const int64_t n1 = (int64_t)s.D * dm.H, n2 = (int64_t)dm.H * dm.O;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n1 + n2; e += (int64_t)gridDim.x * 256) {
        if (e < n1) {
            const int d = (int)(e / dm.H), h = (int)(e % dm.H);
            W1[s.w1off + ((int64_t)(d >> 4) * dm.Hp + h) * 16 + (d & 15)] =
                dimn_init_value(seed, (uint32_t)s.kg, 0u, (uint32_t)e, s.lim1);
        } else {
            const int64_t e2 = e - n1;
            const int h = (int)(e2 / dm.O), o = (int)(e2 % dm.O);
            W2[(int64_t)k * dm.Hp * dm.Op + ((int64_t)(h >> 4) * dm.OT + (o >> 4)) * 256 + (h & 15) * 16 + (o & 15)] =
                dimn_init_value(seed, (uint32_t)s.kg, 1u, (uint32_t)e2, s.lim2);
        }
    }
}

// ---------------------------------------------------------------------------------------
// F1: split-K first layer.  Workgroup = (sub-net k, chunk range [c0,c1)); 4 waves, wave w
// owns hidden tiles [w*NT, w*NT+NT).  P[slot][b][h] = sum_{d in chunks} X[rows[b]][d] W1[d][h]
// ---------------------------------------------------------------------------------------
template <int NT, typename XT>
__global__ __launch_bounds__(256) void k_fwd1(const Work* __restrict__ work, const SubnetDev* __restrict__ sn,
                                              const XT* __restrict__ X, const float* __restrict__ W1,
                                              const int32_t* __restrict__ rows, int b_act,
                                              float* __restrict__ P, Dims dm) {
    const int nwg_ = gridDim.x, xq_ = nwg_ >> 3, xr_ = nwg_ & 7, xcd_ = blockIdx.x & 7;        // a sub-net's D-slices on one XCD (see the ring kernel)
    const Work wk = work[xcd_ * xq_ + (xcd_ < xr_ ? xcd_ : xr_) + (blockIdx.x >> 3)];
    const SubnetDev s = sn[wk.k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int nt0 = wave * NT;

    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const XT* xk = X + s.xoff;
    uint32_t xo[4];
    bool valid[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int b = 16 * mt + li;
        valid[mt] = b < b_act;
        xo[mt] = valid[mt] ? (uint32_t)rows[b] * (uint32_t)s.Dp + 4u * lj : 0u;
    }
    const float* wb = W1 + s.w1off + (int64_t)(16 * nt0 + li) * 16 + 4 * lj;
    const int64_t cstride = (int64_t)dm.Hp * 16;

    for (int c = wk.c0; c < wk.c1; ++c) {
        f32x4 a[4], b[NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            { XRaw<XT> xr_; xr_.load(xk + xo[mt] + 16 * c); a[mt] = valid[mt] ? xr_.get() : (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            b[nt] = (nt0 + nt < dm.HT) ? *(const f32x4*)(wb + c * cstride + nt * 256) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA16(a[mt][r], b[nt][r], acc[mt][nt]);
    }
    float* p = P + (int64_t)wk.slot * DIMN_TB * dm.Hp;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            if (nt0 + nt < dm.HT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p[(16 * mt + 4 * lj + r) * dm.Hp + 16 * (nt0 + nt) + li] = acc[mt][nt][r];
            }
}

// ---------------------------------------------------------------------------------------
// RED: reduce the split-K partials ONCE per sub-net:  A = b1 + sum_slices P ; relu ; dropout
// -> Dd[k][64][Hp].  grid (ceil(64*Hp/1024), K); each thread owns 4 consecutive hidden units.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_act(const SubnetDev* __restrict__ sn, const float* __restrict__ P,
                                                    const float* __restrict__ b1, const uint8_t* __restrict__ mask,
                                                    float* __restrict__ Dd, Dims dm, int b_act, float rate, float scale,
                                                    uint64_t seed, uint32_t epoch_key, uint32_t step_key, int k0,
                                                    int act, float* __restrict__ G) {   // act != relu: G[k][64][Hp] = f'(A) * keep * scale
    const int k = blockIdx.y + k0;
    const SubnetDev s = sn[k];
    const int Hp = dm.Hp;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= DIMN_TB * Hp) return;
    const int b = e / Hp, h = e - b * Hp;
    const float* pk = P + (int64_t)s.slot0 * DIMN_TB * Hp + e;
    const int64_t pstride = (int64_t)DIMN_TB * Hp;
    f32x4 a = *(const f32x4*)(b1 + (int64_t)k * Hp + h);
    {   // S independent 16-byte loads, eight in flight at a time (a dependent add per load would
        // serialise S memory latencies)
        int sl = 0;
        for (; sl + 8 <= s.nslice; sl += 8) {
            f32x4 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = *(const f32x4*)(pk + (sl + i) * pstride);
#pragma unroll
            for (int i = 0; i < 8; ++i) a += t[i];
        }
        for (; sl < s.nslice; ++sl) a += *(const f32x4*)(pk + sl * pstride);
    }
    bool keep[4];
    if (mask) {
#pragma unroll
        for (int r = 0; r < 4; ++r) keep[r] = mask[((int64_t)k * DIMN_TB + b) * Hp + h + r] != 0;
    } else if (rate > 0.f) {
        if ((dm.H & 3) == 0) {
            const dimn_u32x4 rnd = dimn_dropout_block(seed, (uint32_t)s.kg, epoch_key, step_key, (uint32_t)(b * dm.H + h) >> 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) keep[r] = dimn_u01(rnd.v[r]) >= rate;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                keep[r] = (h + r < dm.H) && dimn_dropout_keep(seed, (uint32_t)s.kg, epoch_key, step_key, (uint32_t)(b * dm.H + h + r), rate);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) keep[r] = true;
    }
    f32x4 dd;
    if (act == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float relu = a[r] > 0.f ? a[r] : 0.f;
            dd[r] = (keep[r] && b < b_act) ? relu * scale : 0.f;
        }
    } else {
        f32x4 gg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float f, df;
            hidden_act(act, a[r], f, df);
            const bool on = keep[r] && b < b_act && (h + r) < dm.H;      // padded hidden units stay exactly zero
            dd[r] = on ? f * scale : 0.f;
            gg[r] = on ? df * scale : 0.f;
        }
        *(f32x4*)(G + (int64_t)k * DIMN_TB * Hp + e) = gg;
    }
    *(f32x4*)(Dd + (int64_t)k * DIMN_TB * Hp + e) = dd;
}

// (x, y) of a 2-D grid whose workgroups of one y should share an XCD (and its L2): workgroup number b = x + gridDim.x * y is dispatched to XCD b % 8, so
// the workgroups that land on XCD c take the c-th eighth of the (y-major) list -- the mapping of the work tables of B1F1 and k_mid_pipe.  Round 5 (PMC): the 10
// workgroups of a sub-net in k_mid_bwd sat on 8 XCDs and each fetched the sub-net's 128 KB dZ block from memory -- 47 MB of 126 MB fetched per launch at hidden 300.
__device__ __forceinline__ void xcd_grid_remap(int& bx, int& by) {
    const int total = (int)(gridDim.x * gridDim.y), lin = (int)(blockIdx.x + gridDim.x * blockIdx.y);
    const int q = total >> 3, r = total & 7, xcd = lin & 7;
    const int idx = xcd * q + (xcd < r ? xcd : r) + (lin >> 3);
    by = idx / (int)gridDim.x;
    bx = idx - by * (int)gridDim.x;
}

// ---------------------------------------------------------------------------------------
// MF: middle forward.  Workgroup = (sub-net k, output slice os of 64 columns), 8 waves: wave w
// owns output tile ot = 4*os + (w&3) and the batch rows [32*(w>>2), +32) (two MFMA row tiles).  NTW = 6 (12 waves,
// hidden 300): 40 sub-nets x 6 slices = 240 workgroups run in ONE round of the 256 CUs where 8 slices of 4 tiles
// (320 workgroups at one per CU: the Dd image takes 82 KB of LDS) ran in two.
//  a) Dd[64][Hp] (from k_reduce_act) -> LDS
//  b) Z[:,slice] = Dd W2[:,slice] + b2 ; yhat = softplus ; wMSE ; dZ ; gb2 -> Adam(b2)
// The kernel is latency-bound (one workgroup's serial chain), so the chain is kept short: with
// HTC > 0 (compile-time hidden-tile count) every global load a wave needs -- its W2 column block,
// the targets Y of its batch rows, the bias -- is issued BEFORE the LDS staging; eight waves halve
// the MFMA chain and the transcendental epilogue, which runs on the hardware exp/log/rcp units.
// ---------------------------------------------------------------------------------------
template <int HTC, int NTW = 4>     // NTW output tiles per workgroup = 2 NTW waves (two batch-row halves per tile)
__global__ __launch_bounds__(128 * NTW) void k_mid_fwd(const float* __restrict__ W2,
                                                 float* __restrict__ b2w, float* __restrict__ b2m, float* __restrict__ b2v,
                                                 const float* __restrict__ Y, int64_t n_cells,
                                                 const int32_t* __restrict__ rows, int b_act,
                                                 const float* __restrict__ Dd,
                                                 float* __restrict__ dZ, float* __restrict__ loss_step,
                                                 double* __restrict__ loss_acc, Dims dm, AdamP ap, float inv_n, int loss_binary, int k0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int os, ky;
    xcd_grid_remap(os, ky);                             // the slices of a sub-net on one XCD: its Dd block and the batch rows of Y come from memory once
    const int k = ky + k0;
    const int Hp = dm.Hp, ldd = dm.ldd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int ot = os * NTW + (NTW == 4 ? (wave & 3) : wave % NTW), mh = NTW == 4 ? (wave >> 2) : wave / NTW;
    const bool act = ot < dm.OT;
    const int otc = act ? ot : 0;                       // clamped tile: loads stay in bounds, results are dropped
    const int o = 16 * otc + li;
    const int HT = HTC > 0 ? HTC : dm.HT;

    // ---- loads that depend on nothing: W2 operands, targets, bias ----
    const float* w2 = W2 + (int64_t)k * Hp * dm.Op + (int64_t)otc * 256 + lj * 16 + li;
    float bvr[HTC > 0 ? HTC : 1][4];
    if (HTC > 0) {
#pragma unroll
        for (int ht = 0; ht < HTC; ++ht)
#pragma unroll
            for (int q = 0; q < 4; ++q) bvr[ht][q] = w2[(int64_t)ht * dm.OT * 256 + q * 64];   // W2[h=16ht+4q+lj][o]
    }
    float yv[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = 32 * mh + 16 * mt + 4 * lj + r;
            yv[mt][r] = Y[((int64_t)k * n_cells + rows[b < b_act ? b : 0]) * dm.Op + o];
        }
    const float bias = b2w[(int64_t)k * dm.Op + o];

    // ---- a) stage Dd[64][Hp] into LDS (row stride ldd = 2 mod 32 words: conflict-free column reads) ----
    const float* ddk = Dd + (int64_t)k * DIMN_TB * Hp;
    constexpr int PASS = 128 * NTW * 4;                     // floats one pass of the workgroup moves
    for (int e0 = threadIdx.x * 4; e0 < DIMN_TB * Hp; e0 += 8 * PASS) {     // eight independent 16-byte loads in flight
        f32x4 dd[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = e0 + i * PASS;
            dd[i] = *(const f32x4*)(ddk + (e < DIMN_TB * Hp ? e : 0));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = e0 + i * PASS;
            if (e < DIMN_TB * Hp) {
                const int b = e / Hp, h = e - b * Hp;
                *(float2*)(lds + b * ldd + h) = make_float2(dd[i][0], dd[i][1]);
                *(float2*)(lds + b * ldd + h + 2) = make_float2(dd[i][2], dd[i][3]);
            }
        }
    }
    __syncthreads();

    // ---- b) second layer: 32 batch rows x 16 outputs per wave ----
    f32x4 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow = lds + (32 * mh + li) * ldd + lj;
    if (HTC > 0) {
#pragma unroll
        for (int ht = 0; ht < HTC; ++ht)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = MFMA16(arow[16 * mt * ldd + 16 * ht + 4 * q], bvr[ht][q], acc[mt]);
    } else {
        for (int ht = 0; ht < HT; ++ht) {
            const float* wt = w2 + (int64_t)ht * dm.OT * 256;
            float bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = wt[q * 64];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = MFMA16(arow[16 * mt * ldd + 16 * ht + 4 * q], bv[q], acc[mt]);
        }
    }
    float lsum = 0.f, gb = 0.f;
    const bool col_ok = act && (16 * ot + li) < dm.O;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = 32 * mh + 16 * mt + 4 * lj + r;
            float dz = 0.f;
            if (b < b_act && col_ok) {
                const float z = acc[mt][r] + bias;
                const float y = yv[mt][r];
                const float w = loss_binary ? (y > 0.f ? 1.f : 0.f) : y;   // multinet.py:37-40
                float sp, sg;
                softplus_sigmoid_fast(z, sp, sg);
                const float e = y - sp;
                lsum += w * e * e;
                dz = -2.f * w * e * inv_n * sg;
            }
            if (act) dZ[((int64_t)k * DIMN_TB + b) * dm.Op + o] = dz;
            gb += dz;
        }
    gb += __shfl_xor(gb, 16);
    gb += __shfl_xor(gb, 32);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    __syncthreads();                                    // everybody is done reading Dd from LDS
    if (lj == 0) lds[16 + wave * 16 + li] = gb;         // per-wave bias-gradient partials (32 rows each)
    if (lane == 0) lds[wave] = lsum;
    __syncthreads();
    if (act && mh == 0 && lj == 0) {
        const int64_t i = (int64_t)k * dm.Op + o;
        float w = bias, m = b2m[i], v = b2v[i];
        adam1(w, m, v, lds[16 + wave * 16 + li] + lds[16 + (wave + NTW) * 16 + li], ap);
        b2w[i] = w; b2m[i] = m; b2v[i] = v;
    }
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int wv = 0; wv < 2 * NTW; ++wv) tot += lds[wv];
        loss_step[k * dm.LS + os] = tot;
        if (loss_acc) loss_acc[k * dm.LS + os] += (double)tot;
    }
}

// ---------------------------------------------------------------------------------------
// MB: middle backward.  Workgroup = (sub-net k, NH hidden tiles = 16*NH rows of W2), WV waves;
// wave w owns output tiles [w*otw, w*otw+otw).
//  gW2^T tile = dZ^T Dd  (K = batch)  -> Adam on W2/m/v (tile-blocked, 1 KiB/tile)
//  dD[:,16*NH] = dZ W2^T (OLD W2, K = O split over the WV waves, reduced through LDS)
//  dA = dD * scale * [Dd>0] ; gb1 -> Adam(b1)
// Each wave stages the dZ tile [64 b][16 o] of its current output tile into a PRIVATE 4 KiB LDS
// region (4 x 16-byte loads per lane): the linear image serves both operand forms -- dZ^T
// (ds_read_b32, lane-linear) and dZ rows (ds_read_b128) -- without 64-byte strided gathers.
// State and dZ of the next output tile are prefetched while the current one computes.
// ---------------------------------------------------------------------------------------
template <bool FULL, int NH, int WV, int WPS = (WV == 4 ? 3 : (WV == 16 ? 4 : 2))>   // NH hidden tiles (16 rows of W2 each) per workgroup, WV waves; FULL: HT % NH == 0 and OT == WV*otw; WPS: waves per SIMD to fit
__global__ __launch_bounds__(WV * 64, WPS) void k_mid_bwd(const float* __restrict__ Dd, const float* __restrict__ dZ,
                                                 float* __restrict__ W2, float* __restrict__ M2, float* __restrict__ V2,
                                                 float* __restrict__ b1w, float* __restrict__ b1m, float* __restrict__ b1v,
                                                 float* __restrict__ dA, Dims dm, AdamP ap, float scale, int otw, int k0,
                                                 const float* __restrict__ G) {   // G != NULL: dA = dD * G (activation other than relu)
    constexpr int LDR = 16 * NH + 1;                             // padded row of the reduction buffer
    constexpr int RED = (WV * DIMN_TB * LDR) > WV * 1024 ? (WV * DIMN_TB * LDR) : WV * 1024;                       // floats: cross-wave dD reduction buffer
    __shared__ __attribute__((aligned(16))) float lds[RED];     // first WV x 1024 floats double as the dZ tiles
    int hs, ky;
    xcd_grid_remap(hs, ky);                             // the hidden tiles of a sub-net on one XCD: its dZ and Dd blocks come from memory once
    const int k = ky + k0;
    const int Hp = dm.Hp, Op = dm.Op;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int ht0 = NH * hs;
    const int nht = FULL ? NH : ((dm.HT - ht0) < NH ? (dm.HT - ht0) : NH);
    const float* ddk = Dd + (int64_t)k * DIMN_TB * Hp;
    const float* dzk = dZ + (int64_t)k * DIMN_TB * Op;
    float* tile = lds + wave * 1024;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ot_beg = wave * otw;
    const int ot_end = FULL ? ot_beg + otw : ((wave + 1) * otw < dm.OT ? (wave + 1) * otw : dm.OT);
    const int ot_last = ot_end - 1;

    // epilogue operands requested up front (they do not depend on the tile loop): the Dd values that gate
    // dA = dD*scale*[Dd>0] for this thread's outputs, and the bias state for Adam(b1)
    // 64 x (16*NH) outputs: thread -> column hh = tid % (16*NH), rows b = tid / (16*NH) + RB*i
    constexpr int CW = 16 * NH, RB = (WV * 64) / CW, NI = DIMN_TB / RB;
    const int hh = tid % CW, b0 = tid / CW;
    const int h = CW * hs + hh;
    const int hcl = h < Hp ? h : 0;
    float ddv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) ddv[i] = (G ? G + (int64_t)k * DIMN_TB * Hp : ddk)[(b0 + RB * i) * Hp + hcl];
    const int64_t bidx = (int64_t)k * Hp + (tid < CW ? hcl : 0);
    float b1w0 = b1w[bidx], b1m0 = b1m[bidx], b1v0 = b1v[bidx];

    float ddf[16][NH];    // B operand of gW2: Dd[b=4kb+lj][h=16(ht0+ht)+li] (second tile clamped when absent)
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) ddf[kb][ht] = ddk[(4 * kb + lj) * Hp + 16 * (ht0 + (ht < nht ? ht : 0)) + li];

    f32x4 dacc[4][NH];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) dacc[mt][ht] = zero4;

    const int64_t tbase = (int64_t)k * Hp * Op + li * 16 + 4 * lj;
    auto tidx = [&](int ht, int ot) { return tbase + ((int64_t)(ht0 + (ht < nht ? ht : 0)) * dm.OT + ot) * 256; };
    // dZ tile staging: pass i moves row 16i + lane/4, quarter lane%4
    const float* zsrc = dzk + (lane >> 2) * Op + 4 * (lane & 3);

    // Three NAMED register sets (state of one output tile + its dZ tile) rotate through the loop
    // without copies: the loads of tile ot+2 are issued while tile ot computes, and no wait is ever
    // placed on a load issued in the same iteration (a register copy would force exactly that).
    struct Set { f32x4 w[NH], m[NH], v[NH], zt[4]; };
    auto fetch = [&](Set& st, int ot) {
        const int oc = ot < ot_last ? ot : ot_last;          // clamped: tail prefetches re-read the last tile
#pragma unroll
        for (int i = 0; i < 4; ++i) st.zt[i] = *(const f32x4*)(zsrc + 16 * i * Op + 16 * oc);
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) { const int64_t i = tidx(ht, oc); st.w[ht] = *(const f32x4*)(W2 + i); st.m[ht] = DIMN_LD_MV(M2 + i); st.v[ht] = DIMN_LD_MV(V2 + i); }
    };
    auto step = [&](Set& cur, Set& nx2, int ot) {
        // stage this tile (wave-private, in-order LDS), then request tile ot+2
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(tile + 256 * i + 4 * lane) = cur.zt[i];
        if (WV < 16) fetch(nx2, ot + 2);                   // 16 waves: two tiles per wave, both requested up front
        __builtin_amdgcn_sched_barrier(0);
        f32x4 g[NH];
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) g[ht] = zero4;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float az = tile[64 * kb + lane];                       // dZ^T[o=li][b=4kb+lj]
#pragma unroll
            for (int ht = 0; ht < NH; ++ht) g[ht] = MFMA16(az, ddf[kb][ht], g[ht]);
        }
        f32x4 zf[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) zf[mt] = *(const f32x4*)(tile + (16 * mt + li) * 16 + 4 * lj);   // dZ[b][o=4lj+r]
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) dacc[mt][ht] = MFMA16(zf[mt][r], cur.w[ht][r], dacc[mt][ht]);   // OLD W2
            adam4(cur.w[ht], cur.m[ht], cur.v[ht], g[ht], ap);
            if (FULL || ht < nht) {
                const int64_t i = tidx(ht, ot);
                DIMN_ST_STATE(W2 + i, cur.w[ht]); DIMN_ST_STATE(M2 + i, cur.m[ht]); DIMN_ST_STATE(V2 + i, cur.v[ht]);
            }
        }
    };
    if (ot_beg < ot_end) {
        Set A, B, C;
        fetch(A, ot_beg);
        fetch(B, ot_beg + 1);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb)
#pragma unroll
            for (int ht = 0; ht < NH; ++ht) asm volatile("" : "+v"(ddf[kb][ht]));
#pragma unroll
        for (int ht = 0; ht < NH; ++ht) asm volatile("" : "+v"(A.w[ht]), "+v"(A.m[ht]), "+v"(A.v[ht]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(A.zt[i]));
        int ot = ot_beg;
        for (; ot + 3 <= ot_end; ot += 3) {
            step(A, C, ot);
            step(B, A, ot + 1);
            step(C, B, ot + 2);
        }
        if (ot < ot_end) {
            step(A, C, ot);
            if (ot + 1 < ot_end) step(B, A, ot + 1);
        }
    }
    __syncthreads();                                   // every wave is done with its private tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ht = 0; ht < NH; ++ht)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[(wave * DIMN_TB + 16 * mt + 4 * lj + r) * LDR + 16 * ht + li] = (FULL || ht < nht) ? dacc[mt][ht][r] : 0.f;
    __syncthreads();
    float gsum = 0.f;
    if (h < Hp) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int b = b0 + RB * i;
            float d = 0.f;
#pragma unroll
            for (int wv = 0; wv < WV; ++wv) d += lds[(wv * DIMN_TB + b) * LDR + hh];
            const float da = G ? d * ddv[i] : (ddv[i] > 0.f ? d * scale : 0.f);
            dA[((int64_t)k * DIMN_TB + b) * Hp + h] = da;
            gsum += da;
        }
    }
    __syncthreads();
    lds[b0 * LDR + hh] = gsum;
    __syncthreads();
    if (tid < CW && h < Hp) {
        float gb = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i) gb += lds[i * LDR + hh];
        adam1(b1w0, b1m0, b1v0, gb, ap);
        b1w[bidx] = b1w0; b1m[bidx] = b1m0; b1v[bidx] = b1v0;
    }
}

// ---------------------------------------------------------------------------------------
// The fused second layer (H = 256): workgroup = (sub-net k, output tiles [ot0, ot1)), T = ot1 - ot0 <= 8, one workgroup per CU; the kernel
// is k_mid_pipe (dimn_mid_pipe.h).  (Rounds 2-4 ran it as three workgroup-wide phases -- k_mid_fused: load burst, forward over the slice,
// backward over the slice; 39.9 us against the pipeline's 36.1 in the cfg3 step -- retired in round 6; profiles/HISTORY.md section 2.)
// ---------------------------------------------------------------------------------------
struct MidWork { int32_t k, ot0, ot1, slot, sidx; };   // slot: P2 partial; sidx: slice number inside the sub-net (loss slot)
#define DIMN_MID_TMAX 8

// RED2: dA = (sum over the sub-net's slices of the dD partials) * scale * [Dd > 0] ; gb1 -> Adam(b1).
// grid (ceil(Hp/64), K), 1024 threads: thread -> hidden unit h, 4 batch rows (a latency-bound kernel:
// 16 waves per workgroup keep ~24 independent loads per thread in flight).
__global__ __launch_bounds__(1024) void k_reduce_dd(const int32_t* __restrict__ midk, const float* __restrict__ P2,
                                                    const float* __restrict__ Dd,
                                                    float* __restrict__ b1w, float* __restrict__ b1m, float* __restrict__ b1v,
                                                    float* __restrict__ dA, Dims dm, AdamP ap, float scale, int k0,
                                                    const float* __restrict__ G) {   // G != NULL: dA = dD * G
    __shared__ float gs[16][64];
    const int k = blockIdx.y + k0, Hp = dm.Hp;
    const int tid = threadIdx.x, hh = tid & 63, rg = tid >> 6;
    const int h = 64 * blockIdx.x + hh;
    const int slot0 = midk[2 * k], ns = midk[2 * k + 1];
    float gsum = 0.f;
    float b1w0 = 0.f, b1m0 = 0.f, b1v0 = 0.f;
    const int64_t bi = (int64_t)k * Hp + h;
    if (tid < 64 && h < Hp) { b1w0 = b1w[bi]; b1m0 = b1m[bi]; b1v0 = b1v[bi]; }
    if (h < Hp) {
        const float* p = P2 + ((int64_t)slot0 * DIMN_TB + 4 * rg) * Hp + h;
        const float* dd = (G ? G : Dd) + ((int64_t)k * DIMN_TB + 4 * rg) * Hp + h;
        float* da = dA + ((int64_t)k * DIMN_TB + 4 * rg) * Hp + h;
        float d[4], gate[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { d[i] = 0.f; gate[i] = dd[i * Hp]; }
        int sl = 0;
        for (; sl + 2 <= ns; sl += 2) {                      // slices in ascending order: the sum is order-exact
            float t0[4], t1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { t0[i] = p[((int64_t)sl * DIMN_TB + i) * Hp]; t1[i] = p[((int64_t)(sl + 1) * DIMN_TB + i) * Hp]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = (d[i] + t0[i]) + t1[i];
        }
        for (; sl < ns; ++sl)
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] += p[((int64_t)sl * DIMN_TB + i) * Hp];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = G ? d[i] * gate[i] : (gate[i] > 0.f ? d[i] * scale : 0.f);
            da[i * Hp] = v;
            gsum += v;
        }
    }
    gs[rg][hh] = gsum;
    __syncthreads();
    if (tid < 64 && h < Hp) {
        float gb = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) gb += gs[i][hh];
        adam1(b1w0, b1m0, b1v0, gb, ap);
        b1w[bi] = b1w0; b1m[bi] = b1m0; b1v[bi] = b1v0;
    }
}

// ---------------------------------------------------------------------------------------
// B1F1: first-layer weight gradient + fused Keras-Adam + NEXT step's split-K forward.
// 512 threads = 8 INDEPENDENT waves (wave w owns hidden tiles [w*NT2, w*NT2+NT2)); one
// workgroup per CU; no barrier anywhere in the kernel.
// Per 16-row chunk of W1:  g = X_t^T dA (K = batch) -> Adam on the lane's float4 of W1/m/v
// -> the freshly updated W1 registers are the B operand of  P += X_{t+1}[:,chunk] W1new
// (k-slot trick), so W1 is read ONCE per optimiser step (24 B/param instead of 28) and the
// separate forward launch disappears.
// Each wave stages the X_t / X_{t+1} chunk tiles it needs into its OWN LDS region (the tile
// lines are shared through L1/L2); waves therefore drift apart freely, one wave's MFMAs
// overlap another's Adam VALU work and memory waits (measured: a per-chunk workgroup barrier
// cost ~2x).  State and X tiles of chunk c+1 are prefetched into registers while chunk c
// computes.
// ---------------------------------------------------------------------------------------
template <int NT2, bool FULL, typename XT>   // FULL: every wave's NT2 tiles exist (HT == 8*NT2) -> no predicated memory ops in the loop
__global__ __launch_bounds__(512) void k_w1_update_fwd(const Work* __restrict__ work, const SubnetDev* __restrict__ sn,
                                                       const XT* __restrict__ X, float* __restrict__ W1,
                                                       float* __restrict__ M1, float* __restrict__ V1,
                                                       const int32_t* __restrict__ rows_t, int b_act,
                                                       const int32_t* __restrict__ rows_n, int b_next,
                                                       const float* __restrict__ dA, float* __restrict__ P, Dims dm, AdamP ap) {
    constexpr int XTS = DIMN_TB * 16, XN = DIMN_TB * 20;         // floats per staged tile
    constexpr int WSZ = 2 * (XTS + XN);                          // floats of LDS per wave
    __shared__ __attribute__((aligned(16))) float sm_all[8 * WSZ];
    const int nwg_ = gridDim.x, xq_ = nwg_ >> 3, xr_ = nwg_ & 7, xcd_ = blockIdx.x & 7;        // a sub-net's D-slices on one XCD (see the ring kernel)
    const Work wk = work[xcd_ * xq_ + (xcd_ < xr_ ? xcd_ : xr_) + (blockIdx.x >> 3)];
    const SubnetDev s = sn[wk.k];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int nt0 = wave * NT2;
    const int Hp = dm.Hp;
    float* sm = sm_all + wave * WSZ;                             // [xt0 | xt1 | xn0 | xn1]
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Tiles beyond HT (ragged H) are handled without branches around loads: their addresses are
    // clamped to tile 0 (always valid), their results are never stored.
    bool on[NT2];
    int tcl[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) { on[nt] = FULL || (nt0 + nt < dm.HT); tcl[nt] = on[nt] ? nt0 + nt : 0; }

    float bfr[16][NT2];   // dA[b=4kb+lj][h=16*tile+li]; rows >= b_act are zero
    const float* dak = dA + (int64_t)wk.k * DIMN_TB * Hp;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) bfr[kb][nt] = dak[(4 * kb + lj) * Hp + 16 * tcl[nt] + li];

    // staging: pass i (0..3) of this lane moves row 16i + lane/4, 16-byte quarter lane%4
    const XT* xk = X + s.xoff + 4 * (lane & 3);
    uint32_t xot[4], xon[4];
    bool vt[4], vn[4];
    const bool have_next = b_next > 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = 16 * i + (lane >> 2);
        vt[i] = b < b_act;
        vn[i] = b < b_next;
        xot[i] = (uint32_t)rows_t[vt[i] ? b : 0] * (uint32_t)s.Dp;
        xon[i] = have_next ? (uint32_t)rows_n[vn[i] ? b : 0] * (uint32_t)s.Dp : xot[i];
    }
    const int64_t cstride = (int64_t)Hp * 16;
    int64_t wb[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) wb[nt] = s.w1off + (int64_t)(16 * tcl[nt] + li) * 16 + 4 * lj;

    f32x4 pacc[4][NT2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) pacc[mt][nt] = zero4;

    // prologue: state of chunk c0 into registers, X tiles of chunk c0 into LDS buffer 0
    f32x4 w[NT2], m[NT2], v[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        const int64_t i0 = wb[nt] + wk.c0 * cstride;
        w[nt] = *(const f32x4*)(W1 + i0); m[nt] = DIMN_LD_STATE_MV(M1 + i0); v[nt] = DIMN_LD_STATE_MV(V1 + i0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        XRaw<XT> a, b;
        a.load(xk + xot[i] + 16 * wk.c0);
        b.load(xk + xon[i] + 16 * wk.c0);
        *(f32x4*)(sm + 256 * i + 4 * lane) = vt[i] ? a.get() : zero4;
        *(f32x4*)(sm + 2 * XTS + (16 * i + (lane >> 2)) * 20 + 4 * (lane & 3)) = vn[i] ? b.get() : zero4;
    }
    // Retire every prologue load before the loop: otherwise the waitcnt pass, merging the
    // pre-header state into the loop header, drains the in-loop prefetch with vmcnt(0).
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) asm volatile("" : "+v"(bfr[kb][nt]));
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) asm volatile("" : "+v"(w[nt]), "+v"(m[nt]), "+v"(v[nt]));

    const int clast = wk.c1 - 1;
    for (int c = wk.c0; c < wk.c1; ++c) {
        const int cur = (c - wk.c0) & 1;
        const int cn = c < clast ? c + 1 : clast;            // clamped prefetch (the last one is a harmless re-read)
        XRaw<XT> xa[4], xb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xa[i].load(xk + xot[i] + 16 * cn);
            xb[i].load(xk + xon[i] + 16 * cn);
        }
        f32x4 w1[NT2], m1[NT2], v1[NT2];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int64_t idx = wb[nt] + cn * cstride;
            w1[nt] = *(const f32x4*)(W1 + idx); m1[nt] = DIMN_LD_STATE_MV(M1 + idx); v1[nt] = DIMN_LD_STATE_MV(V1 + idx);
        }
        __builtin_amdgcn_sched_barrier(0);                   // keep the prefetch at the top of the iteration

        // gW1 tile: A = X_t^T[d=li][b=4kb+lj] straight out of this wave's linear LDS tile
        const float* xt = sm + cur * XTS;
        f32x4 g[NT2];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) g[nt] = zero4;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float a = xt[64 * kb + lane];
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) g[nt] = MFMA16(a, bfr[kb][nt], g[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) adam4(w[nt], m[nt], v[nt], g[nt], ap);
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            if (FULL || on[nt]) {
                const int64_t idx = wb[nt] + c * cstride;
                DIMN_ST_STATE(W1 + idx, w[nt]); DIMN_ST_STATE(M1 + idx, m[nt]); DIMN_ST_STATE(V1 + idx, v[nt]);
            }
        }
        if (have_next) {
            const float* xn = sm + 2 * XTS + cur * XN;
            f32x4 af[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = *(const f32x4*)(xn + (16 * mt + li) * 20 + 4 * lj);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) pacc[mt][nt] = MFMA16(af[mt][r], w[nt][r], pacc[mt][nt]);
        }
        // stage the tiles of chunk c+1 into the other buffer (wave-private: LDS ops of one wave
        // execute in order, no barrier needed)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(f32x4*)(sm + (cur ^ 1) * XTS + 256 * i + 4 * lane) = vt[i] ? xa[i].get() : zero4;
            *(f32x4*)(sm + 2 * XTS + (cur ^ 1) * XN + (16 * i + (lane >> 2)) * 20 + 4 * (lane & 3)) = vn[i] ? xb[i].get() : zero4;
        }
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) { w[nt] = w1[nt]; m[nt] = m1[nt]; v[nt] = v1[nt]; }
    }
    if (have_next) {
        float* p = P + (int64_t)wk.slot * DIMN_TB * Hp;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
                if (FULL || on[nt]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[(16 * mt + 4 * lj + r) * Hp + 16 * (nt0 + nt) + li] = pacc[mt][nt][r];
                }
    }
}

// ---------------------------------------------------------------------------------------
// B1F1 "shared staging, deep ring": WAVES waves per workgroup, wave w owns NT2 hidden tiles; the X_t / X_{t+1} chunk tiles are staged ONCE per
// workgroup through LDS (one 16-byte load per staging thread) with one barrier per chunk; with 16 waves of <= 128 VGPRs four waves share each
// SIMD, so MFMA, Adam VALU work and memory waits of different waves overlap inside every chunk.  The chunk loop is unrolled three (DEPTH = 4:
// four) times over NAMED register sets (A, B, C), so that rotating the prefetch ring costs no
// register copies -- a copy would force a wait on the load issued in the same iteration.  The
// state of chunk c+2 and the X tile of chunk c+2 are requested while chunk c computes: two
// chunks (~13 KB per wave) stay in flight.
// ---------------------------------------------------------------------------------------
template <int NT2, typename XT>
struct W1Set { f32x4 w[NT2], m[NT2], v[NT2]; XRaw<XT> x; };

template <int WAVES, int NT2, int DEPTH = 3, int WPS = 1, typename XT = float>   // DEPTH named register sets = DEPTH-1 chunks in flight; WPS: waves per SIMD to fit (co-resident workgroups)
__global__ __launch_bounds__(WAVES * 64, WPS) void k_w1_update_fwd_ring(const Work* __restrict__ work, const SubnetDev* __restrict__ sn,
                                                                  const XT* __restrict__ X, float* __restrict__ W1,
                                                                  float* __restrict__ M1, float* __restrict__ V1,
                                                                  const int32_t* __restrict__ rows_t, int b_act,
                                                                  const int32_t* __restrict__ rows_n, int b_next,
                                                                  const float* __restrict__ dA, float* __restrict__ P, Dims dm, AdamP ap) {
    constexpr int XTS = DIMN_TB * 16, XN = DIMN_TB * 20;
    constexpr int DUMMY = 2 * (XTS + XN);                       // LDS words nobody reads: target of non-staging threads
    __shared__ __attribute__((aligned(16))) float sm[2 * (XTS + XN) + 4 * WAVES * 64];
    // the D-slices of one sub-net on ONE XCD (workgroup b runs on XCD b % 8; the table is sub-net-major): they all read the sub-net's dA block
    const int nwg_ = gridDim.x, xq_ = nwg_ >> 3, xr_ = nwg_ & 7, xcd_ = blockIdx.x & 7;
    const Work wk = work[xcd_ * xq_ + (xcd_ < xr_ ? xcd_ : xr_) + (blockIdx.x >> 3)];
    const SubnetDev s = sn[wk.k];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int nt0 = (blockIdx.y * WAVES + wave) * NT2;
    const int Hp = dm.Hp;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    float bfr[16][NT2];
    const float* dak = dA + (int64_t)wk.k * DIMN_TB * Hp;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) bfr[kb][nt] = dak[(4 * kb + lj) * Hp + 16 * (nt0 + nt) + li];

    // Every thread issues exactly one X load per chunk (no predicated VMEM op in the loop, so the
    // compiler can count vmcnt exactly): threads 0..255 move the X_t tile, 256..511 the X_{t+1} tile,
    // the others repeat those loads (L1 hits) and park the result in a dummy LDS word.
    const bool stager = tid < 512;
    const bool stage_next = (tid & 511) >= 256;
    const bool have_next = b_next > 0;
    const int sb = (tid & 255) >> 2, sq = tid & 3;
    const bool svalid = stage_next ? (sb < b_next) : (sb < b_act);
    const int32_t* srows = (stage_next && have_next) ? rows_n : rows_t;
    const XT* xsrc = X + s.xoff + (int64_t)srows[svalid ? sb : 0] * s.Dp + 4 * sq;
    const int sdst0 = stage_next ? (2 * XTS + sb * 20 + 4 * sq) : (sb * 16 + 4 * sq);
    const int sbuf = stager ? (stage_next ? XN : XTS) : 0;
    const int sdst = stager ? sdst0 : DUMMY + 4 * tid;

    const int64_t cstride = (int64_t)Hp * 16;
    int64_t wb[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) wb[nt] = s.w1off + (int64_t)(16 * (nt0 + nt) + li) * 16 + 4 * lj;

    f32x4 pacc[4][NT2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) pacc[mt][nt] = zero4;

    const int clast = wk.c1 - 1;
    auto fetch = [&](W1Set<NT2, XT>& st, int c) {           // issue the loads of chunk c (clamped)
        const int cc = c < clast ? c : clast;
        // X tile first: one iteration later it is the oldest request of this wave, so waiting for it
        // (in-order vmcnt) leaves the three state loads issued after it in flight
        st.x.load(xsrc + 16 * cc);
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int64_t idx = wb[nt] + cc * cstride;
            st.w[nt] = DIMN_LD_STATE(W1 + idx); st.m[nt] = DIMN_LD_STATE_MV(M1 + idx); st.v[nt] = DIMN_LD_STATE_MV(V1 + idx);
        }
    };
    // one chunk: `cur` holds chunk c, `nx1` chunk c+1 (its X tile is staged into LDS here),
    // `nx2` receives the loads of chunk c+2
    auto step = [&](W1Set<NT2, XT>& cur, W1Set<NT2, XT>& nx1, W1Set<NT2, XT>& nx2, int c) {
        const int par = (c - wk.c0) & 1;
        fetch(nx2, c + DEPTH - 1);                            // nx2 = the set that is free again (chunk c-1's)
        __builtin_amdgcn_sched_barrier(0);
        const float* xt = sm + par * XTS;
        f32x4 g[NT2];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) g[nt] = zero4;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float a = xt[64 * kb + lane];
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) g[nt] = MFMA16(a, bfr[kb][nt], g[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) adam4(cur.w[nt], cur.m[nt], cur.v[nt], g[nt], ap);
        *(f32x4*)(sm + sdst + (par ^ 1) * sbuf) = svalid ? nx1.x.get() : zero4;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int64_t idx = wb[nt] + c * cstride;
            DIMN_ST_STATE(W1 + idx, cur.w[nt]); DIMN_ST_STATE(M1 + idx, cur.m[nt]); DIMN_ST_STATE(V1 + idx, cur.v[nt]);
        }
        if (have_next) {
            const float* xn = sm + 2 * XTS + par * XN;
            f32x4 af[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = *(const f32x4*)(xn + (16 * mt + li) * 20 + 4 * lj);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) pacc[mt][nt] = MFMA16(af[mt][r], cur.w[nt][r], pacc[mt][nt]);
        }
        __syncthreads();
    };

    W1Set<NT2, XT> A, B, C, D4;
    fetch(A, wk.c0);
    fetch(B, wk.c0 + 1);
    if (DEPTH == 4) fetch(C, wk.c0 + 2);
    *(f32x4*)(sm + sdst) = svalid ? A.x.get() : zero4;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) asm volatile("" : "+v"(bfr[kb][nt]));
#if !DIMN_W_PEEL
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
        asm volatile("" : "+v"(A.w[nt]), "+v"(A.m[nt]), "+v"(A.v[nt]));
        asm volatile("" : "+v"(B.w[nt]), "+v"(B.m[nt]), "+v"(B.v[nt]));
    }
#endif
#if DIMN_W_PEEL
    __syncthreads();
    int c = wk.c0;
    if (DEPTH == 3 && c + 3 <= wk.c1) {
        // first triple peeled out of the loop: in straight-line code the waits are placed per use, so chunk c0
        // computes while chunk c0+1 is still arriving (a wait for BOTH before the loop cost part of the burst)
        step(A, B, C, c);
        step(B, C, A, c + 1);
        step(C, A, B, c + 2);
        c += 3;
    }
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {                     // what the loop header inherits is retired here
        asm volatile("" : "+v"(A.w[nt]), "+v"(A.m[nt]), "+v"(A.v[nt]));
        asm volatile("" : "+v"(B.w[nt]), "+v"(B.m[nt]), "+v"(B.v[nt]));
    }
    asm volatile("" : "+v"(A.x.v), "+v"(B.x.v));
#else
    asm volatile("" : "+v"(B.x.v));
    __syncthreads();

    int c = wk.c0;
#endif
    if (DEPTH == 4) {
        asm volatile("" : "+v"(C.x.v));
        for (; c + 4 <= wk.c1; c += 4) {                   // full groups: no conditional memory op inside
            step(A, B, D4, c);
            step(B, C, A, c + 1);
            step(C, D4, B, c + 2);
            step(D4, A, C, c + 3);
        }
        if (c < wk.c1) {                                   // 1..3 chunks left
            step(A, B, D4, c);
            if (c + 1 < wk.c1) step(B, C, A, c + 1);
            if (c + 2 < wk.c1) step(C, D4, B, c + 2);
        }
    } else {
        for (; c + 3 <= wk.c1; c += 3) {                   // full triples: no conditional memory op inside
            step(A, B, C, c);
            step(B, C, A, c + 1);
            step(C, A, B, c + 2);
        }
        if (c < wk.c1) {                                   // 1 or 2 chunks left
            step(A, B, C, c);
            if (c + 1 < wk.c1) step(B, C, A, c + 1);
        }
    }
    if (have_next) {
        float* p = P + (int64_t)wk.slot * DIMN_TB * Hp;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) DIMN_ST_P(&p[(16 * mt + 4 * lj + r) * Hp + 16 * (nt0 + nt) + li], pacc[mt][nt][r]);
    }
}

// ---------------------------------------------------------------------------------------
// Fused forward for 64 rows: model.predict (multinet.py:253,278) and the validation pass.
// grid (row tiles, K).  out != NULL: out[i][k*O + o] = softplus(z).  loss_part != NULL:
// loss_part[k*gridDim.x + tile] = sum w*(y-yhat)^2 over the tile (S9).
// Second layer in the k-slot form: one 16-byte LDS read of an activation row feeds four MFMAs, and the W2 operand comes
// from W2T (k_prep_w2t), where the float4 of a lane is W2[h = 16ht + 4lj .. +3][o = 16ot + li] -- one coalesced 1 KB wave
// request per (hidden tile, output tile), prefetched one tile ahead.
// ---------------------------------------------------------------------------------------
// W2T[k][ot][ht][lane][r] = W2 tile (ht, ot) [h = 4*(lane>>4) + r][o = lane & 15]   (the tile-native W2 is [h][o])
__global__ __launch_bounds__(256) void k_prep_w2t(const float* __restrict__ W2, float* __restrict__ W2T, Dims dm) {
    const int k = blockIdx.y;
    const int64_t n2 = (int64_t)dm.Hp * dm.Op;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n2 / 4; e += (int64_t)gridDim.x * 256) {
        const int lane = (int)(e & 63);
        const int64_t tile = e >> 6;                             // = ot * HT + ht
        const int ot = (int)(tile / dm.HT), ht = (int)(tile - (int64_t)ot * dm.HT);
        const float* src = W2 + (int64_t)k * n2 + ((int64_t)ht * dm.OT + ot) * 256 + (4 * (lane >> 4)) * 16 + (lane & 15);
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = src[16 * r];
        *(f32x4*)(W2T + (int64_t)k * n2 + e * 4) = v;
    }
}

#ifdef DIMN_PRED_TL   // tools/predict_timeline.py: phase clocks of every wave, summed over the workgroups of a launch
__device__ unsigned long long g_pred_tl[8];
#define PRED_STAMP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&g_pred_tl[i], t_ - tl_t); tl_t = t_; }
#define PRED_TL_DECL unsigned long long tl_t = __builtin_amdgcn_s_memtime();
#else
#define PRED_STAMP(i)
#define PRED_TL_DECL
#endif
#define DIMN_PRED_XS (3 * DIMN_TB * 16)   // floats of the X staging ring behind the activations in LDS
// 64 rows per workgroup, 4 waves of NT hidden tiles each.  First layer: per 16-deep chunk a wave requests ITS 16 rows of the
// X tile (one 16-byte load per lane) and its NT W1 tiles; the X tile goes through a three-stage LDS ring (one barrier per chunk)
// and every wave reads all four row tiles from there.  Measured on the way here (50k cells x 40 sub-nets): every wave loading
// the whole X tile itself -- 8 vector loads per 64 MFMAs -- ran at 0.69 of the matrix pipe with two or three chunks in flight,
// and just the same with every load an L1 hit: bound by the vector-memory instruction rate of the CU, not by latency or bytes
// (8 waves x 2 hidden tiles, at 64 or 128 rows: 12 loads per 64 MFMAs, 1.6x slower).  With the ring: 0.72; the same loop with
// its loads removed 0.81, without its barrier 0.83 -- the rest is the second layer and the output epilogue (tools/predict_timeline.py).
// MT: 16-row tiles per workgroup -- 4 (64 rows), or 2 / 1 where 64-row tiles would leave most CUs without a workgroup (round 5: the validation pass of a small
// problem -- configs[1]: 250 rows x 10 sub-nets = 40 workgroups, 0.31 ms each epoch, 8 % of the impute; with 16-row tiles 160 workgroups of a quarter of the
// matrix work each; the four waves then stage the same 16 rows, identical values to identical places).
template <int NT, typename XT, int MT = 4>
__global__ __launch_bounds__(256) void k_predict(const SubnetDev* __restrict__ sn, const XT* __restrict__ X,
                                                 const float* __restrict__ W1, const float* __restrict__ b1,
                                                 const float* __restrict__ W2, const float* __restrict__ b2,
                                                 const int32_t* __restrict__ rows, int64_t n_rows,
                                                 float* __restrict__ out, const float* __restrict__ Y, int64_t n_cells,
                                                 float* __restrict__ loss_part, Dims dm, int loss_binary, int act) {
    constexpr int NW = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int k = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * (16 * MT);
    const SubnetDev s = sn[k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int nt0 = wave * NT;
    const int Hp = dm.Hp, ldp = dm.ldp;
    float* xs = lds + 16 * MT * ldp;                         // X staging ring [3][16 MT rows][16]
    PRED_TL_DECL

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this wave's rows of the X tile (rows past n_rows read row 0 and are dropped at the end: every load unconditional)
    const int64_t irow = r0 + 16 * (wave % MT) + li;
    const int64_t xrow = irow < n_rows ? (rows ? (int64_t)rows[irow] : irow) : 0;
    const XT* xk = X + s.xoff + xrow * s.Dp + 4 * lj;
    const int64_t cstride = (int64_t)Hp * 16;
    const float* wbt[NT];                                    // tiles past HT are clamped: loaded, never used
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int t = (nt0 + nt) < dm.HT ? (nt0 + nt) : (dm.HT - 1);
        wbt[nt] = W1 + s.w1off + (int64_t)(16 * t + li) * 16 + 4 * lj;
    }
    struct WSet { f32x4 b[NT]; };
    auto clampc = [&](int c) { return c < s.nchunk ? c : s.nchunk - 1; };
    auto fetch_w = [&](WSet& o, int c) {
        const int cc = clampc(c);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o.b[nt] = *(const f32x4*)(wbt[nt] + cc * cstride);
    };
    XRaw<XT> xr;                                             // the X piece in flight (chunk c + 1 at the top of iteration c)
    float* xw = xs + (16 * (wave % MT) + li) * 16 + 4 * lj;  // where this lane's piece goes in a stage
    const float* xrd = xs + li * 16 + 4 * lj;                // row tile mt of a stage: + 256 * mt
    // iteration c (stage st = c % 3):  X(c+1) -> stage st+1;  request X(c+2), W(c+2);  barrier;  MFMAs of chunk c from stage st, W(c)
    auto step = [&](WSet& wcur, WSet& wnew, int c, int st) {
        *(f32x4*)(xw + ((st + 1) % 3) * (16 * MT * 16)) = xr.get();
        xr.load(xk + 16 * clampc(c + 2));
        fetch_w(wnew, c + 2);
        __builtin_amdgcn_sched_barrier(0);                   // the requests leave before this chunk's MFMAs (hipcc sinks them otherwise)
        __syncthreads();
        f32x4 a4[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a4[mt] = *(const f32x4*)(xrd + st * (16 * MT * 16) + 256 * mt);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA16(a4[mt][r], wcur.b[nt][r], acc[mt][nt]);
        __builtin_amdgcn_sched_barrier(0);
    };
    WSet W0, W1s, W2s;                                       // three named sets: the W1 tiles of two chunks in flight
    xr.load(xk);
    fetch_w(W0, 0);
    *(f32x4*)xw = xr.get();                                  // chunk 0 -> stage 0
    xr.load(xk + 16 * clampc(1));
    fetch_w(W1s, 1);
    int c = 0;
    for (; c + 3 <= s.nchunk; c += 3) {
        step(W0, W2s, c, 0);
        step(W1s, W0, c + 1, 1);
        step(W2s, W1s, c + 2, 2);
    }
    if (c < s.nchunk) step(W0, W2s, c, 0);
    if (c + 1 < s.nchunk) step(W1s, W0, c + 1, 1);
    PRED_STAMP(0)
    // bias + relu -> LDS (dropout is identity at inference, S3/S12)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        if (nt0 + nt < dm.HT) {
            const int h = 16 * (nt0 + nt) + li;
            const float bias = b1[(int64_t)k * Hp + h];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[mt][nt][r] + bias;
                    float f = v > 0.f ? v : 0.f;
                    if (act != 0) { f = hidden_act_call(act, v); if (h >= dm.H) f = 0.f; }     // (out of line: 64 inlined switches were 100 KB of code)
                    lds[(16 * mt + 4 * lj + r) * ldp + h] = f;
                }
        }
    PRED_STAMP(1)
    __syncthreads();
    PRED_STAMP(2)

    float lsum = 0.f;
    const int HT = dm.HT;
    const float* arow = lds + li * ldp + 4 * lj;                 // activations [b = 16mt + li][h = 16ht + 4lj ..]
    for (int ot = wave; ot < dm.OT; ot += NW) {
        f32x4 z[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) z[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* wt = W2 + ((int64_t)k * dm.OT + ot) * HT * 256 + lane * 4;      // W2T: the tiles (ht, ot), ht = 0.., contiguous
        auto tile = [&](const f32x4 bq, int ht) {
            f32x4 a4[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a4[mt] = *(const f32x4*)(arow + 16 * mt * ldp + 16 * ht);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) z[mt] = MFMA16(a4[mt][r], bq[r], z[mt]);
        };
        // two named operand registers, loop unrolled x2: the request of tile ht+1 leaves before tile ht feeds the MFMAs
        f32x4 q0 = *(const f32x4*)wt, q1;
        int ht = 0;
        for (; ht + 2 <= HT; ht += 2) {
            q1 = *(const f32x4*)(wt + (ht + 1) * 256);
            __builtin_amdgcn_sched_barrier(0);
            tile(q0, ht);
            __builtin_amdgcn_sched_barrier(0);
            q0 = *(const f32x4*)(wt + (ht + 2 < HT ? ht + 2 : ht + 1) * 256);
            __builtin_amdgcn_sched_barrier(0);
            tile(q1, ht + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ht < HT) tile(q0, ht);
        PRED_STAMP(3)
        const int o = 16 * ot + li;
        if (o < dm.O) {
            const float bias = b2[(int64_t)k * dm.Op + o];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t i = r0 + 16 * mt + 4 * lj + r;
                    if (i < n_rows) {
                        const float yh = softplus_out(z[mt][r] + bias);
                        if (out) __builtin_nontemporal_store(yh, &out[(i * dm.K + k) * dm.O + o]);
                        if (loss_part) {
                            const int64_t row = rows ? (int64_t)rows[i] : i;
                            const float y = Y[((int64_t)k * n_cells + row) * dm.Op + o];
                            const float w = loss_binary ? (y > 0.f ? 1.f : 0.f) : y;
                            const float e = y - yh;
                            lsum += w * e * e;
                        }
                    }
                }
        }
        PRED_STAMP(4)
    }
    PRED_STAMP(5)
    if (loss_part) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
        __syncthreads();
        if (lane == 0) lds[wave] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) tot += lds[wv];
            loss_part[(int64_t)k * gridDim.x + blockIdx.x] = tot;
        }
    }
}


// ---------------------------------------------------------------------------------------
// predict() post-processing on the device (reference deepimpute/multinet.py:282-305), one workgroup per cell:
//   predicted[gene] = mean over the gene's target slots of the network outputs (float32, slot order; the reference's
//                     groupby(columns).mean()), genes no sub-net predicts keep log1p(raw);
//   > ceiling (= 2 * max log1p(raw)) or NaN -> 0;  expm1 back to counts (float64);
//   policy 1 "restore": observed counts win wherever raw > 0;  2 "max": max(raw, imputed);  0: none.
// The prediction row [S] is staged once in LDS (coalesced) and gathered per gene from there; raw and the result are
// streamed coalesced in float64.  goff[g+1] / gslot[S]: the slots of every output column (CSR, ascending slot order).
// ---------------------------------------------------------------------------------------
template <typename RT>      // raw counts: float64 streamed from the host, or float32 resident on the device (dimn_counts: exact integers)
__global__ __launch_bounds__(512) void k_impute_finish(const float* __restrict__ pred, int64_t S, int64_t pred_row0,
                                                       const RT* __restrict__ raw, int64_t n_rows, int64_t g,
                                                       const int32_t* __restrict__ goff, const int32_t* __restrict__ gslot,
                                                       double ceiling, int policy, int lds_stage, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float prow[];
    for (int64_t i = blockIdx.x; i < n_rows; i += gridDim.x) {
        const float* pr = pred + (pred_row0 + i) * S;
        if (lds_stage) {
            __syncthreads();
            for (int64_t c = threadIdx.x; c < S; c += 512) prow[c] = pr[c];
            __syncthreads();
        }
        const float* src = lds_stage ? prow : pr;
        const RT* rr = raw + i * g;
        double* orow = out + i * g;
        for (int64_t j = threadIdx.x; j < g; j += 512) {
            const double x = (double)rr[j];
            const int s0 = goff[j], s1 = goff[j + 1];
            double v;
            if (s1 > s0) {
                float acc = src[gslot[s0]];
                for (int s = s0 + 1; s < s1; ++s) acc += src[gslot[s]];
                acc /= (float)(s1 - s0);
                v = (double)acc;
            } else {
                v = log1p(x);
            }
            if (v > ceiling || v != v) v = 0.0;
            v = expm1(v);
            if (policy == 1) v = x > 0.0 ? x : v;
            else if (policy == 2) v = x > v ? x : v;
            orow[j] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------
// predict()'s post-processing, policy "restore", counts resident on the device (round 5): only the entries the policy CHANGES cross PCIe.
// `restore` returns the observed count wherever it is positive (multinet.py:296-299) and the caller still holds those counts in its own
// float64 frame, so the device sends, per cell, just the finished values of the ZERO entries, packed in column order; the host copies its
// frame and drops them into the zeros (dimn_impute_finish_restore).  For the 35 % zeros of the bench matrix that is 2.8 GB instead of 8.
//   k_row_zeros:  zeros[i] = number of zero counts in row i            (one wave per row)
//   k_impute_finish_zeros: out[base[i] - base[row0] + p] = finished value of the p-th zero of row i (the arithmetic of k_impute_finish)
// A wave owns a contiguous eighth of the row's columns; its first output position is the number of zeros in the eighths before it, so
// the order inside a row is column order and no barrier sits in the packing loop.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_zeros(const float* __restrict__ counts, int64_t n_rows, int64_t g, int32_t* __restrict__ zeros) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_rows) return;
    const float* rr = counts + i * g;
    int z = 0;
    for (int64_t j = lane; j < g; j += 64) z += rr[j] == 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
    if (lane == 0) zeros[i] = z;
}

__global__ __launch_bounds__(512) void k_impute_finish_zeros(const float* __restrict__ pred, int64_t S, int64_t pred_row0,
                                                             const float* __restrict__ raw, int64_t n_rows, int64_t g,
                                                             const int32_t* __restrict__ goff, const int32_t* __restrict__ gslot,
                                                             const int64_t* __restrict__ base, double ceiling, int lds_stage,
                                                             double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float prow[];
    __shared__ int wz[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c0 = g * wave / 8, c1 = g * (wave + 1) / 8;           // this wave's columns
    for (int64_t i = blockIdx.x; i < n_rows; i += gridDim.x) {
        const float* pr = pred + (pred_row0 + i) * S;
        const float* rr = raw + i * g;
        __syncthreads();
        if (lds_stage) for (int64_t c = threadIdx.x; c < S; c += 512) prow[c] = pr[c];
        int z = 0;
        for (int64_t j = c0 + lane; j < c1; j += 64) z += rr[j] == 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
        if (lane == 0) wz[wave] = z;
        __syncthreads();
        int64_t pos = base[i] - base[0];
        for (int w = 0; w < wave; ++w) pos += wz[w];
        const float* src = lds_stage ? prow : pr;
        double* orow = out + pos;
        for (int64_t jb = c0; jb < c1; jb += 64) {
            const int64_t j = jb + lane;
            const bool zero = j < c1 && rr[j] == 0.0f;
            const unsigned long long m = __ballot(zero);
            if (zero) {
                const int s0 = goff[j], s1 = goff[j + 1];
                double v = 0.0;                                           // log1p(0): a gene no sub-net predicts
                if (s1 > s0) {
                    float acc = src[gslot[s0]];
                    for (int s = s0 + 1; s < s1; ++s) acc += src[gslot[s]];
                    acc /= (float)(s1 - s0);
                    v = (double)acc;
                }
                if (v > ceiling || v != v) v = 0.0;
                orow[__popcll(m & ((1ull << lane) - 1ull))] = expm1(v);
            }
            orow += __popcll(m);
        }
    }
}


// Held-out metrics of fit() (reference deepimpute/multinet.py:251-262): over the validation cells and every target slot
// with a positive observed value, the sums that give Pearson r and the MSE between truth (log1p counts) and prediction.
// pred [n_rows][K*O] (dimn_predict_device over the validation rows), Y the gathered targets.  sums[7] (double,
// zeroed by the caller): count, Sx, Sy, Sxx, Syy, Sxy, S(x-y)^2  with x = truth, y = prediction.
__global__ __launch_bounds__(256) void k_val_metrics(const float* __restrict__ pred, const float* __restrict__ Y, const int32_t* __restrict__ rows,
                                                     int64_t n_rows, int64_t n_cells, Dims dm, double* __restrict__ sums) {
    __shared__ double red[4][7];
    const int64_t S = (int64_t)dm.K * dm.O;
    double a[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = blockIdx.x; i < n_rows; i += gridDim.x) {
        const int64_t row = rows ? rows[i] : i;
        for (int64_t c = threadIdx.x; c < S; c += 256) {
            const int k = (int)(c / dm.O), o = (int)(c - (int64_t)k * dm.O);
            const double x = Y[((int64_t)k * n_cells + row) * dm.Op + o], y = pred[i * S + c];
            if (x > 0.0) { a[0] += 1.0; a[1] += x; a[2] += y; a[3] += x * x; a[4] += y * y; a[5] += x * y; a[6] += (x - y) * (x - y); }
        }
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[q] += __shfl_xor(a[q], off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 7; ++q) red[wave][q] = a[q];
    __syncthreads();
    if (threadIdx.x < 7) atomicAdd(&sums[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}


// ---------------------------------------------------------------------------------------
// Inference on the bf16 matrix cores (precision DIMN_PREC_BF16; BASELINE configs[4]): model.predict / the validation
// pass with v_mfma_f32_16x16x16_bf16 -- operands bfloat16, accumulation fp32.  Operand maps of that instruction (lane l,
// li = l&15, lj = l>>4): A[i = li][k = 4lj..4lj+3], B[k = 4lj..4lj+3][j = li], C/D as the fp32 one (col = li, row =
// 4lj + reg): the four k's of a lane are exactly what the k-slot trick loads as ONE 8-byte piece, so every operand of
// this kernel is a single 8-byte load and one MFMA covers a whole 16-deep chunk (4x fewer matrix instructions than fp32,
// each ~4x faster).  X is the bf16 arena; W1b / W2t are bf16 images of the fp32 master weights made by k_prep_bf16
// before the launch (W1b: chunk-blocked like W1; W2t: [o][h], so a lane's four h's are contiguous); the hidden
// activations are rounded to bf16 when they are staged in LDS.  Biases, softplus, the loss: fp32.
// ---------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_prep_bf16(const SubnetDev* __restrict__ sn, const float* __restrict__ W1, const float* __restrict__ W2,
                                                   bf16_t* __restrict__ W1b, bf16_t* __restrict__ W2t, Dims dm) {
    const int k = blockIdx.y;
    const SubnetDev s = sn[k];
    const int64_t n1 = (int64_t)s.Dp * dm.Hp, n2 = (int64_t)dm.Hp * dm.Op;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n1 + n2; e += (int64_t)gridDim.x * 256) {
        if (e < n1) {
            W1b[s.w1off + e] = f32_to_bf16(W1[s.w1off + e]);
        } else {
            const int64_t e2 = e - n1;                           // output index: [o][h]
            const int o = (int)(e2 / dm.Hp), h = (int)(e2 - (int64_t)o * dm.Hp);
            W2t[(int64_t)k * n2 + e2] = f32_to_bf16(W2[(int64_t)k * n2 + ((int64_t)(h >> 4) * dm.OT + (o >> 4)) * 256 + (h & 15) * 16 + (o & 15)]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// model.predict / the validation pass on the bf16 matrix cores (precision bf16), round 3: 128 rows per workgroup, both GEMMs on
// v_mfma_f32_16x16x32_bf16 (32-deep: half the instructions and operand fetches per flop of the 16-deep form).
//   * 4 waves; wave w owns hidden columns [64w, 64w + 64) of the first layer for all 128 rows (8 x 4 accumulator tiles) and, in
//     the second layer, four output tiles per pass -- 32 matrix instructions per 32-deep step against 8 LDS reads (A) and four
//     16-byte global loads (B, from the bf16 weight images in L2).  Round 2's kernel (64 rows, 16-deep, every operand an 8-byte
//     global load) issued one vector-memory instruction per matrix instruction: 215 TFLOP/s, VMEM-issue bound.
//   * X tile of every step staged ONCE per workgroup through an LDS ring, since round 4 by the LDS DMA in 64-deep steps (below);
//     the hidden activations [128][Hp] are rounded to bf16 into the same LDS (aliasing the ring).
//   * Round 4, found with per-workgroup phase stamps (tools/pb_trace.sh), 5.87 -> 3.97 ms on 50k cells x 40 sub-nets: the kernel was 250 KB
//     of code (the activation switch inlined 128 times) against a 64 KB instruction cache; its output epilogue waited for the previous
//     block's store at every second block and branched per element; its first layer held a barrier every 32 matrix instructions.
//   * A lane's eight k's of an operand are eight CONSECUTIVE k's on both operands -- which of the 32 k's of the instruction they
//     occupy does not matter for a dot product (the k-slot argument of this file), so no layout of the 32-deep instruction is
//     assumed beyond "A and B use the same one".
// W1b: bf16 image of the chunk-blocked W1 ([Dp/16][Hp][16]); W2t: bf16 [Op][Hp] (k_prep_bf16).  loss_part: [K][lp_stride]
// slots of 64-row tiles (the host sums them); a 128-row workgroup fills two.
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8n __attribute__((ext_vector_type(8)));
#define DIMN_PB_M 128
#define DIMN_PB_XST 16384              // bytes of one stage of the X ring: 128 rows x 64 k's
#ifndef DIMN_PB_NT
#define DIMN_PB_NT 0        // plain 16-byte stores: 5.82 vs 6.04 ms with non-temporal ones
#endif
#if DIMN_PB_NT
#define DIMN_PB_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define DIMN_PB_STORE(p, v) (*(p) = (v))
#endif
#ifndef DIMN_PB_WPS
#define DIMN_PB_WPS 2
#endif
// (Rounds 4-5 had diagnostic builds of this kernel -- per-workgroup phase stamps, store / load ablations: that is how its instruction-cache
//  problem was found; their results are profiles/r04_predict_bf16_phases.txt and r05_predict_bf16_ablation.txt.  Retired in round 6.)
template <bool FAST, bool LOSS>
__global__ __launch_bounds__(256, DIMN_PB_WPS) void k_predict_bf16(const SubnetDev* __restrict__ sn, const bf16_t* __restrict__ X,
                                                         const bf16_t* __restrict__ W1b, const float* __restrict__ b1,
                                                         const bf16_t* __restrict__ W2t, const float* __restrict__ b2,
                                                         const int32_t* __restrict__ rows, int64_t n_rows,
                                                         float* __restrict__ out, const float* __restrict__ Y, int64_t n_cells,
                                                         float* __restrict__ loss_part, int64_t lp_stride, Dims dm, int loss_binary, int act,
                                                         const bf16_t* __restrict__ zeros) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the LDS address space below makes the host pass drop the stub silently otherwise)
    extern __shared__ __attribute__((aligned(1024))) unsigned char pl_lds[];                 // X ring: 4 stages of DIMN_PB_XST bytes
    bf16_t* ddl = (bf16_t*)pl_lds;                               // hidden activations [128][ld2] (aliases the ring once the first layer is done)
    const int Hp = dm.Hp, Hq = (Hp + 31) & ~31;                  // hidden width padded to whole 32-deep steps
    const int ld2 = Hq + 8;
    float* redl = (float*)(pl_lds + (size_t)DIMN_PB_M * ld2 * 2);   // [4] loss partials
    const int k = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * DIMN_PB_M;
    const SubnetDev s = sn[k];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    // ---- first layer: A = X W1 over 64-deep steps ----
    // X goes from global memory straight into LDS (global_load_lds: no staging registers, no ds_write) in FRAGMENT ORDER: a stage of the
    // ring is [8 row tiles][2 k-halves] blocks of 1 KB, lane (li, lj) of block (rt, kh) holding row 16 rt + li, k's 64 step + 32 kh + 8 lj ..+7
    // -- the operand of one matrix instruction is ds_read_b128 at lane * 16, conflict-free.  Wave w brings in blocks 4w .. 4w + 3 of every
    // stage.  One barrier per PAIR of steps (128 matrix instructions per wave): with one per 32-deep step the loop took 62 us per workgroup
    // against 38 with the barrier removed (round 4, phase stamps); W1b stays a direct global -> register operand (private to its wave).
    // Chunks past the sub-net's last one are fetched from 1 KB of zeros.
    f32x4 acc[8][4];
#pragma unroll
    for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = zero4;
    const int nsteps = (s.nchunk + 3) >> 2;
    const bf16_t* xsrc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int b = 4 * wave + u, rt = b >> 1, kh = b & 1;
        const int64_t i = r0 + 16 * rt + li;
        const int64_t src = i < n_rows ? (rows ? (int64_t)rows[i] : i) : 0;            // rows past the end read row 0 and are dropped
        xsrc[u] = X + s.xoff + src * s.Dp + 32 * kh + 8 * lj;
    }
    const bf16_t* zp = zeros + 8 * lane;
    // X(sa) -> stage, X(sb) -> stage + 1, requested together: the two steps' pieces of a row are 256 contiguous bytes, asked for back to back
    // (a predictor row is ~5 KB: a visit of 128 bytes per row and step left the HBM pages half used)
    auto xissue2 = [&](int sa, int sb, int stage) {
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int which = 0; which < 2; ++which)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int u = 2 * uu + e, step = which ? sb : sa;
                    const int c = 4 * step + 2 * e + (lj >> 1);
                    const bf16_t* g = c < s.nchunk ? xsrc[u] + 64 * step : zp;
                    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(pl_lds + (stage + which) * DIMN_PB_XST + (4 * wave + u) * 1024), 16, 0, 0);
                }
    };
    // W1 operand of column tile ct, k-half kh of a step: eight consecutive d's of hidden unit h = 64 wave + 16 ct + li from chunk 4 step + 2 kh + (lj >> 1).
    // Address = a wave-uniform chunk base (one of two, chosen by lj >> 1) + ONE 32-bit lane offset + 512 ct as the instruction's immediate:
    // sixteen 64-bit lane addresses per step cost 32 registers this loop does not have.  Not masked: a hidden column beyond Hp only produces
    // accumulators nobody reads (dimn_create leaves 8 KB of zeros behind the image for the reads past the last unit of the last chunk); a
    // chunk past the sub-net's last one is clamped to it and meets the zeros on the X side.
    const unsigned cstride_b = (unsigned)Hp * 32u;               // bytes of one chunk of W1b
    const int hw = 64 * wave + li;
    const unsigned char* wsub = (const unsigned char*)(W1b + s.w1off) + (unsigned)(hw < Hp ? hw : 0) * 32u + 16u * (lj & 1);
    const bool odd_half = (lj >> 1) != 0;
    auto bload = [&](int step, u32x4v (&b)[2][4]) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            int c0 = 4 * step + 2 * kh, c1 = c0 + 1;
            c0 = c0 < s.nchunk ? c0 : s.nchunk - 1;
            c1 = c1 < s.nchunk ? c1 : s.nchunk - 1;
            const unsigned char* wa = wsub + (size_t)(odd_half ? c1 : c0) * cstride_b;
            // (as inline asm, with the wait for them written by hand in one_step: across the loop's back edge the compiler's own counting gives
            //  up and waits for EVERY outstanding request -- the X stage requested a moment ago included -- before the step's first matrix
            //  instruction.  The price: the compiler does not know that these registers are written LATER -- every set that is requested must
            //  stay live until a wait that names it, or its registers are handed to something else while the data is still on its way)
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:512\n\t"
                         "global_load_dwordx4 %2, %4, off offset:1024\n\tglobal_load_dwordx4 %3, %4, off offset:1536"
                         : "=&v"(b[kh][0]), "=&v"(b[kh][1]), "=&v"(b[kh][2]), "=&v"(b[kh][3]) : "v"(wa) : "memory");
        }
    };
    // the 16 operand blocks of a step two at a time, the next pair requested before the eight matrix instructions of this one (left to
    // itself the scheduler hoists LDS reads until the accumulators spill)
    auto mma1 = [&](const unsigned char* st, const u32x4v (&b)[2][4]) {
        auto rd = [&](int j) { return *(const bf16x8n*)(st + (2 * (j & 7) + (j >> 3)) * 1024 + lane * 16); };       // j = 8 kh + rt
        bf16x8n A[2][2];
        A[0][0] = rd(0); A[0][1] = rd(1);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            if (p < 7) { A[(p + 1) & 1][0] = rd(2 * p + 2); A[(p + 1) & 1][1] = rd(2 * p + 3); }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * p + e, kh = j >> 3, rt = j & 7;
#pragma unroll
                // A^T tile = W1b (16 hidden units x 32 d) x X^T (32 d x 16 rows): a lane's accumulator is then four CONSECUTIVE HIDDEN UNITS of one
                // row (acc[rt][ct][r] = A[16 rt + li][64 wave + 16 ct + 4 lj + r]) -- one packed conversion and one 8-byte LDS store per tile below
                for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8n, b[kh][ct]), A[p & 1][e], acc[rt][ct], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        // Steps in pairs (s even, s + 1), ring of four stages, stage of X(t) = t mod 4.  Request queue of a wave at the top of a pair, oldest
        // first: X(s), X(s + 1) x 8 (requested at the top of the pair before), W(s) x 8.  "All but 8 done" = this wave's part of both X stages
        // has landed; the barrier makes that true of every wave's part and says that every wave has finished the pair before, whose stages
        // X(s + 2), X(s + 3) then overwrite -- ONE barrier per 128 k's.  W(s) lives in B0, W(s + 1) in B1, each requested one step ahead; the
        // waits for them are counted by hand (see bload).  Requests past the last step repeat it (into stages nobody reads), so that the
        // counts hold in every pair; an odd step count skips the second half of the last pair.
        u32x4v B0[2][4], B1[2][4];
        const int last = nsteps - 1;
        auto clampi = [&](int v) { return v < last ? v : last; };
        int half = 0;                                            // stages 2 half, 2 half + 1 hold this pair
        xissue2(0, clampi(1), 0);
        bload(0, B0);
        for (int step = 0; step < nsteps; step += 2) {
            __builtin_amdgcn_s_waitcnt(0x0f78);                  // vmcnt(8)
            __builtin_amdgcn_s_barrier();
            bload(clampi(step + 1), B1);
            xissue2(clampi(step + 2), clampi(step + 3), half ? 0 : 2);
            // queue now: W(step) x 8, W(step + 1) x 8, X x 8
            asm volatile("s_waitcnt vmcnt(16)" : "+v"(B0[0][0]), "+v"(B0[0][1]), "+v"(B0[0][2]), "+v"(B0[0][3]),
                                                 "+v"(B0[1][0]), "+v"(B0[1][1]), "+v"(B0[1][2]), "+v"(B0[1][3]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma1(pl_lds + (2 * half) * DIMN_PB_XST, B0);
            __builtin_amdgcn_sched_barrier(0);
            if (step + 1 < nsteps) {
                bload(clampi(step + 2), B0);
                // queue now: W(step + 1) x 8, X x 8, W(step + 2) x 8
                asm volatile("s_waitcnt vmcnt(16)" : "+v"(B1[0][0]), "+v"(B1[0][1]), "+v"(B1[0][2]), "+v"(B1[0][3]),
                                                     "+v"(B1[1][0]), "+v"(B1[1][1]), "+v"(B1[1][2]), "+v"(B1[1][3]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mma1(pl_lds + (2 * half + 1) * DIMN_PB_XST, B1);
                __builtin_amdgcn_sched_barrier(0);
            }
            half ^= 1;
        }
        // (both operand sets named: the last step requested one of them for a step that does not exist)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(B0[0][0]), "+v"(B0[0][1]), "+v"(B0[0][2]), "+v"(B0[0][3]),
                                            "+v"(B0[1][0]), "+v"(B0[1][1]), "+v"(B0[1][2]), "+v"(B0[1][3]) :: "memory");
        asm volatile("" : "+v"(B1[0][0]), "+v"(B1[0][1]), "+v"(B1[0][2]), "+v"(B1[0][3]),
                          "+v"(B1[1][0]), "+v"(B1[1][1]), "+v"(B1[1][2]), "+v"(B1[1][3]) :: "memory");
        __syncthreads();                                         // every wave is done with the ring (and every request has landed): the activations take its place
    }
    // bias + activation, rounded to bf16 into LDS (four hidden units of a row at a time); columns [Hp, Hq) zero.
    // CODE SIZE matters here: this block is unrolled 32 x 4 times.  With the activation switch inlined at every element (tanhf, expm1f,
    // log1pf(expf)) the kernel was 250 KB of code against a 64 KB instruction cache, and a workgroup spent 36 us in this block --
    // one instruction-cache miss per skipped switch -- and 57 + 28 us in the two output epilogues below (round 4, per-workgroup
    // phase stamps).  FAST (relu, O % 4 == 0) holds no switch at all; the other instantiation calls one out-of-line copy.
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int h0 = 64 * wave + 16 * ct + 4 * lj;
        if (h0 < Hq) {
            const f32x4 bias = h0 < Hp ? *(const f32x4*)(b1 + (int64_t)k * Hp + h0) : zero4;
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                f32x4 f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[rt][ct][r] + bias[r];
                    f[r] = FAST ? (v > 0.f ? v : 0.f) : hidden_act_call(act, v);
                    if (h0 + r >= dm.H) f[r] = 0.f;              // (also the columns [H, Hq): their accumulators are not meaningful)
                }
                *(bf16x4*)(ddl + (16 * rt + li) * ld2 + h0) = pk4(f);
            }
        }
    }
    __syncthreads();                                             // (the four waves cover 256 hidden columns: launch_predict sends Hp > 256 to the round-2 kernel)
    // ---- second layer: Z = Dd W2, 16 output tiles per pass (4 per wave) ----
    float lsum0 = 0.f, lsum1 = 0.f;
    const bf16_t* w2k = W2t + (int64_t)k * Hp * dm.Op;
    const int nsteps2 = Hq >> 5;
    for (int ot0 = 0; ot0 < dm.OT; ot0 += 16) {
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = zero4;
        const bf16_t* w2p[4];
        bool ov[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int ot = ot0 + 4 * wave + ct;
            ov[ct] = ot < dm.OT;
            w2p[ct] = w2k + (int64_t)(16 * (ov[ct] ? ot : 0) + li) * Hp;
        }
        // (W2 operands: requested one step ahead into two named sets; a half step beyond Hp re-reads valid weights against the zero
        //  columns [Hp, Hq) of the activations)
        auto b2load = [&](int st, u32x4v (&b)[4]) {
            int h0 = 32 * st + 8 * lj;
            h0 = h0 < Hp ? h0 : 0;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) b[ct] = *(const u32x4v*)(w2p[ct] + h0);
        };
        auto mma2 = [&](int st, const u32x4v (&b)[4]) {
#pragma unroll
            for (int rt = 0; rt < 8; ++rt) {
                const bf16x8n a = *(const bf16x8n*)(ddl + (16 * rt + li) * ld2 + 32 * st + 8 * lj);
#pragma unroll
                // Z^T tile = W2t (A: 16 outputs x 32 hidden) x Dd^T (B: 32 hidden x 16 batch rows): the SAME two register operands as for
                // Z = Dd W2, handed over in the other order -- the accumulator of a lane is then four CONSECUTIVE OUTPUTS of one batch row
                // (acc[rt][ct][r] = Z[b = 16 rt + li][o = 16 ot + 4 lj + r]): one 16-byte store per lane instead of four 4-byte ones
                for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8n, b[ct]), a, acc[rt][ct], 0, 0, 0);
            }
        };
        if constexpr (LOSS) {                                    // (the loss variant has no registers for a third set)
            u32x4v C0[4], C1[4];
            b2load(0, C0);
            int st = 0;
            for (; st + 2 <= nsteps2; st += 2) {
                b2load(st + 1, C1);
                __builtin_amdgcn_sched_barrier(0);
                mma2(st, C0);
                b2load(st + 2 < nsteps2 ? st + 2 : st + 1, C0);
                __builtin_amdgcn_sched_barrier(0);
                mma2(st + 1, C1);
            }
            if (st < nsteps2) mma2(st, C0);
        } else {
            // three named operand sets, requested two steps ahead (one step = 32 matrix instructions per wave does not cover an L2 round trip:
            // with one step of distance this phase took 10-11 us per pass for 3.4 us of matrix-pipe time)
            u32x4v C0[4], C1[4], C2[4];
            const int l2 = nsteps2 - 1;
            auto cl = [&](int v) { return v < l2 ? v : l2; };
            b2load(0, C0);
            b2load(cl(1), C1);
            int st = 0;
            for (; st + 3 <= nsteps2; st += 3) {
                b2load(cl(st + 2), C2);
                __builtin_amdgcn_sched_barrier(0);
                mma2(st, C0);
                b2load(cl(st + 3), C0);
                __builtin_amdgcn_sched_barrier(0);
                mma2(st + 1, C1);
                b2load(cl(st + 4), C1);
                __builtin_amdgcn_sched_barrier(0);
                mma2(st + 2, C2);
            }
            if (st < nsteps2) mma2(st, C0);
            if (st + 1 < nsteps2) mma2(st + 1, C1);
        }
        // epilogue: bias, softplus, store / loss.  A lane holds outputs o0 .. o0+3 of batch row 16 rt + li; the four output tiles of a
        // wave are adjacent (64 outputs = 256 contiguous bytes of a row of `out`), written tile after tile for each row tile
        const bool vec_ok = FAST || (dm.O & 3) == 0;
        f32x4 b2v[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) b2v[ct] = *(const f32x4*)(b2 + (int64_t)k * dm.Op + 16 * (ov[ct] ? ot0 + 4 * wave + ct : 0) + 4 * lj);
#pragma unroll
        for (int rt = 0; rt < 8; ++rt) {
            const int64_t i = r0 + 16 * rt + li;
            const bool row_ok = i < n_rows;
            const int64_t row = row_ok ? (rows ? (int64_t)rows[i] : i) : 0;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int o0 = 16 * (ot0 + 4 * wave + ct) + 4 * lj;
                const bool ok = ov[ct] && row_ok && o0 < dm.O;
                // softplus for every lane, store / loss under `ok`: with the whole block skipped by a branch (as it was) every second block began
                // with a wait for ALL outstanding memory operations -- the previous block's store included, 16 store round trips per pass -- and
                // the scheduler could not move anything across the 32 branch pairs (round 4: 36 + 27 us for the two epilogues of a workgroup)
                const f32x4 zz = acc[rt][ct] + b2v[ct];                                              // (b2 is padded to Op)
                const f32x2 y01 = softplus_out2((f32x2){zz[0], zz[1]}), y23 = softplus_out2((f32x2){zz[2], zz[3]});
                const f32x4 yh = (f32x4){y01.x, y01.y, y23.x, y23.y};
                if (out && ok) {
                    float* dst = out + (i * dm.K + k) * dm.O + o0;
                    if (vec_ok) DIMN_PB_STORE((f32x4*)dst, yh);
                    else
                        for (int r = 0; r < 4; ++r) if (o0 + r < dm.O) dst[r] = yh[r];
                }
                if (LOSS && ok) {
                    const float* yrow = Y + ((int64_t)k * n_cells + row) * dm.Op + o0;
                    float acc_l = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (o0 + r < dm.O) {
                            const float y = yrow[r];
                            const float w = loss_binary ? (y > 0.f ? 1.f : 0.f) : y;
                            const float e = y - yh[r];
                            acc_l += w * e * e;
                        }
                    if (rt < 4) lsum0 += acc_l; else lsum1 += acc_l;
                }
            }
        }
    }
    if (LOSS) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lsum0 += __shfl_xor(lsum0, off); lsum1 += __shfl_xor(lsum1, off); }
        __syncthreads();                                         // (the activations are no longer read: redl may alias nothing, but keep the order explicit)
        if (lane == 0) { redl[wave] = lsum0; redl[4 + wave] = lsum1; }
        __syncthreads();
        if (tid == 0) {
            const int64_t t64 = 2 * (int64_t)blockIdx.x;
            loss_part[(int64_t)k * lp_stride + t64] = redl[0] + redl[1] + redl[2] + redl[3];
            if (t64 + 1 < lp_stride) loss_part[(int64_t)k * lp_stride + t64 + 1] = redl[4] + redl[5] + redl[6] + redl[7];
        }
    }
#endif
}

// the round-2 form (64 rows per workgroup, 16-deep instructions, every operand an 8-byte global load): hidden widths beyond 256
template <int NT>
__global__ __launch_bounds__(256) void k_predict_bf16_r2(const SubnetDev* __restrict__ sn, const bf16_t* __restrict__ X,
                                                      const bf16_t* __restrict__ W1b, const float* __restrict__ b1,
                                                      const bf16_t* __restrict__ W2t, const float* __restrict__ b2,
                                                      const int32_t* __restrict__ rows, int64_t n_rows,
                                                      float* __restrict__ out, const float* __restrict__ Y, int64_t n_cells,
                                                      float* __restrict__ loss_part, Dims dm, int loss_binary, int act) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_lds[];
    bf16_t* ddl = (bf16_t*)pl_lds;                               // Dd [64][ldb] bf16
    const int ldb = dm.Hp + 4;                                   // 8 bytes of padding per row
    float* redl = (float*)(pl_lds + (size_t)DIMN_TB * ldb * 2);  // [4] loss partials
    const int k = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * DIMN_TB;
    const SubnetDev s = sn[k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lj = lane >> 4;
    const int nt0 = wave * NT;
    const int Hp = dm.Hp;

    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16_t* xk = X + s.xoff;
    int64_t xo[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t i = r0 + 16 * mt + li;
        const int64_t row = i < n_rows ? (rows ? (int64_t)rows[i] : i) : 0;      // rows past the end read row 0 and are dropped
        xo[mt] = row * s.Dp + 4 * lj;
    }
    const int64_t cstride = (int64_t)Hp * 16;
    const bf16_t* wbt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int t = (nt0 + nt) < dm.HT ? (nt0 + nt) : (dm.HT - 1);
        wbt[nt] = W1b + s.w1off + (int64_t)(16 * t + li) * 16 + 4 * lj;
    }
    struct Ops { bf16x4 a[4], b[NT]; };
    auto fetch = [&](Ops& o, int c) {
        const int cc = c < s.nchunk ? c : s.nchunk - 1;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) o.a[mt] = *(const bf16x4*)(xk + xo[mt] + 16 * cc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o.b[nt] = *(const bf16x4*)(wbt[nt] + cc * cstride);
    };
    auto mma = [&](const Ops& o) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = MFMA_BF16(o.a[mt], o.b[nt], acc[mt][nt]);
    };
    Ops P0, P1;
    fetch(P0, 0);
    int c = 0;
    for (; c + 2 <= s.nchunk; c += 2) {
        fetch(P1, c + 1);
        mma(P0);
        fetch(P0, c + 2);
        mma(P1);
    }
    if (c < s.nchunk) mma(P0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        if (nt0 + nt < dm.HT) {
            const int h = 16 * (nt0 + nt) + li;
            const float bias = b1[(int64_t)k * Hp + h];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[mt][nt][r] + bias;
                    float f = v > 0.f ? v : 0.f;
                    if (act != 0) { f = hidden_act_call(act, v); if (h >= dm.H) f = 0.f; }     // (out of line: 64 inlined switches were 100 KB of code)
                    ddl[(16 * mt + 4 * lj + r) * ldb + h] = f32_to_bf16(f);
                }
        }
    __syncthreads();

    float lsum = 0.f;
    const bf16_t* w2k = W2t + (int64_t)k * Hp * dm.Op;
    for (int ot = wave; ot < dm.OT; ot += 4) {
        f32x4 z[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) z[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bf16_t* wrow = w2k + (int64_t)(16 * ot + li) * Hp + 4 * lj;         // W2[h = 16ht + 4lj..][o = 16ot + li]
        for (int ht = 0; ht < dm.HT; ++ht) {
            const bf16x4 bq = *(const bf16x4*)(wrow + 16 * ht);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x4 aq = *(const bf16x4*)(ddl + (16 * mt + li) * ldb + 16 * ht + 4 * lj);
                z[mt] = MFMA_BF16(aq, bq, z[mt]);
            }
        }
        const int o = 16 * ot + li;
        if (o < dm.O) {
            const float bias = b2[(int64_t)k * dm.Op + o];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t i = r0 + 16 * mt + 4 * lj + r;
                    if (i < n_rows) {
                        const float yh = softplus_out(z[mt][r] + bias);
                        if (out) __builtin_nontemporal_store(yh, &out[(i * dm.K + k) * dm.O + o]);
                        if (loss_part) {
                            const int64_t row = rows ? (int64_t)rows[i] : i;
                            const float y = Y[((int64_t)k * n_cells + row) * dm.Op + o];
                            const float w = loss_binary ? (y > 0.f ? 1.f : 0.f) : y;
                            const float e = y - yh;
                            lsum += w * e * e;
                        }
                    }
                }
        }
    }
    if (loss_part) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
        if (lane == 0) redl[wave] = lsum;
        __syncthreads();
        if (threadIdx.x == 0) loss_part[(int64_t)k * gridDim.x + blockIdx.x] = redl[0] + redl[1] + redl[2] + redl[3];
    }
}
