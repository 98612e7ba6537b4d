/*
 * dimo.c -- CPU ORACLE for libdimn.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (deepimpute_amd) never imports, links or calls it and fails
 * loudly when the HIP library is missing.
 *
 * What it restates: the arithmetic the reference delegates to Keras/TensorFlow for the
 * MultiNet hot path, written as plain loops in Keras memory layout:
 *   - topology      reference deepimpute/multinet.py:99-103,126-148
 *                   Input(D_k) -> Dense(H, relu) -> Dropout(p) -> Dense(O, softplus)
 *   - loss          multinet.py:36-41   wMSE = reduce_mean(w * (y - yhat)^2), w = y
 *   - optimiser     multinet.py:164     keras.optimizers.Adam(lr): Keras form, eps
 *                   outside the bias correction, one shared step counter
 *   - training loop multinet.py:238-246 model.fit: one permutation per epoch shared by
 *                   all sub-nets, batches incl. the last partial one, validation after
 *                   each epoch, EarlyStopping(val_loss, patience), summed over outputs
 *   - inference     multinet.py:253,278-280 model.predict: forward only
 * plus the upstream semantics S1-S13 listed in SURVEY.md section 8(c).
 *
 * PARITY STATUS: the NN arithmetic is "parity unpinned" against Keras itself:
 * TensorFlow/Keras are not installable here and the reference's tests hold no numeric
 * vector for this path (tests/multinet_test.py:29-33 asserts nothing).  The oracle is
 * pinned instead against torch.autograd in fp64 (tests/golden/kat_steps.npz: single steps,
 * make_kat.py; kat_epochs.npz: a three-epoch trajectory with the Philox streams restated in
 * numpy and checked against the Random123 vectors, make_epochs.py) and analytic known answers; the host shell around it is
 * pinned by fixtures captured from the imported reference (tests/golden/make_shell.py).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC [-DDIMO_REAL=double] (oracle/Makefile).
 * DIMO_REAL=double gives libdimo64.so, used only to pin the formulas against fp64 torch.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dimn.h"
#include "../include/dimn_rng.h"

#ifndef DIMO_REAL
#define DIMO_REAL float
#endif
typedef DIMO_REAL real;

static __thread char g_err[512];
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
const char* dimo_last_error(void) { return g_err; }
int dimo_real_bytes(void) { return (int)sizeof(real); }

typedef struct {
    int D;
    int32_t *pred, *targ;
    real *W1, *b1, *W2, *b2;         /* Keras layout W1[D][H], W2[H][O] */
    real *m[4], *v[4];               /* Adam moments for W1,b1,W2,b2    */
    /* per-step scratch, [B][.] */
    real *x, *a, *dd, *z, *dz, *dA, *y;
    uint8_t* keep;
} subnet;

struct dimo_handle_s {
    dimn_config cfg;
    int K, H, O, B;
    subnet* s;
    float* norm; int64_t n, g;
    int32_t *train_rows, *val_rows; int64_t n_tr, n_val;
    int64_t t;                        /* Adam step counter */
    int act;                          /* hidden activation, DIMN_ACT_* (multinet.py:137) */
    int infer_bf16;                   /* precision bf16: inference/validation GEMM operands rounded to bfloat16 (fp32 accumulate) */
    int train_bf16;                   /* 1 restates k_mid_fused<KEEP, BF>: the three TRAINING GEMMs of the second layer take bf16 operands;
                                         2 restates k_epoch_resident<.., BF>: the two training GEMMs of the first layer as well */
    /* Test instrument (tests/helpers.py, relu flips): up to 8 (sub-net, epoch, step, batch position, hidden unit) whose relu
     * gate is taken on the OTHER side of zero.  Used only where the fp64 replay shows the pre-activation within fp32
     * summation-order error of zero, i.e. where its sign is not defined at fp32 precision. */
    int n_inv; int32_t inv[8][5];
};
typedef struct dimo_handle_s* dimo_handle;

static size_t psize(const struct dimo_handle_s* h, const subnet* s, int which) {
    switch (which) {
        case 0: return (size_t)s->D * h->H;
        case 1: return (size_t)h->H;
        case 2: return (size_t)h->H * h->O;
        default: return (size_t)h->O;
    }
}

int dimo_create(const dimn_config* cfg, const int32_t* D, dimo_handle* out) {
    if (!cfg || !D || !out) return fail(DIMN_ERR_ARG, "null argument");
    if (cfg->n_subnets < 1 || cfg->hidden < 1 || cfg->out_dim < 1 || cfg->batch_size < 1)
        return fail(DIMN_ERR_ARG, "bad sizes");
    if (!(cfg->dropout_rate >= 0.f && cfg->dropout_rate < 1.f))
        return fail(DIMN_ERR_ARG, "dropout_rate must be in [0,1)");
    struct dimo_handle_s* h = calloc(1, sizeof *h);
    h->cfg = *cfg; h->K = cfg->n_subnets; h->H = cfg->hidden; h->O = cfg->out_dim;
    h->B = cfg->batch_size;
    h->s = calloc((size_t)h->K, sizeof(subnet));
    for (int k = 0; k < h->K; ++k) {
        subnet* s = &h->s[k];
        if (D[k] < 1) return fail(DIMN_ERR_ARG, "D[k] < 1");
        s->D = D[k];
        s->pred = calloc((size_t)s->D, sizeof(int32_t));
        s->targ = calloc((size_t)h->O, sizeof(int32_t));
        s->W1 = calloc(psize(h, s, 0), sizeof(real)); s->b1 = calloc(psize(h, s, 1), sizeof(real));
        s->W2 = calloc(psize(h, s, 2), sizeof(real)); s->b2 = calloc(psize(h, s, 3), sizeof(real));
        for (int w = 0; w < 4; ++w) {
            s->m[w] = calloc(psize(h, s, w), sizeof(real));
            s->v[w] = calloc(psize(h, s, w), sizeof(real));
        }
        const size_t B = (size_t)h->B;
        s->x = calloc(B * s->D, sizeof(real)); s->a = calloc(B * h->H, sizeof(real));
        s->dd = calloc(B * h->H, sizeof(real)); s->z = calloc(B * h->O, sizeof(real));
        s->dz = calloc(B * h->O, sizeof(real)); s->dA = calloc(B * h->H, sizeof(real));
        s->y = calloc(B * h->O, sizeof(real)); s->keep = calloc(B * h->H, 1);
    }
    *out = h;
    return DIMN_OK;
}

int dimo_destroy(dimo_handle h) {
    if (!h) return DIMN_OK;
    for (int k = 0; k < h->K; ++k) {
        subnet* s = &h->s[k];
        free(s->pred); free(s->targ); free(s->W1); free(s->b1); free(s->W2); free(s->b2);
        for (int w = 0; w < 4; ++w) { free(s->m[w]); free(s->v[w]); }
        free(s->x); free(s->a); free(s->dd); free(s->z); free(s->dz); free(s->dA); free(s->y);
        free(s->keep);
    }
    free(h->s); free(h->norm); free(h->train_rows); free(h->val_rows); free(h);
    return DIMN_OK;
}

int dimo_set_matrix(dimo_handle h, const float* norm, int64_t n, int64_t g) {
    if (!h || !norm || n < 1 || g < 1) return fail(DIMN_ERR_ARG, "set_matrix: bad argument");
    free(h->norm);
    h->norm = malloc((size_t)n * g * sizeof(float));
    memcpy(h->norm, norm, (size_t)n * g * sizeof(float));
    h->n = n; h->g = g;
    return DIMN_OK;
}

int dimo_set_indices(dimo_handle h, int32_t k, const int32_t* pred, int32_t D_k, const int32_t* targ) {
    if (!h || k < 0 || k >= h->K || !pred || !targ) return fail(DIMN_ERR_ARG, "set_indices: bad argument");
    if (D_k != h->s[k].D) return fail(DIMN_ERR_ARG, "set_indices: D_k differs from create()");
    memcpy(h->s[k].pred, pred, (size_t)D_k * sizeof(int32_t));
    memcpy(h->s[k].targ, targ, (size_t)h->O * sizeof(int32_t));
    return DIMN_OK;
}
int dimo_gather(dimo_handle h, int32_t with_targets) { (void)h; (void)with_targets; return DIMN_OK; }

int dimo_set_split(dimo_handle h, const int32_t* tr, int64_t n_tr, const int32_t* va, int64_t n_val) {
    if (!h || (n_tr > 0 && !tr) || (n_val > 0 && !va)) return fail(DIMN_ERR_ARG, "set_split: bad argument");
    free(h->train_rows); free(h->val_rows);
    h->train_rows = malloc((size_t)(n_tr > 0 ? n_tr : 1) * sizeof(int32_t));
    h->val_rows = malloc((size_t)(n_val > 0 ? n_val : 1) * sizeof(int32_t));
    if (n_tr > 0) memcpy(h->train_rows, tr, (size_t)n_tr * sizeof(int32_t));
    if (n_val > 0) memcpy(h->val_rows, va, (size_t)n_val * sizeof(int32_t));
    h->n_tr = n_tr; h->n_val = n_val;
    return DIMN_OK;
}

/* S1: Glorot-uniform kernels, zero biases (Keras Dense defaults; multinet.py:137,145). */
int dimo_init_weights(dimo_handle h, uint64_t seed) {
    for (int k = 0; k < h->K; ++k) {
        subnet* s = &h->s[k];
        const uint32_t kg = (uint32_t)(h->cfg.subnet_offset + k);
        const float lim1 = (float)sqrt(6.0 / ((double)s->D + h->H));
        const float lim2 = (float)sqrt(6.0 / ((double)h->H + h->O));
        for (size_t e = 0; e < psize(h, s, 0); ++e) s->W1[e] = dimn_init_value(seed, kg, 0, (uint32_t)e, lim1);
        for (size_t e = 0; e < psize(h, s, 2); ++e) s->W2[e] = dimn_init_value(seed, kg, 1, (uint32_t)e, lim2);
        memset(s->b1, 0, psize(h, s, 1) * sizeof(real));
        memset(s->b2, 0, psize(h, s, 3) * sizeof(real));
        for (int w = 0; w < 4; ++w) {
            memset(s->m[w], 0, psize(h, s, w) * sizeof(real));
            memset(s->v[w], 0, psize(h, s, w) * sizeof(real));
        }
    }
    h->t = 0;
    return DIMN_OK;
}

int dimo_reset_optimizer(dimo_handle h) {
    for (int k = 0; k < h->K; ++k)
        for (int w = 0; w < 4; ++w) {
            memset(h->s[k].m[w], 0, psize(h, &h->s[k], w) * sizeof(real));
            memset(h->s[k].v[w], 0, psize(h, &h->s[k], w) * sizeof(real));
        }
    h->t = 0;
    return DIMN_OK;
}
int dimo_get_step_count(dimo_handle h, int64_t* t) { *t = h->t; return DIMN_OK; }

static void cp_in(real* dst, const float* src, size_t n) { for (size_t i = 0; i < n; ++i) dst[i] = (real)src[i]; }
static void cp_out(float* dst, const real* src, size_t n) { for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i]; }

int dimo_set_weights(dimo_handle h, int32_t k, const float* W1, const float* b1, const float* W2, const float* b2) {
    if (!h || k < 0 || k >= h->K) return fail(DIMN_ERR_ARG, "set_weights: bad k");
    subnet* s = &h->s[k];
    cp_in(s->W1, W1, psize(h, s, 0)); cp_in(s->b1, b1, psize(h, s, 1));
    cp_in(s->W2, W2, psize(h, s, 2)); cp_in(s->b2, b2, psize(h, s, 3));
    return DIMN_OK;
}
int dimo_get_weights(dimo_handle h, int32_t k, float* W1, float* b1, float* W2, float* b2) {
    if (!h || k < 0 || k >= h->K) return fail(DIMN_ERR_ARG, "get_weights: bad k");
    subnet* s = &h->s[k];
    cp_out(W1, s->W1, psize(h, s, 0)); cp_out(b1, s->b1, psize(h, s, 1));
    cp_out(W2, s->W2, psize(h, s, 2)); cp_out(b2, s->b2, psize(h, s, 3));
    return DIMN_OK;
}
int dimo_get_adam_state(dimo_handle h, int32_t k, int32_t which, float* W1, float* b1, float* W2, float* b2) {
    if (!h || k < 0 || k >= h->K || which < 0 || which > 1) return fail(DIMN_ERR_ARG, "get_adam_state: bad argument");
    subnet* s = &h->s[k];
    real** src = which == 0 ? s->m : s->v;
    cp_out(W1, src[0], psize(h, s, 0)); cp_out(b1, src[1], psize(h, s, 1));
    cp_out(W2, src[2], psize(h, s, 2)); cp_out(b2, src[3], psize(h, s, 3));
    return DIMN_OK;
}

/* S4: softplus with TensorFlow's fp32 thresholds (log(eps_f32)+2 ~ -13.94). */
static inline real softplus_r(real x) {
    const real thr = (real)13.942385f;
    if (x > thr) return x;
    if (x < -thr) return (real)exp((double)x);
    return (real)log1p(exp((double)x));
}
static inline real sigmoid_r(real x) { return (real)(1.0 / (1.0 + exp(-(double)x))); }

/* Hidden activation f and its derivative f' at pre-activation a (Keras definitions; elu alpha = 1). */
static inline void hidden_act(int act, real a, real* f, real* df) {
    switch (act) {
        case DIMN_ACT_LINEAR: *f = a; *df = 1; break;
        case DIMN_ACT_SIGMOID: { const real s = sigmoid_r(a); *f = s; *df = s * (1 - s); break; }
        case DIMN_ACT_TANH: { const real t = (real)tanh((double)a); *f = t; *df = 1 - t * t; break; }
        case DIMN_ACT_ELU: { const real e = (real)expm1((double)a); *f = a > 0 ? a : e; *df = a > 0 ? (real)1 : e + 1; break; }
        case DIMN_ACT_SOFTPLUS: *f = softplus_r(a); *df = sigmoid_r(a); break;
        /* keras.activations, TF / Keras 2.x: selu (scale, alpha as published), softsign, swish, gelu (erf form), exponential, hard_sigmoid */
        case DIMN_ACT_SELU: { const real e = (real)expm1((double)a); *f = (real)1.0507009873554805 * (a > 0 ? a : (real)1.6732632423543772 * e);
                              *df = (real)1.0507009873554805 * (a > 0 ? (real)1 : (real)1.6732632423543772 * (e + 1)); break; }
        case DIMN_ACT_SOFTSIGN: { const real r = 1 / (1 + (a < 0 ? -a : a)); *f = a * r; *df = r * r; break; }
        case DIMN_ACT_SWISH: { const real s = sigmoid_r(a); *f = a * s; *df = s + a * s * (1 - s); break; }
        case DIMN_ACT_GELU: { const real c = (real)(0.5 * (1.0 + erf((double)a * 0.70710678118654752))); *f = a * c;
                              *df = c + a * (real)(0.3989422804014327 * exp(-0.5 * (double)a * (double)a)); break; }
        case DIMN_ACT_EXPONENTIAL: { const real e = (real)exp((double)a); *f = e; *df = e; break; }
        case DIMN_ACT_HARD_SIGMOID: { const real y = (real)0.2 * a + (real)0.5; *f = y < 0 ? 0 : (y > 1 ? (real)1 : y);
                                      *df = (a > (real)-2.5 && a < (real)2.5) ? (real)0.2 : 0; break; }
        default: *f = a > 0 ? a : 0; *df = a > 0 ? (real)1 : (real)0; break;     /* relu */
    }
}

int dimo_set_activation(dimo_handle h, int32_t activation) {
    if (!h) return fail(DIMN_ERR_ARG, "null handle");
    if (activation < DIMN_ACT_RELU || activation > DIMN_ACT_LAST) return fail(DIMN_ERR_UNSUP, "unknown activation id");
    h->act = activation;
    return DIMN_OK;
}

/* bfloat16 rounding (nearest even) of an fp32 value, as a float */
static inline float bf16_round(float f) {
    union { float f; uint32_t u; } c; c.f = f;
    c.u = (c.u + 0x7fffu + ((c.u >> 16) & 1u)) & 0xffff0000u;
    return c.f;
}
int dimo_set_inference_bf16(struct dimo_handle_s* h, int32_t on) { h->infer_bf16 = on != 0; return DIMN_OK; }
int dimo_set_training_bf16(struct dimo_handle_s* h, int32_t on) { h->train_bf16 = on < 0 ? 0 : (on > 2 ? 2 : on); return DIMN_OK; }
int dimo_invert_gate(struct dimo_handle_s* h, int32_t k, int32_t epoch, int32_t step, int32_t b, int32_t unit) {
    if (!h || h->n_inv >= 8 || k < 0 || k >= h->K || unit < 0 || unit >= h->H) return DIMN_ERR_ARG;
    const int32_t rec[5] = {k, epoch, step, b, unit};
    memcpy(h->inv[h->n_inv++], rec, sizeof rec);
    return DIMN_OK;
}

/* Forward of one row of one sub-net.  x[D] in; a[H] pre-activation, dd[H] hidden output
 * after relu(+dropout when keep != NULL), z[O] pre-softplus out. */
static void forward_row(const struct dimo_handle_s* h, const subnet* s, const real* x,
                        const uint8_t* keep, real* a, real* dd, real* z) {
    const int H = h->H, O = h->O, D = s->D;
    const real scale = keep ? (real)(1.0f / (1.0f - h->cfg.dropout_rate)) : (real)1;
    const int q2 = (!keep && h->infer_bf16) || (keep && h->train_bf16);        /* second layer on the bf16 matrix cores: operands rounded, fp32 accumulate */
    const int q = (!keep && h->infer_bf16) || (keep && h->train_bf16 == 2);    /* first layer likewise (inference; training on the resident bf16 kernel) */
    for (int j = 0; j < H; ++j) a[j] = 0;
    for (int d = 0; d < D; ++d) {                       /* S1: a = x W1 + b1 */
        const real xv = x[d];
        const real* w = s->W1 + (size_t)d * H;
        if (q) { const real xq = (real)bf16_round((float)xv); for (int j = 0; j < H; ++j) a[j] += xq * (real)bf16_round((float)w[j]); }
        else for (int j = 0; j < H; ++j) a[j] += xv * w[j];
    }
    for (int j = 0; j < H; ++j) {
        a[j] += s->b1[j];
        real r, dr;
        hidden_act(h->act, a[j], &r, &dr);              /* S2: relu (or the configured activation) */
        dd[j] = keep ? (keep[j] ? r * scale : 0) : r;   /* S3: dropout (train) / identity */
    }
    for (int o = 0; o < O; ++o) z[o] = 0;
    for (int j = 0; j < H; ++j) {
        const real dv = q2 ? (real)bf16_round((float)dd[j]) : dd[j];
        const real* w = s->W2 + (size_t)j * O;
        if (q2) for (int o = 0; o < O; ++o) z[o] += dv * (real)bf16_round((float)w[o]);
        else for (int o = 0; o < O; ++o) z[o] += dv * w[o];
    }
    for (int o = 0; o < O; ++o) z[o] += s->b2[o];
}

/* cfg.precision == DIMN_PREC_BF16: the predictor blocks are STORED in bfloat16 (targets are not) */
static inline void load_x(const struct dimo_handle_s* h, const subnet* s, int64_t row, real* x) {
    const float* r = h->norm + (size_t)row * h->g;
    if (h->cfg.precision == DIMN_PREC_BF16) for (int d = 0; d < s->D; ++d) x[d] = (real)bf16_round(r[s->pred[d]]);
    else for (int d = 0; d < s->D; ++d) x[d] = (real)r[s->pred[d]];
}

/* S7: Keras-form Adam (TF ResourceApplyAdam): alpha = lr*sqrt(1-b2^t)/(1-b1^t);
 * m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= alpha*m/(sqrt(v)+eps). */
static inline void adam1(real* w, real* m, real* v, real g, real alpha, real omb1, real omb2, real eps) {
    *m += (g - *m) * omb1;
    *v += (g * g - *v) * omb2;
    *w -= (*m * alpha) / ((real)sqrt((double)*v) + eps);
}

static double adam_alpha(const dimn_config* c, int64_t t) {
    return (double)c->learning_rate * sqrt(1.0 - pow((double)c->beta2, (double)t)) /
           (1.0 - pow((double)c->beta1, (double)t));
}

int dimo_train_step(dimo_handle h, const int32_t* rows, int32_t b_act, const uint8_t* keep_mask,
                    int32_t epoch_key, int32_t step_key, float* loss_out) {
    if (!h || !rows || b_act < 1 || b_act > h->B) return fail(DIMN_ERR_ARG, "train_step: bad batch");
    if (!h->norm) return fail(DIMN_ERR_STATE, "train_step: set_matrix first");
    const int H = h->H, O = h->O;
    const int64_t t = h->t + 1;
    const real alpha = (real)(float)adam_alpha(&h->cfg, t);
    const real omb1 = (real)(1.0f - h->cfg.beta1), omb2 = (real)(1.0f - h->cfg.beta2);
    const real eps = (real)h->cfg.eps;
    const float rate = h->cfg.dropout_rate;
    const real scale = (real)(1.0f / (1.0f - rate));
    const real inv_n = (real)(1.0 / ((double)b_act * O));   /* S6: actual batch size */

    /* Every loop nest below is parallel over (sub-net, row) pairs so that all host cores are
     * used even when K is small; per-element summation order does not depend on the thread
     * count, so results are deterministic. */
    const int K = h->K;
    double* lrow = calloc((size_t)K * b_act, sizeof(double));
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < K; ++k)
        for (int b = 0; b < b_act; ++b) {
            subnet* s = &h->s[k];
            const int D = s->D;
            const uint32_t kg = (uint32_t)(h->cfg.subnet_offset + k);
            real* x = s->x + (size_t)b * D;
            uint8_t* keep = s->keep + (size_t)b * H;
            load_x(h, s, rows[b], x);
            for (int j = 0; j < H; ++j) {
                if (keep_mask) keep[j] = keep_mask[((size_t)k * b_act + b) * H + j];
                else if (rate > 0.f)
                    keep[j] = (uint8_t)dimn_dropout_keep(h->cfg.seed, kg, (uint32_t)epoch_key,
                                                         (uint32_t)step_key, (uint32_t)(b * H + j), rate);
                else keep[j] = 1;
            }
            real* a = s->a + (size_t)b * H; real* dd = s->dd + (size_t)b * H;
            real* z = s->z + (size_t)b * O; real* dz = s->dz + (size_t)b * O;
            forward_row(h, s, x, keep, a, dd, z);
            for (int i = 0; i < h->n_inv; ++i)              /* test instrument: a pre-activation at fp32 noise level, taken on the other side */
                if (h->inv[i][0] == k && h->inv[i][1] == epoch_key && h->inv[i][2] == step_key && h->inv[i][3] == b) {
                    const int j = h->inv[i][4];
                    a[j] = a[j] > 0 ? (real)-1e-30 : (real)1e-30;
                    dd[j] = (keep[j] && a[j] > 0) ? a[j] * scale : 0;
                }
            const float* yr = h->norm + (size_t)rows[b] * h->g;
            double loss = 0;
            for (int o = 0; o < O; ++o) {
                const real y = (real)yr[s->targ[o]];
                const real w = h->cfg.loss_binary ? (real)(y > 0) : y;   /* multinet.py:37-40 */
                const real yh = softplus_r(z[o]);
                const real e = y - yh;
                loss += (double)(w * e * e);
                /* dL/dz = -2 w (y - yhat) / (B_act*O) * sigmoid(z)  (zero wherever w == 0) */
                dz[o] = (real)-2 * w * e * inv_n * sigmoid_r(z[o]);
            }
            lrow[(size_t)k * b_act + b] = loss;
        }
    if (loss_out)
        for (int k = 0; k < K; ++k) {
            double loss = 0;
            for (int b = 0; b < b_act; ++b) loss += lrow[(size_t)k * b_act + b];
            loss_out[k] = (float)(loss / ((double)b_act * O));
        }
    free(lrow);

    /* backward through the output layer, using the OLD W2 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < K; ++k)
        for (int b = 0; b < b_act; ++b) {
            subnet* s = &h->s[k];
            real* dA = s->dA + (size_t)b * H;
            const real* dz = s->dz + (size_t)b * O;
            const real* a = s->a + (size_t)b * H;
            const uint8_t* keep = s->keep + (size_t)b * H;
            for (int j = 0; j < H; ++j) {
                const real* w = s->W2 + (size_t)j * O;
                real acc = 0;
                if (h->train_bf16) for (int o = 0; o < O; ++o) acc += (real)bf16_round((float)dz[o]) * (real)bf16_round((float)w[o]);
                else for (int o = 0; o < O; ++o) acc += dz[o] * w[o];
                if (h->act == DIMN_ACT_RELU) {
                    dA[j] = (keep[j] && a[j] > 0) ? acc * scale : 0;
                } else {
                    real f, df;
                    hidden_act(h->act, a[j], &f, &df);
                    dA[j] = keep[j] ? acc * scale * df : 0;
                }
            }
        }
    /* gW2 = dd^T dz ; Adam */
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < H; ++j) {
            subnet* s = &h->s[k];
            for (int o = 0; o < O; ++o) {
                real g = 0;
                if (h->train_bf16)
                    for (int b = 0; b < b_act; ++b) g += (real)bf16_round((float)s->dd[(size_t)b * H + j]) * (real)bf16_round((float)s->dz[(size_t)b * O + o]);
                else
                    for (int b = 0; b < b_act; ++b) g += s->dd[(size_t)b * H + j] * s->dz[(size_t)b * O + o];
                const size_t e = (size_t)j * O + o;
                adam1(&s->W2[e], &s->m[2][e], &s->v[2][e], g, alpha, omb1, omb2, eps);
            }
        }
    /* gb2 = sum_b dz ; gb1 = sum_b dA ; Adam */
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) {
        subnet* s = &h->s[k];
        for (int o = 0; o < O; ++o) {
            real g = 0;
            for (int b = 0; b < b_act; ++b) g += s->dz[(size_t)b * O + o];
            adam1(&s->b2[o], &s->m[3][o], &s->v[3][o], g, alpha, omb1, omb2, eps);
        }
        for (int j = 0; j < H; ++j) {
            real g = 0;
            for (int b = 0; b < b_act; ++b) g += s->dA[(size_t)b * H + j];
            adam1(&s->b1[j], &s->m[1][j], &s->v[1][j], g, alpha, omb1, omb2, eps);
        }
    }
    /* gW1 = x^T dA ; Adam.  Parallel over (sub-net, block of 16 input rows). */
    int maxD = 0;
    for (int k = 0; k < K; ++k) maxD = h->s[k].D > maxD ? h->s[k].D : maxD;
    const int nblk = (maxD + 15) / 16;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < K; ++k)
        for (int blk = 0; blk < nblk; ++blk) {
            subnet* s = &h->s[k];
            const int D = s->D;
            const int d1 = (blk + 1) * 16 < D ? (blk + 1) * 16 : D;
            for (int d = blk * 16; d < d1; ++d) {
                real* wrow = s->W1 + (size_t)d * H; real* mrow = s->m[0] + (size_t)d * H;
                real* vrow = s->v[0] + (size_t)d * H;
                real gr[H];
                for (int j = 0; j < H; ++j) gr[j] = 0;
                for (int b = 0; b < b_act; ++b) {
                    const real xv = s->x[(size_t)b * D + d];
                    const real* da = s->dA + (size_t)b * H;
                    if (h->train_bf16 == 2) { const real xq = (real)bf16_round((float)xv); for (int j = 0; j < H; ++j) gr[j] += xq * (real)bf16_round((float)da[j]); }
                    else for (int j = 0; j < H; ++j) gr[j] += xv * da[j];
                }
                for (int j = 0; j < H; ++j) adam1(&wrow[j], &mrow[j], &vrow[j], gr[j], alpha, omb1, omb2, eps);
            }
        }
    h->t = t;
    return DIMN_OK;
}

/* Test instrument: dL/dz and z of sub-net k, [b_act][O], as the LAST dimo_train_step left them (multinet.py:36-41 under
 * model.fit: dL/dz = dwMSE/dyhat * sigmoid(z)); tests/test_oracle_kat.py divides the two and compares with central differences of
 * the REFERENCE's own wMSE (tests/golden/make_wmse.py). */
int dimo_debug_last_dz(dimo_handle h, int32_t k, int32_t b_act, double* dz_out, double* z_out) {
    if (!h || k < 0 || k >= h->K || b_act < 1 || b_act > h->B || !dz_out || !z_out) return fail(DIMN_ERR_ARG, "debug_last_dz: bad argument");
    const subnet* s = &h->s[k];
    for (size_t i = 0; i < (size_t)b_act * h->O; ++i) { dz_out[i] = (double)s->dz[i]; z_out[i] = (double)s->z[i]; }
    return DIMN_OK;
}

int dimo_epoch_permutation(uint64_t seed, int32_t epoch, int64_t n, int32_t* perm) {
    dimn_fill_permutation(seed, (uint32_t)epoch, n, perm);
    return DIMN_OK;
}

/* S8: one permutation per epoch shared by all sub-nets; last partial batch kept. */
int dimo_train_epoch(dimo_handle h, int32_t epoch, const int32_t* perm, double* train_loss) {
    if (!h || !h->train_rows || h->n_tr < 1) return fail(DIMN_ERR_STATE, "train_epoch: set_split first");
    int32_t* p = NULL;
    if (!perm) {
        p = malloc((size_t)h->n_tr * sizeof(int32_t));
        dimn_fill_permutation(h->cfg.seed, (uint32_t)epoch, h->n_tr, p);
        perm = p;
    }
    int32_t* rows = malloc((size_t)h->B * sizeof(int32_t));
    float* lb = malloc((size_t)h->K * sizeof(float));
    double* acc = calloc((size_t)h->K, sizeof(double));
    int step = 0, rc = DIMN_OK;
    for (int64_t i0 = 0; i0 < h->n_tr; i0 += h->B, ++step) {
        const int b_act = (int)((h->n_tr - i0) < h->B ? (h->n_tr - i0) : h->B);
        for (int b = 0; b < b_act; ++b) rows[b] = h->train_rows[perm[i0 + b]];
        rc = dimo_train_step(h, rows, b_act, NULL, epoch, step, lb);
        if (rc) break;
        for (int k = 0; k < h->K; ++k) acc[k] += (double)lb[k] * b_act;
    }
    if (train_loss) for (int k = 0; k < h->K; ++k) train_loss[k] = acc[k] / (double)h->n_tr;
    free(rows); free(lb); free(acc); free(p);
    return rc;
}

/* S9: validation, no dropout; element mean over the whole validation set per sub-net. */
int dimo_val_loss(dimo_handle h, double* val_loss) {
    if (!h || !val_loss) return fail(DIMN_ERR_ARG, "val_loss: null");
    if (!h->val_rows || h->n_val < 1) return fail(DIMN_ERR_STATE, "val_loss: no validation rows");
    double* part = calloc((size_t)h->K * h->n_val, sizeof(double));
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < h->K; ++k)
        for (int64_t i = 0; i < h->n_val; ++i) {
            subnet* s = &h->s[k];
            real* x = malloc((size_t)s->D * sizeof(real));
            real* a = malloc((size_t)h->H * sizeof(real)); real* dd = malloc((size_t)h->H * sizeof(real));
            real* z = malloc((size_t)h->O * sizeof(real));
            load_x(h, s, h->val_rows[i], x);
            forward_row(h, s, x, NULL, a, dd, z);
            const float* yr = h->norm + (size_t)h->val_rows[i] * h->g;
            double acc = 0;
            for (int o = 0; o < h->O; ++o) {
                const real y = (real)yr[s->targ[o]];
                const real w = h->cfg.loss_binary ? (real)(y > 0) : y;
                const real e = y - softplus_r(z[o]);
                acc += (double)(w * e * e);
            }
            part[(size_t)k * h->n_val + i] = acc;
            free(x); free(a); free(dd); free(z);
        }
    for (int k = 0; k < h->K; ++k) {
        double acc = 0;
        for (int64_t i = 0; i < h->n_val; ++i) acc += part[(size_t)k * h->n_val + i];
        val_loss[k] = acc / ((double)h->n_val * h->O);
    }
    free(part);
    return DIMN_OK;
}

/* S10/S11: EarlyStopping(monitor='val_loss', patience), min mode, min_delta 0, strict <,
 * last-epoch weights kept; monitored value = SUM over sub-nets (S5/S9). */
int dimo_fit(dimo_handle h, int32_t max_epochs, int32_t patience, double* loss_hist, double* val_hist,
             int32_t* epochs_run) {
    double* tl = malloc((size_t)h->K * sizeof(double));
    double* vl = malloc((size_t)h->K * sizeof(double));
    double best = INFINITY; int wait = 0, e = 0, rc = DIMN_OK;
    for (e = 0; e < max_epochs; ++e) {
        if ((rc = dimo_train_epoch(h, e, NULL, tl))) break;
        if ((rc = dimo_val_loss(h, vl))) break;
        double st = 0, sv = 0;
        for (int k = 0; k < h->K; ++k) { st += tl[k]; sv += vl[k]; }
        if (loss_hist) loss_hist[e] = st;
        if (val_hist) val_hist[e] = sv;
        if (sv < best) { best = sv; wait = 0; }
        else if (++wait >= patience) { ++e; break; }
    }
    if (epochs_run) *epochs_run = e;
    free(tl); free(vl);
    return rc;
}

/* S12: model.predict -- forward only; out[n_rows][K*O] = np.hstack(predicted). */
int dimo_predict(dimo_handle h, const int32_t* rows, int64_t n_rows, float* out) {
    if (!h || !out || n_rows < 0) return fail(DIMN_ERR_ARG, "predict: bad argument");
    if (!h->norm) return fail(DIMN_ERR_STATE, "predict: set_matrix first");
    const int K = h->K, O = h->O;
#pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < K; ++k)
        for (int64_t i = 0; i < n_rows; ++i) {
            subnet* s = &h->s[k];
            real* x = malloc((size_t)s->D * sizeof(real));
            real* a = malloc((size_t)h->H * sizeof(real)); real* dd = malloc((size_t)h->H * sizeof(real));
            real* z = malloc((size_t)O * sizeof(real));
            const int64_t row = rows ? rows[i] : i;
            load_x(h, s, row, x);
            forward_row(h, s, x, NULL, a, dd, z);
            float* o = out + ((size_t)i * K + k) * O;
            for (int c = 0; c < O; ++c) o[c] = (float)softplus_r(z[c]);
            free(x); free(a); free(dd); free(z);
        }
    return DIMN_OK;
}
