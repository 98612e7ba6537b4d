/*
 * dimo_general.c -- CPU ORACLE of libdimn's GENERAL path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as dimo.c:
 * only tests/ may load it; the product never does).
 *
 * Restates, as plain loops in Keras memory layout, what the reference's build() makes Keras do for ANY architecture list
 * (deepimpute/multinet.py:132-146: for every sub-net a chain of Dense(neurons, activation) / Dropout(rate) layers, then
 * Dense(sub_outputdim, softplus)), any batch size (multinet.py:69) and the losses reachable through `loss`
 * (multinet.py:150-162): wMSE, wMSE(binary), keras mean_squared_error, mean_absolute_error.  Adam, the epoch loop,
 * validation and predict are those of dimo.c (S5-S12 of SURVEY.md section 8c); the dropout layer j (j-th layer with a
 * rate > 0) draws from the Philox stream keyed (seed, sub-net, epoch, step | j << 24, element).
 *
 * PARITY STATUS: as dimo.c -- unpinned against Keras (not installable here); pinned against torch-fp64 autograd by
 * tests/golden/kat_general.npz (make_general.py: two hidden layers, batch 100, four losses, Philox masks restated in numpy).
 *
 * Build: oracle/Makefile -> libdimo_gen.so (float) and libdimo_gen64.so (-DDIMO_REAL=double).
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
static int fail(int code, const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }
const char* dimog_last_error(void) { return g_err; }
int dimog_real_bytes(void) { return (int)sizeof(real); }

typedef struct {
    int D;
    int32_t *pred, *targ;
    real **W, **b, **mW, **vW, **mb, **vb;     /* [L+1] */
    real **h, **gate, **dz;                    /* [L+1]: layer outputs [B][w_l], gates, dZ (last: output layer) */
    real* x;                                   /* [B][D] */
} gsub;

struct dimog_s {
    dimn_config cfg;
    int K, O, B, L, loss;
    float in_rate;                             /* rate of a Dropout layer before the first Dense layer (0: none) */
    dimn_layer* layers;
    int* width;                                /* [L+1] */
    gsub* s;
    float* norm; int64_t n, g;
    int32_t *train_rows, *val_rows; int64_t n_tr, n_val;
    int64_t t;
};
typedef struct dimog_s* dimog_handle;

static int in_of(const struct dimog_s* h, const gsub* s, int l) { return l == 0 ? s->D : h->width[l - 1]; }

int dimog_create(const dimn_config* cfg, const int32_t* D, const dimn_layer* layers, int32_t L, int32_t loss, dimog_handle* out) {
    if (!cfg || !D || !layers || L < 1 || !out) return fail(DIMN_ERR_ARG, "create: bad argument");
    float in_rate = 0.f;                 /* a leading entry with neurons == 0: a Dropout layer before the first Dense layer (multinet.py:139-141) */
    if (layers[0].neurons == 0) {
        in_rate = layers[0].dropout_rate;
        if (!(in_rate > 0.f && in_rate < 1.f) || L < 2) return fail(DIMN_ERR_ARG, "create: bad input-dropout entry");
        ++layers; --L;
    }
    struct dimog_s* h = calloc(1, sizeof *h);
    h->in_rate = in_rate;
    h->cfg = *cfg; h->K = cfg->n_subnets; h->O = cfg->out_dim; h->B = cfg->batch_size; h->L = L; h->loss = loss;
    h->layers = malloc((size_t)L * sizeof(dimn_layer)); memcpy(h->layers, layers, (size_t)L * sizeof(dimn_layer));
    h->width = malloc((size_t)(L + 1) * sizeof(int));
    for (int l = 0; l < L; ++l) h->width[l] = layers[l].neurons;
    h->width[L] = h->O;
    h->s = calloc((size_t)h->K, sizeof(gsub));
    const size_t PB = (size_t)(h->B > 256 ? h->B : 256);
    for (int k = 0; k < h->K; ++k) {
        gsub* s = &h->s[k];
        s->D = D[k];
        s->pred = calloc((size_t)s->D, 4); s->targ = calloc((size_t)h->O, 4);
        real*** arrs[] = {&s->W, &s->b, &s->mW, &s->vW, &s->mb, &s->vb, &s->h, &s->gate, &s->dz};
        for (size_t a = 0; a < sizeof arrs / sizeof arrs[0]; ++a) *arrs[a] = calloc((size_t)L + 1, sizeof(real*));
        for (int l = 0; l <= L; ++l) {
            const size_t nw = (size_t)in_of(h, s, l) * h->width[l], nb = (size_t)h->width[l];
            s->W[l] = calloc(nw, sizeof(real)); s->mW[l] = calloc(nw, sizeof(real)); s->vW[l] = calloc(nw, sizeof(real));
            s->b[l] = calloc(nb, sizeof(real)); s->mb[l] = calloc(nb, sizeof(real)); s->vb[l] = calloc(nb, sizeof(real));
            s->h[l] = calloc(PB * nb, sizeof(real)); s->gate[l] = calloc(PB * nb, sizeof(real)); s->dz[l] = calloc(PB * nb, sizeof(real));
        }
        s->x = calloc(PB * (size_t)s->D, sizeof(real));
    }
    *out = h;
    return DIMN_OK;
}

int dimog_destroy(dimog_handle h) {
    if (!h) return DIMN_OK;
    for (int k = 0; k < h->K; ++k) {
        gsub* s = &h->s[k];
        for (int l = 0; l <= h->L; ++l) {
            free(s->W[l]); free(s->b[l]); free(s->mW[l]); free(s->vW[l]); free(s->mb[l]); free(s->vb[l]); free(s->h[l]); free(s->gate[l]); free(s->dz[l]);
        }
        free(s->W); free(s->b); free(s->mW); free(s->vW); free(s->mb); free(s->vb); free(s->h); free(s->gate); free(s->dz);
        free(s->pred); free(s->targ); free(s->x);
    }
    free(h->s); free(h->layers); free(h->width); free(h->norm); free(h->train_rows); free(h->val_rows); free(h);
    return DIMN_OK;
}

int dimog_set_matrix(dimog_handle h, const float* norm, int64_t n, int64_t g) {
    free(h->norm);
    h->norm = malloc((size_t)n * g * sizeof(float));
    memcpy(h->norm, norm, (size_t)n * g * sizeof(float));
    h->n = n; h->g = g;
    return DIMN_OK;
}
int dimog_set_indices(dimog_handle h, int32_t k, const int32_t* pred, int32_t D_k, const int32_t* targ) {
    if (k < 0 || k >= h->K || D_k != h->s[k].D) return fail(DIMN_ERR_ARG, "set_indices: bad argument");
    memcpy(h->s[k].pred, pred, (size_t)D_k * 4); memcpy(h->s[k].targ, targ, (size_t)h->O * 4);
    return DIMN_OK;
}
int dimog_set_split(dimog_handle h, const int32_t* tr, int64_t n_tr, const int32_t* va, int64_t n_val) {
    free(h->train_rows); free(h->val_rows);
    h->train_rows = malloc((size_t)(n_tr > 0 ? n_tr : 1) * 4); h->val_rows = malloc((size_t)(n_val > 0 ? n_val : 1) * 4);
    if (n_tr > 0) memcpy(h->train_rows, tr, (size_t)n_tr * 4);
    if (n_val > 0) memcpy(h->val_rows, va, (size_t)n_val * 4);
    h->n_tr = n_tr; h->n_val = n_val;
    return DIMN_OK;
}

int dimog_init_weights(dimog_handle h, uint64_t seed) {        /* Glorot uniform per layer, zero biases, zero Adam state */
    for (int k = 0; k < h->K; ++k) {
        gsub* s = &h->s[k];
        const uint32_t kg = (uint32_t)(h->cfg.subnet_offset + k);
        for (int l = 0; l <= h->L; ++l) {
            const int in = in_of(h, s, l), out = h->width[l];
            const float lim = (float)sqrt(6.0 / ((double)in + out));
            for (size_t e = 0; e < (size_t)in * out; ++e) { s->W[l][e] = dimn_init_value(seed, kg, (uint32_t)l, (uint32_t)e, lim); s->mW[l][e] = 0; s->vW[l][e] = 0; }
            for (int j = 0; j < out; ++j) { s->b[l][j] = 0; s->mb[l][j] = 0; s->vb[l][j] = 0; }
        }
    }
    h->t = 0;
    return DIMN_OK;
}
int dimog_get_step_count(dimog_handle h, int64_t* t) { *t = h->t; return DIMN_OK; }

int dimog_set_layer_weights(dimog_handle h, int32_t k, int32_t l, const float* W, const float* b) {
    if (k < 0 || k >= h->K || l < 0 || l > h->L) return fail(DIMN_ERR_ARG, "set_layer_weights: bad argument");
    gsub* s = &h->s[k];
    for (size_t e = 0; e < (size_t)in_of(h, s, l) * h->width[l]; ++e) s->W[l][e] = (real)W[e];
    for (int j = 0; j < h->width[l]; ++j) s->b[l][j] = (real)b[j];
    return DIMN_OK;
}
int dimog_get_layer_weights(dimog_handle h, int32_t k, int32_t l, int32_t which, float* W, float* b) {
    if (k < 0 || k >= h->K || l < 0 || l > h->L || which < 0 || which > 2) return fail(DIMN_ERR_ARG, "get_layer_weights: bad argument");
    gsub* s = &h->s[k];
    const real* sw = which == 0 ? s->W[l] : (which == 1 ? s->mW[l] : s->vW[l]);
    const real* sb = which == 0 ? s->b[l] : (which == 1 ? s->mb[l] : s->vb[l]);
    for (size_t e = 0; e < (size_t)in_of(h, s, l) * h->width[l]; ++e) W[e] = (float)sw[e];
    for (int j = 0; j < h->width[l]; ++j) b[j] = (float)sb[j];
    return DIMN_OK;
}

static inline real softplus_r(real x) {
    const real thr = (real)13.942385f;
    if (x > thr) return x;
    if (x < -thr) return (real)exp((double)x);
    return (real)log1p(exp((double)x));
}
static inline real sigmoid_r(real x) { return (real)(1.0 / (1.0 + exp(-(double)x))); }
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
        default: *f = a > 0 ? a : 0; *df = a > 0 ? (real)1 : (real)0; break;
    }
}

/* forward of `cnt` rows of sub-net k; train: dropout with the Philox masks of (epoch, step) */
static void forward(struct dimog_s* h, int k, const int32_t* rows, int cnt, int train, uint32_t epoch, uint32_t step) {
    gsub* s = &h->s[k];
    const uint32_t kg = (uint32_t)(h->cfg.subnet_offset + k);
    const float irate = train ? h->in_rate : 0.f;         /* S3 on the inputs: x * (1 / (1 - p)) * [u >= p], dropout ordinal 0 */
    const real iscale = irate > 0.f ? (real)(1.0f / (1.0f - irate)) : (real)1;
    for (int b = 0; b < cnt; ++b) {
        const float* r = h->norm + (size_t)rows[b] * h->g;
        for (int d = 0; d < s->D; ++d) {
            real v = (real)r[s->pred[d]];
            if (irate > 0.f) v = dimn_dropout_keep(h->cfg.seed, kg, epoch, step & 0xFFFFFFu, (uint32_t)(b * s->D + d), irate) ? v * iscale : 0;
            s->x[(size_t)b * s->D + d] = v;
        }
    }
    int dl = h->in_rate > 0.f ? 1 : 0;
    for (int l = 0; l <= h->L; ++l) {
        const int in = in_of(h, s, l), out = h->width[l];
        const real* prev = l == 0 ? s->x : s->h[l - 1];
        const float rate = (l < h->L && train) ? h->layers[l].dropout_rate : 0.f;
        const real scale = rate > 0.f ? (real)(1.0f / (1.0f - rate)) : (real)1;
        for (int b = 0; b < cnt; ++b) {
            real* o = s->h[l] + (size_t)b * out;
            for (int j = 0; j < out; ++j) o[j] = 0;
            for (int i = 0; i < in; ++i) {
                const real xv = prev[(size_t)b * in + i];
                const real* w = s->W[l] + (size_t)i * out;
                for (int j = 0; j < out; ++j) o[j] += xv * w[j];
            }
            for (int j = 0; j < out; ++j) {
                const real a = o[j] + s->b[l][j];
                if (l == h->L) { o[j] = a; continue; }                 /* output layer: pre-softplus */
                real f, df;
                hidden_act(h->layers[l].activation, a, &f, &df);
                int keep = 1;
                if (rate > 0.f) keep = dimn_dropout_keep(h->cfg.seed, kg, epoch, (step & 0xFFFFFFu) | ((uint32_t)dl << 24), (uint32_t)(b * out + j), rate);
                o[j] = keep ? f * scale : 0;
                s->gate[l][(size_t)b * out + j] = keep ? df * scale : 0;
            }
        }
        if (l < h->L && h->layers[l].dropout_rate > 0.f) ++dl;
    }
}

/* loss term and dL/dyhat * N of one element */
static inline void loss_term(int loss, real y, real yh, real* term, real* dy) {
    const real e = y - yh;
    if (loss == DIMN_LOSS_MAE) { *term = (real)fabs((double)e); *dy = e > 0 ? (real)-1 : (e < 0 ? (real)1 : (real)0); return; }
    /* keras.losses (2.x, epsilon 1e-7): mean_squared_logarithmic_error, logcosh, huber (delta 1), poisson -- element terms; the mean
     * over the last axis and over the batch is the caller's 1 / N */
    if (loss == DIMN_LOSS_MSLE) {
        const real a = yh > (real)1e-7 ? yh : (real)1e-7, yy = y > (real)1e-7 ? y : (real)1e-7;
        const real d = (real)log1p((double)yy) - (real)log1p((double)a);
        *term = d * d; *dy = yh > (real)1e-7 ? (real)-2 * d / (a + 1) : 0; return;
    }
    if (loss == DIMN_LOSS_LOGCOSH) {
        const double x = -(double)e;
        *term = (real)(x + log1p(exp(-2.0 * x)) - 0.69314718055994531); *dy = (real)tanh(x); return;
    }
    if (loss == DIMN_LOSS_HUBER) {
        const real ae = e < 0 ? -e : e;
        *term = ae <= 1 ? (real)0.5 * e * e : ae - (real)0.5; *dy = ae <= 1 ? -e : (e > 0 ? (real)-1 : (real)1); return;
    }
    if (loss == DIMN_LOSS_POISSON) { *term = yh - y * (real)log((double)yh + 1e-7); *dy = 1 - y / (yh + (real)1e-7); return; }
    const real w = loss == DIMN_LOSS_WMSE ? y : (loss == DIMN_LOSS_WMSE_BINARY ? (real)(y > 0) : (real)1);
    *term = w * e * e; *dy = (real)-2 * w * e;
}

static inline void adam1(real* w, real* m, real* v, real g, real alpha, real omb1, real omb2, real eps) {
    *m += (g - *m) * omb1;
    *v += (g * g - *v) * omb2;
    *w -= (*m * alpha) / ((real)sqrt((double)*v) + eps);
}

int dimog_train_step(dimog_handle h, const int32_t* rows, int32_t b_act, int32_t epoch_key, int32_t step_key, float* loss_out) {
    if (!h || !rows || b_act < 1 || b_act > h->B) return fail(DIMN_ERR_ARG, "train_step: bad batch");
    const int64_t t = h->t + 1;
    const real alpha = (real)(float)((double)h->cfg.learning_rate * sqrt(1.0 - pow((double)h->cfg.beta2, (double)t)) / (1.0 - pow((double)h->cfg.beta1, (double)t)));
    const real omb1 = (real)(1.0f - h->cfg.beta1), omb2 = (real)(1.0f - h->cfg.beta2), eps = (real)h->cfg.eps;
    const real inv_n = (real)(1.0 / ((double)b_act * h->O));
#pragma omp parallel for schedule(dynamic)
    for (int k = 0; k < h->K; ++k) {
        gsub* s = &h->s[k];
        const int L = h->L, O = h->O;
        forward(h, k, rows, b_act, 1, (uint32_t)epoch_key, (uint32_t)step_key);
        double loss = 0;
        for (int b = 0; b < b_act; ++b) {
            const float* yr = h->norm + (size_t)rows[b] * h->g;
            for (int o = 0; o < O; ++o) {
                const real z = s->h[L][(size_t)b * O + o], y = (real)yr[s->targ[o]];
                real term, dy;
                loss_term(h->loss, y, softplus_r(z), &term, &dy);
                loss += (double)term;
                s->dz[L][(size_t)b * O + o] = dy * inv_n * sigmoid_r(z);
            }
        }
        if (loss_out) loss_out[k] = (float)(loss / ((double)b_act * O));
        for (int l = L; l >= 0; --l) {
            const int in = in_of(h, s, l), out = h->width[l];
            const real* prev = l == 0 ? s->x : s->h[l - 1];
            if (l >= 1)                                          /* dZ_{l-1} = (dZ_l W_l^T) * gate_{l-1}, with the OLD W_l */
                for (int b = 0; b < b_act; ++b)
                    for (int i = 0; i < in; ++i) {
                        real acc = 0;
                        for (int j = 0; j < out; ++j) acc += s->dz[l][(size_t)b * out + j] * s->W[l][(size_t)i * out + j];
                        s->dz[l - 1][(size_t)b * in + i] = acc * s->gate[l - 1][(size_t)b * in + i];
                    }
            for (int i = 0; i < in; ++i)
                for (int j = 0; j < out; ++j) {
                    real g = 0;
                    for (int b = 0; b < b_act; ++b) g += prev[(size_t)b * in + i] * s->dz[l][(size_t)b * out + j];
                    const size_t e = (size_t)i * out + j;
                    adam1(&s->W[l][e], &s->mW[l][e], &s->vW[l][e], g, alpha, omb1, omb2, eps);
                }
            for (int j = 0; j < out; ++j) {
                real g = 0;
                for (int b = 0; b < b_act; ++b) g += s->dz[l][(size_t)b * out + j];
                adam1(&s->b[l][j], &s->mb[l][j], &s->vb[l][j], g, alpha, omb1, omb2, eps);
            }
        }
    }
    h->t = t;
    return DIMN_OK;
}

int dimog_epoch_permutation(uint64_t seed, int32_t epoch, int64_t n, int32_t* perm) { dimn_fill_permutation(seed, (uint32_t)epoch, n, perm); return DIMN_OK; }

int dimog_train_epoch(dimog_handle h, int32_t epoch, const int32_t* perm, double* train_loss) {
    if (!h->train_rows || h->n_tr < 1) return fail(DIMN_ERR_STATE, "train_epoch: set_split first");
    int32_t* p = NULL;
    if (!perm) { p = malloc((size_t)h->n_tr * 4); dimn_fill_permutation(h->cfg.seed, (uint32_t)epoch, h->n_tr, p); perm = p; }
    int32_t* rows = malloc((size_t)h->B * 4);
    float* lb = malloc((size_t)h->K * sizeof(float));
    double* acc = calloc((size_t)h->K, sizeof(double));
    int step = 0, rc = DIMN_OK;
    for (int64_t i0 = 0; i0 < h->n_tr; i0 += h->B, ++step) {
        const int b_act = (int)((h->n_tr - i0) < h->B ? (h->n_tr - i0) : h->B);
        for (int b = 0; b < b_act; ++b) rows[b] = h->train_rows[perm[i0 + b]];
        if ((rc = dimog_train_step(h, rows, b_act, epoch, step, lb))) break;
        for (int k = 0; k < h->K; ++k) acc[k] += (double)lb[k] * b_act;
    }
    if (train_loss) for (int k = 0; k < h->K; ++k) train_loss[k] = acc[k] / (double)h->n_tr;
    free(rows); free(lb); free(acc); free(p);
    return rc;
}

int dimog_val_loss(dimog_handle h, double* val_loss) {
    if (!h->val_rows || h->n_val < 1) return fail(DIMN_ERR_STATE, "val_loss: no validation rows");
#pragma omp parallel for schedule(dynamic)
    for (int k = 0; k < h->K; ++k) {
        gsub* s = &h->s[k];
        double acc = 0;
        for (int64_t i0 = 0; i0 < h->n_val; i0 += 256) {
            const int cnt = (int)((h->n_val - i0) < 256 ? (h->n_val - i0) : 256);
            forward(h, k, h->val_rows + i0, cnt, 0, 0, 0);
            for (int b = 0; b < cnt; ++b) {
                const float* yr = h->norm + (size_t)h->val_rows[i0 + b] * h->g;
                for (int o = 0; o < h->O; ++o) {
                    real term, dy;
                    loss_term(h->loss, (real)yr[s->targ[o]], softplus_r(s->h[h->L][(size_t)b * h->O + o]), &term, &dy);
                    acc += (double)term;
                }
            }
        }
        val_loss[k] = acc / ((double)h->n_val * h->O);
    }
    return DIMN_OK;
}

int dimog_fit(dimog_handle h, int32_t max_epochs, int32_t patience, double* loss_hist, double* val_hist, int32_t* epochs_run) {
    double* tl = malloc((size_t)h->K * sizeof(double)); double* vl = malloc((size_t)h->K * sizeof(double));
    double best = INFINITY; int wait = 0, e = 0, rc = DIMN_OK;
    for (e = 0; e < max_epochs; ++e) {
        if ((rc = dimog_train_epoch(h, e, NULL, tl))) break;
        if ((rc = dimog_val_loss(h, vl))) break;
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

int dimog_predict(dimog_handle h, const int32_t* rows, int64_t n_rows, float* out) {
#pragma omp parallel for schedule(dynamic)
    for (int k = 0; k < h->K; ++k) {
        gsub* s = &h->s[k];
        int32_t idx[256];
        for (int64_t i0 = 0; i0 < n_rows; i0 += 256) {
            const int cnt = (int)((n_rows - i0) < 256 ? (n_rows - i0) : 256);
            for (int b = 0; b < cnt; ++b) idx[b] = rows ? rows[i0 + b] : (int32_t)(i0 + b);
            forward(h, k, idx, cnt, 0, 0, 0);
            for (int b = 0; b < cnt; ++b)
                for (int o = 0; o < h->O; ++o) out[((size_t)(i0 + b) * h->K + k) * h->O + o] = (float)softplus_r(s->h[h->L][(size_t)b * h->O + o]);
        }
    }
    return DIMN_OK;
}
