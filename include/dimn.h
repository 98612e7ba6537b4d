/*
 * dimn.h -- C ABI of libdimn, the MI355X-native replacement for the Keras/TensorFlow
 * seam of DeepImpute's MultiNet (K independent Dense->Dropout->Dense sub-networks,
 * wMSE loss, Keras-form Adam, global early stopping).
 *
 * The reference (lanagarmire/deepimpute) has NO FFI/plugin interface: its hot path is the
 * eight Keras calls multinet.py makes.  Every entry point below names the reference
 * call(s) it replaces (file:line into the reference tree).  Host code (Python, numpy
 * only) binds these through ctypes; see INTEGRATION.md for the binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain C types only; the caller owns every host buffer; the library copies in /
 *     writes into caller-allocated outputs and never keeps a host pointer past the call.
 *   - every function returns 0 on success, <0 on error; dimn_last_error() gives the
 *     message (thread-local).  No exceptions cross the ABI, nothing calls exit().
 *   - a handle is bound to ONE GPU (cfg.device_id) and is not thread-safe.
 *   - weights cross the ABI in Keras layout: kernel W[in][out] row-major fp32, bias[out].
 *   - all arithmetic is fp32 (reference: multinet.py:217,273 cast to float32; Keras
 *     default floatx) on the f32 MFMA path of gfx950.
 */
#ifndef DIMN_H
#define DIMN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIMN_OK 0
#define DIMN_ERR_ARG (-1)     /* bad argument / call order            */
#define DIMN_ERR_HIP (-2)     /* HIP runtime error (message has code) */
#define DIMN_ERR_STATE (-3)   /* object not ready (missing set_* call)*/
#define DIMN_ERR_UNSUP (-4)   /* configuration not supported          */
#define DIMN_ERR_COMM (-5)    /* RCCL error                           */

#define DIMN_MAX_BATCH 64     /* rows per optimiser step handled by one MFMA row tile */

typedef struct dimn_handle_s* dimn_handle;

/*
 * Hyper-parameters of one group of sub-networks.  Replaces the arguments of
 * MultiNet.build() (multinet.py:126-167): Dense(hidden, relu) -> Dropout(rate, seed)
 * -> Dense(out_dim, softplus), compiled with keras.optimizers.Adam(lr) and wMSE
 * (multinet.py:36-41,164-165).  beta1/beta2/eps are Keras's Adam defaults
 * (0.9, 0.999, 1e-7; eps OUTSIDE the bias correction).
 */
typedef struct dimn_config {
    int32_t n_subnets;        /* K_local: sub-networks owned by this handle             */
    int32_t subnet_offset;    /* global index of local sub-net 0 (RNG keys use the
                                 global index, so results do not depend on sharding)    */
    int32_t hidden;           /* H  (architecture[0].neurons, multinet.py:101)          */
    int32_t out_dim;          /* O  (sub_outputdim, multinet.py:75)                     */
    int32_t batch_size;       /* B  (multinet.py:69); 1..DIMN_MAX_BATCH                 */
    int32_t device_id;        /* HIP device ordinal                                     */
    float dropout_rate;       /* p  (architecture[1].rate, multinet.py:102); 0 = none   */
    float learning_rate;      /* multinet.py:68                                         */
    float beta1, beta2, eps;  /* Keras Adam                                             */
    int32_t loss_binary;      /* wMSE(binary=True) weights 1[y>0] (multinet.py:37-38)   */
    uint64_t seed;            /* multinet.py:77; keys the Philox streams                */
    int32_t precision;        /* DIMN_PREC_F32 (reference: float32 everywhere) or DIMN_PREC_BF16: the gathered predictor
                                 blocks X_k are stored in bfloat16 (round to nearest even) and the inference / validation
                                 GEMMs run on the bf16 matrix cores with fp32 accumulation; the TRAINING GEMMs take bf16
                                 operands (fp32 accumulation) wherever the handle runs a kernel that has the bf16
                                 matrix-core variant -- dimn_training_precision() / dimn_path_info() say which; weights
                                 (fp32 master copies), Adam state, targets and every accumulation stay fp32
                                 (BASELINE configs[4]: "bf16 MFMA with fp32 accumulate") */
} dimn_config;
#define DIMN_PREC_F32 0
#define DIMN_PREC_BF16 1

const char* dimn_last_error(void);
/* ABI version; bumped on any signature change. */
int dimn_abi_version(void);
/* Number of HIP devices visible to this process (0 when there is none: not an error).  Host code uses it
 * to place one process per GPU; the reference has no analogue (multinet.py:222-223 sizes CPU threads). */
int dimn_device_count(int32_t* n);

/* build(inputdims) (multinet.py:126-148,226): D[k] = predictor count of sub-net k.  The tuned kernels: one hidden
 * Dense layer of at most 384 units (+ Dropout), batch <= DIMN_MAX_BATCH, wMSE -- loadDefaultArchitecture() and the CLI
 * defaults.  Everything else build() accepts goes through dimn_create_general. */
int dimn_create(const dimn_config* cfg, const int32_t* D, dimn_handle* out);

/* build(inputdims) for ANY architecture list and loss (multinet.py:135-143: a sequence of Dense / Dropout layers;
 * :150-162: loss by name; parser.py:50-66: any batch size / hidden width): the hidden Dense layers in order, each with
 * the rate of the Dropout layer behind it (0 = none); Dense(out_dim, softplus) is implied (multinet.py:145).  A Dropout layer BEFORE
 * the first Dense layer (dropout on the inputs; ABI 8) is a leading entry with neurons == 0 and its rate: it is dropout layer 0.
 * cfg->hidden / dropout_rate / loss_binary are ignored here.  Every other entry point works on such a handle; weights of
 * models with more than one hidden layer move through dimn_set/get_layer_weights.  Dropout layer j (j-th layer with a
 * rate > 0) draws from the Philox stream keyed (seed, sub-net, epoch, step | j << 24, element): j = 0 is the stream
 * of the tuned kernels, so both paths train the default architecture on identical masks. */
typedef struct dimn_layer {
    int32_t neurons;          /* units of the Dense layer                        */
    int32_t activation;       /* DIMN_ACT_*                                      */
    float dropout_rate;       /* rate of the Dropout layer that follows; 0: none */
} dimn_layer;
#define DIMN_LOSS_WMSE 0          /* multinet.py:36-41                                  */
#define DIMN_LOSS_WMSE_BINARY 1   /* wMSE(binary=True)                                  */
#define DIMN_LOSS_MSE 2           /* keras.losses.mean_squared_error                    */
#define DIMN_LOSS_MAE 3           /* keras.losses.mean_absolute_error                   */
/* ABI 8: the other element-wise keras.losses a regression on log1p counts can be given (Keras 2.x definitions, epsilon = 1e-7) */
#define DIMN_LOSS_MSLE 4          /* mean_squared_logarithmic_error: (log(max(y, eps) + 1) - log(max(yhat, eps) + 1))^2 */
#define DIMN_LOSS_LOGCOSH 5       /* logcosh: log(cosh(yhat - y)) = x + softplus(-2x) - log 2                           */
#define DIMN_LOSS_HUBER 6         /* huber, delta = 1: e^2 / 2 for |e| <= 1, |e| - 1/2 beyond                           */
#define DIMN_LOSS_POISSON 7       /* poisson: yhat - y log(yhat + eps)                                                  */
#define DIMN_LOSS_LAST DIMN_LOSS_POISSON
int dimn_create_general(const dimn_config* cfg, const int32_t* D, const dimn_layer* layers, int32_t n_layers,
                        int32_t loss, dimn_handle* out);
/* Keras-layout kernel W[in][out] and bias[out] of dense layer `layer` (0 .. n_layers; the last one is the output layer);
 * get: which = 0 weights, 1 Adam m, 2 Adam v. */
int dimn_set_layer_weights(dimn_handle h, int32_t k, int32_t layer, const float* W, const float* b);
int dimn_get_layer_weights(dimn_handle h, int32_t k, int32_t layer, int32_t which, float* W, float* b);
int dimn_destroy(dimn_handle h);
/* Device memory policy (no reference analogue: Keras/TF own their allocator).  Blocks of >= 32 MB -- the matrix, the gathered
 * X_k / Y_k arenas (multinet.py:231-235), the resident counts, predictions, correlation temporaries -- are kept by the PROCESS
 * when a handle / counts object releases them and are handed to the next request they fit: memory that was hipFree'd earlier
 * in the same process comes back from hipMalloc slowly (the driver wipes it first), which made the hand-over of a second fit()
 * cost 0.2-0.6 s instead of 0.02 s.  dimn_release_cached_memory() gives every idle block back to the driver;
 * DIMN_ARENA_CACHE_GB caps what is kept (default 48, 0: nothing); a request that fails empties the cache and is retried with its
 * exact size.  ABI 7.  The host side calls it from MultiNet.close() (deepimpute_amd.release_cached_memory()). */
int dimn_release_cached_memory(void);
/* out2[0] = bytes of idle blocks the cache holds, out2[1] = bytes of cached-class blocks currently owned by handles.  ABI 8. */
int dimn_cached_memory_info(int64_t* out2);
/* Start-up cost moved out of fit(): creates the HIP context of the device and pins the process-wide bounce buffers (~0.1 s).
 * Idempotent, thread-safe; host code calls it from a helper thread as soon as it knows which GPU it will use.  ABI 7. */
int dimn_warm_up(int32_t device_id);

/*
 * The shared log1p matrix (multinet.py:217 `norm_data`, :271 `norm_raw`), row-major
 * [n_cells][n_genes] fp32.  Copied to HBM once; replaces the K pairs of pandas
 * `.loc[cells, genes].values` host copies (multinet.py:231-235, 273-274).
 */
int dimn_set_matrix(dimn_handle h, const float* norm, int64_t n_cells, int64_t n_genes);
/* The same matrix STREAMED from host memory in row blocks (pinned bounce buffers, copy of one block overlapping the
 * gather of the previous one): the device holds only the gathered X_k / Y_k blocks, never the matrix -- for matrices
 * that do not fit beside their own gathered copy (BASELINE configs[4], 1M x 30k).  Replaces dimn_set_matrix +
 * dimn_gather; every dimn_set_indices must have been made. */
int dimn_set_matrix_streamed(dimn_handle h, const float* norm, int64_t n_cells, int64_t n_genes, int32_t with_targets);
/* Where the streamed hand-over of this handle starts: row block part * NB / parts of the NB blocks, wrapping around.  The ranks of a one-node
 * job that read ONE host copy of the matrix (the reference materialises norm_data per process, multinet.py:217, and has one process; here
 * rank r of w calls (r, w)) then walk different pages of it at any moment.  Default (0, 1): from the first row. */
int dimn_set_stream_order(dimn_handle h, int32_t part, int32_t parts);
/* Column lists of sub-net k: predictors (multinet.py:362) and targets (:338-342),
 * as column indices into the matrix. */
int dimn_set_indices(dimn_handle h, int32_t k, const int32_t* pred_idx, int32_t D_k,
                     const int32_t* targ_idx /* [O] */);
/* Device gather of X_k[n][D_k] and (with_targets) Y_k[n][O] for every sub-net from the
 * shared matrix.  Must follow set_matrix + all set_indices. */
int dimn_gather(dimn_handle h, int32_t with_targets);
/* train_cells / test_cells (multinet.py:228-229) as row indices into the matrix. */
int dimn_set_split(dimn_handle h, const int32_t* train_rows, int64_t n_train,
                   const int32_t* val_rows, int64_t n_val);

/* Glorot-uniform kernels, zero biases (Keras Dense defaults, multinet.py:137,145),
 * Philox stream keyed by (seed, global sub-net, layer, element); zeroes Adam state. */
int dimn_init_weights(dimn_handle h, uint64_t seed);
/* Keras-layout weight I/O (replaces model.save_weights / load_weights,
 * multinet.py:114,122).  set_weights leaves Adam state untouched. */
int dimn_set_weights(dimn_handle h, int32_t k, const float* W1 /*[D_k][H]*/,
                     const float* b1 /*[H]*/, const float* W2 /*[H][O]*/,
                     const float* b2 /*[O]*/);
int dimn_get_weights(dimn_handle h, int32_t k, float* W1, float* b1, float* W2, float* b2);
/* Adam moments (which: 0 = m, 1 = v) in the same layout; test/checkpoint hook. */
int dimn_get_adam_state(dimn_handle h, int32_t k, int32_t which, float* W1, float* b1,
                        float* W2, float* b2);
/* Hidden-layer activation (reference deepimpute/multinet.py:137: Dense(neurons, activation=layer['activation']);
 * the default architecture, the CLI and every reference test use 'relu').  Call before training / inference;
 * a handle starts as DIMN_ACT_RELU.  Returns DIMN_ERR_UNSUP for an unknown id. */
#define DIMN_ACT_RELU 0
#define DIMN_ACT_LINEAR 1
#define DIMN_ACT_SIGMOID 2
#define DIMN_ACT_TANH 3
#define DIMN_ACT_ELU 4       /* alpha = 1 (Keras default) */
#define DIMN_ACT_SOFTPLUS 5
/* ABI 8: the rest of keras.activations' element-wise names (TF / Keras 2.x definitions; `softmax` is not element-wise and stays loud) */
#define DIMN_ACT_SELU 6          /* scale * (x > 0 ? x : alpha * expm1(x)), scale = 1.0507009873554805, alpha = 1.6732632423543772 */
#define DIMN_ACT_SOFTSIGN 7      /* x / (1 + |x|)                                     */
#define DIMN_ACT_SWISH 8         /* x * sigmoid(x)                                    */
#define DIMN_ACT_GELU 9          /* 0.5 x (1 + erf(x / sqrt 2)): keras gelu(approximate=False) */
#define DIMN_ACT_EXPONENTIAL 10  /* exp(x)                                            */
#define DIMN_ACT_HARD_SIGMOID 11 /* clip(0.2 x + 0.5, 0, 1) (Keras 2.x)               */
#define DIMN_ACT_LAST DIMN_ACT_HARD_SIGMOID
int dimn_set_activation(dimn_handle h, int32_t activation);

int dimn_reset_optimizer(dimn_handle h);
/* Adam step counter t (shared by all variables, as in Keras). */
int dimn_get_step_count(dimn_handle h, int64_t* t);

/*
 * One optimiser step on an injected batch (the body of model.fit's inner loop,
 * multinet.py:238-244): forward, wMSE, backward, Adam for every local sub-net.
 *   rows      [b_act] matrix row indices, b_act <= batch_size
 *   keep_mask NULL -> Philox mask keyed (seed, epoch_key, step_key, sub-net, b, h);
 *             else uint8 [K_local][b_act][H], 1 = keep (test hook)
 *   loss_out  NULL or [K_local]: wMSE of this batch per sub-net
 */
int dimn_train_step(dimn_handle h, const int32_t* rows, int32_t b_act,
                    const uint8_t* keep_mask, int32_t epoch_key, int32_t step_key,
                    float* loss_out);
/*
 * One epoch of model.fit (multinet.py:238-244): one permutation of the train rows shared
 * by all sub-nets, batches of batch_size including the last partial one.
 *   perm        NULL -> dimn_epoch_permutation(seed, epoch); else [n_train] permutation
 *   train_loss  NULL or [K_local]: sample-weighted mean of the batch losses
 */
int dimn_train_epoch(dimn_handle h, int32_t epoch, const int32_t* perm, double* train_loss);
/* Validation pass (no dropout): val_loss[k] = mean over n_val*O elements of w*(y-yhat)^2. */
int dimn_val_loss(dimn_handle h, double* val_loss /* [K_local] */);
/*
 * Whole model.fit(...) with EarlyStopping(monitor='val_loss', patience)
 * (multinet.py:238-246) for a single-rank job: strict-< improvement on the SUM over
 * sub-nets, stop after `patience` consecutive non-improving epochs, last-epoch weights.
 *   loss_hist/val_hist  NULL or [max_epochs]
 */
int dimn_fit(dimn_handle h, int32_t max_epochs, int32_t patience, double* loss_hist,
             double* val_hist, int32_t* epochs_run);

/*
 * model.predict(X_list) (multinet.py:253,278-280): forward only, dropout = identity.
 *   rows  NULL -> all matrix rows 0..n_rows-1; else [n_rows] row indices
 *   out   host [n_rows][K_local*O] row-major = np.hstack(predicted)
 */
int dimn_predict(dimn_handle h, const int32_t* rows, int64_t n_rows, float* out);
/* Same, result left in HBM; *dev_out stays valid until the next predict/destroy. */
int dimn_predict_device(dimn_handle h, const int32_t* rows, int64_t n_rows, void** dev_out);

/*
 * predict()'s post-processing (multinet.py:282-305) as a device epilogue over the LAST dimn_predict_device result
 * (from_gathered != 0: over root's gathered matrix of dimn_comm_gather_predictions): per output gene the mean of its
 * target slots (float32, slot order), genes without a slot keep log1p(raw); values above `ceiling`
 * (2 * max log1p(raw), multinet.py:292) or NaN -> 0; expm1; policy 1 "restore" / 2 "max" / 0 none against raw.
 *   raw        host [n_rows][g] float64 (the observed counts, columns in output order), streamed in by row blocks;
 *              NULL: the resident counts of dimn_set_matrix_counts (same cells, same columns)
 *   gene_off   [g+1], gene_slot [gene_off[g]]: the prediction slots (columns of np.hstack(predicted)) of every gene
 *   out        host [n_rows][g] float64, streamed out by row blocks (pinned bounce buffers, copies overlap the kernel)
 */
int dimn_impute_finish(dimn_handle h, const double* raw, int64_t n_rows, int64_t g, const int32_t* gene_off,
                       const int32_t* gene_slot, int32_t policy, double ceiling, int32_t from_gathered, double* out);
/* The same for policy "restore" (multinet.py:296-299: every positive observed count is returned as it is) over the RESIDENT counts of
 * dimn_set_matrix_counts, with `observed` = the caller's own float64 frame of those counts: only the finished values of the ZERO entries
 * cross PCIe (packed per row, column order); the library copies `observed` into `out` and drops them in -- 2.8 GB instead of 8 GB device
 * to host for the 35 % zeros of the 50k x 20k bench matrix.  Same values as dimn_impute_finish(raw = NULL, policy = 1), bit for bit.
 * DIMN_ERR_STATE when `observed` is not the resident matrix (some row holds a different number of zeros): call dimn_impute_finish.
 * *observed_checksum (may be NULL) = dimn_counts_checksum of `observed`, computed on the way (every element is read anyway): equal to the
 * checksum dimn_counts_create returned <=> `observed` is, bit for bit, the frame that was uploaded -- no separate pass over 8 GB.  ABI 8. */
int dimn_impute_finish_restore(dimn_handle h, const void* observed, int32_t observed_dtype, int64_t n_rows, int64_t g, const int32_t* gene_off,
                               const int32_t* gene_slot, double ceiling, int32_t from_gathered, double* out, uint64_t* observed_checksum);
/* element types of a host count matrix: float64, or int64 -- what pd.read_csv (deepImpute.py:13) makes of a count CSV.  An int64 matrix
 * stands for the float64 matrix of the same numbers: checksums hash the bit patterns of (double)v, outputs are float64.  ABI 8. */
#define DIMN_DTYPE_F64 0
#define DIMN_DTYPE_I64 1

/*
 * The held-out metrics fit() reports (multinet.py:251-262: Pearson r and MSE between the validation cells' target
 * values and their predictions, over the entries with a positive observed value) as seven sums computed on the device:
 * out7 = count, Sx, Sy, Sxx, Syy, Sxy, S(x-y)^2 with x = truth (log1p counts), y = prediction.
 */
int dimn_val_metrics(dimn_handle h, double* out7);

/* The epoch permutation the library uses when perm == NULL (host Fisher-Yates over a
 * Philox stream); exported so callers/tests can reproduce the batch order. */
int dimn_epoch_permutation(uint64_t seed, int32_t epoch, int64_t n, int32_t* perm_out);

/* Wait for all queued GPU work of this handle. */
int dimn_synchronize(dimn_handle h);
/* Times (ms, HIP events on the stream each launch went to) accumulated since the last call
 * with reset != 0, over every sub-net lane: out8 = [0] sum of per-lane step times, [1] number of
 * (lane, step) pairs, [2] sum of W1-update kernel times, [3] W1-update launches, [4] sum of the
 * ALGORITHMIC bytes of those launches, [5] number of lanes, [6] sum of the durations of the
 * register-resident epoch launches (few sub-nets per GPU), [7] optimiser steps they ran.
 * bench.py's live roofline figure. */
int dimn_get_timers(dimn_handle h, double* out8, int32_t reset);

/* The operand format of the TRAINING GEMMs of the second layer (multinet.py:139-146 under model.fit, :238): DIMN_PREC_BF16 when
 * the handle was created with precision bf16 and runs the fused second-layer kernel with its bf16 matrix-core variant
 * (Dd, W2, dZ rounded to nearest even per GEMM, fp32 accumulation, fp32 master weights and Adam state), else DIMN_PREC_F32.
 * The first layer's training GEMMs and the optimiser are fp32 on every path.  ABI 4. */
int dimn_training_precision(dimn_handle h);
int dimn_set_profiling(dimn_handle h, int32_t on);
/* Which kernels the library chose for this handle (the automatic decision of dimn_create; DESIGN.md has the table): out8 =
 * [0] 0 streaming kernels / 1 register-resident epoch kernel / 2 general path, [1] resident: launches (sub-net groups) per epoch,
 * [2] resident: D-splits per hidden tile, [3] streaming: 1 fused second layer / 0 two kernels, [4] its slices per sub-net,
 * [5] the fused kernel's form ("mid_kernel"): 2 tile pipeline (k_mid_pipe, fp32 or bf16 operands; 1 / 0 were the three-phase kernel of ABI <= 8, retired),
 * [6] training GEMMs on the bf16 matrix cores (1: the second layer's three, fused kernel; 2: those of both layers, resident kernel; 0: none),
 * [7] first-layer kernel (1 ring, H = 256; 3 four-set ring with one hidden tile per wave, 8 .. 24 hidden tiles other than 16; 0 generic; 2 was the
 * shared-staging kernel of hidden 300, retired in ABI 9).  A handle whose resident launch had to be undone reports 0 from then on.  ABI 5; [5] = 2 since ABI 7. */
int dimn_path_info(dimn_handle h, int32_t* out8);

/* ---- multi-GPU: sub-nets sharded over ranks, RCCL over xGMI (no reference analogue:
 * the reference is single-process, multinet.py:222-223 only sets TF CPU threads) ---- */
#define DIMN_COMM_ID_BYTES 128
int dimn_comm_unique_id(uint8_t* id /* [DIMN_COMM_ID_BYTES] */);
int dimn_comm_init(dimn_handle h, const uint8_t* id, int32_t n_ranks, int32_t rank);
/* What the communicator itself reports: out2 = {ncclCommCount, ncclCommUserRank} (bench.py prints it, so a multi-GPU line
 * names the ranks RCCL really connected).  ABI 5. */
int dimn_comm_info(dimn_handle h, int32_t* out2);
/* In-place sum over ranks of a small host vector (per-epoch val-loss for the global
 * early-stopping decision, multinet.py:242-243). */
int dimn_comm_allreduce_sum(dimn_handle h, double* v, int32_t n);
/* Gather the last dimn_predict_device result of every rank into the full matrix
 * [n_rows][K_global*O] in ROOT's HBM (column block of rank r at r's subnet_offset*O;
 * counts[r] = K_local of rank r): ncclSend/ncclRecv straight to root, one xGMI link per
 * peer.  If out != NULL on root it is also copied to that host buffer. */
int dimn_comm_gather_predictions(dimn_handle h, int64_t n_rows, const int32_t* counts,
                                 int32_t root, float* out);
int dimn_comm_destroy(dimn_handle h);
/* The same gather on ONE GPU, with device-to-device copies where ncclSend / ncclRecv would run: handles[r] plays rank r (all on the
 * root's device, each holding a dimn_predict_device result over n_rows).  Sizing, offsets, the strided placement into
 * [n_rows][K_global*O] and the hand-over to dimn_impute_finish(from_gathered) are dimn_comm_gather_predictions' own code; RCCL
 * itself is not exercised.  Test / bring-up aid for single-GPU machines.  ABI 7. */
int dimn_comm_gather_loopback(const dimn_handle* handles, int32_t n_ranks, int64_t n_rows, const int32_t* counts, int32_t root, float* out);

/* ---- next row: get_distance_matrix (multinet.py:20-34) ------------------------
 * out[g][g] = np.abs(np.corrcoef(X.T)) with NaN -> 0, X host row-major fp64 [n][g] (the candidate
 * predictor columns of the raw counts).  fp64 MFMA on the device, numpy's order of operations.
 * Needs no handle (fit() calls it before the network exists). */
int dimn_abs_corrcoef(int32_t device_id, const double* X, int64_t n, int64_t g, double* out);

/* ---- next row: setPredictors (multinet.py:344-365) fused behind the correlation ----------
 * For every target gene of every sub-net: the `ntop` (<= 16) columns of X with the largest |Pearson r| to it,
 * among the columns that are not targets of the same sub-net; the g x g matrix never leaves the GPU.
 *   targ_pos  [K][O] column positions (into X) of the sub-nets' targets
 *   col_rank  [g]    position of each column in label-sorted order (np.setdiff1d order: the tie rule), < 0 = not a candidate
 *   out_idx   [K][O][ntop] picks, best first, -1 where fewer candidates exist
 * The host turns each sub-net's O*ntop picks into its predictor list (first-occurrence unique, multinet.py:362). */
int dimn_select_predictors(int32_t device_id, const double* X, int64_t n, int64_t g, const int32_t* targ_pos,
                           int32_t K, int32_t O, const int32_t* col_rank, int32_t ntop, int32_t* out_idx);

/* ---- next row: the CSV edges of the CLI, multi-threaded host code (no GPU needed) -------------
 * deepImpute.py:13 pd.read_csv(inputFile, index_col=0) for the input the tool is specified for -- a rectangular matrix of
 * raw integer counts with unquoted labels.  Two passes: dimn_csv_scan gives the shape and the size of the label buffer,
 * dimn_csv_read fills values[n_rows][n_cols] (int64) and `labels` = index name, column labels, row labels, each
 * NUL-terminated.  Returns DIMN_ERR_UNSUP for anything else (decimal / empty / quoted fields, ragged rows): the caller
 * then uses pandas itself, whose float parser this code does not imitate. */
int dimn_csv_scan(const char* path, int64_t* n_rows, int64_t* n_cols, int64_t* label_bytes);
int dimn_csv_read(const char* path, int64_t n_rows, int64_t n_cols, int64_t* values, char* labels, int64_t label_bytes);
/* deepImpute.py:35 imputed.to_csv(output): float64 in Python's repr() form (shortest round trip; scientific iff the decimal
 * exponent is < -4 or >= 16), NaN -> "", labels NUL-separated; byte-identical to DataFrame.to_csv for unquoted labels. */
int dimn_csv_write(const char* path, const double* values, int64_t n_rows, int64_t n_cols, const char* index_name,
                   const char* col_labels, const char* row_labels);

/* ---- the raw count matrix resident on the device (extension of the drop-in; the reference passes the same frame through numpy
 * four times: multinet.py:191 var / mean, :20-34 corrcoef, :216 log1p, :292-303 restore).  ABI 5. ----------
 * dimn_counts_create: raw host [n][g] float64 -> device float32, uploaded ONCE through pinned buffers; every value must be a
 *   non-negative integer <= 2^22 (exact in float32), else DIMN_ERR_UNSUP and the caller keeps the host path.  *vmax = the matrix
 *   maximum (multinet.py:55, :292), *checksum = a position-dependent 64-bit sum of the float64 bit patterns (dimn_counts_checksum
 *   recomputes it from a host frame: "is this the frame that was uploaded?").
 * dimn_counts_select_predictors: dimn_select_predictors with X = columns pool_cols[pool_n] of the resident matrix (converted to
 *   float64 on the device: same numbers, no 8 GB upload).
 * dimn_set_matrix_counts: replaces dimn_set_matrix: norm = lut[count], lut[v] = float32(log1p(v)) for v = 0 .. lut_n - 1 as the
 *   CALLER computed it (numpy on the host: bit-identical to np.log1p(raw).astype(float32), multinet.py:216-217); the handle
 *   remembers the counts, and dimn_impute_finish(raw = NULL) then takes the observed values from them.  The counts object must
 *   outlive every handle bound to it. */
typedef struct dimn_counts_s* dimn_counts;
int dimn_counts_create(int32_t device_id, const double* raw, int64_t n, int64_t g, double* vmax, uint64_t* checksum, dimn_counts* out);
int dimn_counts_checksum(const double* raw, int64_t n, int64_t g, uint64_t* checksum);
/* the same two for a matrix of element type `dtype` (DIMN_DTYPE_F64 / DIMN_DTYPE_I64; ABI 8): an integer frame is uploaded and
 * checked in place -- no float64 copy of it on the host (8 GB at 50k x 20k, 0.07 s to make and 0.35 s to unmap again). */
int dimn_counts_create_typed(int32_t device_id, const void* raw, int32_t dtype, int64_t n, int64_t g, double* vmax, uint64_t* checksum, dimn_counts* out);
int dimn_counts_checksum_typed(const void* raw, int32_t dtype, int64_t n, int64_t g, uint64_t* checksum);
int dimn_counts_destroy(dimn_counts c);
int dimn_counts_select_predictors(dimn_counts c, const int32_t* pool_cols, int64_t pool_n, const int32_t* targ_pos, int32_t K, int32_t O,
                                  const int32_t* col_rank, int32_t ntop, int32_t* out_idx);
/* the same selection as two calls (the matrix product needs only the candidate pool, so it can run while the host still ranks
 * genes): dimn_counts_corr leaves |corr| of the pool on the device, dimn_counts_topk selects from it and frees it */
int dimn_counts_corr(dimn_counts c, const int32_t* pool_cols, int64_t pool_n);
int dimn_counts_topk(dimn_counts c, const int32_t* targ_pos, int32_t K, int32_t O, const int32_t* col_rank, int32_t ntop, int32_t* out_idx);
/* frees what dimn_counts_corr left when the caller's selection takes another path after all (pool_n^2 * 8 bytes).  ABI 7. */
int dimn_counts_corr_drop(dimn_counts c);
int dimn_set_matrix_counts(dimn_handle h, dimn_counts c, const float* lut, int64_t lut_n);
/* ABI 6.  The correlation of resident counts below 65536 runs EXACTLY on the int8 matrix cores (integer numerator and radicands,
 * one rounding each into float64; csrc/dimn_counts_dev.h), larger counts on the float64 kernel; DIMN_CORR_I8=0 forces the latter.
 * dimn_counts_corr_read (tests / diagnostics): the matrix dimn_counts_corr left on the device, out[pool_n][pool_n].
 * dimn_counts_gene_stats: multinet.py:191 `raw.var()` / `raw.mean()` of the resident counts and the column extremes, each [g] or
 *   NULL -- the additions pandas performs, in pandas' order, one thread per gene: equal to DataFrame.mean() / .var() to the bit
 *   (the float32 counts convert to float64 exactly). */
int dimn_counts_corr_read(dimn_counts c, double* out, int64_t pool_n);
int dimn_counts_gene_stats(dimn_counts c, double* mean, double* var, double* cmin, double* cmax);

/* ---- the per-gene statistics fit() orders genes by (multinet.py:191 `raw.var() / (1 + raw.mean())`), host code ----------
 * mean[g], var[g] (ddof 1; NULL: skipped) of the columns of a[n][ld] in pandas' own order of operations (two sequential sums
 * over the rows per column, no fused multiply-add: bit-identical to DataFrame.mean() / .var() on a NaN-free float64 frame),
 * *vmax = the matrix maximum (multinet.py:292), *has_nan != 0: a NaN was seen (the caller then uses pandas: skipna).
 * threads <= 0: up to 64.  ABI 5. */
int dimn_col_stats(const double* a, int64_t n, int64_t g, int64_t ld, double* mean, double* var, double* vmax,
                   int32_t* has_nan, int32_t threads);
/* the same numbers as two calls, so that other work can run between the two sweeps over the matrix: first mean[g], nanvar's own
 * average avg[g], the per-column minimum / maximum, the matrix maximum and the NaN flag; then var[g] from those averages */
int dimn_col_stats_first(const double* a, int64_t n, int64_t g, int64_t ld, double* mean, double* avg, double* cmin, double* cmax,
                         double* vmax, int32_t* has_nan, int32_t threads);
int dimn_col_stats_var(const double* a, int64_t n, int64_t g, int64_t ld, const double* avg, double* var, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif /* DIMN_H */
