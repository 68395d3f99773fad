/* cmoop.h -- C ABI of libcmoop_hip.so: the MI355X-native population-fitness evaluator.
 *
 * This is the drop-in boundary for ONE path of sumansamui/CMOOP_Audio_Processing:
 *     compute_objectives_and_constraints(population) -> evaluate_individual(hparams)
 *       -> build_model / compile / fit / evaluate / predict / calculate_fpr / compute_model_size_mb
 * (reference nsga_penalty.py:225-442, sa_nsga_penalty.py:137-253 and their copies in
 * mobo_penalty.py and ablation_study/).  The reference is pure Python over TensorFlow;
 * it has no FFI of its own, so the functions below are what a ctypes binding of that
 * path binds (INTEGRATION.md shows the stub).  Plain pointers and sizes only: device
 * buffers are raw HIP device pointers (e.g. torch.Tensor.data_ptr()), no torch types.
 *
 * Every function returns 0 on success, non-zero on failure; cmoop_last_error() returns
 * the message of the calling thread's last failure.  The reference's own error
 * convention is "any exception kills the run" (no try/except around
 * nsga_penalty.py:383); the Python shim re-raises these codes as RuntimeError.
 */
#ifndef CMOOP_H
#define CMOOP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMOOP_ABI_VERSION 3

/* topologies hidden behind the reference's build_model(hparams) */
#define CMOOP_VARIANT_A 0 /* "deep":    nsga_penalty.py:225-334, mobo_penalty.py:128-194 */
#define CMOOP_VARIANT_B 1 /* "shallow": sa_nsga_penalty.py:137-177 and the sa_/psi_/init_ ablations */

/* calculate_fpr variants */
#define CMOOP_FPR_V1 0       /* nsga_penalty.py:351-364 (= sa_nsga_penalty.py:189-202) */
#define CMOOP_FPR_V1_QUIRK 1 /* V1 with y_true = argmax of an (N,1) array == zeros, nsga_penalty.py:387 */
#define CMOOP_FPR_V3 2       /* ablation_study/sa_nsga_local.py:138-141 */

/* arithmetic of the MFMA conv/dense GEMM kernels (first conv, C_in = 1, and the classifier layer are always fp32).
 * The reference trains in fp32 (TF default, no mixed-precision policy set anywhere); DEFAULT/FP32 is that.
 * BF16X3 and BF16 are opt-in: BF16X3 = fp32 operands split exactly into 3 bf16, six bf16 MFMA terms
 * (fp32-accurate); BF16 = operands rounded to bf16, fp32 accumulation (BASELINE.json configs[4] "bf16 train"). */
#define CMOOP_GEMM_DEFAULT 0 /* exact fp32 unless the environment variable CMOOP_GEMM_MODE=bf16x3|bf16 overrides */
#define CMOOP_GEMM_BF16X3 2
#define CMOOP_GEMM_BF16 3

/* Evaluation protocol: the module-level constants the reference's evaluate_individual
 * closes over (nsga_penalty.py:176-179) plus the Keras defaults it relies on. */
typedef struct cmoop_config {
    int32_t variant;       /* CMOOP_VARIANT_* */
    int32_t classes;       /* CLASSES           nsga_penalty.py:176 (10) / sa_nsga_penalty.py:102 (11) */
    int32_t epochs;        /* EPOCHS            :177 */
    int32_t batch;         /* BATCH_SIZE        :178 */
    int32_t patience;      /* PATIENCE          :179 */
    int32_t early_stop;    /* 1: EarlyStopping(monitor='val_loss') (:382); 0: run all epochs (throughput mode) */
    int32_t restore_best;  /* restore_best_weights: 0 in nsga_penalty.py:382, 1 in sa_nsga_penalty.py:215 */
    int32_t acc_readout;   /* 0: history['val_accuracy'][-1] (:384); 1: model.evaluate (sa_nsga_penalty.py:219) */
    int32_t fpr_variant;   /* CMOOP_FPR_* */
    int32_t shuffle;       /* Model.fit shuffle=True default */
    int32_t eval_batch;    /* rows per inference launch (results do not depend on it) */
    int32_t n_slots;       /* candidates in flight per GPU, each on its own HIP stream */
    int32_t profile_every; /* >0: HIP-event-time the MFMA GEMM launches of every n-th train step */
    int32_t gemm_mode;     /* CMOOP_GEMM_*: arithmetic of the MFMA conv/dense GEMMs (0 = library default = exact fp32) */
    double lr;             /* 1e-3: optimizer='adam' (:377); LEARNING_RATE (:162) is unused by the reference */
    double beta1, beta2, adam_eps; /* Keras Adam defaults .9 / .999 / 1e-7 */
    double bn_eps, bn_momentum;    /* Keras BatchNormalization defaults 1e-3 / .99 */
    double dropout;        /* 0.3 (nsga_penalty.py:323) */
} cmoop_config;

/* The module globals X_train, y_train, X_validation, y_validation (nsga_penalty.py:167),
 * resident in HBM: features [n, T, F] fp32 (the trailing channel axis of :151-153 is
 * implicit), labels [n] int32. */
typedef struct cmoop_dataset {
    const float* x_train;
    const int32_t* y_train;
    int64_t n_train;
    const float* x_val;
    const int32_t* y_val;
    int64_t n_val;
    int32_t T, F;
} cmoop_dataset;

int cmoop_abi_version(void);
const char* cmoop_last_error(void);
void cmoop_config_default(cmoop_config* cfg);

/* compute_model_size_mb's count_params() without building a model (nsga_penalty.py:337-344);
 * gene = {filters, kernel_size, use_bn, residual_blocks, fc_layers, use_dropout}. */
int cmoop_param_count(const int32_t gene[6], int32_t variant, int32_t classes, int64_t* out);
int cmoop_fwd_flops(const int32_t gene[6], int32_t variant, int32_t classes, int32_t T, int32_t F, double* out);

/* compute_objectives_and_constraints' inner loop (nsga_penalty.py:426-427): evaluate n
 * candidates; outputs are host arrays of length n (any may be NULL). */
int cmoop_eval_population(const cmoop_config* cfg, const cmoop_dataset* ds, const int32_t* genes /* [n][6] */,
                          const uint32_t* seeds /* [n] */, int32_t n, double* acc, double* size_mb, double* fpr,
                          int32_t* epochs_run, double* val_loss, double* seconds);

/* The same loop drained through a caller-supplied queue: the library's worker threads (cfg.n_slots of them,
 * concurrently) call next(ctx) for the index in [0,n) of the next candidate to train; a negative return ends that
 * worker.  Several processes (one per GPU) that share one counter therefore drain ONE longest-first queue, which
 * balances the early-stopped protocol where epochs run are unknown in advance (the reference's loop is serial,
 * nsga_penalty.py:426-427; it has no counterpart).  evaluated[i] = 1 for the candidates this call trained; the
 * other outputs of un-evaluated candidates are left untouched. */
typedef int32_t (*cmoop_next_fn)(void* ctx);
int cmoop_eval_population_pull(const cmoop_config* cfg, const cmoop_dataset* ds, const int32_t* genes /* [n][6] */,
                               const uint32_t* seeds /* [n] */, int32_t n, cmoop_next_fn next, void* ctx, double* acc,
                               double* size_mb, double* fpr, int32_t* epochs_run, double* val_loss, double* seconds,
                               int32_t* evaluated /* [n], required */);

/* host-only: does every conv layer of this candidate at `batch` rows per launch (pass max(batch, eval_batch)) stay inside
 * the kernels' 32-bit byte offsets (each activation / kernel tensor below 2^29 elements)?  Non-zero + message if not;
 * cmoop_net_create and the population calls make the same check before they allocate anything. */
int cmoop_plan_check(const int32_t gene[6], int32_t variant, int32_t T, int32_t F, int32_t batch);

/* host-only: the launch-path variant (kernel instantiation as rocprofv3 names it + "+sk" / "+stats" / "+tab" / "+slabs",
 * see cmoop_profile_variant) the TRAINER uses for one conv layer at this batch: op 0 forward (want_stats: the layer
 * feeds a BatchNorm), 1 dgrad, 2 wgrad.  Pure arithmetic on the shape -- tests enumerate every layer of every gene with
 * it and require a GPU parity case for each variant. */
int cmoop_conv_launch_plan(int32_t op, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride,
                           int32_t want_stats, char* name, int32_t name_cap);

/* host-only: the halo-tiled direct convolution (stride-1 KS x KS layers) stages the input rows of a 128- / 256-pixel tile in LDS;
 * rows_bound = rows its LDS image is sized for (closed form), rows_needed = the most rows any tile of this geometry really
 * spans, rows_stageable = rows the per-thread staging slots can hold.  All 0 when the geometry runs on the implicit GEMM.
 * Tests sweep geometries and require rows_needed <= rows_bound <= rows_stageable. */
int cmoop_halo_tile_check(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t* rows_bound,
                          int32_t* rows_needed, int32_t* rows_stageable);

/* host-only: number of row slices the weight-gradient kernel splits a conv/dense layer into (workspace sizing;
 * NOT monotone in B -- tests pin that the trainer sizes its slab workspace for the worst batch 1..cfg.batch) */
int cmoop_wgrad_slices(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t* out);

/* calculate_fpr on host label arrays (nsga_penalty.py:351-364 and variants) */
int cmoop_calculate_fpr(const int32_t* y_true, const int32_t* y_pred, int64_t n, int32_t classes, int32_t fpr_variant,
                        double* out);

/* ---- audio front end (north-star addition; the reference loads pre-extracted features,
 *      nsga_penalty.py:64-71) and prepare_dataset's StandardScaler (nsga_penalty.py:103-141) */
int cmoop_logmel(const float* wav_dev /* [n_clips][n_samples] */, int64_t n_clips, int32_t n_samples,
                 float* out_dev /* [n_clips][1+n_samples/160][40] */);
/* optional MFCC features: DCT-II, ortho-normalised, along the mel axis of log-mel rows; first n_mfcc coefficients.
 * The reference ships no front end; its comment at ablation_study/sa_nsga_init.py:68 calls the stored features MFCCs. */
int cmoop_mfcc(const float* logmel_dev /* [rows][n_mels] */, int64_t rows, int32_t n_mels, int32_t n_mfcc,
               float* out_dev /* [rows][n_mfcc] */);
int cmoop_standardize_fit(const float* x_dev, int64_t rows, int32_t cols, double* mean_host, double* scale_host);
int cmoop_standardize_apply(float* x_dev, int64_t rows, int32_t cols, const double* mean_host, const double* scale_host);

/* ---- HIP-event profile of the MFMA GEMM kernels sampled during cmoop_eval_population
 *      (cfg.profile_every): one entry per kernel instantiation, named as rocprofv3 names it */
int cmoop_profile_reset(void);
int cmoop_profile_count(int32_t* out);
int cmoop_profile_entry(int32_t i, char* name, int32_t name_cap, int64_t* launches, double* total_ms, double* total_flops);
/* launch-path variants of the sampled launches: instantiation name + "+sk" (split-K slabs and combine) / "+bal" /
 * "+stats" (BatchNorm statistics in the epilogue) / "+tab" (row-table operand loader) / "+slabs" (wgrad row slices) */
int cmoop_profile_variant_count(int32_t* out);
int cmoop_profile_variant(int32_t i, char* name, int32_t name_cap);

/* ---- single-candidate session (parity tests, smoke): one net on the library's stream */
typedef struct cmoop_net cmoop_net;
int cmoop_net_create(const int32_t gene[6], const cmoop_config* cfg, int32_t T, int32_t F, uint32_t seed, cmoop_net** out);
int cmoop_net_destroy(cmoop_net* net);
int cmoop_net_total_params(cmoop_net* net, int64_t* out);
int cmoop_net_get_params(cmoop_net* net, float* host);       /* canonical order, see genes.py */
int cmoop_net_set_params(cmoop_net* net, const float* host);
int cmoop_net_get_grads(cmoop_net* net, float* host);
int cmoop_net_train_step(cmoop_net* net, const float* x_dev, const int32_t* y_dev, const int32_t* idx_dev, int64_t row0,
                         int32_t B);
int cmoop_net_evaluate(cmoop_net* net, const float* x_dev, const int32_t* y_dev, int64_t n, double* loss_sum,
                       int64_t* correct, int32_t* preds_dev);
int cmoop_net_train_metrics(cmoop_net* net, double* loss_sum, int64_t* correct, int32_t reset);
/* Full training state of the net (host arrays of cmoop_net_total_params floats; any pointer may be NULL): parameters in
 * canonical order INCLUDING the BatchNorm moving statistics, Adam's m and v in the same layout (zero in the
 * non-trainable slots), optimizer.iterations and the global train-step count that keys the dropout masks.  With these a
 * checker can re-synchronise with the GPU at every epoch boundary of Model.fit (nsga_penalty.py:383) instead of
 * comparing two long, chaotic fp32 trajectories at their ends. */
int cmoop_net_get_state(cmoop_net* net, float* params, float* adam_m, float* adam_v, int64_t* iterations, int64_t* steps);
int cmoop_net_set_state(cmoop_net* net, const float* params, const float* adam_m, const float* adam_v, int64_t iterations,
                        int64_t steps);
/* rows the resident training tensor holds: gathered row indices are clamped into [0, n_rows) (0 = unknown, no clamp) */
int cmoop_net_set_gather_rows(cmoop_net* net, int64_t n_rows);
/* ONE epoch of Model.fit on the trainer's own path: epoch permutation of (seed, epoch) computed on the device when
 * cfg.shuffle, ceil(n_train / batch) steps driven by the device-resident step state, last partial batch kept. */
int cmoop_net_run_epoch(cmoop_net* net, const float* x_train_dev, const int32_t* y_train_dev, int64_t n_train, int32_t epoch);
/* evaluate_individual's fit + read-outs (nsga_penalty.py:377-392) on THIS net -- the body of cmoop_eval_population's
 * per-candidate work -- with the per-epoch validation history (first hist_cap epochs), the epoch EarlyStopping took its
 * best weights from (-1: none / early_stop off) and the epochs run. */
int cmoop_net_fit(cmoop_net* net, const cmoop_dataset* ds, int32_t hist_cap, double* val_loss_hist, double* val_acc_hist,
                  int32_t* epochs_run, int32_t* best_epoch, double* acc, double* fpr, double* val_loss);
int cmoop_epoch_permutation(uint32_t seed, uint32_t epoch, int64_t n, int32_t* out_host);
/* the same permutation computed on the GPU (what the trainer uses: no host sort / H2D copy per epoch); n <= 262144 */
int cmoop_epoch_permutation_device(uint32_t seed, uint32_t epoch, int64_t n, int32_t* out_dev);

/* ---- kernel-level entry points (parity tests and the roofline leg of bench.py).
 *      conv: y[B,OH,OW,Cout] = SAME-conv(x[B,H,W,Cin], w[Cout][KS][KS][Cin]) + bias, optional ReLU */
int cmoop_conv_fwd(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int32_t B, int32_t H,
                   int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t relu);
/* dx = dgrad(dy) (optionally masked by x > 0), dw[Cout][KS][KS][Cin], db[Cout] */
int cmoop_conv_bwd(const float* x_dev, const float* w_dev, const float* dy_dev, float* dx_dev, float* dw_dev, float* db_dev,
                   int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t mask_relu);
/* The same two operations launched EXACTLY as the trainer launches them (Net::forward / Net::backward): row-table operand
 * loader, split-K workspace, flip-transposed dgrad operand prepared up front, weight-gradient slabs + fixed-order slice
 * sum, and -- forward, col_sum != NULL -- the BatchNorm batch statistics taken in the conv epilogue (per-tile partials
 * summed here in float64 into col_sum / col_sumsq [Cout]; *stats_fused = 0 when the launch was split-K and the
 * statistics came from the stand-alone reduction, as in the trainer).  Cin must be a power of two >= 16. */
int cmoop_conv_fwd_trainer(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int32_t B, int32_t H,
                           int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t relu, double* col_sum,
                           double* col_sumsq, int32_t* stats_fused);
int cmoop_conv_bwd_trainer(const float* x_dev, const float* w_dev, const float* dy_dev, float* dx_dev, float* dw_dev,
                           float* db_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride,
                           int32_t mask_relu);
/* kernel instantiations (+ launch-path variant suffixes, see cmoop_profile_variant) the calling thread's last
 * cmoop_conv_fwd* / cmoop_conv_bwd* call launched, ';'-separated */
int cmoop_last_kernels(char* buf, int32_t cap);
/* average ms per launch over `iters` back-to-back launches (HIP events on the library stream);
 * mode 0: forward implicit GEMM, 1: dgrad implicit GEMM (y holds dY, x receives dX), 2: wgrad MFMA kernel */
int cmoop_conv_time(int32_t mode, const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int32_t B,
                    int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t iters, double* avg_ms);
/* MLP-head layers (Dense+ReLU ladder, nsga_penalty.py:306-330) on the dense.hip kernels: y[M][N] = x[M][K] w[N][K]^T + b
 * (optional ReLU); dx (optionally masked by x > 0), dw[N][K], db[N].  K a multiple of 16. */
int cmoop_dense_fwd(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int32_t M, int32_t N, int32_t K,
                    int32_t relu);
int cmoop_dense_bwd(const float* x_dev, const float* w_dev, const float* dy_dev, float* dx_dev, float* dw_dev, float* db_dev,
                    int32_t M, int32_t N, int32_t K, int32_t mask_relu);
int cmoop_maxpool_fwd(const float* x_dev, float* y_dev, uint8_t* arg_dev, int32_t B, int32_t H, int32_t W, int32_t C);
int cmoop_maxpool_bwd(const float* dy_dev, const uint8_t* arg_dev, const float* y_dev, float* dx_dev, int32_t B, int32_t H,
                      int32_t W, int32_t C, int32_t mask_y_pos);
int cmoop_device_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif /* CMOOP_H */
