// Host-side launchers of every HIP kernel in libcmoop_hip.so (gfx950 / CDNA4).
// All tensors are fp32, activations NHWC, conv/dense kernels in the canonical
// [C_out][kh][kw][C_in] layout (K = kh*kw*C_in contiguous) -- see genes.py.
#pragma once
#include "common.h"
#include <string>

namespace cmoop {

// ---------------------------------------------------------------------------
// Implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_16x16x4_f32).
// Replaces the Keras Conv2D / Dense fwd+bwd the reference delegates to TF
// (nsga_penalty.py:255-330 via Model.fit, :383).
// ---------------------------------------------------------------------------
struct ConvGeom {
    int B, H, W, Cin;        // input  [B,H,W,Cin]
    int OH, OW, Cout;        // output [B,OH,OW,Cout]
    int KH, KW, stride;
    int pad_t, pad_l;        // TF "SAME": total//2 on top/left, remainder bottom/right
    int M() const { return B * OH * OW; }
    int K() const { return KH * KW * Cin; }
};

// Arithmetic of the MFMA GEMM kernels.  GEMM_FP32 (exact v_mfma_f32_16x16x4_f32) is the product default;
// the bf16 matrix-core modes are opt-in (cmoop_config.gemm_mode or CMOOP_GEMM_MODE=bf16x3|bf16), see gemm.hip.
enum GemmMode { GEMM_DEFAULT = -1, GEMM_FP32 = 0, GEMM_FP32_DMA = 1 /* kernel-internal: fp32 with LDS-DMA operand loads */, GEMM_BF16X3 = 2, GEMM_BF16 = 3,
                GEMM_FP32_HALO = 4 /* kernel-internal: fp32, halo-tiled direct convolution (halo_fwd_kernel) */ };
int gemm_mode_default();   // CMOOP_GEMM_MODE, else GEMM_FP32

struct GemmEpilogue {
    int mode = GEMM_DEFAULT;       // GemmMode of this launch (GEMM_DEFAULT: gemm_mode_default())
    const float* bias = nullptr;   // + bias[col]
    int relu = 0;                  // max(v, 0)
    const float* mask = nullptr;   // v = mask[out] > 0 ? v * mask_scale : 0   (ReLU/dropout backward)
    float mask_scale = 1.f;
    int accumulate = 0;            // out += v
    int out_stride = 1;            // >1: row (b,oh,ow) is stored at (b, oh*s, ow*s) of [B,OHf,OWf,N]
    int OHf = 0, OWf = 0;
    int dropout = 0;               // inverted dropout keyed by fmix32(drop_prefix ^ (row*N+col))
    uint32_t drop_prefix = 0, drop_thr = 0;
    float drop_scale = 1.f;
    // optional: column partials (sum, sum of squares) of the stored output, one per M tile: [tiles][2][Cout] floats --
    // the BatchNorm batch statistics computed in the producing conv's epilogue.  Needs room for cdiv(M, 64) tiles.
    float* stats = nullptr;
};

// Y[m][n] = sum_k im2col(X)[m][k] * Wt[n][k]  (+ epilogue).  Cin must be a power of two >= 16.
// optional start/stop events.  ext: filled with the kernel's own begin/end by hipExtLaunchKernelGGL
// (exact even with other streams in flight); !ext: plain hipEventRecord pair around the launch
// (what works under rocprofv3, whose tool library crashes on ext launches in ROCm 7.2).
struct GemmTiming {
    hipEvent_t start = nullptr, stop = nullptr;
    bool ext = true;
};
// returns the instantiation code mode*1e8 + BM*100000 + BN*100 + BK of the kernel that was launched
// splitk_ws (optional, >= igemm_splitk_workspace(g) floats): lets under-filled grids split the K axis
// stats_blocks (with e.stats): receives the number of M-tile partials written, 0 when the launch was split-K (no fused statistics)
// rowtab / tab_rows (optional): the row table of THIS geometry (launch_build_rowtab), covering at least every 128-row tile
// the launch touches: the kernel then reads each row's offset / padding mask instead of deriving them
// flags_out (optional): GEMM_FLAG_* of the path the launch took (parity-coverage bookkeeping)
enum GemmFlags { GEMM_FLAG_SPLITK = 1, GEMM_FLAG_STATS = 2, GEMM_FLAG_ROWTAB = 4, GEMM_FLAG_BALANCED = 8, GEMM_FLAG_SLABS = 16 };
int launch_igemm_fwd(const float* X, const float* Wt, float* Y, const ConvGeom& g,
                     const GemmEpilogue& e, hipStream_t s, const GemmTiming* tm = nullptr,
                     float* splitk_ws = nullptr, size_t splitk_ws_floats = 0, int* stats_blocks = nullptr,
                     const void* rowtab = nullptr, int tab_rows = 0, int* flags_out = nullptr);
size_t igemm_splitk_workspace(const ConvGeom& g);
// host-only twins of the two launchers' choices (no HIP call): the instantiation code and GemmFlags a launch of this
// geometry takes -- ws_floats: split-K workspace the caller would pass (0: none); S: wgrad row slices (wgrad_slices)
int igemm_fwd_plan(const ConvGeom& g, const GemmEpilogue& e, size_t ws_floats, bool want_stats, bool have_rowtab, int* flags_out);
int igemm_wgrad_plan(const ConvGeom& g, int S, int mode, bool have_rowtab, int* flags_out);
// host-only (tests): rows the halo kernel's LDS image is sized for, the most rows any tile of this geometry spans, and the rows
// the per-thread staging slots can hold -- need <= bound <= items_cap must hold for every eligible geometry; all 0 if not eligible
void halo_rows_bound_and_need(const ConvGeom& g, int* bound, int* need, int* items_cap);
// throws when a tensor of this geometry is beyond the kernels' 32-bit BYTE offsets (buffer descriptors, row tables):
// B*H*W*Cin (+ padding bias) and M*Cout must stay below 2^29 elements
void igemm_check_range(const ConvGeom& g);
// kernel instantiation name as rocprofv3 prints it; cls 0: launch_igemm_fwd's return code, 1: launch_igemm_wgrad's
std::string gemm_kernel_name(int cls, int code);

// dWt[n][k] = sum_m dY[m][n] * im2col(X)[m][k], split over S row-slices into P[S][N][K].
int wgrad_slices(const ConvGeom& g);
// returns the instantiation code mode*1e7 + (64-row chunks ? 1e6 : 0) + BCO*1000 + BKI of the kernel that was launched
// Pbias (optional): [S][N] per-slice column sums of dY (the bias gradient), fused into the first K tile's blocks
// slab_stride: floats between consecutive slices of P and of Pbias (0 = N*K, bias slabs packed [S][N]);
// the trainer lays slices out as [S][N*K + N] so one reduction yields kernel and bias gradients.
// rowtab (optional, fp32 kernel): the layer's row table (launch_build_rowtab, tab_rows = rowtab_rows(geometry it was
// built for) >= this launch's rows rounded up to 32) -- switches the gather to buffer loads without per-element
// divisions / bounds arithmetic.  A table built for the full batch serves every smaller batch of the same layer.
int launch_igemm_wgrad(const float* X, const float* dY, float* P, const ConvGeom& g, int S, hipStream_t s,
                       const GemmTiming* tm = nullptr, float* Pbias = nullptr, size_t slab_stride = 0,
                       int mode = GEMM_DEFAULT, const void* rowtab = nullptr, int tab_rows = 0, int* flags_out = nullptr);
int rowtab_rows(const ConvGeom& g);                                           // entries (8 bytes each) the table needs
void launch_build_rowtab(const ConvGeom& g, void* tab, hipStream_t s);        // needs KH*KW <= 32
// out[i] = sum_s P[s][i]  (fixed order -> deterministic)
void launch_reduce_slices(const float* P, float* out, int S, int64_t n, hipStream_t s, int64_t stride = 0);
// Wd[ci][KH-1-kh][KW-1-kw][co] = W[co][kh][kw][ci]   (operand of the dgrad implicit GEMM)
void launch_flip_transpose(const float* W, float* Wd, int Cout, int KH, int KW, int Cin, hipStream_t s);
// the same for every conv layer of a net in ONE launch per train step (table rows in device memory)
struct FlipEntry { int64_t w_off, wd_off; int Cout, KH, KW, Cin; };
void launch_flip_transpose_all(const float* params, float* wd_all, const FlipEntry* table_dev, int layers, int64_t max_elems,
                               hipStream_t s);

// ---------------------------------------------------------------------------
// Device-resident step state of a candidate's training run.  Every per-step quantity a kernel needs (the batch's first
// row in the epoch permutation, the step counter that keys the dropout masks, the optimiser iteration that selects
// Adam's bias-corrected step size) is read from here, so a train step's launch sequence has NO host-side arguments
// that change from step to step and can be captured once as a hipGraph and replayed (net.hip).
// ---------------------------------------------------------------------------
struct StepState {
    long long row0;        // first row of the current batch in idx (advanced by the batch size after every step)
    unsigned step;         // global train step (dropout counter)
    unsigned iter;         // optimizer.iterations BEFORE this step's update (alpha_table[iter] is its step size)
};
void launch_step_advance(StepState* st, int batch, hipStream_t s);
// w[i] = (float)(2 * (fmix32(prefix ^ i) >> 8) - 2^24) * scale: the seeded glorot-uniform twin of oracle/rng.py; constant fill
void launch_glorot_init(float* w, int64_t n, uint32_t prefix, float scale, hipStream_t s);
void launch_fill(float* w, float v, int64_t n, hipStream_t s);

// ---------------------------------------------------------------------------
// Dense layers of the MLP head (dense.hip): M = batch rows, K = C_in (multiple of 16), N = units (any).
// One workgroup per 16x16 output tile, operands read straight from global memory in the MFMA lane layout,
// fixed-order 4-wave reduction: no split-K slabs, no flip-transposed weights, no slice reduction.
// mode GEMM_BF16 rounds both operands to bf16 (fp32 accumulation); every other mode is exact fp32.
// ---------------------------------------------------------------------------
// dropout: st == null -> mask keyed by drop_prefix; st != null -> by rng_prefix(drop_seed, drop_stream, st->step) (graph replay)
void launch_dense_fwd(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int relu,
                      int dropout, uint32_t drop_prefix, uint32_t drop_thr, float drop_scale, int mode, hipStream_t s,
                      const StepState* st = nullptr, uint32_t drop_seed = 0, uint32_t drop_stream = 0);
// dX[m][k] = sum_n dY[m][n] W[n][k]; mask != null: dX = mask > 0 ? dX * mask_scale : 0 (ReLU / dropout backward of the layer's input)
void launch_dense_dgrad(const float* dY, const float* W, float* dX, int M, int N, int K, const float* mask, float mask_scale,
                        int mode, hipStream_t s);
// dW[n][k] = sum_m dY[m][n] X[m][k], dB[n] = sum_m dY[m][n]
void launch_dense_wgrad(const float* X, const float* dY, float* dW, float* dB, int M, int N, int K, int mode, hipStream_t s);
// both of the above in one launch (same arithmetic, bit-identical results): the trainer's backward of a hidden dense layer
void launch_dense_bwd(const float* X, const float* dY, const float* W, float* dW, float* dB, float* dX, int M, int N, int K,
                      const float* mask, float mask_scale, int mode, hipStream_t s);

// ---------------------------------------------------------------------------
// First layer (C_in = 1, K = 9 or 25: too small for MFMA) -- direct conv on the VALU.
// X is the resident feature tensor [N_total, H, W]; `idx` (may be null) gathers the
// batch rows, fusing Keras' shuffle+batch gather (nsga_penalty.py:383) into the load.
// ---------------------------------------------------------------------------
// st (optional): the batch's first row comes from st->row0 instead of row0
// n_rows (optional, > 0): rows the resident tensor holds -- a gathered row index is clamped into [0, n_rows) so that a
// corrupt idx can never address outside X (the only producer of idx is the device permutation; this is a fault fence)
// stats / stats_blocks (optional): column partials (sum, sum of squares) of the stored output, [blocks][2][Cout] -- written by the
// matrix-core form only (*stats_blocks = 0 otherwise: the caller then runs the stand-alone reduction)
void launch_conv1_fwd(const float* X, const int32_t* idx, int64_t row0, const float* Wt, const float* bias,
                      float* Y, int B, int H, int W, int Cout, int KS, int relu, hipStream_t s, const StepState* st = nullptr,
                      int64_t n_rows = 0, float* stats = nullptr, int* stats_blocks = nullptr);
int conv1_wgrad_blocks(int B, int H, int W);
// P[blk][Cout*(KS*KS) + Cout]: per-block partial kernel grads then bias grads
void launch_conv1_wgrad(const float* X, const int32_t* idx, int64_t row0, const float* dY, float* P,
                        int B, int H, int W, int Cout, int KS, hipStream_t s, const StepState* st = nullptr, int64_t n_rows = 0);

// ---------------------------------------------------------------------------
// Per-channel reductions over the M rows of an [M][C] tensor (C % 4 == 0).
// Two-stage and order-fixed: `blocks` partials then a double-precision finalize.
// ---------------------------------------------------------------------------
int colreduce_blocks(int64_t M, int C);
// P[blk][2][C] = (sum x, sum x^2)
void launch_colstats(const float* X, float* P, int64_t M, int C, int blocks, hipStream_t s);
// P[blk][2][C] = (sum dy, sum dy*xhat), xhat = (x-mean)*invstd
void launch_bn_bwd_reduce(const float* dY, const float* X, const float* mean, const float* invstd,
                          float* P, int64_t M, int C, int blocks, hipStream_t s);
// BN train finalize: batch mean / biased var -> (mean, invstd, scale, shift), moving stats update
void launch_bn_finalize(const float* P, int blocks, int64_t M, int C, const float* gamma, const float* beta,
                        float* moving_mean, float* moving_var, float* mean, float* invstd, float* scale,
                        float* shift, float eps, float momentum, float one_minus_momentum, hipStream_t s);
// BN inference: scale/shift from the moving statistics
void launch_bn_eval_prepare(const float* gamma, const float* beta, const float* moving_mean,
                            const float* moving_var, float* scale, float* shift, int C, float eps, hipStream_t s);
// y = x*scale[c] + shift[c]  (optional ReLU)
void launch_scale_shift(const float* X, float* Y, const float* scale, const float* shift, int64_t M, int C,
                        int relu, hipStream_t s);
// BN backward apply: dgamma/dbeta from the partials (written by block 0), then
// dx = gamma*invstd*(dy - sum_dy/M - xhat*sum_dyxhat/M), optionally masked by (x > 0)
void launch_bn_bwd_apply(const float* dY, const float* X, const float* mean, const float* invstd,
                         const float* gamma, const float* P, int blocks, float* dX, float* dgamma, float* dbeta,
                         int64_t M, int C, int mask_x_pos, hipStream_t s);
// output layer helpers (C_out = classes: not a multiple of 4 / power of two)
void launch_colsum_small(const float* X, float* out, int M, int C, hipStream_t s);
void launch_dense_dgrad_small(const float* dY, const float* W, float* dX, int M, int N, int K, const float* mask,
                              float scale, hipStream_t s);
// out[c] = sum_blk P[blk][0][c]  (bias gradients)
void launch_colsum_finalize(const float* P, int blocks, int C, float* out, hipStream_t s);

// ---------------------------------------------------------------------------
// Pool / residual / GAP / loss / optimiser
// ---------------------------------------------------------------------------
void launch_maxpool_fwd(const float* X, float* Y, uint8_t* arg, int B, int H, int W, int C, hipStream_t s);
void launch_maxpool_bwd(const float* dY, const uint8_t* arg, const float* Y, float* dX, int B, int H, int W, int C,
                        int mask_y_pos, hipStream_t s);
// BatchNorm-apply (+ReLU) + MaxPool in one pass and the matching backward (the full-resolution normalised tensor and
// its gradient are never materialised); results bit-identical to scale_shift + maxpool_fwd / maxpool_bwd + bn_bwd_*.
void launch_bn_pool_fwd(const float* X, float* Y, uint8_t* arg, const float* scale, const float* shift, int B, int H, int W,
                        int C, int relu, hipStream_t s);
void launch_bn_pool_bwd_reduce(const float* g_pooled, const uint8_t* arg, const float* X, const float* mean, const float* invstd,
                               float* P, int B, int H, int W, int C, int blocks, hipStream_t s);
void launch_bn_pool_bwd_apply(const float* g_pooled, const uint8_t* arg, const float* X, const float* mean, const float* invstd,
                              const float* gamma, const float* P, int blocks, float* dX, float* dgamma, float* dbeta, int B,
                              int H, int W, int C, int mask_x_pos, hipStream_t s);
void launch_add_relu(const float* A, const float* Bt, float* Y, int64_t n, hipStream_t s);
void launch_gap_fwd(const float* X, float* Y, int B, int HW, int C, hipStream_t s);
void launch_gap_bwd(const float* dY, const float* X, float* dX, int B, int HW, int C, hipStream_t s);
// softmax + clipped sparse CE (+ gradient wrt logits when dZ != null); adds into
// acc[0] (double: sum of per-sample losses) and acc[1] (as int64: correct); writes preds when non-null.
void launch_softmax_ce(const float* Z, const int32_t* labels, const int32_t* idx, int64_t row0, int B, int C,
                       float* dZ, double* acc, int32_t* preds, hipStream_t s, const StepState* st = nullptr, int64_t n_rows = 0);
// st != null: alpha = alpha_table[st->iter] (host-precomputed per iteration: the Keras step size in double, rounded once)
void launch_adam(float* w, const float* g, float* m, float* v, int64_t n, float alpha, float c1, float c2,
                 float eps, hipStream_t s, const StepState* st = nullptr, const float* alpha_table = nullptr);
// One optimiser launch for the whole parameter arena that ALSO finishes the weight gradients: the arena is cut into
// segments; a plain segment reads its gradient from g[], a slab segment sums the S row-slice partials its weight-gradient
// kernel left in `slab` (same fixed order as reduce_slices_kernel), stores the sum to g[] and applies Adam to it.
struct AdamSeg {
    int64_t off = 0, n = 0;        // arena range [off, off + n)
    const float* slab = nullptr;   // null: plain segment
    int64_t stride = 0;            // floats between consecutive slices
    int32_t S = 0;                 // slices
    int32_t block0 = 0;            // first workgroup of the segment
};
constexpr int ADAM_MAX_SEGS = 64;
struct AdamSegTable {
    int32_t count = 0, blocks = 0;
    AdamSeg seg[ADAM_MAX_SEGS];
};
// fills block0 / blocks from off, n, slab of the first `count` entries (segments must tile the arena in order)
void adam_segments_finalize(AdamSegTable& tab);
void launch_adam_segments(float* w, float* g, float* m, float* v, const AdamSegTable& tab, float alpha, float c1, float c2,
                          float eps, hipStream_t s, const StepState* st = nullptr, const float* alpha_table = nullptr);
// device twin of epoch_permutation (net.h): out[rank of key_i] = i; n <= EPOCH_PERMUTATION_DEVICE_MAX (O(n^2) rank sort)
constexpr int64_t EPOCH_PERMUTATION_DEVICE_MAX = 262144;
void launch_epoch_permutation(uint32_t seed, uint32_t epoch, int64_t n, int32_t* out, hipStream_t s);
void launch_confusion(const int32_t* y_true, const int32_t* y_pred, int64_t n, int C, int force_true_zero,
                      int64_t* cm, hipStream_t s);

// ---------------------------------------------------------------------------
// Audio front end (north-star addition; no reference counterpart)
// ---------------------------------------------------------------------------
struct FrontendCfg {
    int sr = 16000, n_fft = 512, win = 400, hop = 160, n_mels = 40;
    float fmin = 20.f, fmax = 7600.f, log_eps = 1e-6f;
};
struct FrontendTables;   // device-resident twiddles / window / sparse mel weights
FrontendTables* frontend_tables_create(const FrontendCfg& c);
void frontend_tables_destroy(FrontendTables* t);
void launch_logmel(const float* wav, int64_t n_clips, int n_samples, float* out, const FrontendTables* t, hipStream_t s);
void launch_mfcc(const float* X, float* Y, int64_t rows, int n_mels, int n_mfcc, hipStream_t s);   // DCT-II ortho along the mel axis
void launch_standardize(float* X, const double* mean, const double* scale, int64_t rows, int C, hipStream_t s);
void colstats_finalize_f64(const float* P, int blocks, int64_t M, int C, double* mean, double* scale, hipStream_t s);

}  // namespace cmoop
