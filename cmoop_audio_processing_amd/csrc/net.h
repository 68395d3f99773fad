// One candidate CNN resident on the GPU: plan (from the six genes), parameter /
// Adam arenas, forward, backward, optimiser step, inference.  The MI355X-native
// replacement of build_model + model.fit/evaluate/predict
// (/root/reference/nsga_penalty.py:225-334,375-388; sa_nsga_penalty.py:137-177,211-221).
#pragma once
#include "kernels.h"
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

namespace cmoop {

struct NetConfig {
    int variant = 0, classes = 10, epochs = 300, batch = 64, patience = 5;
    int early_stop = 1, restore_best = 0, acc_readout = 0, fpr_variant = 0, shuffle = 1;
    int eval_batch = 256, n_slots = 8, profile_every = 0;
    int gemm_mode = GEMM_FP32;   // resolved GemmMode of the MFMA layers (never GEMM_DEFAULT here)
    double lr = 1e-3, beta1 = 0.9, beta2 = 0.999, adam_eps = 1e-7, bn_eps = 1e-3, bn_momentum = 0.99, dropout = 0.3;
};

struct Dataset {
    const float* x_train = nullptr; const int32_t* y_train = nullptr; int64_t n_train = 0;
    const float* x_val = nullptr;   const int32_t* y_val = nullptr;   int64_t n_val = 0;
    int T = 0, F = 0;
};

// host-side closed forms (bit-exact twins of genes.py)
int64_t param_count(const int32_t g[6], int variant, int classes);
double fwd_flops_per_sample(const int32_t g[6], int variant, int classes, int T, int F);
void validate_gene(const int32_t g[6]);

// HIP-event sampling of the MFMA GEMM kernels inside the timed region (bench.py roofline)
struct ProfileEntry { double ms = 0, flops = 0; long long launches = 0; };
struct ProfileTotals {   // keyed by kernel instantiation name, e.g. "igemm_fwd_kernel<128,32>"
    std::mutex mu;
    std::map<std::string, ProfileEntry> by_kernel;
    // launch-path variants seen: name + "+sk" (split-K slabs + combine) / "+stats" (BatchNorm statistics epilogue) /
    // "+tab" (row-table operand loader) / "+bal" / "+slabs" (several row slices) -- what the parity-coverage test compares
    std::set<std::string> variants;
    void reset() { std::lock_guard<std::mutex> l(mu); by_kernel.clear(); variants.clear(); }
};
std::string gemm_variant_name(int cls, int code, int flags);
ProfileTotals& profile_totals();

struct GemmHook {   // brackets every MFMA GEMM launch (HIP-event sampling)
    virtual const GemmTiming* begin(int cls, double flops) = 0;   // null: do not time this launch
    virtual void end(int code, int flags) = 0;   // code: instantiation (gemm_kernel_name), flags: GemmFlags of the path taken
    virtual ~GemmHook() {}
};
// wgrad_ws holds wgrad_ws_floats floats; the slice count is clamped to what fits (never written past)
void conv_backward_weights(const float* X, const float* dY, float* dW, float* dB, const ConvGeom& g, float* wgrad_ws,
                           size_t wgrad_ws_floats, hipStream_t s, GemmHook* hook, int mode = GEMM_DEFAULT,
                           const void* rowtab = nullptr, int tab_rows = 0, AdamSeg* defer = nullptr);
// defer != null (requires dB == dW + Cout*K): the slice sum is left to the optimiser launch -- *defer receives the slab
// pointer / stride / slice count (slab stays null when one slice wrote dW, dB in place)
// wd_ready: wd_ws already holds the flip-transposed weights (the trainer refreshes all layers in one launch per step)
// rowtab_d: row table of the DGRAD geometry (see dgrad_geometry), optional
void conv_backward_data(const float* dY, const float* W, float* dX, const ConvGeom& g, float* wd_ws, const float* mask,
                        float mask_scale, int accumulate, hipStream_t s, GemmHook* hook, float* sk_ws = nullptr,
                        size_t sk_floats = 0, int mode = GEMM_DEFAULT, bool wd_ready = false, const void* rowtab_d = nullptr,
                        int rowtab_d_rows = 0);
// geometry of the implicit GEMM that computes dX from dY for forward geometry g (stride 1: flipped SAME padding;
// the strided 1x1 skip projection: a 1x1 GEMM over the output pixels, scattered by the epilogue)
ConvGeom dgrad_geometry(const ConvGeom& g);

// cached device allocations (net.hip): get may return stale contents, free never blocks on other streams
void* pool_alloc(size_t bytes);
void pool_free(void* p);
void* pool_alloc_pinned(size_t bytes);
void pool_free_pinned(void* p);

struct Act {
    float* data = nullptr;
    float* grad = nullptr;
    int H = 0, W = 0, C = 0;
    bool own_grad = false;
    bool virt = false;         // never materialised (BatchNorm output consumed by a fused max-pool)
    size_t per_sample() const { return (size_t)H * W * C; }
};

enum OpKind { OP_CONV1, OP_CONV, OP_BN, OP_POOL, OP_ADDRELU, OP_GAP, OP_DENSE };

struct Op {
    OpKind kind;
    int in = -1, in2 = -1, out = -1;
    // conv / dense
    int KS = 1, stride = 1, Cin = 0, Cout = 0;
    int relu = 0, need_dgrad = 1, in_is_relu = 0, dgrad_accumulate = 0, dropout_layer = -1;
    float in_mask_scale = 1.f;
    int gemm_mode = GEMM_FP32;   // arithmetic of this layer's three GEMMs
    void* rowtab = nullptr;    // conv: row table of the layer (forward tile prologue and weight-gradient gather), max(batch, eval_batch) samples
    int rowtab_rows = 0;
    void* rowtab_d = nullptr;  // conv: row table of the layer's dgrad geometry (dY as input, flipped padding), train batch
    int rowtab_d_rows = 0;
    int64_t w_off = 0, b_off = 0, wd_off = -1;   // wd_off: this layer's slice of the flip-transposed copy (dgrad operand)
    int64_t slab_off = -1;     // conv / first conv: this layer's weight-gradient slabs in the candidate's slab arena
    size_t slab_floats = 0;    // (kept until the optimiser launch sums them), sized for the worst train batch
    int tensor_index = 0;   // canonical index of the kernel tensor (RNG init stream)
    // bn
    int64_t gamma_off = 0, beta_off = 0, mm_off = 0, mv_off = 0;
    int relu_after = 0, mask_in_pos = 0;
    int fuse_pool = 0;         // bn: the next op is the max-pool of this BN's output -> one fused kernel, output never materialised
    int feeds_bn = 0;          // conv: the next op is the BatchNorm of this conv's output (statistics fused into the epilogue)
    float* bn_buf = nullptr;   // mean | invstd | scale | shift, each [C]
    // pool
    uint8_t* arg = nullptr;
    int mask_y_pos = 0;
    int fused_into_bn = 0;     // pool: executed inside the preceding BatchNorm's fused kernels
};

class Net : public GemmHook {
  public:
    const GemmTiming* begin(int cls, double flops) override;
    void end(int code, int flags) override;
    Net(const int32_t gene[6], const NetConfig& cfg, int T, int F, uint32_t seed, hipStream_t stream);
    ~Net();
    Net(const Net&) = delete;
    Net& operator=(const Net&) = delete;

    int64_t total_params() const { return n_params_; }
    void get_params(float* host);
    void set_params(const float* host);
    void get_grads(float* host);
    void snapshot_params();   // device copy (restore_best_weights)
    void restore_snapshot();
    // full training state: parameters (BatchNorm moving statistics included), Adam m / v, optimizer.iterations and the
    // global step that keys the dropout masks -- what a checker needs to re-synchronise with this net at an epoch boundary
    void get_state(float* params, float* m, float* v, long long* iterations, long long* steps);
    void set_state(const float* params, const float* m, const float* v, long long iterations, long long steps);
    // rows of the resident tensor the next train steps gather from (0: unknown, no clamp)
    void set_gather_rows(int64_t n) { gather_rows_ = n; }
    // ONE epoch of Model.fit on the production path (device permutation of (seed, epoch) when cfg.shuffle, device
    // StepState steps, last partial batch kept); idx_scratch: n_train int32 on the device
    void run_epoch(const float* X, const int32_t* y, int64_t n_train, int epoch, int32_t* idx_scratch);

    // fwd + bwd + Adam on rows idx[row0 .. row0+B) (idx may be null -> rows row0..)
    void train_step(const float* X, const int32_t* y, const int32_t* idx, int64_t row0, int B);
    // The fit loop's form of the same step: the batch position, dropout counter and Adam iteration live in a device
    // StepState (kernels.h), so a full-batch step has no per-step host arguments; with CMOOP_GRAPH=1 it is captured ONCE
    // as a hipGraph and replayed (opt-in: measured no faster than eager launches, see begin_fit).
    void begin_fit(int64_t total_steps);      // uploads Adam's per-iteration step sizes, sets the device state to (0, step, iterations)
    void begin_epoch();                       // state.row0 = 0
    void train_step_stateful(const float* X, const int32_t* y, const int32_t* idx, int B);
    // inference over n rows of (X, y); returns sum of per-sample losses and #correct, fills preds (device, may be null)
    void evaluate(const float* X, const int32_t* y, int64_t n, double* loss_sum, long long* correct, int32_t* preds);
    void read_train_metrics(double* loss_sum, long long* correct, bool reset);
    void drain_profile();
    hipStream_t stream() const { return stream_; }
    uint32_t seed() const { return seed_; }
    const NetConfig& config() const { return cfg_; }
    int feature_T() const { return T_; }
    int feature_F() const { return F_; }
    long long steps_done() const { return step_; }
    long long iterations_done() const { return iterations_; }

  private:
    void build_plan();
    void forward(const float* X, const int32_t* idx, int64_t row0, int B, bool train, const StepState* st = nullptr);
    void backward(const float* X, const int32_t* idx, int64_t row0, int B, const StepState* st = nullptr);
    void step_body(const float* X, const int32_t* y, const int32_t* idx, int64_t row0, int B, const StepState* st);
    ConvGeom geom_of(const Op& op, int B) const;
    void run_gemm(int cls, const float* X, const float* Wt, float* Y, const ConvGeom& g, const GemmEpilogue& e, int* stats_blocks = nullptr,
                  const void* rowtab = nullptr, int tab_rows = 0);
    float* dalloc(size_t floats);

    int32_t gene_[6];
    NetConfig cfg_;
    int T_, F_, Bmax_;
    uint32_t seed_;
    hipStream_t stream_;
    std::vector<Act> acts_;
    std::vector<Op> ops_;
    std::vector<void*> allocs_;
    int64_t n_params_ = 0;
    float *params_ = nullptr, *grads_ = nullptr, *adam_m_ = nullptr, *adam_v_ = nullptr, *snap_ = nullptr;
    float *wgrad_ws_ = nullptr, *wd_ws_ = nullptr, *red_ws_ = nullptr, *splitk_ws_ = nullptr;
    StepState* st_dev_ = nullptr;       // device step state (train_step_stateful)
    float* alpha_tab_ = nullptr;        // Adam step size per iteration
    int64_t alpha_tab_n_ = 0, host_row0_ = 0, gather_rows_ = 0;
    hipGraphExec_t graph_exec_ = nullptr;   // the captured full-batch train step
    bool graph_ok_ = true;
    FlipEntry* flip_table_ = nullptr;   // device table of the conv layers whose dgrad needs flip-transposed weights
    int flip_layers_ = 0;
    int64_t flip_max_elems_ = 0;
    std::vector<AdamSeg> slab_segs_;    // this step's unreduced weight-gradient slabs (backward fills, the optimiser launch consumes)
    size_t wgrad_ws_floats_ = 0, wd_ws_floats_ = 0, red_ws_floats_ = 0, splitk_ws_floats_ = 0;
    double* acc_train_ = nullptr;   // [2]: loss sum, correct (int64 bits)
    double* acc_eval_ = nullptr;
    int logits_ = -1;
    long long step_ = 0, iterations_ = 0;
    bool profiling_now_ = false, hook_live_ = false;
    int fused_stats_blocks_ = 0;   // > 0: the conv just launched left this many BatchNorm statistic partials in red_ws_
    struct EvPair { GemmTiming t; double flops; int cls, code, flags; };
    std::vector<EvPair> ev_pool_;
    size_t ev_used_ = 0;
};

struct EvalResult {
    double acc = 0, size_mb = 0, fpr = 0, val_loss = 0, seconds = 0;
    int epochs_run = 0;
    int evaluated = 0;   // 1 when THIS call trained the candidate (always, unless a pull callback handed it to another rank)
};

// per-epoch record of a fit (tests: the early-stopping decisions are checked against the validation-loss history)
struct FitHistory {
    std::vector<double> val_loss, val_acc;
    int best_epoch = -1;
};
// Model.fit + EarlyStopping + the read-outs on an EXISTING net (the body of run_candidate); seed keys the epoch shuffle
EvalResult fit_and_read_out(Net& net, const NetConfig& cfg, const Dataset& ds, uint32_t seed, FitHistory* hist = nullptr);
// train-to-early-stop + readouts for one candidate (evaluate_individual, nsga_penalty.py:368-395)
EvalResult run_candidate(const int32_t gene[6], const NetConfig& cfg, const Dataset& ds, uint32_t seed, hipStream_t stream);
// host-only: every implicit-GEMM conv geometry of a candidate at batch B (plan walk without device memory); throws
// through igemm_check_range when a layer is beyond the kernels' 32-bit byte offsets
void check_plan_ranges(const int32_t gene[6], int variant, int T, int F, int B);
// the population loop (compute_objectives_and_constraints, nsga_penalty.py:418-442): n_slots
// candidates in flight on their own HIP streams, longest-first
// pull (optional): called by the worker threads (concurrently) for the next candidate index, < 0 = queue exhausted --
// lets several ranks drain ONE longest-first queue (cross-rank dynamic scheduling, evaluator.py); without it the
// n candidates are taken longest-first from a process-local counter.
void eval_population(const NetConfig& cfg, const Dataset& ds, const int32_t* genes, const uint32_t* seeds, int n,
                     EvalResult* out, const std::function<int()>& pull = {});

double fpr_from_confusion(const int64_t* cm, int C, int variant);
void epoch_permutation(uint32_t seed, uint32_t epoch, int64_t n, int32_t* out);

}  // namespace cmoop
