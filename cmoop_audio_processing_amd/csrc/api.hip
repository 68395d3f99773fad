// extern "C" boundary of libcmoop_hip.so -- see include/cmoop.h for the contract.
#include "../../include/cmoop.h"
#include "net.h"

#include <cstring>
#include <mutex>
#include <vector>

using namespace cmoop;

static thread_local std::string g_err;
static thread_local std::string g_last_kernels;   // cmoop_last_kernels

template <class F>
static int guard(F&& f) {
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    } catch (...) {
        g_err = "unknown error";
        return 2;
    }
}

constexpr int MAX_DEVICES = 16;

// one library stream per (calling thread, device): switching devices neither leaks nor reuses the other device's stream
static hipStream_t lib_stream() {
    static thread_local hipStream_t streams[MAX_DEVICES] = {nullptr};
    int dev = 0;
    CMOOP_HIP(hipGetDevice(&dev));
    CMOOP_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device index out of range");
    if (!streams[dev]) CMOOP_HIP(hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking));
    return streams[dev];
}

// front-end tables (twiddles / window / sparse mel weights): one set per device, created once under a lock
static const FrontendTables* frontend_tables_for_current_device() {
    static std::mutex mu;
    static FrontendTables* tables[MAX_DEVICES] = {nullptr};
    int dev = 0;
    CMOOP_HIP(hipGetDevice(&dev));
    CMOOP_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device index out of range");
    std::lock_guard<std::mutex> l(mu);
    if (!tables[dev]) tables[dev] = frontend_tables_create(FrontendCfg());
    return tables[dev];
}

static NetConfig to_cfg(const cmoop_config* c) {
    CMOOP_REQUIRE(c != nullptr, "config is NULL");
    NetConfig n;
    n.variant = c->variant; n.classes = c->classes; n.epochs = c->epochs; n.batch = c->batch; n.patience = c->patience;
    n.early_stop = c->early_stop; n.restore_best = c->restore_best; n.acc_readout = c->acc_readout;
    n.fpr_variant = c->fpr_variant; n.shuffle = c->shuffle; n.eval_batch = c->eval_batch; n.n_slots = c->n_slots;
    n.profile_every = c->profile_every;
    n.gemm_mode = c->gemm_mode == CMOOP_GEMM_DEFAULT ? gemm_mode_default() : c->gemm_mode;
    n.lr = c->lr; n.beta1 = c->beta1; n.beta2 = c->beta2; n.adam_eps = c->adam_eps; n.bn_eps = c->bn_eps;
    n.bn_momentum = c->bn_momentum; n.dropout = c->dropout;
    CMOOP_REQUIRE(n.variant == 0 || n.variant == 1, "variant must be CMOOP_VARIANT_A or _B");
    CMOOP_REQUIRE(n.fpr_variant >= 0 && n.fpr_variant <= 2, "bad fpr_variant");
    CMOOP_REQUIRE(n.epochs >= 0 && n.patience >= 0 && n.batch >= 1 && n.eval_batch >= 1, "bad epochs/patience/batch");
    CMOOP_REQUIRE(n.dropout >= 0.0 && n.dropout < 1.0, "dropout must be in [0,1)");
    CMOOP_REQUIRE(n.gemm_mode == GEMM_FP32 || n.gemm_mode == GEMM_BF16X3 || n.gemm_mode == GEMM_BF16, "bad gemm_mode");
    return n;
}

static ConvGeom make_geom(int B, int H, int W, int Cin, int Cout, int KS, int stride) {
    ConvGeom g;
    g.B = B; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.KH = g.KW = KS; g.stride = stride;
    g.OH = (H + stride - 1) / stride; g.OW = (W + stride - 1) / stride;
    g.pad_t = std::max((g.OH - 1) * stride + KS - H, 0) / 2;
    g.pad_l = std::max((g.OW - 1) * stride + KS - W, 0) / 2;
    return g;
}

// records the launch-path variant names of the GEMM launches a kernel-level call makes (cmoop_last_kernels)
struct RecordingHook : GemmHook {
    int cls = 0;
    std::string* out;
    explicit RecordingHook(std::string* o) : out(o) {}
    const GemmTiming* begin(int c, double) override { cls = c; return nullptr; }
    void end(int code, int flags) override {
        if (!out->empty()) *out += ";";
        *out += gemm_variant_name(cls, code, flags);
    }
};

extern "C" {

int cmoop_abi_version(void) { return CMOOP_ABI_VERSION; }
const char* cmoop_last_error(void) { return g_err.c_str(); }

void cmoop_config_default(cmoop_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->variant = CMOOP_VARIANT_A; c->classes = 10; c->epochs = 300; c->batch = 64; c->patience = 5;
    c->early_stop = 1; c->restore_best = 0; c->acc_readout = 0; c->fpr_variant = CMOOP_FPR_V1; c->shuffle = 1;
    c->eval_batch = 256; c->n_slots = 8; c->profile_every = 0;
    c->lr = 1e-3; c->beta1 = 0.9; c->beta2 = 0.999; c->adam_eps = 1e-7; c->bn_eps = 1e-3; c->bn_momentum = 0.99;
    c->dropout = 0.3;
}

int cmoop_param_count(const int32_t gene[6], int32_t variant, int32_t classes, int64_t* out) {
    return guard([&] { *out = param_count(gene, variant, classes); });
}
int cmoop_fwd_flops(const int32_t gene[6], int32_t variant, int32_t classes, int32_t T, int32_t F, double* out) {
    return guard([&] { *out = fwd_flops_per_sample(gene, variant, classes, T, F); });
}

int cmoop_eval_population(const cmoop_config* cfg, const cmoop_dataset* ds, const int32_t* genes, const uint32_t* seeds,
                          int32_t n, double* acc, double* size_mb, double* fpr, int32_t* epochs_run, double* val_loss,
                          double* seconds) {
    return guard([&] {
        CMOOP_REQUIRE(ds && genes && seeds, "NULL argument");
        CMOOP_REQUIRE(n >= 0, "negative population size");
        NetConfig c = to_cfg(cfg);
        Dataset d;
        d.x_train = ds->x_train; d.y_train = ds->y_train; d.n_train = ds->n_train;
        d.x_val = ds->x_val; d.y_val = ds->y_val; d.n_val = ds->n_val; d.T = ds->T; d.F = ds->F;
        CMOOP_REQUIRE(n == 0 || (d.x_train && d.y_train && d.x_val && d.y_val), "dataset pointers are NULL");
        for (int i = 0; i < n; ++i) check_plan_ranges(genes + 6 * i, c.variant, d.T, d.F, std::max(c.batch, c.eval_batch));
        std::vector<EvalResult> r(n);
        eval_population(c, d, genes, seeds, n, r.data());
        for (int i = 0; i < n; ++i) {
            if (acc) acc[i] = r[i].acc;
            if (size_mb) size_mb[i] = r[i].size_mb;
            if (fpr) fpr[i] = r[i].fpr;
            if (epochs_run) epochs_run[i] = r[i].epochs_run;
            if (val_loss) val_loss[i] = r[i].val_loss;
            if (seconds) seconds[i] = r[i].seconds;
        }
    });
}

int cmoop_eval_population_pull(const cmoop_config* cfg, const cmoop_dataset* ds, const int32_t* genes, const uint32_t* seeds,
                               int32_t n, cmoop_next_fn next, void* ctx, double* acc, double* size_mb, double* fpr,
                               int32_t* epochs_run, double* val_loss, double* seconds, int32_t* evaluated) {
    return guard([&] {
        CMOOP_REQUIRE(ds && genes && seeds && next && evaluated, "NULL argument");
        CMOOP_REQUIRE(n >= 0, "negative population size");
        NetConfig c = to_cfg(cfg);
        Dataset d;
        d.x_train = ds->x_train; d.y_train = ds->y_train; d.n_train = ds->n_train;
        d.x_val = ds->x_val; d.y_val = ds->y_val; d.n_val = ds->n_val; d.T = ds->T; d.F = ds->F;
        CMOOP_REQUIRE(n == 0 || (d.x_train && d.y_train && d.x_val && d.y_val), "dataset pointers are NULL");
        for (int i = 0; i < n; ++i) check_plan_ranges(genes + 6 * i, c.variant, d.T, d.F, std::max(c.batch, c.eval_batch));
        std::vector<EvalResult> r(n);
        eval_population(c, d, genes, seeds, n, r.data(), [&]() { return (int)next(ctx); });
        for (int i = 0; i < n; ++i) {
            evaluated[i] = r[i].evaluated;
            if (acc) acc[i] = r[i].acc;
            if (size_mb) size_mb[i] = r[i].size_mb;
            if (fpr) fpr[i] = r[i].fpr;
            if (epochs_run) epochs_run[i] = r[i].epochs_run;
            if (val_loss) val_loss[i] = r[i].val_loss;
            if (seconds) seconds[i] = r[i].seconds;
        }
    });
}

int cmoop_plan_check(const int32_t gene[6], int32_t variant, int32_t T, int32_t F, int32_t batch) {
    return guard([&] {
        CMOOP_REQUIRE(gene && (variant == 0 || variant == 1) && T >= 1 && F >= 1 && batch >= 1, "plan_check: bad arguments");
        check_plan_ranges(gene, variant, T, F, batch);
    });
}

int cmoop_conv_launch_plan(int32_t op, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride,
                           int32_t want_stats, char* name, int32_t name_cap) {
    return guard([&] {
        CMOOP_REQUIRE(name && name_cap > 0 && op >= 0 && op <= 2, "launch_plan: bad arguments");
        CMOOP_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cin >= 16 && Cout >= 1 && KS >= 1 && stride >= 1, "bad conv shape");
        const ConvGeom g = make_geom(B, H, W, Cin, Cout, KS, stride);
        const bool tab = KS * KS <= 32;              // Net::build_plan builds the layer's row tables under this condition
        int flags = 0, code = 0, cls = 0;
        GemmEpilogue e;
        if (op == 0) {
            code = igemm_fwd_plan(g, e, igemm_splitk_workspace(g), want_stats != 0, tab, &flags);
        } else if (op == 1) {
            CMOOP_REQUIRE(ilog2_exact(Cout) >= 4, "launch_plan: the dgrad of this layer does not run on the MFMA kernel");
            const ConvGeom gd = dgrad_geometry(g);
            if (stride != 1) { e.out_stride = stride; e.OHf = H; e.OWf = W; e.accumulate = 1; }
            code = igemm_fwd_plan(gd, e, igemm_splitk_workspace(gd), false, tab, &flags);
        } else {
            cls = 1;
            code = igemm_wgrad_plan(g, wgrad_slices(g), GEMM_DEFAULT, tab, &flags);
        }
        std::snprintf(name, name_cap, "%s", gemm_variant_name(cls, code, flags).c_str());
    });
}

int cmoop_halo_tile_check(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t* rows_bound,
                          int32_t* rows_needed, int32_t* rows_stageable) {
    return guard([&] {
        CMOOP_REQUIRE(rows_bound && rows_needed && rows_stageable && B >= 1 && H >= 1 && W >= 1 && Cin >= 16 && Cout >= 1 && KS >= 1, "bad conv shape");
        int b = 0, n = 0, c = 0;
        halo_rows_bound_and_need(make_geom(B, H, W, Cin, Cout, KS, 1), &b, &n, &c);
        *rows_bound = b; *rows_needed = n; *rows_stageable = c;
    });
}

int cmoop_wgrad_slices(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t* out) {
    return guard([&] {
        CMOOP_REQUIRE(out && B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && Cout >= 1 && KS >= 1 && stride >= 1, "bad conv shape");
        *out = wgrad_slices(make_geom(B, H, W, Cin, Cout, KS, stride));
    });
}

int cmoop_calculate_fpr(const int32_t* y_true, const int32_t* y_pred, int64_t n, int32_t classes, int32_t fpr_variant,
                        double* out) {
    return guard([&] {
        CMOOP_REQUIRE(classes >= 1 && classes <= 4096, "bad class count");
        std::vector<int64_t> cm((size_t)classes * classes, 0);
        for (int64_t i = 0; i < n; ++i) {
            const int a = fpr_variant == CMOOP_FPR_V1_QUIRK ? 0 : y_true[i];
            const int b = y_pred[i];
            if (a >= 0 && a < classes && b >= 0 && b < classes) cm[(size_t)a * classes + b] += 1;
        }
        *out = fpr_from_confusion(cm.data(), classes, fpr_variant == CMOOP_FPR_V3 ? 2 : 0);
    });
}

// ---- front end ---------------------------------------------------------------
int cmoop_logmel(const float* wav_dev, int64_t n_clips, int32_t n_samples, float* out_dev) {
    return guard([&] {
        const FrontendTables* tables = frontend_tables_for_current_device();
        CMOOP_REQUIRE(n_samples >= 160, "clip shorter than one hop");
        hipStream_t s = lib_stream();
        launch_logmel(wav_dev, n_clips, n_samples, out_dev, tables, s);
        CMOOP_HIP(hipStreamSynchronize(s));
    });
}

int cmoop_mfcc(const float* logmel_dev, int64_t rows, int32_t n_mels, int32_t n_mfcc, float* out_dev) {
    return guard([&] {
        CMOOP_REQUIRE(rows >= 0 && (rows == 0 || (logmel_dev && out_dev && logmel_dev != out_dev)), "mfcc: out of place, rows >= 0");
        hipStream_t s = lib_stream();
        launch_mfcc(logmel_dev, out_dev, rows, n_mels, n_mfcc, s);
        CMOOP_HIP(hipStreamSynchronize(s));
    });
}

int cmoop_standardize_fit(const float* x_dev, int64_t rows, int32_t cols, double* mean_host, double* scale_host) {
    return guard([&] {
        CMOOP_REQUIRE(rows >= 1 && cols >= 4 && cols % 4 == 0, "standardize: cols must be a multiple of 4");
        hipStream_t s = lib_stream();
        const int nb = colreduce_blocks(rows, cols);
        float* P = nullptr;
        double* ms = nullptr;
        CMOOP_HIP(hipMalloc(&P, (size_t)nb * 2 * cols * 4));
        CMOOP_HIP(hipMalloc(&ms, (size_t)2 * cols * 8));
        launch_colstats(x_dev, P, rows, cols, nb, s);
        colstats_finalize_f64(P, nb, rows, cols, ms, ms + cols, s);
        CMOOP_HIP(hipMemcpyAsync(mean_host, ms, cols * 8, hipMemcpyDeviceToHost, s));
        CMOOP_HIP(hipMemcpyAsync(scale_host, ms + cols, cols * 8, hipMemcpyDeviceToHost, s));
        CMOOP_HIP(hipStreamSynchronize(s));
        hipFree(P); hipFree(ms);
    });
}

int cmoop_standardize_apply(float* x_dev, int64_t rows, int32_t cols, const double* mean_host, const double* scale_host) {
    return guard([&] {
        hipStream_t s = lib_stream();
        double* ms = nullptr;
        CMOOP_HIP(hipMalloc(&ms, (size_t)2 * cols * 8));
        CMOOP_HIP(hipMemcpyAsync(ms, mean_host, cols * 8, hipMemcpyHostToDevice, s));
        CMOOP_HIP(hipMemcpyAsync(ms + cols, scale_host, cols * 8, hipMemcpyHostToDevice, s));
        launch_standardize(x_dev, ms, ms + cols, rows, cols, s);
        CMOOP_HIP(hipStreamSynchronize(s));
        hipFree(ms);
    });
}

// ---- profile -------------------------------------------------------------------
int cmoop_profile_reset(void) { return guard([] { profile_totals().reset(); }); }
int cmoop_profile_count(int32_t* out) {
    return guard([&] {
        ProfileTotals& t = profile_totals();
        std::lock_guard<std::mutex> l(t.mu);
        *out = (int32_t)t.by_kernel.size();
    });
}
int cmoop_profile_entry(int32_t i, char* name, int32_t name_cap, int64_t* launches, double* total_ms, double* total_flops) {
    return guard([&] {
        ProfileTotals& t = profile_totals();
        std::lock_guard<std::mutex> l(t.mu);
        CMOOP_REQUIRE(i >= 0 && i < (int)t.by_kernel.size() && name_cap > 0, "profile entry out of range");
        auto it = t.by_kernel.begin();
        std::advance(it, i);
        std::snprintf(name, name_cap, "%s", it->first.c_str());
        *launches = it->second.launches; *total_ms = it->second.ms; *total_flops = it->second.flops;
    });
}

int cmoop_profile_variant_count(int32_t* out) {
    return guard([&] {
        ProfileTotals& t = profile_totals();
        std::lock_guard<std::mutex> l(t.mu);
        *out = (int32_t)t.variants.size();
    });
}
int cmoop_profile_variant(int32_t i, char* name, int32_t name_cap) {
    return guard([&] {
        ProfileTotals& t = profile_totals();
        std::lock_guard<std::mutex> l(t.mu);
        CMOOP_REQUIRE(i >= 0 && i < (int)t.variants.size() && name_cap > 0, "profile variant out of range");
        auto it = t.variants.begin();
        std::advance(it, i);
        std::snprintf(name, name_cap, "%s", it->c_str());
    });
}

// ---- session -------------------------------------------------------------------
struct cmoop_net {
    Net* net;
};

int cmoop_net_create(const int32_t gene[6], const cmoop_config* cfg, int32_t T, int32_t F, uint32_t seed, cmoop_net** out) {
    return guard([&] {
        NetConfig c = to_cfg(cfg);
        auto* h = new cmoop_net;
        h->net = nullptr;
        try {
            h->net = new Net(gene, c, T, F, seed, lib_stream());
        } catch (...) {
            delete h;
            throw;
        }
        *out = h;
    });
}
int cmoop_net_destroy(cmoop_net* h) {
    return guard([&] {
        if (!h) return;
        if (h->net) { hipStreamSynchronize(h->net->stream()); delete h->net; }
        delete h;
    });
}
int cmoop_net_total_params(cmoop_net* h, int64_t* out) { return guard([&] { *out = h->net->total_params(); }); }
int cmoop_net_get_params(cmoop_net* h, float* host) { return guard([&] { h->net->get_params(host); }); }
int cmoop_net_set_params(cmoop_net* h, const float* host) { return guard([&] { h->net->set_params(host); }); }
int cmoop_net_get_grads(cmoop_net* h, float* host) { return guard([&] { h->net->get_grads(host); }); }
int cmoop_net_train_step(cmoop_net* h, const float* x, const int32_t* y, const int32_t* idx, int64_t row0, int32_t B) {
    return guard([&] {
        h->net->train_step(x, y, idx, row0, B);
        CMOOP_HIP(hipStreamSynchronize(h->net->stream()));
        h->net->drain_profile();     // cfg.profile_every > 0: the sampled launches enter cmoop_profile_entry / _variant
    });
}
int cmoop_net_get_state(cmoop_net* h, float* params, float* adam_m, float* adam_v, int64_t* iterations, int64_t* steps) {
    return guard([&] {
        long long it = 0, st = 0;
        h->net->get_state(params, adam_m, adam_v, &it, &st);
        if (iterations) *iterations = it;
        if (steps) *steps = st;
    });
}
int cmoop_net_set_state(cmoop_net* h, const float* params, const float* adam_m, const float* adam_v, int64_t iterations,
                        int64_t steps) {
    return guard([&] { h->net->set_state(params, adam_m, adam_v, iterations, steps); });
}
int cmoop_net_set_gather_rows(cmoop_net* h, int64_t n_rows) {
    return guard([&] {
        CMOOP_REQUIRE(n_rows >= 0, "negative row count");
        h->net->set_gather_rows(n_rows);
    });
}
int cmoop_net_run_epoch(cmoop_net* h, const float* x, const int32_t* y, int64_t n_train, int32_t epoch) {
    return guard([&] {
        CMOOP_REQUIRE(x && y && n_train >= 1, "run_epoch: NULL / empty training split");
        int32_t* idx = static_cast<int32_t*>(pool_alloc((size_t)n_train * 4));
        try {
            h->net->run_epoch(x, y, n_train, epoch, idx);
            CMOOP_HIP(hipStreamSynchronize(h->net->stream()));
            h->net->drain_profile();
        } catch (...) {
            hipStreamSynchronize(h->net->stream());
            pool_free(idx);
            throw;
        }
        pool_free(idx);
    });
}
int cmoop_net_fit(cmoop_net* h, const cmoop_dataset* ds, int32_t hist_cap, double* val_loss_hist, double* val_acc_hist,
                  int32_t* epochs_run, int32_t* best_epoch, double* acc, double* fpr, double* val_loss) {
    return guard([&] {
        CMOOP_REQUIRE(ds && ds->x_train && ds->y_train && ds->x_val && ds->y_val, "fit: dataset pointers are NULL");
        Dataset d;
        d.x_train = ds->x_train; d.y_train = ds->y_train; d.n_train = ds->n_train;
        d.x_val = ds->x_val; d.y_val = ds->y_val; d.n_val = ds->n_val; d.T = ds->T; d.F = ds->F;
        CMOOP_REQUIRE(d.T == h->net->feature_T() && d.F == h->net->feature_F(), "fit: dataset feature shape differs from the net's");
        FitHistory hist;
        const EvalResult r = fit_and_read_out(*h->net, h->net->config(), d, h->net->seed(), &hist);
        for (int i = 0; i < hist_cap && i < (int)hist.val_loss.size(); ++i) {
            if (val_loss_hist) val_loss_hist[i] = hist.val_loss[i];
            if (val_acc_hist) val_acc_hist[i] = hist.val_acc[i];
        }
        if (epochs_run) *epochs_run = r.epochs_run;
        if (best_epoch) *best_epoch = hist.best_epoch;
        if (acc) *acc = r.acc;
        if (fpr) *fpr = r.fpr;
        if (val_loss) *val_loss = r.val_loss;
    });
}
int cmoop_net_evaluate(cmoop_net* h, const float* x, const int32_t* y, int64_t n, double* loss_sum, int64_t* correct,
                       int32_t* preds_dev) {
    return guard([&] {
        long long c = 0;
        h->net->evaluate(x, y, n, loss_sum, &c, preds_dev);
        *correct = c;
    });
}
int cmoop_net_train_metrics(cmoop_net* h, double* loss_sum, int64_t* correct, int32_t reset) {
    return guard([&] {
        long long c = 0;
        h->net->read_train_metrics(loss_sum, &c, reset != 0);
        *correct = c;
    });
}
int cmoop_epoch_permutation(uint32_t seed, uint32_t epoch, int64_t n, int32_t* out_host) {
    return guard([&] { epoch_permutation(seed, epoch, n, out_host); });
}

int cmoop_epoch_permutation_device(uint32_t seed, uint32_t epoch, int64_t n, int32_t* out_dev) {
    return guard([&] {
        hipStream_t s = lib_stream();
        launch_epoch_permutation(seed, epoch, n, out_dev, s);
        CMOOP_HIP(hipStreamSynchronize(s));
    });
}

// ---- kernel-level ---------------------------------------------------------------

int cmoop_conv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t H, int32_t W, int32_t Cin,
                   int32_t Cout, int32_t KS, int32_t stride, int32_t relu) {
    return guard([&] {
        g_last_kernels.clear();
        hipStream_t s = lib_stream();
        if (Cin == 1) {
            CMOOP_REQUIRE(stride == 1 && bias, "first-layer conv: stride 1 with bias");
            launch_conv1_fwd(x, nullptr, 0, w, bias, y, B, H, W, Cout, KS, relu, s);
        } else {
            GemmEpilogue e;
            e.bias = bias; e.relu = relu;
            const ConvGeom g = make_geom(B, H, W, Cin, Cout, KS, stride);
            const size_t skf = igemm_splitk_workspace(g);
            float* sk = nullptr;
            if (skf) CMOOP_HIP(hipMalloc(&sk, skf * 4));
            int flags = 0;
            const int code = launch_igemm_fwd(x, w, y, g, e, s, nullptr, sk, skf, nullptr, nullptr, 0, &flags);
            g_last_kernels = gemm_variant_name(0, code, flags);
            CMOOP_HIP(hipStreamSynchronize(s));
            if (sk) hipFree(sk);
        }
        CMOOP_HIP(hipStreamSynchronize(s));
    });
}

int cmoop_conv_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int32_t B, int32_t H,
                   int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t mask_relu) {
    return guard([&] {
        hipStream_t s = lib_stream();
        if (Cin == 1) {
            const int nb = conv1_wgrad_blocks(B, H, W);
            const int64_t per = (int64_t)Cout * (KS * KS + 1);
            float *P = nullptr, *tmp = nullptr;
            CMOOP_HIP(hipMalloc(&P, (size_t)nb * per * 4));
            CMOOP_HIP(hipMalloc(&tmp, (size_t)per * 4));
            launch_conv1_wgrad(x, nullptr, 0, dy, P, B, H, W, Cout, KS, s);
            launch_reduce_slices(P, tmp, nb, per, s);
            CMOOP_HIP(hipMemcpyAsync(dw, tmp, (size_t)Cout * KS * KS * 4, hipMemcpyDeviceToDevice, s));
            CMOOP_HIP(hipMemcpyAsync(db, tmp + (size_t)Cout * KS * KS, (size_t)Cout * 4, hipMemcpyDeviceToDevice, s));
            CMOOP_HIP(hipStreamSynchronize(s));
            hipFree(P); hipFree(tmp);
            return;
        }
        const ConvGeom g = make_geom(B, H, W, Cin, Cout, KS, stride);
        float *wg = nullptr, *red = nullptr, *wd = nullptr, *sk = nullptr;
        const size_t wg_floats = (size_t)wgrad_slices(g) * g.Cout * (g.K() + 1);
        CMOOP_HIP(hipMalloc(&wg, wg_floats * 4));
        CMOOP_HIP(hipMalloc(&red, ((size_t)1024 * 2 * Cout + 2 * Cout + 64) * 4));
        CMOOP_HIP(hipMalloc(&wd, (size_t)g.Cout * g.K() * 4));
        void* tab = nullptr;
        const int tab_rows = (KS * KS <= 32) ? rowtab_rows(g) : 0;
        if (tab_rows) {
            CMOOP_HIP(hipMalloc(&tab, (size_t)tab_rows * 8));
            launch_build_rowtab(g, tab, s);
        }
        g_last_kernels.clear();
        RecordingHook hook(&g_last_kernels);
        conv_backward_weights(x, dy, dw, db, g, wg, wg_floats, s, &hook, GEMM_DEFAULT, tab, tab_rows);
        if (dx) {
            int accumulate = 0;
            if (stride != 1) {
                CMOOP_HIP(hipMemsetAsync(dx, 0, (size_t)B * H * W * Cin * 4, s));
                accumulate = 1;
            }
            ConvGeom gd = g;
            gd.H = g.OH; gd.W = g.OW; gd.Cin = g.Cout; gd.Cout = g.Cin; gd.OH = g.H; gd.OW = g.W;
            const size_t skf = (stride == 1 && Cout >= 16 && (Cout & (Cout - 1)) == 0) ? igemm_splitk_workspace(gd) : 0;
            if (skf) CMOOP_HIP(hipMalloc(&sk, skf * 4));
            conv_backward_data(dy, w, dx, g, wd, mask_relu ? x : nullptr, 1.f, accumulate, s, &hook, sk, skf);
        }
        CMOOP_HIP(hipStreamSynchronize(s));
        hipFree(wg); hipFree(red); hipFree(wd);
        if (sk) hipFree(sk);
        if (tab) hipFree(tab);
    });
}

// ---- the same two operations launched as the trainer launches them -------------------
// ---- the same two operations launched as the trainer launches them -------------------
int cmoop_conv_fwd_trainer(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t H, int32_t W,
                           int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t relu, double* col_sum,
                           double* col_sumsq, int32_t* stats_fused) {
    return guard([&] {
        g_last_kernels.clear();
        hipStream_t s = lib_stream();
        const ConvGeom g = make_geom(B, H, W, Cin, Cout, KS, stride);
        const int64_t M = g.M();
        GemmEpilogue e;
        e.bias = bias; e.relu = relu;
        float *sk = nullptr, *red = nullptr;
        void* tab = nullptr;
        const size_t skf = igemm_splitk_workspace(g);
        const bool want_stats = col_sum != nullptr && col_sumsq != nullptr;
        CMOOP_REQUIRE(!want_stats || Cout % 4 == 0, "statistics need Cout % 4 == 0");
        auto cleanup = [&] { hipStreamSynchronize(s); if (sk) hipFree(sk); if (red) hipFree(red); if (tab) hipFree(tab); };
        try {
            if (skf) CMOOP_HIP(hipMalloc(&sk, skf * 4));
            const int tab_rows = (KS * KS <= 32) ? rowtab_rows(g) : 0;     // Net::build_plan's condition
            if (tab_rows) {
                CMOOP_HIP(hipMalloc(&tab, (size_t)tab_rows * 8));
                launch_build_rowtab(g, tab, s);
            }
            size_t red_floats = 0;
            if (want_stats) {   // Net::build_plan's sizing of the statistics partials
                const size_t blocks = std::max<size_t>((size_t)colreduce_blocks(M, Cout), (size_t)cdiv64(M, 64));
                red_floats = blocks * 2 * Cout + 2 * Cout;
                CMOOP_HIP(hipMalloc(&red, red_floats * 4));
                e.stats = red;
            }
            int nb = 0, flags = 0;
            const int code = launch_igemm_fwd(x, w, y, g, e, s, nullptr, sk, skf, want_stats ? &nb : nullptr, tab, tab_rows, &flags);
            g_last_kernels = gemm_variant_name(0, code, flags);
            if (want_stats) {
                if (stats_fused) *stats_fused = nb > 0 ? 1 : 0;
                if (nb == 0) {          // split-K launch: the trainer falls back to the stand-alone reduction
                    nb = colreduce_blocks(M, Cout);
                    launch_colstats(y, red, M, Cout, nb, s);
                }
                std::vector<float> hp((size_t)nb * 2 * Cout);
                CMOOP_HIP(hipMemcpyAsync(hp.data(), red, hp.size() * 4, hipMemcpyDeviceToHost, s));
                CMOOP_HIP(hipStreamSynchronize(s));
                for (int c = 0; c < Cout; ++c) { col_sum[c] = 0.0; col_sumsq[c] = 0.0; }
                for (int b = 0; b < nb; ++b)
                    for (int c = 0; c < Cout; ++c) {
                        col_sum[c] += (double)hp[((size_t)b * 2) * Cout + c];
                        col_sumsq[c] += (double)hp[((size_t)b * 2 + 1) * Cout + c];
                    }
            }
            CMOOP_HIP(hipStreamSynchronize(s));
        } catch (...) {
            cleanup();
            throw;
        }
        cleanup();
    });
}

int cmoop_conv_bwd_trainer(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int32_t B, int32_t H,
                           int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t stride, int32_t mask_relu) {
    return guard([&] {
        g_last_kernels.clear();
        hipStream_t s = lib_stream();
        const ConvGeom g = make_geom(B, H, W, Cin, Cout, KS, stride);
        float *wg = nullptr, *wd = nullptr, *sk = nullptr, *dwb = nullptr;
        void *tab = nullptr, *tab_d = nullptr;
        auto cleanup = [&] {
            hipStreamSynchronize(s);
            if (wg) hipFree(wg); if (wd) hipFree(wd); if (sk) hipFree(sk); if (dwb) hipFree(dwb); if (tab) hipFree(tab); if (tab_d) hipFree(tab_d);
        };
        try {
            RecordingHook hook(&g_last_kernels);
            const size_t NK = (size_t)g.Cout * g.K();
            const size_t wg_floats = (size_t)wgrad_slices(g) * g.Cout * (g.K() + 1);
            CMOOP_HIP(hipMalloc(&wg, wg_floats * 4));
            CMOOP_HIP(hipMalloc(&dwb, (NK + g.Cout) * 4));          // the trainer's arena layout: bias gradient directly after the kernel gradient
            const int tab_rows = (KS * KS <= 32) ? rowtab_rows(g) : 0;
            if (tab_rows) {
                CMOOP_HIP(hipMalloc(&tab, (size_t)tab_rows * 8));
                launch_build_rowtab(g, tab, s);
            }
            conv_backward_weights(x, dy, dwb, dwb + NK, g, wg, wg_floats, s, &hook, GEMM_DEFAULT, tab, tab_rows);
            CMOOP_HIP(hipMemcpyAsync(dw, dwb, NK * 4, hipMemcpyDeviceToDevice, s));
            CMOOP_HIP(hipMemcpyAsync(db, dwb + NK, (size_t)g.Cout * 4, hipMemcpyDeviceToDevice, s));
            if (dx) {
                int accumulate = 0;
                if (stride != 1) {
                    CMOOP_HIP(hipMemsetAsync(dx, 0, (size_t)B * H * W * Cin * 4, s));
                    accumulate = 1;
                }
                const bool mfma_dgrad = ilog2_exact(Cout) >= 4;
                const ConvGeom gd = dgrad_geometry(g);
                int tab_d_rows = 0;
                size_t skf = 0;
                if (mfma_dgrad) {
                    CMOOP_HIP(hipMalloc(&wd, NK * 4));
                    launch_flip_transpose(w, wd, Cout, KS, KS, Cin, s);       // the trainer's one-launch-per-step refresh, for one layer
                    if (KS * KS <= 32) {
                        tab_d_rows = rowtab_rows(gd);
                        CMOOP_HIP(hipMalloc(&tab_d, (size_t)tab_d_rows * 8));
                        launch_build_rowtab(gd, tab_d, s);
                    }
                    skf = igemm_splitk_workspace(gd);      // the trainer passes its workspace to every dgrad, the skip projection's too
                    if (skf) CMOOP_HIP(hipMalloc(&sk, skf * 4));
                }
                conv_backward_data(dy, w, dx, g, wd, mask_relu ? x : nullptr, 1.f, accumulate, s, &hook, sk, skf, GEMM_DEFAULT, mfma_dgrad,
                                   tab_d, tab_d_rows);
            }
            CMOOP_HIP(hipStreamSynchronize(s));
        } catch (...) {
            cleanup();
            throw;
        }
        cleanup();
    });
}

int cmoop_last_kernels(char* buf, int32_t cap) {
    return guard([&] {
        CMOOP_REQUIRE(buf && cap > 0, "last_kernels: no buffer");
        std::snprintf(buf, cap, "%s", g_last_kernels.c_str());
    });
}

int cmoop_conv_time(int32_t mode, const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t H,
                    int32_t W, int32_t Cin, int32_t Cout, int32_t KS, int32_t iters, double* avg_ms) {
    // mode 0: forward implicit GEMM; 1: dgrad implicit GEMM (y = dY [B,H,W,Cout] in, x = dX out);
    // mode 2: wgrad MFMA kernel only (y = dY in, partials into a scratch buffer)
    return guard([&] {
        hipStream_t s = lib_stream();
        const ConvGeom g = make_geom(B, H, W, Cin, Cout, KS, 1);
        float *wd = nullptr, *wg = nullptr, *sk = nullptr;
        ConvGeom gd = g;
        GemmEpilogue e;
        const int S = wgrad_slices(g);
        if (mode == 0) e.bias = bias;
        if (mode == 1) {
            CMOOP_HIP(hipMalloc(&wd, (size_t)g.Cout * g.K() * 4));
            launch_flip_transpose(w, wd, Cout, KS, KS, Cin, s);
            gd.H = g.OH; gd.W = g.OW; gd.Cin = Cout; gd.Cout = Cin; gd.OH = g.H; gd.OW = g.W;
            gd.pad_t = KS - 1 - g.pad_t; gd.pad_l = KS - 1 - g.pad_l;
        }
        void* tab = nullptr;   // the layer's row table, as the trainer passes it (forward / wgrad: forward geometry; dgrad: its own)
        const int tab_rows = (KS * KS <= 32 && Cin >= 16) ? rowtab_rows(mode == 1 ? gd : g) : 0;
        if (mode == 2) CMOOP_HIP(hipMalloc(&wg, (size_t)S * g.Cout * g.K() * 4));
        if (tab_rows) {
            CMOOP_HIP(hipMalloc(&tab, (size_t)tab_rows * 8));
            launch_build_rowtab(mode == 1 ? gd : g, tab, s);
        }
        const size_t skf = mode == 0 ? igemm_splitk_workspace(g) : (mode == 1 ? igemm_splitk_workspace(gd) : 0);
        if (skf) CMOOP_HIP(hipMalloc(&sk, skf * 4));
        auto once = [&]() {
            if (mode == 0) launch_igemm_fwd(x, w, y, g, e, s, nullptr, sk, skf, nullptr, tab, tab_rows);
            else if (mode == 1) launch_igemm_fwd(y, wd, const_cast<float*>(x), gd, e, s, nullptr, sk, skf, nullptr, tab, tab_rows);
            else launch_igemm_wgrad(x, y, wg, g, S, s, nullptr, nullptr, 0, GEMM_DEFAULT, tab, tab_rows);
        };
        for (int i = 0; i < 3; ++i) once();
        hipEvent_t a, b;
        CMOOP_HIP(hipEventCreate(&a));
        CMOOP_HIP(hipEventCreate(&b));
        CMOOP_HIP(hipEventRecord(a, s));
        for (int i = 0; i < iters; ++i) once();
        CMOOP_HIP(hipEventRecord(b, s));
        CMOOP_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        CMOOP_HIP(hipEventElapsedTime(&ms, a, b));
        hipEventDestroy(a); hipEventDestroy(b);
        if (wd) hipFree(wd);
        if (wg) hipFree(wg);
        if (sk) hipFree(sk);
        if (tab) hipFree(tab);
        *avg_ms = (double)ms / std::max(1, iters);
    });
}

int cmoop_dense_fwd(const float* x, const float* w, const float* bias, float* y, int32_t M, int32_t N, int32_t K, int32_t relu) {
    return guard([&] {
        hipStream_t s = lib_stream();
        launch_dense_fwd(x, w, bias, y, M, N, K, relu, 0, 0u, 0u, 1.f, GEMM_DEFAULT, s);
        CMOOP_HIP(hipStreamSynchronize(s));
    });
}
int cmoop_dense_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int32_t M, int32_t N, int32_t K,
                    int32_t mask_relu) {
    return guard([&] {
        hipStream_t s = lib_stream();
        launch_dense_wgrad(x, dy, dw, db, M, N, K, GEMM_DEFAULT, s);
        if (dx) launch_dense_dgrad(dy, w, dx, M, N, K, mask_relu ? x : nullptr, 1.f, GEMM_DEFAULT, s);
        CMOOP_HIP(hipStreamSynchronize(s));
    });
}
int cmoop_maxpool_fwd(const float* x, float* y, uint8_t* arg, int32_t B, int32_t H, int32_t W, int32_t C) {
    return guard([&] {
        hipStream_t s = lib_stream();
        launch_maxpool_fwd(x, y, arg, B, H, W, C, s);
        CMOOP_HIP(hipStreamSynchronize(s));
    });
}
int cmoop_maxpool_bwd(const float* dy, const uint8_t* arg, const float* y, float* dx, int32_t B, int32_t H, int32_t W,
                      int32_t C, int32_t mask_y_pos) {
    return guard([&] {
        hipStream_t s = lib_stream();
        launch_maxpool_bwd(dy, arg, y, dx, B, H, W, C, mask_y_pos, s);
        CMOOP_HIP(hipStreamSynchronize(s));
    });
}
int cmoop_device_synchronize(void) { return guard([] { CMOOP_HIP(hipDeviceSynchronize()); }); }

}  // extern "C"
