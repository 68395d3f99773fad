// extern "C" boundary of libcmoop_hip.so -- see include/cmoop.h for the contract.
#include "../../include/cmoop.h"
#include "net.h"

#include <cstring>
#include <mutex>
#include <vector>

using namespace cmoop;

static thread_local std::string g_err;

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
            launch_igemm_fwd(x, w, y, g, e, s, nullptr, sk, skf);
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
        conv_backward_weights(x, dy, dw, db, g, wg, wg_floats, s, nullptr, GEMM_DEFAULT, tab, tab_rows);
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
            conv_backward_data(dy, w, dx, g, wd, mask_relu ? x : nullptr, 1.f, accumulate, s, nullptr, sk, skf);
        }
        CMOOP_HIP(hipStreamSynchronize(s));
        hipFree(wg); hipFree(red); hipFree(wd);
        if (sk) hipFree(sk);
        if (tab) hipFree(tab);
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
