// Audio front end: framing -> Hann -> 512-pt real FFT -> |.|^2 -> sparse Slaney mel -> log.
// North-star addition beneath the reference's data loader (the reference ships
// pre-extracted features only: nsga_penalty.py:64-71); the
// algorithm restates librosa.feature.melspectrogram (requirements.txt:80) and is
// checked against oracle/frontend.py.
//
// One 256-thread workgroup per clip, each wave owns one frame at a time.  A frame is one
// coalesced 2 KB segment of the clip; the 3.2x overlap between consecutive frames is served
// by L2, so HBM sees each clip once (64 000 B in, T*n_mels*4 B out).
// The 512 real samples are packed into 256 complex points (even + i*odd) and transformed by a
// 256-point radix-4 decimation-in-frequency FFT held in registers: each lane owns one radix-4
// butterfly per stage (4 stages), with three exchanges through the wave's own LDS scratch
// (padded so every ds_read/write_b64 is conflict-free) and every twiddle / window value
// precomputed per lane in registers outside the frame loop.  The real-FFT split
// X[k] = E[k] + W^k O[k] follows in LDS-skewed natural order, then the sparse mel: each band is one
// lane's short loop, the widest bands split over two lanes, summed in fixed order.
// Only wave-level synchronisation (LDS operations of one wave complete in issue order).
// Measured 2.19 ms for 30 000 one-second clips = 1.10 TB/s of algorithmic bytes (13.7 % of the HBM roofline;
// the 15 kFLOP of fp32 VALU work per 792-byte frame, not HBM, bounds it); the radix-2 LDS FFT it replaces took 10.3 ms.
#include "kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace cmoop {

struct FrontendTables {
    FrontendCfg cfg;
    float* tw = nullptr;       // [n_fft/2][2] cos, -sin
    float* win = nullptr;      // [n_fft] padded periodic Hann
    float* melw = nullptr;     // sparse weights, band after band
    int* meltask = nullptr;    // [64][3] first bin, count, weight offset of each lane's task ; then [64] second task of a band or -1
    int nnz = 0;
};

static double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

FrontendTables* frontend_tables_create(const FrontendCfg& c) {
    CMOOP_REQUIRE(c.n_fft == 512, "front end: n_fft must be 512");
    CMOOP_REQUIRE(c.n_mels >= 1 && c.n_mels <= 64 && c.win <= c.n_fft, "front end: n_mels <= 64, win <= n_fft");
    auto* t = new FrontendTables;
    t->cfg = c;
    const int half = c.n_fft / 2, nb = half + 1;
    std::vector<float> tw(2 * half), win(c.n_fft, 0.f);
    for (int k = 0; k < half; ++k) {
        const double a = -2.0 * M_PI * k / c.n_fft;
        tw[2 * k] = (float)std::cos(a);
        tw[2 * k + 1] = (float)std::sin(a);
    }
    const int lpad = (c.n_fft - c.win) / 2;
    for (int n = 0; n < c.win; ++n) win[lpad + n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / c.win));
    // Slaney mel basis, norm='slaney' (librosa.filters.mel)
    std::vector<double> mel_f(c.n_mels + 2);
    const double m_lo = hz_to_mel(c.fmin), m_hi = hz_to_mel(c.fmax);
    for (int i = 0; i < c.n_mels + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (c.n_mels + 1));
    std::vector<float> w;
    std::vector<int> first_bin(c.n_mels), count(c.n_mels), start(c.n_mels);
    for (int i = 0; i < c.n_mels; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        int first = -1, cnt = 0;
        start[i] = (int)w.size();
        for (int b = 0; b < nb; ++b) {
            const double f = (double)b * (c.sr / 2.0) / half;
            const double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
            const double v = std::max(0.0, std::min(lower, upper)) * enorm;
            if (v > 0.0) {
                if (first < 0) first = b;
                w.push_back((float)v);     // bins of one triangle are contiguous
                ++cnt;
            }
        }
        first_bin[i] = first < 0 ? 0 : first;
        count[i] = cnt;
    }
    // lane tasks: task i < n_mels = band i; the spare lanes take the second half of the widest bands
    std::vector<int> task(64 * 3 + 64, 0);
    for (int i = 0; i < 64; ++i) task[192 + i] = -1;
    for (int i = 0; i < c.n_mels; ++i) { task[3 * i] = first_bin[i]; task[3 * i + 1] = count[i]; task[3 * i + 2] = start[i]; }
    std::vector<int> order(c.n_mels);
    for (int i = 0; i < c.n_mels; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return count[a] > count[b]; });
    int next = c.n_mels;
    for (int oi = 0; oi < c.n_mels && next < 64; ++oi) {
        const int b = order[oi];
        if (count[b] < 8) break;
        const int h = count[b] / 2;                  // first task keeps bins [0, h), second [h, count)
        task[3 * b + 1] = h;
        task[3 * next] = first_bin[b] + h; task[3 * next + 1] = count[b] - h; task[3 * next + 2] = start[b] + h;
        task[192 + b] = next;
        ++next;
    }
    t->nnz = (int)w.size();
    if (w.empty()) w.push_back(0.f);
    CMOOP_HIP(hipMalloc(&t->tw, tw.size() * 4));
    CMOOP_HIP(hipMalloc(&t->win, win.size() * 4));
    CMOOP_HIP(hipMalloc(&t->melw, w.size() * 4));
    CMOOP_HIP(hipMalloc(&t->meltask, task.size() * 4));
    CMOOP_HIP(hipMemcpy(t->tw, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
    CMOOP_HIP(hipMemcpy(t->win, win.data(), win.size() * 4, hipMemcpyHostToDevice));
    CMOOP_HIP(hipMemcpy(t->melw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CMOOP_HIP(hipMemcpy(t->meltask, task.data(), task.size() * 4, hipMemcpyHostToDevice));
    CMOOP_REQUIRE(t->nnz <= 1024, "front end: mel table too large");
    return t;
}

void frontend_tables_destroy(FrontendTables* t) {
    if (!t) return;
    hipFree(t->tw); hipFree(t->win); hipFree(t->melw); hipFree(t->meltask);
    delete t;
}

constexpr int NFFT = 512, NC = 256;   // real points, packed complex points

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Each wave owns its scratch, so the phases only need the wave's own LDS writes to be visible to its
// other lanes: LDS operations of one wave complete in issue order; the fences only stop the compiler
// from moving the reads above the writes.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ f32x2 cmul(const f32x2 a, const f32x2 w) {
    return f32x2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
}
// radix-4 forward butterfly: b_j = sum_p a_p (-i)^(p j)
__device__ __forceinline__ void bfly4(const f32x2 a0, const f32x2 a1, const f32x2 a2, const f32x2 a3, f32x2& b0, f32x2& b1,
                                      f32x2& b2, f32x2& b3) {
    const f32x2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    const f32x2 nid = f32x2{d13.y, -d13.x};   // -i * d13
    b0 = s02 + s13;
    b2 = s02 - s13;
    b1 = d02 + nid;
    b3 = d02 - nid;
}

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wav, int n_samples, float* __restrict__ out,
                                                     int T, int hop, int n_mels, float log_eps,
                                                     const float* __restrict__ g_tw, const float* __restrict__ g_win,
                                                     const float* __restrict__ g_melw, const int* __restrict__ g_task,
                                                     int nnz) {
    constexpr int XB = 320;    // complex slots of the exchange buffer (pitch-20 / pitch-5 layouts, skewed spectrum)
    constexpr int PB = 264;    // power spectrum bins 0..256
    __shared__ float s_melw[1024];
    __shared__ int s_task[256];
    __shared__ __attribute__((aligned(16))) f32x2 s_x[4][XB];
    __shared__ float s_p[4][PB];
    __shared__ float s_part[4][64];

    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const float* clip = wav + (size_t)blockIdx.x * n_samples;
    for (int i = t; i < nnz; i += 256) s_melw[i] = g_melw[i];
    s_task[t] = g_task[t];
    __syncthreads();

    // ---- per-lane constants, fixed over the frames -------------------------------------------------
    const f32x2* tw = reinterpret_cast<const f32x2*>(g_tw);   // tw[k] = exp(-2 pi i k / 512), k < 256
    auto tw512 = [&](int k) {                                 // k < 512: W^(k + 256) = -W^k
        const f32x2 v = tw[k & 255];
        return (k & 256) ? -v : v;
    };
    f32x2 w0[3], w1[3], w2[3], wp[4];
    float win[8];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        w0[j - 1] = tw512(2 * ((lane * j) & 255));           // W_256^(l j)
        w1[j - 1] = tw512(8 * (((lane & 15) * j) & 63));     // W_64^(n j),  n = l & 15
        w2[j - 1] = tw512(32 * (((lane & 3) * j) & 15));     // W_16^(n j),  n = l & 3
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        wp[u] = tw[lane + 64 * u];                            // W_512^k, k = lane + 64 u
        win[2 * u] = g_win[2 * (lane + 64 * u)];
        win[2 * u + 1] = g_win[2 * (lane + 64 * u) + 1];
    }
    const int j1 = lane >> 4, n1 = lane & 15;                 // stage 1: sub-FFT j1 of 64 points, butterfly n1
    const int s2 = lane >> 2, n2 = lane & 3;                  // stage 2: sub-FFT s2 of 16 points, butterfly n2
    const int klow = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);   // stage 3: output bins klow + 64 j4
    const int task_first = s_task[3 * lane], task_cnt = s_task[3 * lane + 1], task_w = s_task[3 * lane + 2];
    const int task2 = s_task[192 + lane];

    f32x2* xb = s_x[wave];
    float* pw = s_p[wave];
    float* part = s_part[wave];
    const int iters = (T + 3) >> 2;
    for (int it = 0; it < iters; ++it) {
        const int frame = it * 4 + wave;
        if (frame >= T) break;                                // wave-uniform; no workgroup barrier below
        // windowed frame, centre-padded with zeros (pad_mode='constant'); z[m] = x[2m] + i x[2m+1]
        const int base = frame * hop - NFFT / 2;
        f32x2 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i0 = base + 2 * (lane + 64 * u);
            const float x0 = (i0 >= 0 && i0 < n_samples) ? clip[i0] : 0.f;
            const float x1 = (i0 + 1 >= 0 && i0 + 1 < n_samples) ? clip[i0 + 1] : 0.f;
            a[u] = f32x2{x0 * win[2 * u], x1 * win[2 * u + 1]};
        }
        // stage 0: points l + 64 p  ->  four 64-point sequences y_j[l]
        bfly4(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
        xb[lane] = b[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) xb[64 * j + lane] = cmul(b[j], w0[j - 1]);
        wave_sync();
        // stage 1: y_j[n + 16 p] -> sixteen 16-point sequences, stored with pitch 20
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p] = xb[64 * j1 + n1 + 16 * p];
        wave_sync();
        bfly4(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
        xb[(4 * j1) * 20 + n1] = b[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) xb[(4 * j1 + j) * 20 + n1] = cmul(b[j], w1[j - 1]);
        wave_sync();
        // stage 2: u[n + 4 p] -> sixty-four 4-point sequences, stored with pitch 5
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p] = xb[20 * s2 + n2 + 4 * p];
        wave_sync();
        bfly4(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
        xb[(4 * s2) * 5 + n2] = b[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) xb[(4 * s2 + j) * 5 + n2] = cmul(b[j], w2[j - 1]);
        wave_sync();
        // stage 3: 4-point transforms; Z[k], k = klow + 64 j4, stored in natural order skewed by k >> 4
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p] = xb[5 * lane + p];
        wave_sync();
        bfly4(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = klow + 64 * j;
            xb[k + (k >> 4)] = b[j];
        }
        wave_sync();
        // real-FFT split and power spectrum: X[k] = E[k] + W_512^k O[k], E = (Z[k] + conj Z[256-k]) / 2, O = (Z[k] - conj Z[256-k]) / 2i
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = lane + 64 * u, kc = (NC - k) & (NC - 1);
            const f32x2 z = xb[k + (k >> 4)], zc = xb[kc + (kc >> 4)];
            const f32x2 e = f32x2{0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y)};
            const f32x2 o = f32x2{0.5f * (z.y + zc.y), -0.5f * (z.x - zc.x)};
            const f32x2 x = e + cmul(o, wp[u]);
            pw[k] = x.x * x.x + x.y * x.y;
            if (k == 0) {                                     // bin 256: E[0] - O[0]
                const float xn = e.x - o.x;
                pw[NC] = xn * xn;
            }
        }
        wave_sync();
        // sparse mel: one short loop per lane task, second halves of the widest bands on the spare lanes
        {
            float acc = 0.f;
            for (int i = 0; i < task_cnt; ++i) acc = fmaf(s_melw[task_w + i], pw[task_first + i], acc);
            part[lane] = acc;
        }
        wave_sync();
        if (lane < n_mels) {
            float acc = part[lane];
            if (task2 >= 0) acc += part[task2];
            out[((size_t)blockIdx.x * T + frame) * n_mels + lane] = logf(acc + log_eps);
        }
        wave_sync();
    }
}

void launch_logmel(const float* wav, int64_t n_clips, int n_samples, float* out, const FrontendTables* t, hipStream_t s) {
    if (n_clips == 0) return;
    const FrontendCfg& c = t->cfg;
    const int T = 1 + n_samples / c.hop;
    hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)n_clips), dim3(256), 0, s, wav, n_samples, out, T, c.hop, c.n_mels,
                       c.log_eps, t->tw, t->win, t->melw, t->meltask, t->nnz);
    CMOOP_HIP(hipGetLastError());
}

// MFCC option ("optional DCT-II ortho -> 40 MFCC"; the reference's own comment calls its features MFCCs,
// sa_nsga_init.py:68): out[row][k] = s_k * sum_f x[row][f] * cos(pi (f + 1/2) k / n), s_0 = sqrt(1/n), s_k = sqrt(2/n)
// -- scipy.fft.dct(type=2, norm="ortho") along the mel axis, first n_out coefficients.  HBM-bound (rows = clips x frames):
// a workgroup stages 32 rows and the n x n basis (built in double, once per workgroup) in LDS, 8 threads per row.
constexpr int MFCC_MAX = 64, MFCC_ROWS = 32;
__global__ __launch_bounds__(256) void mfcc_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t rows, int n,
                                                   int n_out) {
    __shared__ float basis[MFCC_MAX * (MFCC_MAX + 1)];
    __shared__ float xs[MFCC_ROWS * (MFCC_MAX + 1)];
    const int t = threadIdx.x;
    for (int i = t; i < n_out * n; i += 256) {
        const int k = i / n, f = i - k * n;
        const double sk = k == 0 ? sqrt(1.0 / n) : sqrt(2.0 / n);
        basis[k * (MFCC_MAX + 1) + f] = (float)(sk * cos(M_PI * (f + 0.5) * k / n));
    }
    const int64_t row0 = (int64_t)blockIdx.x * MFCC_ROWS;
    for (int i = t; i < MFCC_ROWS * n; i += 256) {
        const int r = i / n, f = i - r * n;
        xs[r * (MFCC_MAX + 1) + f] = row0 + r < rows ? X[(row0 + r) * n + f] : 0.f;
    }
    __syncthreads();
    const int r = t >> 3;
    if (row0 + r >= rows) return;
    for (int k = t & 7; k < n_out; k += 8) {
        float acc = 0.f;
        for (int f = 0; f < n; ++f) acc = fmaf(xs[r * (MFCC_MAX + 1) + f], basis[k * (MFCC_MAX + 1) + f], acc);
        Y[(row0 + r) * n_out + k] = acc;
    }
}

void launch_mfcc(const float* X, float* Y, int64_t rows, int n_mels, int n_mfcc, hipStream_t s) {
    CMOOP_REQUIRE(n_mels >= 1 && n_mels <= MFCC_MAX && n_mfcc >= 1 && n_mfcc <= n_mels, "mfcc: 1 <= n_mfcc <= n_mels <= 64");
    if (rows == 0) return;
    hipLaunchKernelGGL(mfcc_kernel, dim3((unsigned)cdiv64(rows, MFCC_ROWS)), dim3(256), 0, s, X, Y, rows, n_mels, n_mfcc);
    CMOOP_HIP(hipGetLastError());
}

// StandardScaler (nsga_penalty.py:103-141): mean / sqrt(biased var) per mel bin over N*T rows
__global__ void colstats_f64_kernel(const float* __restrict__ P, int blocks, int64_t M, int C, double* __restrict__ mean,
                                    double* __restrict__ scale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < blocks; ++b) {
        s1 += (double)P[(size_t)b * 2 * C + c];
        s2 += (double)P[(size_t)b * 2 * C + C + c];
    }
    const double mu = s1 / (double)M;
    double var = s2 / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    double sc = sqrt(var);
    if (sc == 0.0) sc = 1.0;
    mean[c] = mu;
    scale[c] = sc;
}

void colstats_finalize_f64(const float* P, int blocks, int64_t M, int C, double* mean, double* scale, hipStream_t s) {
    hipLaunchKernelGGL(colstats_f64_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, P, blocks, M, C, mean, scale);
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void standardize_kernel(float* __restrict__ X, const double* __restrict__ mean,
                                                          const double* __restrict__ scale, int64_t n, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        X[i] = (float)(((double)X[i] - mean[c]) / scale[c]);
    }
}

void launch_standardize(float* X, const double* mean, const double* scale, int64_t rows, int C, hipStream_t s) {
    const int64_t n = rows * C;
    if (n == 0) return;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, 256), 16384));
    hipLaunchKernelGGL(standardize_kernel, dim3(grid), dim3(256), 0, s, X, mean, scale, n, C);
    CMOOP_HIP(hipGetLastError());
}

}  // namespace cmoop
