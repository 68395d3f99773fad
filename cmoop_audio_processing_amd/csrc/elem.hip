// HBM-bound kernels of the candidate-training path: first-layer direct conv,
// BatchNorm (train / eval / backward), max-pool, residual add, GAP, softmax-CE,
// Adam, confusion matrix.  All loads/stores are float4 along the channel axis of
// NHWC tensors (coalesced 16 B/lane); every cross-thread sum has a fixed order so
// a training run is bit-reproducible.
#include "kernels.h"
#include <cstdlib>
#include <algorithm>

namespace cmoop {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// row of the resident tensor a batch position maps to: idx[pos] (the epoch permutation) or pos itself; with n_rows > 0
// the result is clamped into [0, n_rows): a corrupt index can then mis-train but never fault
__device__ __forceinline__ int64_t gather_row(const int32_t* __restrict__ idx, int64_t pos, int64_t n_rows) {
    int64_t src = idx ? (int64_t)idx[pos] : pos;
    if (n_rows > 0) src = src < 0 ? 0 : (src >= n_rows ? n_rows - 1 : src);
    return src;
}

// ===========================================================================
// first layer: C_in = 1 direct conv (Keras Conv2D(filters, k, padding='same') on
// the (T,F,1) input, nsga_penalty.py:255 / sa_nsga_penalty.py:151)
// ===========================================================================
// Thread = 4 consecutive pixels of one row x 4 output channels: each input row segment (4 + KS - 1 values)
// is loaded once and reused by the KS horizontal taps of all four pixels; weights come from LDS as float4.
template <int KS>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ X, const int32_t* __restrict__ idx,
                                                        int64_t row0, const float* __restrict__ Wt,
                                                        const float* __restrict__ bias, float* __restrict__ Y, int B,
                                                        int H, int W, int Cout, int relu, const StepState* __restrict__ st,
                                                        int64_t n_rows) {
    constexpr int TAPS = KS * KS, SEG = 4 + KS - 1, p = (KS - 1) >> 1;
    __shared__ __attribute__((aligned(16))) float Ws[TAPS * 64];
    if (st) row0 = st->row0;
    const int t = threadIdx.x;
    for (int i = t; i < TAPS * Cout; i += 256) {
        int co = i / TAPS, tap = i - co * TAPS;
        Ws[tap * Cout + co] = Wt[i];
    }
    __syncthreads();
    const int TPG = Cout >> 2, GPB = 256 / TPG;
    const int W4 = (W + 3) >> 2;
    const int64_t group = (int64_t)blockIdx.x * GPB + t / TPG;
    const int c4 = t % TPG;
    if (group >= (int64_t)B * H * W4) return;
    const int b = (int)(group / (H * W4)), r = (int)(group - (int64_t)b * H * W4);
    const int h = r / W4, w0 = (r - h * W4) * 4;
    const int64_t src = gather_row(idx, row0 + b, n_rows);
    const float* xb = X + src * (int64_t)H * W;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 4 * c4);
    f32x4 acc[4] = {bv, bv, bv, bv};
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
        const int ih = h + kh - p;
        if ((unsigned)ih >= (unsigned)H) continue;
        float xr[SEG];
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
            const int iw = w0 - p + j;
            xr[j] = (unsigned)iw < (unsigned)W ? xb[ih * W + iw] : 0.f;
        }
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(&Ws[(kh * KS + kw) * Cout + 4 * c4]);
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const float xv = xr[px + kw];
                acc[px][0] = fmaf(xv, w4[0], acc[px][0]); acc[px][1] = fmaf(xv, w4[1], acc[px][1]);
                acc[px][2] = fmaf(xv, w4[2], acc[px][2]); acc[px][3] = fmaf(xv, w4[3], acc[px][3]);
            }
        }
    }
    const int64_t pix0 = ((int64_t)b * H + h) * W + w0;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        if (w0 + px >= W) break;
        f32x4 v = acc[px];
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *reinterpret_cast<f32x4*>(Y + (pix0 + px) * Cout + 4 * c4) = v;
    }
}

// ---------------------------------------------------------------------------
// The same layer on the matrix core [r3]: out[pixel][co] = sum_tap x[pixel + tap] W[co][tap] is a GEMM with K = KS*KS (9 / 25,
// padded to 12 / 28 with zero weights).  It is not the 0.8 GFLOP that matter -- the layer is bound by its 66 MB output -- but the
// 400 FMAs per thread of the VALU form above execute on the SIMDs the other candidates' MFMA kernels are issuing on.  Here a
// workgroup stages the single-channel input halo of 256 consecutive output pixels (a few KB; rows of the virtual tall image as
// in halo_fwd_kernel, the shuffle gather resolved once per halo ROW), every lane reads its K values straight from that image
// (one ds_read_b32 per MFMA operand), the weights sit in registers, and the epilogue is the straight-line one of the halo
// kernel with the BatchNorm statistics partials riding along (no stand-alone statistics pass after the first layer).
// ---------------------------------------------------------------------------
struct Conv1Dev {
    int B, H, W, Cout, M, WP, VH, rows_max;
    uint32_t hw_magic, hw_shift, w_magic, w_shift, wp_magic, wp_shift, vh_magic, vh_shift;
};
static void c1_fastdiv_init(int d, uint32_t* magic, uint32_t* shift) {     // Granlund-Montgomery, exact for n < 2^31 (as in gemm.hip)
    uint32_t sh = 0;
    while ((1u << sh) < (uint32_t)d) ++sh;
    *shift = sh;
    *magic = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << sh) - (uint64_t)d)) / (uint64_t)d + 1);
}
__device__ __forceinline__ int c1_fastdiv(int n, uint32_t magic, uint32_t shift) {
    return (int)((__umulhi((uint32_t)n, magic) + (uint32_t)n) >> shift);
}

template <int KS, int CT>   // CT = Cout / 16
__global__ __launch_bounds__(256) void conv1_fwd_mfma_kernel(const float* __restrict__ X, const int32_t* __restrict__ idx, int64_t row0,
                                                             const float* __restrict__ Wt, const float* __restrict__ bias,
                                                             float* __restrict__ Y, Conv1Dev g, int relu,
                                                             const StepState* __restrict__ st, int64_t n_rows,
                                                             float* __restrict__ stats) {
    constexpr int R = KS / 2, T = KS * KS, NJ = (T + 3) / 4, BM = 256, RT = 4, NIT = 4;
    extern __shared__ __attribute__((aligned(16))) float c1_lds[];
    float* const img = c1_lds;                                                   // [rows_max][WP]
    long long* const row_src = reinterpret_cast<long long*>(c1_lds + g.rows_max * g.WP);   // element offset of each halo row's x = 0, or -1
    if (st) row0 = st->row0;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * BM, N = g.Cout;
    auto vrow = [&](int m, int& x) {
        const int b = c1_fastdiv(m, g.hw_magic, g.hw_shift), r = m - b * (g.H * g.W);
        const int y = c1_fastdiv(r, g.w_magic, g.w_shift);
        x = r - y * g.W;
        return b * g.VH + y;
    };
    int xd;
    const int vbase = vrow(m0, xd), vlast = vrow(min(m0 + BM, g.M) - 1, xd);
    const int nrows = vlast - vbase + 1 + 2 * R;
    if (t < g.rows_max) {
        const int v = vbase - R + t, vc = max(v, 0);
        const int b = c1_fastdiv(vc, g.vh_magic, g.vh_shift), y = vc - b * g.VH;
        const bool ok = t < nrows && v >= 0 && b < g.B && y < g.H;
        row_src[t] = ok ? gather_row(idx, row0 + b, n_rows) * (long long)(g.H * g.W) + (long long)y * g.W : -1ll;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = t + 256 * it;
        if (i < g.rows_max * g.WP) {
            const int hy = c1_fastdiv(i, g.wp_magic, g.wp_shift), x = i - hy * g.WP - R;
            const long long src = row_src[hy];
            img[i] = (src >= 0 && (unsigned)x < (unsigned)g.W) ? X[src + x] : 0.f;
        }
    }
    // weights: lane (co = 16 ct + lr, k slot q) holds tap 4 j + q of its output channel for the j-th MFMA (zero past the window)
    float bw[CT][NJ];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < NJ; ++j) bw[ct][j] = (4 * j + q < T) ? Wt[(ct * 16 + lr) * T + 4 * j + q] : 0.f;
    // this lane's tap offsets inside the halo image, and its four pixels
    int tap_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int tap = min(4 * j + q, T - 1), ky = tap / KS;
        tap_off[j] = ky * g.WP + (tap - ky * KS);
    }
    int a_base[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        int x;
        const int v = vrow(min(m0 + wave * 64 + rt * 16 + lr, g.M - 1), x);
        a_base[rt] = (v - vbase) * g.WP + x;
    }
    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float a = img[a_base[rt] + tap_off[j]];
            if (4 * j + q >= T) a = 0.f;                   // the padded K slots: 0 x W(0) even if the image held a non-finite value
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[ct][j], acc[rt][ct], 0, 0, 0);
        }

    // epilogue (C/D map of 16x16x4: col = lane & 15, row = 4 (lane >> 4) + reg); whole tiles take the straight-line form
    float csum[CT], csq[CT], bias_v[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { csum[ct] = 0.f; csq[ct] = 0.f; bias_v[ct] = bias[ct * 16 + lr]; }
    if (m0 + BM <= g.M) {
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, g.M * N * 4, 0x00020000);
        const uint32_t e_lane = (uint32_t)((m0 + wave * 64 + q * 4) * N + lr) * 4u;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int soff = (rt * 16 + r) * N * 4;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    float v = acc[rt][ct][r] + bias_v[ct];
                    v = relu ? fmaxf(v, 0.f) : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)(e_lane + ct * 64), soff, 0);
                    csum[ct] += v; csq[ct] += v * v;
                }
            }
    } else {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wave * 64 + rt * 16 + q * 4 + r;
                if (row >= g.M) continue;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    float v = acc[rt][ct][r] + bias_v[ct];
                    v = relu ? fmaxf(v, 0.f) : v;
                    Y[(size_t)row * N + ct * 16 + lr] = v;
                    csum[ct] += v; csq[ct] += v * v;
                }
            }
    }
    if (stats) {   // column partials of this tile: lane's rows -> the wave's four row groups (shuffle) -> the four waves (LDS), fixed order
        __syncthreads();
        float* red = c1_lds;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            csum[ct] += __shfl_xor(csum[ct], 16, 64); csq[ct] += __shfl_xor(csq[ct], 16, 64);
            csum[ct] += __shfl_xor(csum[ct], 32, 64); csq[ct] += __shfl_xor(csq[ct], 32, 64);
            if (q == 0) { red[(wave * N + ct * 16 + lr) * 2] = csum[ct]; red[(wave * N + ct * 16 + lr) * 2 + 1] = csq[ct]; }
        }
        __syncthreads();
        if (t < N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a += red[(w * N + t) * 2]; b += red[(w * N + t) * 2 + 1]; }
            stats[((size_t)blockIdx.x * 2) * N + t] = a;
            stats[((size_t)blockIdx.x * 2 + 1) * N + t] = b;
        }
    }
}

void launch_conv1_fwd(const float* X, const int32_t* idx, int64_t row0, const float* Wt, const float* bias, float* Y,
                      int B, int H, int W, int Cout, int KS, int relu, hipStream_t s, const StepState* st, int64_t n_rows,
                      float* stats, int* stats_blocks) {
    CMOOP_REQUIRE(Cout % 4 == 0 && Cout <= 64 && 64 % (Cout / 4) == 0 && (KS == 3 || KS == 5), "conv1: unsupported shape");
    if (stats_blocks) *stats_blocks = 0;
    // matrix-core form (default; CMOOP_CONV1_MFMA=0 keeps the VALU kernel): C_out a multiple of 16, halo image within the staging slots
    static const bool mfma_on = [] { const char* v = std::getenv("CMOOP_CONV1_MFMA"); return !(v && v[0] == '0'); }();
    {
        const int R = KS / 2, wp = (W + KS - 1 + 7) / 8 * 8;
        const int span = (256 - 2 + W) / W + 1 + ((256 - 1) / (H * W) + 1) * R, rows = span + 2 * R;
        const int64_t M = (int64_t)B * H * W;
        if (mfma_on && (Cout == 16 || Cout == 32 || Cout == 64) && rows * wp <= 4 * 256 && rows <= 256 && M > 0 && M * Cout < (1ll << 29)) {
            Conv1Dev g;
            g.B = B; g.H = H; g.W = W; g.Cout = Cout; g.M = (int)M; g.WP = wp; g.VH = H + R; g.rows_max = rows;
            c1_fastdiv_init(H * W, &g.hw_magic, &g.hw_shift);
            c1_fastdiv_init(W, &g.w_magic, &g.w_shift);
            c1_fastdiv_init(wp, &g.wp_magic, &g.wp_shift);
            c1_fastdiv_init(H + R, &g.vh_magic, &g.vh_shift);
            const size_t lds = std::max((size_t)rows * wp * 4 + (size_t)rows * 8, (size_t)4 * Cout * 2 * 4);
            const dim3 grid((unsigned)cdiv64(M, 256));
#define CMOOP_C1(KS_, CT_) hipLaunchKernelGGL((conv1_fwd_mfma_kernel<KS_, CT_>), grid, dim3(256), lds, s, X, idx, row0, Wt, bias, Y, g, relu, st, n_rows, stats)
            if (KS == 3) { if (Cout == 16) CMOOP_C1(3, 1); else if (Cout == 32) CMOOP_C1(3, 2); else CMOOP_C1(3, 4); }
            else         { if (Cout == 16) CMOOP_C1(5, 1); else if (Cout == 32) CMOOP_C1(5, 2); else CMOOP_C1(5, 4); }
#undef CMOOP_C1
            CMOOP_HIP(hipGetLastError());
            if (stats && stats_blocks) *stats_blocks = (int)grid.x;
            return;
        }
    }
    const int GPB = 256 / (Cout / 4);
    const int64_t groups = (int64_t)B * H * ((W + 3) / 4);
    if (groups == 0) return;
    const dim3 grid((unsigned)cdiv64(groups, GPB));
    if (KS == 3) hipLaunchKernelGGL(conv1_fwd_kernel<3>, grid, dim3(256), 0, s, X, idx, row0, Wt, bias, Y, B, H, W, Cout, relu, st, n_rows);
    else hipLaunchKernelGGL(conv1_fwd_kernel<5>, grid, dim3(256), 0, s, X, idx, row0, Wt, bias, Y, B, H, W, Cout, relu, st, n_rows);
    CMOOP_HIP(hipGetLastError());
}

// partial slabs of the first layer's weight gradient: one per workgroup = (sample, chunk of image rows); ~512 chip-wide
static int conv1_row_chunks(int B, int H) { return std::max(1, std::min(H, 512 / std::max(B, 1))); }
int conv1_wgrad_blocks(int B, int H, int W) {
    (void)W;
    return std::max(1, B) * conv1_row_chunks(B, H);
}

template <int KS>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ X, const int32_t* __restrict__ idx,
                                                          int64_t row0, const float* __restrict__ dY,
                                                          float* __restrict__ P, int B, int H, int W, int Cout,
                                                          const StepState* __restrict__ st, int64_t n_rows) {
    constexpr int TAPS = KS * KS, SEG = 4 + KS - 1, p = (KS - 1) >> 1;
    __shared__ float red[4 * (TAPS + 1) * 64];
    if (st) row0 = st->row0;
    const int t = threadIdx.x;
    const int TPG = Cout >> 2, GPB = 256 / TPG;
    const int gl = t / TPG, c4 = t % TPG;
    const int W4 = (W + 3) >> 2;
    const int64_t groups = (int64_t)B * H * W4;
    f32x4 acc[TAPS + 1];
#pragma unroll
    for (int i = 0; i <= TAPS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int64_t group = (int64_t)blockIdx.x * GPB + gl; group < groups; group += (int64_t)gridDim.x * GPB) {
        const int b = (int)(group / (H * W4)), r = (int)(group - (int64_t)b * H * W4);
        const int h = r / W4, w0 = (r - h * W4) * 4;
        const int64_t src = gather_row(idx, row0 + b, n_rows);
        const float* xb = X + src * (int64_t)H * W;
        const int64_t pix0 = ((int64_t)b * H + h) * W + w0;
        f32x4 dy[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            dy[px] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (w0 + px < W) dy[px] = *reinterpret_cast<const f32x4*>(dY + (pix0 + px) * Cout + 4 * c4);
            acc[TAPS] += dy[px];
        }
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            const int ih = h + kh - p;
            if ((unsigned)ih >= (unsigned)H) continue;
            float xr[SEG];
#pragma unroll
            for (int j = 0; j < SEG; ++j) {
                const int iw = w0 - p + j;
                xr[j] = (unsigned)iw < (unsigned)W ? xb[ih * W + iw] : 0.f;
            }
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                f32x4& a = acc[kh * KS + kw];
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const float xv = xr[px + kw];
                    a[0] = fmaf(xv, dy[px][0], a[0]); a[1] = fmaf(xv, dy[px][1], a[1]);
                    a[2] = fmaf(xv, dy[px][2], a[2]); a[3] = fmaf(xv, dy[px][3], a[3]);
                }
            }
        }
    }
    // lanes sharing c4 sit TPG apart inside the wave: butterfly down to TPG
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int i = 0; i <= TAPS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[i][j];
            for (int off = 32; off >= TPG; off >>= 1) v += __shfl_xor(v, off, 64);
            acc[i][j] = v;
        }
    if (lane < TPG) {
#pragma unroll
        for (int i = 0; i <= TAPS; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(wave * (TAPS + 1) + i) * 64 + 4 * c4 + j] = acc[i][j];
    }
    __syncthreads();
    float* Pb = P + (size_t)blockIdx.x * (Cout * (TAPS + 1));
    for (int i = t; i < (TAPS + 1) * Cout; i += 256) {
        const int tap = i / Cout, co = i - tap * Cout;
        float v = red[(0 * (TAPS + 1) + tap) * 64 + co] + red[(1 * (TAPS + 1) + tap) * 64 + co];
        v += red[(2 * (TAPS + 1) + tap) * 64 + co];
        v += red[(3 * (TAPS + 1) + tap) * 64 + co];
        if (tap < TAPS) Pb[co * TAPS + tap] = v;        // kernel grad, canonical [co][tap]
        else Pb[Cout * TAPS + co] = v;                  // bias grad
    }
}

// The same gradient on the matrix cores (Cout in {16, 32, 64}: the search space): dW[tap][co] = sum over pixels of
// x[pixel + tap] * dY[pixel][co] is a [taps x pixels] x [pixels x Cout] product with a 66 MB reduction axis for 1.7 k
// outputs.  One v_mfma_f32_16x16x4_f32 reduces FOUR pixels: lane (lr, q) supplies A[tap 16 tt + lr][pixel q] -- its own
// tap's shifted input value, a gather from the 16 KB image (L1) -- and B[pixel q][column lr]; the bias gradient is the
// extra "tap" whose input is the constant 1.  dY is read once, as one 16 / 8 / 4-byte load per lane of NCH consecutive
// channels: MFMA j of a group uses component j, so column lr of accumulator j IS channel NCH*lr + j (a permutation of
// the output columns, undone when the tile is written).  Workgroup = (sample, row chunk); its four waves take rows
// h, h+4, ...; U groups of 4 pixels have their loads in flight together.  The VALU form above took 49 us per step on
// 64 filters k5 (104 accumulator registers, 40 scalar loads per 400 FMAs); the floor here is the 66 MB read of dY.
template <int KS, int NCH, int U>
__global__ __launch_bounds__(256) void conv1_wgrad_mfma_kernel(const float* __restrict__ X, const int32_t* __restrict__ idx,
                                                               int64_t row0, const float* __restrict__ dY,
                                                               float* __restrict__ P, int H, int W, int RC,
                                                               const StepState* __restrict__ st, int64_t n_rows) {
    constexpr int TAPS = KS * KS, TT = (TAPS + 1 + 15) / 16, pad = (KS - 1) >> 1, Cout = 16 * NCH;
    typedef float dyvec __attribute__((ext_vector_type(NCH)));
    __shared__ __attribute__((aligned(16))) float red[3 * TT * NCH * 256];
    if (st) row0 = st->row0;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int b = blockIdx.x / RC, rc = blockIdx.x - b * RC;
    const int rows_per = (H + RC - 1) / RC;
    const int h_begin = rc * rows_per, h_end = min(H, h_begin + rows_per);
    const int64_t src = gather_row(idx, row0 + b, n_rows);
    const float* xb = X + src * (int64_t)H * W;
    const float* dyb = dY + (int64_t)b * H * W * Cout + NCH * lr;
    int dh[TT], dw[TT], kind[TT];   // kind 0: a real tap (input shifted by dh, dw); 1: the bias row (input 1); 2: unused row
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int tap = tt * 16 + lr;
        kind[tt] = tap < TAPS ? 0 : (tap == TAPS ? 1 : 2);
        dh[tt] = tap < TAPS ? tap / KS - pad : 0;
        dw[tt] = tap < TAPS ? tap % KS - pad : 0;
    }
    f32x4 acc[TT][NCH];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int j = 0; j < NCH; ++j) acc[tt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int h = h_begin + wave; h < h_end; h += 4) {
        const float* xrow[TT];
        bool rok[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int ih = h + dh[tt];
            rok[tt] = kind[tt] == 0 && (unsigned)ih < (unsigned)H;
            xrow[tt] = xb + (rok[tt] ? ih : 0) * W + dw[tt];
        }
        const float* dyrow = dyb + (int64_t)h * W * Cout;
        for (int w0 = 0; w0 < W; w0 += 4 * U) {
            float a[U][TT];
            dyvec d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int w = w0 + 4 * u + q;
                const bool ok = w < W;
#pragma unroll
                for (int j = 0; j < NCH; ++j) d[u][j] = 0.f;
                if (ok) d[u] = *reinterpret_cast<const dyvec*>(dyrow + (int64_t)w * Cout);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const bool in = ok && rok[tt] && (unsigned)(w + dw[tt]) < (unsigned)W;
                    const float xv = in ? xrow[tt][w] : 0.f;
                    a[u][tt] = kind[tt] == 1 ? 1.f : xv;      // where the pixel is outside the row, dY is 0
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int j = 0; j < NCH; ++j)
                        acc[tt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][tt], d[u][j], acc[tt][j], 0, 0, 0);
        }
    }
    // waves 1..3 park their tiles, wave 0 adds them in wave order and writes the slab
    if (wave > 0) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                *reinterpret_cast<f32x4*>(&red[(((wave - 1) * TT + tt) * NCH + j) * 256 + lane * 4]) = acc[tt][j];
    }
    __syncthreads();
    if (wave != 0) return;
    float* Pb = P + (size_t)blockIdx.x * (Cout * (TAPS + 1));
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f32x4 v = acc[tt][j];
#pragma unroll
            for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const f32x4*>(&red[((w * TT + tt) * NCH + j) * 256 + lane * 4]);
            const int co = NCH * lr + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tap = tt * 16 + 4 * q + r;
                if (tap < TAPS) Pb[co * TAPS + tap] = v[r];           // kernel grad, canonical [co][tap]
                else if (tap == TAPS) Pb[Cout * TAPS + co] = v[r];    // bias grad
            }
        }
}

template <int KS, int NCH>
static void launch_conv1_wgrad_mfma(const float* X, const int32_t* idx, int64_t row0, const float* dY, float* P, int B, int H,
                                    int W, hipStream_t s, const StepState* st, int64_t n_rows) {
    const int RC = conv1_row_chunks(B, H), W4 = (W + 3) / 4;
    const dim3 grid((unsigned)(B * RC));
    // groups of 4 pixels per row, U at a time: pick the U that wastes no masked group (W = 40: 10 groups = 2 x 5)
    if (W4 % 5 == 0)
        hipLaunchKernelGGL((conv1_wgrad_mfma_kernel<KS, NCH, 5>), grid, dim3(256), 0, s, X, idx, row0, dY, P, H, W, RC, st, n_rows);
    else if (W4 % 3 == 0 && W4 % 4 != 0)
        hipLaunchKernelGGL((conv1_wgrad_mfma_kernel<KS, NCH, 3>), grid, dim3(256), 0, s, X, idx, row0, dY, P, H, W, RC, st, n_rows);
    else
        hipLaunchKernelGGL((conv1_wgrad_mfma_kernel<KS, NCH, 4>), grid, dim3(256), 0, s, X, idx, row0, dY, P, H, W, RC, st, n_rows);
}

void launch_conv1_wgrad(const float* X, const int32_t* idx, int64_t row0, const float* dY, float* P, int B, int H,
                        int W, int Cout, int KS, hipStream_t s, const StepState* st, int64_t n_rows) {
    CMOOP_REQUIRE(Cout % 4 == 0 && Cout <= 64 && 64 % (Cout / 4) == 0, "conv1 wgrad: unsupported Cout");
    if (B == 0) return;
    if ((Cout == 16 || Cout == 32 || Cout == 64) && (KS == 3 || KS == 5)) {
        const int nch = Cout / 16;
        if (KS == 3) {
            if (nch == 1) launch_conv1_wgrad_mfma<3, 1>(X, idx, row0, dY, P, B, H, W, s, st, n_rows);
            else if (nch == 2) launch_conv1_wgrad_mfma<3, 2>(X, idx, row0, dY, P, B, H, W, s, st, n_rows);
            else launch_conv1_wgrad_mfma<3, 4>(X, idx, row0, dY, P, B, H, W, s, st, n_rows);
        } else {
            if (nch == 1) launch_conv1_wgrad_mfma<5, 1>(X, idx, row0, dY, P, B, H, W, s, st, n_rows);
            else if (nch == 2) launch_conv1_wgrad_mfma<5, 2>(X, idx, row0, dY, P, B, H, W, s, st, n_rows);
            else launch_conv1_wgrad_mfma<5, 4>(X, idx, row0, dY, P, B, H, W, s, st, n_rows);
        }
        CMOOP_HIP(hipGetLastError());
        return;
    }
    const int nb = conv1_wgrad_blocks(B, H, W);
    if (KS == 3) hipLaunchKernelGGL(conv1_wgrad_kernel<3>, dim3(nb), dim3(256), 0, s, X, idx, row0, dY, P, B, H, W, Cout, st, n_rows);
    else if (KS == 5) hipLaunchKernelGGL(conv1_wgrad_kernel<5>, dim3(nb), dim3(256), 0, s, X, idx, row0, dY, P, B, H, W, Cout, st, n_rows);
    else CMOOP_REQUIRE(false, "conv1 wgrad: kernel size must be 3 or 5");
    CMOOP_HIP(hipGetLastError());
}

// ===========================================================================
// per-channel reductions over [M][C]
// ===========================================================================
int colreduce_blocks(int64_t M, int C) {
    const int lanes = C / 4;
    const int rpp = std::max(1, 256 / lanes);
    int64_t nb = cdiv64(M, (int64_t)rpp * 16);
    return (int)std::max<int64_t>(1, std::min<int64_t>(1024, nb));
}

struct StatsOp {   // (sum x, sum x^2)
    const float* X;
    __device__ void operator()(size_t off, int, f32x4& s0, f32x4& s1) const {
        f32x4 x = *reinterpret_cast<const f32x4*>(X + off);
        s0 += x;
        s1 += x * x;
    }
};
struct BnBwdOp {   // (sum dy, sum dy * xhat)
    const float* dY; const float* X; const float* mean; const float* invstd;
    __device__ void operator()(size_t off, int c, f32x4& s0, f32x4& s1) const {
        f32x4 dy = *reinterpret_cast<const f32x4*>(dY + off);
        f32x4 x = *reinterpret_cast<const f32x4*>(X + off);
        f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        s0 += dy;
        s1 += dy * ((x - mu) * is);
    }
};

template <class Op>
__global__ __launch_bounds__(256) void colreduce_kernel(Op op, float* __restrict__ P, int64_t M, int C, int64_t rows_per_block) {
    __shared__ __attribute__((aligned(16))) float red[256 * 8];
    const int t = threadIdx.x;
    const int lanes = C >> 2;
    const int rpp = 256 / lanes;
    const int rl = t / lanes, cl = t - rl * lanes;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
    const int64_t rend = min(M, rbeg + rows_per_block);
    if (rl < rpp)
        for (int64_t r = rbeg + rl; r < rend; r += rpp) op((size_t)r * C + 4 * cl, 4 * cl, s0, s1);
    *reinterpret_cast<f32x4*>(&red[t * 8]) = s0;
    *reinterpret_cast<f32x4*>(&red[t * 8 + 4]) = s1;
    __syncthreads();
    if (rl == 0 && cl < lanes) {
        for (int r = 1; r < rpp; ++r) {
            s0 += *reinterpret_cast<const f32x4*>(&red[(r * lanes + cl) * 8]);
            s1 += *reinterpret_cast<const f32x4*>(&red[(r * lanes + cl) * 8 + 4]);
        }
        float* Pb = P + (size_t)blockIdx.x * 2 * C;
        *reinterpret_cast<f32x4*>(Pb + 4 * cl) = s0;
        *reinterpret_cast<f32x4*>(Pb + C + 4 * cl) = s1;
    }
}

static void check_colreduce(int64_t M, int C) {
    CMOOP_REQUIRE(C % 4 == 0 && C / 4 <= 256 && C >= 4, "colreduce: C must be a multiple of 4, <= 1024");
    (void)M;
}

void launch_colstats(const float* X, float* P, int64_t M, int C, int blocks, hipStream_t s) {
    check_colreduce(M, C);
    StatsOp op{X};
    hipLaunchKernelGGL((colreduce_kernel<StatsOp>), dim3(blocks), dim3(256), 0, s, op, P, M, C, cdiv64(M, blocks));
    CMOOP_HIP(hipGetLastError());
}

void launch_bn_bwd_reduce(const float* dY, const float* X, const float* mean, const float* invstd, float* P,
                          int64_t M, int C, int blocks, hipStream_t s) {
    check_colreduce(M, C);
    BnBwdOp op{dY, X, mean, invstd};
    hipLaunchKernelGGL((colreduce_kernel<BnBwdOp>), dim3(blocks), dim3(256), 0, s, op, P, M, C, cdiv64(M, blocks));
    CMOOP_HIP(hipGetLastError());
}

// Per-channel totals of the [blocks][2][C] partials a reduction (or a conv epilogue) left: PS_CH channels per workgroup,
// 64 lanes per channel, lane l sums partials l, l+64, ... into four independent double accumulators (eight loads in
// flight: the 4 040 partials of a 101x40 layer are 16 rounds, not 126), then lane 0 adds the 64 lane sums in lane order.
constexpr int PS_CH = 4;
__device__ __forceinline__ bool partial_sums(const float* __restrict__ P, int blocks, int C, double* s1, double* s2,
                                             int* c_out) {
    __shared__ double sh[2][PS_CH][64];
    const int cl = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * PS_CH + cl;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        int blk = lane;
        for (; blk + 192 < blocks; blk += 256) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] += (double)P[(size_t)(blk + 64 * u) * 2 * C + c];
                b[u] += (double)P[(size_t)(blk + 64 * u) * 2 * C + C + c];
            }
        }
        for (; blk < blocks; blk += 64) {
            a[0] += (double)P[(size_t)blk * 2 * C + c];
            b[0] += (double)P[(size_t)blk * 2 * C + C + c];
        }
    }
    sh[0][cl][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    sh[1][cl][lane] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    *c_out = c;
    if (lane != 0 || c >= C) return false;
    double x = 0.0, y = 0.0;
    for (int l = 0; l < 64; ++l) { x += sh[0][cl][l]; y += sh[1][cl][l]; }
    *s1 = x; *s2 = y;
    return true;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ P, int blocks, int64_t M, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ mm, float* __restrict__ mv, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
                                   float eps, float momentum, float one_minus_momentum) {
    double s1, s2;
    int c;
    if (!partial_sums(P, blocks, C, &s1, &s2, &c)) return;
    const double mu = s1 / (double)M;
    double var = s2 / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    const float muf = (float)mu, varf = (float)var;
    const float is = (float)(1.0 / sqrt((double)varf + (double)eps));
    const float sc = gamma[c] * is;
    mean[c] = muf;
    invstd[c] = is;
    scale[c] = sc;
    shift[c] = beta[c] - muf * sc;
    mm[c] = mm[c] * momentum + muf * one_minus_momentum;
    mv[c] = mv[c] * momentum + varf * one_minus_momentum;
}

void launch_bn_finalize(const float* P, int blocks, int64_t M, int C, const float* gamma, const float* beta,
                        float* moving_mean, float* moving_var, float* mean, float* invstd, float* scale, float* shift,
                        float eps, float momentum, float omm, hipStream_t s) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, PS_CH)), dim3(256), 0, s, P, blocks, M, C, gamma, beta, moving_mean,
                       moving_var, mean, invstd, scale, shift, eps, momentum, omm);
    CMOOP_HIP(hipGetLastError());
}

__global__ void bn_eval_prepare_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ mm, const float* __restrict__ mv,
                                       float* __restrict__ scale, float* __restrict__ shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = (float)(1.0 / sqrt((double)mv[c] + (double)eps));
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - mm[c] * sc;
}

void launch_bn_eval_prepare(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                            float* scale, float* shift, int C, float eps, hipStream_t s) {
    hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, gamma, beta, moving_mean, moving_var,
                       scale, shift, C, eps);
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void scale_shift_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int64_t n4, int C, int relu) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        f32x4 x = *reinterpret_cast<const f32x4*>(X + i * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = x[j] * sc[j] + sh[j];
            y[j] = relu ? fmaxf(v, 0.f) : v;
        }
        *reinterpret_cast<f32x4*>(Y + i * 4) = y;
    }
}

static inline unsigned ew_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, 256), 8192)); }

void launch_scale_shift(const float* X, float* Y, const float* scale, const float* shift, int64_t M, int C, int relu,
                        hipStream_t s) {
    CMOOP_REQUIRE(C % 4 == 0, "scale_shift: C % 4");
    const int64_t n4 = M * C / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(scale_shift_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, X, Y, scale, shift, n4, C, relu);
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ P, int blocks, int C,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
    double s1, s2;
    int c;
    if (!partial_sums(P, blocks, C, &s1, &s2, &c)) return;
    sums[c] = (float)s1;
    sums[C + c] = (float)s2;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, float* __restrict__ dX,
                                                           int64_t n4, int C, float invM, int mask_x_pos) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const f32x4 dy = *reinterpret_cast<const f32x4*>(dY + i * 4);
        const f32x4 x = *reinterpret_cast<const f32x4*>(X + i * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + c);
        const f32x4 s2 = *reinterpret_cast<const f32x4*>(sums + C + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (x[j] - mu[j]) * is[j];
            float v = ga[j] * is[j] * (dy[j] - s1[j] * invM - xh * (s2[j] * invM));
            if (mask_x_pos && !(x[j] > 0.f)) v = 0.f;
            o[j] = v;
        }
        *reinterpret_cast<f32x4*>(dX + i * 4) = o;
    }
}

void launch_bn_bwd_apply(const float* dY, const float* X, const float* mean, const float* invstd, const float* gamma,
                         const float* P, int blocks, float* dX, float* dgamma, float* dbeta, int64_t M, int C,
                         int mask_x_pos, hipStream_t s) {
    // sums live right after the partials (caller reserves 2*C floats there)
    float* sums = const_cast<float*>(P) + (size_t)blocks * 2 * C;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, PS_CH)), dim3(256), 0, s, P, blocks, C, sums, dgamma, dbeta);
    CMOOP_HIP(hipGetLastError());
    const int64_t n4 = M * C / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, dY, X, mean, invstd, gamma, sums, dX, n4,
                       C, (float)(1.0 / (double)M), mask_x_pos);
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ P, int blocks, int C,
                                                              float* __restrict__ out) {
    double s1, s2;
    int c;
    if (!partial_sums(P, blocks, C, &s1, &s2, &c)) return;
    out[c] = (float)s1;
}

void launch_colsum_finalize(const float* P, int blocks, int C, float* out, hipStream_t s) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(C, PS_CH)), dim3(256), 0, s, P, blocks, C, out);
    CMOOP_HIP(hipGetLastError());
}

// tiny helpers for the output layer (C_out = classes is not a multiple of 4 / power of two)
__global__ void colsum_small_kernel(const float* __restrict__ X, float* __restrict__ out, int M, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += X[(size_t)m * C + c];
    out[c] = s;
}

void launch_colsum_small(const float* X, float* out, int M, int C, hipStream_t s) {
    hipLaunchKernelGGL(colsum_small_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, X, out, M, C);
    CMOOP_HIP(hipGetLastError());
}

// dX[m][i] = sum_o dY[m][o] * W[o][i], optionally masked by (mask[m][i] > 0) * scale
__global__ __launch_bounds__(256) void dense_dgrad_small_kernel(const float* __restrict__ dY, const float* __restrict__ W,
                                                                float* __restrict__ dX, int M, int N, int K,
                                                                const float* __restrict__ mask, float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M * K) return;
    const int m = i / K, k = i - m * K;
    float s = 0.f;
    for (int o = 0; o < N; ++o) s = fmaf(dY[(size_t)m * N + o], W[(size_t)o * K + k], s);
    if (mask) s = mask[i] > 0.f ? s * scale : 0.f;
    dX[i] = s;
}

void launch_dense_dgrad_small(const float* dY, const float* W, float* dX, int M, int N, int K, const float* mask,
                              float scale, hipStream_t s) {
    if (M * K == 0) return;
    hipLaunchKernelGGL(dense_dgrad_small_kernel, dim3(cdiv(M * K, 256)), dim3(256), 0, s, dY, W, dX, M, N, K, mask, scale);
    CMOOP_HIP(hipGetLastError());
}

// ===========================================================================
// MaxPooling2D((2,2), strides 2, padding='same'): out = ceil(n/2), the window is
// clipped at the bottom/right edge (TF pads with -inf).  First max wins ties.
// ===========================================================================
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                          uint8_t* __restrict__ arg, int B, int H, int W, int C, int OH,
                                                          int OW) {
    const int C4 = C >> 2;
    const int64_t n = (int64_t)B * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        f32x4 best;
        uchar4 a = {0, 0, 0, 0};
        bool first = true;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const int ih = 2 * oh + (pos >> 1), iw = 2 * ow + (pos & 1);
            if (ih >= H || iw >= W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(X + ((size_t)(b * H + ih) * W + iw) * C + 4 * c4);
            if (first) { best = v; first = false; continue; }
            if (v[0] > best[0]) { best[0] = v[0]; a.x = pos; }
            if (v[1] > best[1]) { best[1] = v[1]; a.y = pos; }
            if (v[2] > best[2]) { best[2] = v[2]; a.z = pos; }
            if (v[3] > best[3]) { best[3] = v[3]; a.w = pos; }
        }
        *reinterpret_cast<f32x4*>(Y + i * 4) = best;
        *reinterpret_cast<uchar4*>(arg + i * 4) = a;
    }
}

void launch_maxpool_fwd(const float* X, float* Y, uint8_t* arg, int B, int H, int W, int C, hipStream_t s) {
    CMOOP_REQUIRE(C % 4 == 0, "maxpool: C % 4");
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t n = (int64_t)B * OH * OW * (C / 4);
    if (n == 0) return;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, s, X, Y, arg, B, H, W, C, OH, OW);
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dY, const uint8_t* __restrict__ arg,
                                                          const float* __restrict__ Y, float* __restrict__ dX, int B,
                                                          int H, int W, int C, int OH, int OW, int mask_y_pos) {
    const int C4 = C >> 2;
    const int64_t n = (int64_t)B * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        f32x4 g = *reinterpret_cast<const f32x4*>(dY + i * 4);
        if (mask_y_pos) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(Y + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!(y[j] > 0.f)) g[j] = 0.f;
        }
        const uchar4 a = *reinterpret_cast<const uchar4*>(arg + i * 4);
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const int ih = 2 * oh + (pos >> 1), iw = 2 * ow + (pos & 1);
            if (ih >= H || iw >= W) continue;
            f32x4 o;
            o[0] = a.x == pos ? g[0] : 0.f; o[1] = a.y == pos ? g[1] : 0.f;
            o[2] = a.z == pos ? g[2] : 0.f; o[3] = a.w == pos ? g[3] : 0.f;
            *reinterpret_cast<f32x4*>(dX + ((size_t)(b * H + ih) * W + iw) * C + 4 * c4) = o;
        }
    }
}

void launch_maxpool_bwd(const float* dY, const uint8_t* arg, const float* Y, float* dX, int B, int H, int W, int C,
                        int mask_y_pos, hipStream_t s) {
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t n = (int64_t)B * OH * OW * (C / 4);
    if (n == 0) return;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, s, dY, arg, Y, dX, B, H, W, C, OH, OW,
                       mask_y_pos);
    CMOOP_HIP(hipGetLastError());
}

// ===========================================================================
// BatchNorm-apply (+ReLU) + MaxPool 2x2 SAME in one pass, and its backward twins: the normalised full-resolution
// tensor and its gradient are never written.  Arithmetic and tie rules are those of scale_shift_kernel followed by
// maxpool_fwd_kernel (bit-identical results); backward: dY_full[b,ih,iw,c] = (arg[b,oh,ow,c] == pos) ? g[b,oh,ow,c] : 0
// is formed on the fly where bn_bwd_reduce / bn_bwd_apply would have read the materialised tensor.
// ===========================================================================
__global__ __launch_bounds__(256) void bn_pool_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                          uint8_t* __restrict__ arg, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int B, int H, int W, int C,
                                                          int OH, int OW, int relu) {
    const int C4 = C >> 2;
    const int64_t n = (int64_t)B * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int ow = (int)(p % OW); p /= OW;
        const int oh = (int)(p % OH);
        const int b = (int)(p / OH);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * c4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + 4 * c4);
        f32x4 best;
        uchar4 a = {0, 0, 0, 0};
        bool first = true;
#pragma unroll
        for (int pos = 0; pos < 4; ++pos) {
            const int ih = 2 * oh + (pos >> 1), iw = 2 * ow + (pos & 1);
            if (ih >= H || iw >= W) continue;
            const f32x4 x = *reinterpret_cast<const f32x4*>(X + ((size_t)(b * H + ih) * W + iw) * C + 4 * c4);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = x[j] * sc[j] + sh[j];
                v[j] = relu ? fmaxf(t, 0.f) : t;
            }
            if (first) { best = v; first = false; continue; }
            if (v[0] > best[0]) { best[0] = v[0]; a.x = pos; }
            if (v[1] > best[1]) { best[1] = v[1]; a.y = pos; }
            if (v[2] > best[2]) { best[2] = v[2]; a.z = pos; }
            if (v[3] > best[3]) { best[3] = v[3]; a.w = pos; }
        }
        *reinterpret_cast<f32x4*>(Y + i * 4) = best;
        *reinterpret_cast<uchar4*>(arg + i * 4) = a;
    }
}

void launch_bn_pool_fwd(const float* X, float* Y, uint8_t* arg, const float* scale, const float* shift, int B, int H, int W,
                        int C, int relu, hipStream_t s) {
    CMOOP_REQUIRE(C % 4 == 0, "bn_pool: C % 4");
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    const int64_t n = (int64_t)B * OH * OW * (C / 4);
    if (n == 0) return;
    hipLaunchKernelGGL(bn_pool_fwd_kernel, dim3(ew_grid(n)), dim3(256), 0, s, X, Y, arg, scale, shift, B, H, W, C, OH, OW, relu);
    CMOOP_HIP(hipGetLastError());
}

// gradient of the (never materialised) BN output at full-resolution row r = (b, ih, iw), channels c..c+3
struct PooledGrad {
    const float* g; const uint8_t* arg;
    int H, W, OH, OW, C;
    __device__ __forceinline__ f32x4 at(int64_t row, int c) const {
        const int iw = (int)(row % W);
        const int64_t t = row / W;
        const int ih = (int)(t % H), b = (int)(t / H);
        const size_t o = (((size_t)b * OH + (ih >> 1)) * OW + (iw >> 1)) * C + c;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + o);
        const uchar4 a = *reinterpret_cast<const uchar4*>(arg + o);
        const int pos = ((ih & 1) << 1) | (iw & 1);
        return f32x4{a.x == pos ? gv[0] : 0.f, a.y == pos ? gv[1] : 0.f, a.z == pos ? gv[2] : 0.f, a.w == pos ? gv[3] : 0.f};
    }
};

struct BnBwdPooledOp {   // (sum dy, sum dy * xhat) with dy scattered from the pooled gradient
    PooledGrad pg; const float* X; const float* mean; const float* invstd;
    __device__ void operator()(size_t off, int c, f32x4& s0, f32x4& s1) const {
        const f32x4 dy = pg.at((int64_t)(off / pg.C), c);
        const f32x4 x = *reinterpret_cast<const f32x4*>(X + off);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        s0 += dy;
        s1 += dy * ((x - mu) * is);
    }
};

void launch_bn_pool_bwd_reduce(const float* g_pooled, const uint8_t* arg, const float* X, const float* mean, const float* invstd,
                               float* P, int B, int H, int W, int C, int blocks, hipStream_t s) {
    const int64_t M = (int64_t)B * H * W;
    check_colreduce(M, C);
    BnBwdPooledOp op{PooledGrad{g_pooled, arg, H, W, (H + 1) / 2, (W + 1) / 2, C}, X, mean, invstd};
    hipLaunchKernelGGL((colreduce_kernel<BnBwdPooledOp>), dim3(blocks), dim3(256), 0, s, op, P, M, C, cdiv64(M, blocks));
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(PooledGrad pg, const float* __restrict__ X,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ sums, float* __restrict__ dX,
                                                                int64_t n4, int C, float invM, int mask_x_pos) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const f32x4 dy = pg.at((i * 4) / C, c);
        const f32x4 x = *reinterpret_cast<const f32x4*>(X + i * 4);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 s1 = *reinterpret_cast<const f32x4*>(sums + c);
        const f32x4 s2 = *reinterpret_cast<const f32x4*>(sums + C + c);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (x[j] - mu[j]) * is[j];
            float v = ga[j] * is[j] * (dy[j] - s1[j] * invM - xh * (s2[j] * invM));
            if (mask_x_pos && !(x[j] > 0.f)) v = 0.f;
            o[j] = v;
        }
        *reinterpret_cast<f32x4*>(dX + i * 4) = o;
    }
}

void launch_bn_pool_bwd_apply(const float* g_pooled, const uint8_t* arg, const float* X, const float* mean, const float* invstd,
                              const float* gamma, const float* P, int blocks, float* dX, float* dgamma, float* dbeta, int B,
                              int H, int W, int C, int mask_x_pos, hipStream_t s) {
    const int64_t M = (int64_t)B * H * W;
    float* sums = const_cast<float*>(P) + (size_t)blocks * 2 * C;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, PS_CH)), dim3(256), 0, s, P, blocks, C, sums, dgamma, dbeta);
    CMOOP_HIP(hipGetLastError());
    const int64_t n4 = M * C / 4;
    if (n4 == 0) return;
    PooledGrad pg{g_pooled, arg, H, W, (H + 1) / 2, (W + 1) / 2, C};
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, pg, X, mean, invstd, gamma, sums, dX, n4, C,
                       (float)(1.0 / (double)M), mask_x_pos);
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void add_relu_kernel(const float* __restrict__ A, const float* __restrict__ Bt,
                                                       float* __restrict__ Y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(A + i * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(Bt + i * 4);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = fmaxf(a[j] + b[j], 0.f);
        *reinterpret_cast<f32x4*>(Y + i * 4) = y;
    }
}

void launch_add_relu(const float* A, const float* Bt, float* Y, int64_t n, hipStream_t s) {
    CMOOP_REQUIRE(n % 4 == 0, "add_relu: n % 4");
    if (n == 0) return;
    hipLaunchKernelGGL(add_relu_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, s, A, Bt, Y, n / 4);
    CMOOP_HIP(hipGetLastError());
}

// GlobalAveragePooling2D: one workgroup per sample; lanes = channel quads, the remaining threads split the pixels;
// float4 loads, fixed-order LDS reduction over the pixel slices (the old one-thread-per-(b,c) loop took 69 us on a
// 26x10x32 tensor -- 12 % of a small candidate's step)
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y, int HW, int C,
                                                      float inv) {
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int t = threadIdx.x, lanes = C >> 2;
    const int slices = 256 / lanes;
    const int sl = t / lanes, cl = t - sl * lanes;
    const float* x = X + (size_t)blockIdx.x * HW * C + 4 * cl;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (sl < slices)
        for (int p = sl; p < HW; p += slices) s += *reinterpret_cast<const f32x4*>(x + (size_t)p * C);
    *reinterpret_cast<f32x4*>(&red[t * 4]) = s;
    __syncthreads();
    if (sl == 0) {
        for (int r = 1; r < slices; ++r) s += *reinterpret_cast<const f32x4*>(&red[(r * lanes + cl) * 4]);
        *reinterpret_cast<f32x4*>(Y + (size_t)blockIdx.x * C + 4 * cl) = s * inv;
    }
}

void launch_gap_fwd(const float* X, float* Y, int B, int HW, int C, hipStream_t s) {
    if (B == 0) return;
    CMOOP_REQUIRE(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0, "gap: C must be 4 x a divisor of 256");
    hipLaunchKernelGGL(gap_fwd_kernel, dim3(B), dim3(256), 0, s, X, Y, HW, C, (float)(1.0 / HW));
    CMOOP_HIP(hipGetLastError());
}

// seeded glorot-uniform initial weights / constant fill (Net::build_plan)
__global__ __launch_bounds__(256) void glorot_init_kernel(float* __restrict__ w, int64_t n, uint32_t prefix, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int u24 = (int)(fmix32(prefix ^ (uint32_t)i) >> 8);
        w[i] = (float)(2 * u24 - 16777216) * scale;
    }
}
void launch_glorot_init(float* w, int64_t n, uint32_t prefix, float scale, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(glorot_init_kernel, dim3(ew_grid(n)), dim3(256), 0, s, w, n, prefix, scale);
    CMOOP_HIP(hipGetLastError());
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ w, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) w[i] = v;
}
void launch_fill(float* w, float v, int64_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(256), 0, s, w, v, n);
    CMOOP_HIP(hipGetLastError());
}

__global__ void step_advance_kernel(StepState* st, int batch) {
    st->row0 += batch;
    st->step += 1;
    st->iter += 1;
}
void launch_step_advance(StepState* st, int batch, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, st, batch);
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                      float* __restrict__ dX, int64_t n4, int HW, int C, float inv) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i * 4;
        const int c = (int)(e % C);
        const int64_t b = e / ((int64_t)HW * C);
        const f32x4 g = *reinterpret_cast<const f32x4*>(dY + b * C + c);
        const f32x4 x = *reinterpret_cast<const f32x4*>(X + e);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = x[j] > 0.f ? g[j] * inv : 0.f;
        *reinterpret_cast<f32x4*>(dX + e) = o;
    }
}

void launch_gap_bwd(const float* dY, const float* X, float* dX, int B, int HW, int C, hipStream_t s) {
    const int64_t n4 = (int64_t)B * HW * C / 4;
    if (n4 == 0) return;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(ew_grid(n4)), dim3(256), 0, s, dY, X, dX, n4, HW, C, (float)(1.0 / HW));
    CMOOP_HIP(hipGetLastError());
}

// ===========================================================================
// softmax + Keras-3 sparse_categorical_crossentropy(from_logits=False):
//   p = softmax(z); pc = clip(p, 1e-7, 1-1e-7); loss = -(log pc_y - log sum_j pc_j)
// (nsga_penalty.py:377-379 via the TF backend).  One block; rows strided over
// lanes; wavefront (shuffle) reductions for the loss / correct counters.
// ===========================================================================
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ Z, const int32_t* __restrict__ labels,
                                                         const int32_t* __restrict__ idx, int64_t row0, int B, int C,
                                                         float* __restrict__ dZ, double* __restrict__ acc,
                                                         int32_t* __restrict__ preds, const StepState* __restrict__ st,
                                                         int64_t n_rows) {
    __shared__ double lsum[4];
    if (st) row0 = st->row0;
    __shared__ int csum[4];
    const int t = threadIdx.x;
    const float lo = 1e-7f, hi = 1.0f - 1e-7f;
    double myloss = 0.0;
    int mycorrect = 0;
    for (int r = t; r < B; r += 256) {
        const float* z = Z + (size_t)r * C;
        const int y = labels[gather_row(idx, row0 + r, n_rows)];
        float mx = z[0];
        int am = 0;
        for (int j = 1; j < C; ++j)
            if (z[j] > mx) { mx = z[j]; am = j; }
        float se = 0.f;
        for (int j = 0; j < C; ++j) se += expf(z[j] - mx);
        float S = 0.f, py = 0.f, pyc = 1.f;
        for (int j = 0; j < C; ++j) {
            const float p = expf(z[j] - mx) / se;
            const float pc = fminf(fmaxf(p, lo), hi);
            S += pc;
            if (j == y) { py = p; pyc = pc; }
        }
        (void)py;
        myloss += (double)(-(logf(pyc) - logf(S)));
        mycorrect += (am == y);
        if (preds) preds[r] = am;
        if (dZ) {
            // q_j = gate_j * (1/S - [j==y]/pc_y); dz_i = p_i * (q_i - sum_j p_j q_j), then / B
            float dot = 0.f;
            for (int j = 0; j < C; ++j) {
                const float p = expf(z[j] - mx) / se;
                const float gate = (p >= lo && p <= hi) ? 1.f : 0.f;
                const float qj = gate * (1.f / S - (j == y ? 1.f / pyc : 0.f));
                dot += p * qj;
            }
            const float invB = 1.f / (float)B;
            for (int j = 0; j < C; ++j) {
                const float p = expf(z[j] - mx) / se;
                const float gate = (p >= lo && p <= hi) ? 1.f : 0.f;
                const float qj = gate * (1.f / S - (j == y ? 1.f / pyc : 0.f));
                dZ[(size_t)r * C + j] = p * (qj - dot) * invB;
            }
        }
    }
    // wavefront reduction (fixed butterfly order: deterministic), then the four wave sums in wave order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        myloss += __shfl_xor(myloss, off, 64);
        mycorrect += __shfl_xor(mycorrect, off, 64);
    }
    if ((t & 63) == 0) { lsum[t >> 6] = myloss; csum[t >> 6] = mycorrect; }
    __syncthreads();
    if (t == 0 && acc) {
        acc[0] += ((lsum[0] + lsum[1]) + lsum[2]) + lsum[3];
        reinterpret_cast<long long*>(acc)[1] += (csum[0] + csum[1]) + (csum[2] + csum[3]);
    }
}

void launch_softmax_ce(const float* Z, const int32_t* labels, const int32_t* idx, int64_t row0, int B, int C, float* dZ,
                       double* acc, int32_t* preds, hipStream_t s, const StepState* st, int64_t n_rows) {
    if (B == 0) return;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(1), dim3(256), 0, s, Z, labels, idx, row0, B, C, dZ, acc, preds, st, n_rows);
    CMOOP_HIP(hipGetLastError());
}

// Every operation of the update is a separately rounded IEEE single operation, in the order the reference's CPU path and
// the oracle apply them (oracle/net.py train_step): no fused multiply-add.  Left to the compiler, the contraction of
// m + (g - m) * c1 differed BETWEEN TWO KERNELS of this file (fused in one, mul + add in the other), an ulp apart --
// enough to change the predictions of a BatchNorm + dropout candidate 100 steps later.
__device__ __forceinline__ void adam_update(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v, int64_t i,
                                            float gi, float alpha, float c1, float c2, float eps) {
#pragma clang fp contract(off)
    float mi = m[i], vi = v[i];
    const float dm = (gi - mi) * c1;
    mi = mi + dm;
    const float gg = gi * gi;
    const float dv = (gg - vi) * c2;
    vi = vi + dv;
    m[i] = mi;
    v[i] = vi;
    const float num = mi * alpha;
    const float den = sqrtf(vi) + eps;
    w[i] = w[i] - num / den;
}

// Keras-form Adam: m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= m*alpha/(sqrt(v)+eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float alpha, float c1, float c2,
                                                   float eps, const StepState* __restrict__ st,
                                                   const float* __restrict__ alpha_table) {
    if (st) alpha = alpha_table[st->iter];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        adam_update(w, m, v, i, g[i], alpha, c1, c2, eps);
}

void launch_adam(float* w, const float* g, float* m, float* v, int64_t n, float alpha, float c1, float c2, float eps,
                 hipStream_t s, const StepState* st, const float* alpha_table) {
    if (n == 0) return;
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, s, w, g, m, v, n, alpha, c1, c2, eps, st, alpha_table);
    CMOOP_HIP(hipGetLastError());
}

// slab segment: 64 column groups of VEC floats x 4 slice lanes per workgroup; lane sl sums slices sl, sl+4, ... with four
// independent accumulators of eight (the order of reduce_slices_kernel, gemm.hip), lane 0 combines, stores g and updates w/m/v
template <int VEC>
__device__ __forceinline__ void adam_slab_block(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m,
                                                float* __restrict__ v, const AdamSeg sg, int b, float (*red)[256],
                                                float alpha, float c1, float c2, float eps) {
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = ((int64_t)b * 64 + e) * VEC;
    float acc[8][VEC];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[u][j] = 0.f;
    if (i < sg.n) {
        int s = sl;
        for (; s + 28 < sg.S; s += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* src = sg.slab + (size_t)(s + 4 * u) * sg.stride + i;
                if constexpr (VEC == 4) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(src);
                    acc[u][0] += q[0]; acc[u][1] += q[1]; acc[u][2] += q[2]; acc[u][3] += q[3];
                } else {
                    acc[u][0] += src[0];
                }
            }
        }
        for (; s < sg.S; s += 4) {
            const float* src = sg.slab + (size_t)s * sg.stride + i;
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[0][j] += src[j];
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        red[sl][e * VEC + j] = ((acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j])) + ((acc[4][j] + acc[5][j]) + (acc[6][j] + acc[7][j]));
    __syncthreads();
    if (sl == 0 && i < sg.n) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float gi = ((red[0][e * VEC + j] + red[1][e * VEC + j]) + red[2][e * VEC + j]) + red[3][e * VEC + j];
            g[sg.off + i + j] = gi;
            adam_update(w, m, v, sg.off + i + j, gi, alpha, c1, c2, eps);
        }
    }
}

constexpr int ADAM_PLAIN_PER_BLOCK = 1024;

__global__ __launch_bounds__(256) void adam_segments_kernel(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, const AdamSegTable tab, float alpha, float c1,
                                                            float c2, float eps, const StepState* __restrict__ st,
                                                            const float* __restrict__ alpha_table) {
    __shared__ float red[4][256];
    if (st) alpha = alpha_table[st->iter];
    int si = 0;
    while (si + 1 < tab.count && (int)blockIdx.x >= tab.seg[si + 1].block0) ++si;
    const AdamSeg sg = tab.seg[si];
    const int b = (int)blockIdx.x - sg.block0;
    if (sg.slab == nullptr) {
        const int64_t base = (int64_t)b * ADAM_PLAIN_PER_BLOCK;
#pragma unroll
        for (int k = 0; k < ADAM_PLAIN_PER_BLOCK / 256; ++k) {
            const int64_t i = base + k * 256 + threadIdx.x;
            if (i < sg.n) adam_update(w, m, v, sg.off + i, g[sg.off + i], alpha, c1, c2, eps);
        }
        return;
    }
    const bool vec = (sg.n % 4 == 0) && (sg.stride % 4 == 0) && (sg.off % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(sg.slab) % 16 == 0);
    if (vec) adam_slab_block<4>(w, g, m, v, sg, b, red, alpha, c1, c2, eps);
    else adam_slab_block<1>(w, g, m, v, sg, b, red, alpha, c1, c2, eps);
}

void adam_segments_finalize(AdamSegTable& tab) {
    CMOOP_REQUIRE(tab.count >= 0 && tab.count <= ADAM_MAX_SEGS, "adam segment table overflow");
    int64_t blocks = 0, pos = tab.count ? tab.seg[0].off : 0;
    for (int i = 0; i < tab.count; ++i) {
        AdamSeg& sg = tab.seg[i];
        CMOOP_REQUIRE(sg.off == pos && sg.n > 0, "adam segments must tile the arena in order");
        pos += sg.n;
        sg.block0 = (int32_t)blocks;
        if (sg.slab) {
            const bool vec = (sg.n % 4 == 0) && (sg.stride % 4 == 0) && (sg.off % 4 == 0) &&
                             (reinterpret_cast<uintptr_t>(sg.slab) % 16 == 0);
            blocks += vec ? cdiv64(sg.n / 4, 64) : cdiv64(sg.n, 64);
        } else {
            blocks += cdiv64(sg.n, ADAM_PLAIN_PER_BLOCK);
        }
    }
    CMOOP_REQUIRE(blocks < (int64_t)1 << 30, "adam grid too large");
    tab.blocks = (int32_t)blocks;
}

void launch_adam_segments(float* w, float* g, float* m, float* v, const AdamSegTable& tab, float alpha, float c1, float c2,
                          float eps, hipStream_t s, const StepState* st, const float* alpha_table) {
    if (tab.blocks == 0) return;
    hipLaunchKernelGGL(adam_segments_kernel, dim3((unsigned)tab.blocks), dim3(256), 0, s, w, g, m, v, tab, alpha, c1, c2, eps, st,
                       alpha_table);
    CMOOP_HIP(hipGetLastError());
}

// Epoch shuffle on the device (Keras fit(shuffle=True), nsga_penalty.py:383; seeded here): the permutation that sorts
// the keys (fmix32(prefix ^ i) << 32 | i), i.e. exactly what the host twin (net.hip epoch_permutation, oracle/rng.py)
// produces with std::sort -- computed as a rank sort: out[#{j : key_j < key_i}] = i.  O(n^2) compares on LDS tiles
// (0.6 G for n = 24 000: tens of microseconds chip-wide), no host sort, no H2D copy, no stream stall.
__global__ __launch_bounds__(256) void epoch_permutation_kernel(uint32_t prefix, int n, int32_t* __restrict__ out) {
    __shared__ uint32_t hs[1024];
    const int t = threadIdx.x;
    const int i = blockIdx.x * 256 + t;
    const uint32_t hi = fmix32(prefix ^ (uint32_t)i);
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) hs[t + 256 * u] = fmix32(prefix ^ (uint32_t)(j0 + t + 256 * u));
        __syncthreads();
        const int lim = min(1024, n - j0);
        for (int j = 0; j < lim; ++j) {
            const uint32_t hj = hs[j];
            rank += (hj < hi || (hj == hi && j0 + j < i)) ? 1 : 0;
        }
        __syncthreads();
    }
    if (i < n) out[rank] = i;
}

void launch_epoch_permutation(uint32_t seed, uint32_t epoch, int64_t n, int32_t* out, hipStream_t s) {
    CMOOP_REQUIRE(n >= 0 && n <= EPOCH_PERMUTATION_DEVICE_MAX, "device epoch permutation: n too large (use the host twin)");
    if (n == 0) return;
    hipLaunchKernelGGL(epoch_permutation_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s,
                       rng_prefix(seed, STREAM_SHUFFLE, epoch), (int)n, out);
    CMOOP_HIP(hipGetLastError());
}

// confusion_matrix(y_true, y_pred, labels=range(C)) (nsga_penalty.py:355): LDS histogram
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t* __restrict__ yt, const int32_t* __restrict__ yp,
                                                        int64_t n, int C, int force_true_zero, long long* __restrict__ cm) {
    extern __shared__ int hist[];
    for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const int a = force_true_zero ? 0 : yt[i];
        const int b = yp[i];
        if ((unsigned)a < (unsigned)C && (unsigned)b < (unsigned)C) atomicAdd(&hist[a * C + b], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256) cm[i] = hist[i];
}

void launch_confusion(const int32_t* y_true, const int32_t* y_pred, int64_t n, int C, int force_true_zero, int64_t* cm,
                      hipStream_t s) {
    CMOOP_REQUIRE(C * C * 4 <= 64 * 1024, "confusion: too many classes");
    hipLaunchKernelGGL(confusion_kernel, dim3(1), dim3(256), C * C * sizeof(int), s, y_true, y_pred, n, C,
                       force_true_zero, reinterpret_cast<long long*>(cm));
    CMOOP_HIP(hipGetLastError());
}

}  // namespace cmoop
