// Dense layers of the MLP head (GlobalAveragePooling -> Dense+ReLU ladder [+Dropout] -> Dense(CLASSES),
// /root/reference/nsga_penalty.py:306-330): forward, input gradient and weight gradient.
//
// M = batch rows (<= a few hundred), K = C_in in {32..512}, N = units in {512,256,128,64,classes}: far too small
// for the tiled implicit-GEMM kernels (gemm.hip), which needed a split-K slab + combine pass forward, and a
// flip-transposed weight copy + split-K dgrad + sliced wgrad + slice reduction backward: 7 launches per layer per
// step at ~1 % of the MFMA peak.  Here each layer is 3 launches (forward; dgrad; wgrad with the bias gradient):
// one 256-thread workgroup owns ONE 16x16 output tile, its 4 waves split the reduction axis, every operand
// fragment is read straight from global memory in the MFMA's own lane layout (no LDS staging, no transposed
// copies), and the four partial tiles are summed through LDS in a fixed order (deterministic).
//
// v_mfma_f32_16x16x4_f32 operand layout: lane (lr = lane & 15, q = lane >> 4) supplies A[row lr][k q] and
// B[k q][col lr]; it receives D[row 4q + r][col lr] in register r.
#include "kernels.h"
#include <mutex>

namespace cmoop {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// GEMM_BF16 ("bf16 train"): operands rounded to bf16 (round-to-nearest-even); products of bf16 values are exact in fp32,
// so feeding the rounded values to the fp32 MFMA IS the mode's definition (fp32 accumulation).
__device__ __forceinline__ float rbf(float x, bool on) {
    if (!on) return x;
    const bf16x2v h = __builtin_convertvector((f32x2v){x, 0.f}, bf16x2v);
    return (float)h[0];
}
__device__ __forceinline__ f32x4 rbf4(f32x4 v, bool on) {
    if (!on) return v;
    return f32x4{rbf(v[0], true), rbf(v[1], true), rbf(v[2], true), rbf(v[3], true)};
}

// sum the four waves' partial tiles in wave order; result valid in wave 0
__device__ __forceinline__ f32x4 reduce_waves(f32x4 acc, float* red /* [3][256] */, int wave, int lane) {
    if (wave > 0) *reinterpret_cast<f32x4*>(&red[((wave - 1) * 64 + lane) * 4]) = acc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) acc += *reinterpret_cast<const f32x4*>(&red[(w * 64 + lane) * 4]);
    }
    return acc;
}

struct DenseEpi {
    const float* bias;
    int relu, dropout;
    uint32_t drop_prefix, drop_thr;
    float drop_scale;
    const StepState* st;              // non-null: drop_prefix = rng_prefix(drop_seed, drop_stream, st->step)
    uint32_t drop_seed, drop_stream;
};

// Y[m][n] = act(sum_k X[m][k] * W[n][k] + bias[n])
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                        float* __restrict__ Y, int M, int N, int K, DenseEpi e, int bf16,
                                                        const float* __restrict__ zeros) {
    __shared__ __attribute__((aligned(16))) float red[3 * 256];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int n_tiles = (N + 15) >> 4;
    const int m0 = (blockIdx.x / n_tiles) * 16, n0 = (blockIdx.x % n_tiles) * 16;
    const bool aok = m0 + lr < M, bok = n0 + lr < N;
    const float* xa = aok ? X + (size_t)(m0 + lr) * K + 4 * q : zeros;
    const float* wb = bok ? W + (size_t)(n0 + lr) * K + 4 * q : zeros;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool rb = bf16 != 0;
    // 16-k groups kk = wave, wave + 4, ...: four waves stream disjoint quarters of K; four groups' loads are issued
    // before their MFMAs so one global round trip covers 64 k's per wave
    const int KK = K >> 4;
    for (int base = wave; base < KK; base += 16) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = base + 4 * u;
            const bool ok = kk < KK;
            a[u] = *reinterpret_cast<const f32x4*>((ok && aok) ? xa + (size_t)kk * 16 : zeros);
            b[u] = *reinterpret_cast<const f32x4*>((ok && bok) ? wb + (size_t)kk * 16 : zeros);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 av = rbf4(a[u], rb), bv4 = rbf4(b[u], rb);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv4[j], acc, 0, 0, 0);
        }
    }
    acc = reduce_waves(acc, red, wave, lane);
    if (wave != 0) return;
    const int col = n0 + lr;
    if (col >= N) return;
    const float bv = e.bias ? e.bias[col] : 0.f;
    const uint32_t prefix = (e.dropout && e.st) ? rng_prefix(e.drop_seed, e.drop_stream, e.st->step) : e.drop_prefix;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + 4 * q + r;
        if (row >= M) continue;
        float v = acc[r] + bv;
        if (e.relu) v = fmaxf(v, 0.f);
        if (e.dropout) {
            const uint32_t u24 = fmix32(prefix ^ (uint32_t)((size_t)row * N + col)) >> 8;
            v = (u24 >= e.drop_thr) ? v * e.drop_scale : 0.f;
        }
        Y[(size_t)row * N + col] = v;
    }
}

// dX[m][k] = sum_n dY[m][n] * W[n][k], optionally masked: mask[m][k] > 0 ? v * scale : 0
__device__ __forceinline__ void dense_dgrad_body(const int bid, float* red, const float* __restrict__ dY, const float* __restrict__ W,
                                                 float* __restrict__ dX, int M, int N, int K,
                                                 const float* __restrict__ mask, float scale, int bf16) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int k_tiles = K >> 4;
    const int m0 = (bid / k_tiles) * 16, k0 = (bid % k_tiles) * 16;
    const bool aok = m0 + lr < M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool rb = bf16 != 0;
    const int n_groups = (N + 15) >> 4;
    for (int g = wave; g < n_groups; g += 4) {
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = g * 16 + 4 * q + j;
            const bool ok = n < N;
            a[j] = (ok && aok) ? dY[(size_t)(m0 + lr) * N + n] : 0.f;
            b[j] = ok ? W[(size_t)n * K + k0 + lr] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rbf(a[j], rb), rbf(b[j], rb), acc, 0, 0, 0);
    }
    acc = reduce_waves(acc, red, wave, lane);
    if (wave != 0) return;
    const int col = k0 + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + 4 * q + r;
        if (row >= M) continue;
        float v = acc[r];
        const size_t off = (size_t)row * K + col;
        if (mask) v = mask[off] > 0.f ? v * scale : 0.f;
        dX[off] = v;
    }
}

// dW[n][k] = sum_m dY[m][n] * X[m][k];  dB[n] = sum_m dY[m][n] (by the k-tile-0 workgroups)
__global__ __launch_bounds__(256) void dense_dgrad_kernel(const float* __restrict__ dY, const float* __restrict__ W,
                                                          float* __restrict__ dX, int M, int N, int K,
                                                          const float* __restrict__ mask, float scale, int bf16) {
    __shared__ __attribute__((aligned(16))) float red[3 * 256];
    dense_dgrad_body(blockIdx.x, red, dY, W, dX, M, N, K, mask, scale, bf16);
}

// dW[n][k] = sum_m dY[m][n] * X[m][k];  dB[n] = sum_m dY[m][n] (by the k-tile-0 workgroups)
__device__ __forceinline__ void dense_wgrad_body(const int bid, float* red, const float* __restrict__ X, const float* __restrict__ dY,
                                                 float* __restrict__ dW, float* __restrict__ dB, int M, int N, int K, int bf16) {
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int k_tiles = K >> 4;
    const int n0 = (bid / k_tiles) * 16, k0 = (bid % k_tiles) * 16;
    const bool nok = n0 + lr < N;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool rb = bf16 != 0;
    const int m_groups = (M + 15) >> 4;
    for (int g = wave; g < m_groups; g += 4) {
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = g * 16 + 4 * q + j;
            const bool ok = m < M;
            a[j] = (ok && nok) ? dY[(size_t)m * N + n0 + lr] : 0.f;
            b[j] = ok ? X[(size_t)m * K + k0 + lr] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rbf(a[j], rb), rbf(b[j], rb), acc, 0, 0, 0);
    }
    acc = reduce_waves(acc, red, wave, lane);
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 4 * q + r;
            if (n < N) dW[(size_t)n * K + k0 + lr] = acc[r];
        }
    }
    if (k0 == 0 && wave == 1 && lane < 16 && n0 + lane < N) {   // bias gradient: plain fp32 column sum in row order
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += dY[(size_t)m * N + n0 + lane];
        dB[n0 + lane] = s;
    }
}

__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                          float* __restrict__ dW, float* __restrict__ dB, int M, int N,
                                                          int K, int bf16) {
    __shared__ __attribute__((aligned(16))) float red[3 * 256];
    dense_wgrad_body(blockIdx.x, red, X, dY, dW, dB, M, N, K, bf16);
}

// weight gradient and data gradient of one dense layer in ONE launch (both read dY; the first wgrad_blocks workgroups are the
// weight-gradient tiles, the rest the data-gradient tiles): the same two bodies, bit-identical results, one dispatch fewer per layer
__global__ __launch_bounds__(256) void dense_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                        const float* __restrict__ W, float* __restrict__ dW, float* __restrict__ dB,
                                                        float* __restrict__ dX, int M, int N, int K, const float* __restrict__ mask,
                                                        float scale, int bf16, int wgrad_blocks) {
    __shared__ __attribute__((aligned(16))) float red[3 * 256];
    if ((int)blockIdx.x < wgrad_blocks) dense_wgrad_body(blockIdx.x, red, X, dY, dW, dB, M, N, K, bf16);
    else dense_dgrad_body((int)blockIdx.x - wgrad_blocks, red, dY, W, dX, M, N, K, mask, scale, bf16);
}

static const float* dense_zero_page() {
    static thread_local const float* page[16] = {nullptr};
    int dev = 0;
    CMOOP_HIP(hipGetDevice(&dev));
    CMOOP_REQUIRE(dev >= 0 && dev < 16, "device index out of range");
    if (!page[dev]) {
        static std::mutex mu;
        static const float* shared[16] = {nullptr};
        std::lock_guard<std::mutex> l(mu);
        if (!shared[dev]) {
            float* p = nullptr;
            CMOOP_HIP(hipMalloc(&p, 256));
            CMOOP_HIP(hipMemset(p, 0, 256));
            shared[dev] = p;
        }
        page[dev] = shared[dev];
    }
    return page[dev];
}

static void check_dense(int M, int N, int K) {
    CMOOP_REQUIRE(M >= 0 && N >= 1 && K >= 16 && K % 16 == 0, "dense: C_in must be a multiple of 16");
    CMOOP_REQUIRE((int64_t)M * K < (1ll << 31) && (int64_t)M * N < (1ll << 31) && (int64_t)N * K < (1ll << 31), "dense: tensor too large");
}

void launch_dense_fwd(const float* X, const float* W, const float* bias, float* Y, int M, int N, int K, int relu,
                      int dropout, uint32_t drop_prefix, uint32_t drop_thr, float drop_scale, int mode, hipStream_t s,
                      const StepState* st, uint32_t drop_seed, uint32_t drop_stream) {
    check_dense(M, N, K);
    if (M == 0) return;
    DenseEpi e{bias, relu, dropout, drop_prefix, drop_thr, drop_scale, st, drop_seed, drop_stream};
    const unsigned grid = (unsigned)(cdiv(M, 16) * cdiv(N, 16));
    hipLaunchKernelGGL(dense_fwd_kernel, dim3(grid), dim3(256), 0, s, X, W, Y, M, N, K, e, mode == GEMM_BF16 ? 1 : 0, dense_zero_page());
    CMOOP_HIP(hipGetLastError());
}

void launch_dense_dgrad(const float* dY, const float* W, float* dX, int M, int N, int K, const float* mask, float mask_scale,
                        int mode, hipStream_t s) {
    check_dense(M, N, K);
    if (M == 0) return;
    const unsigned grid = (unsigned)(cdiv(M, 16) * (K / 16));
    hipLaunchKernelGGL(dense_dgrad_kernel, dim3(grid), dim3(256), 0, s, dY, W, dX, M, N, K, mask, mask_scale, mode == GEMM_BF16 ? 1 : 0);
    CMOOP_HIP(hipGetLastError());
}

void launch_dense_bwd(const float* X, const float* dY, const float* W, float* dW, float* dB, float* dX, int M, int N, int K,
                      const float* mask, float mask_scale, int mode, hipStream_t s) {
    check_dense(M, N, K);
    const unsigned wg = (unsigned)(cdiv(N, 16) * (K / 16)), dg = (unsigned)(cdiv(M, 16) * (K / 16));
    hipLaunchKernelGGL(dense_bwd_kernel, dim3(wg + dg), dim3(256), 0, s, X, dY, W, dW, dB, dX, M, N, K, mask, mask_scale,
                       mode == GEMM_BF16 ? 1 : 0, (int)wg);
    CMOOP_HIP(hipGetLastError());
}

void launch_dense_wgrad(const float* X, const float* dY, float* dW, float* dB, int M, int N, int K, int mode, hipStream_t s) {
    check_dense(M, N, K);
    const unsigned grid = (unsigned)(cdiv(N, 16) * (K / 16));
    hipLaunchKernelGGL(dense_wgrad_kernel, dim3(grid), dim3(256), 0, s, X, dY, dW, dB, M, N, K, mode == GEMM_BF16 ? 1 : 0);
    CMOOP_HIP(hipGetLastError());
}

}  // namespace cmoop
