// Implicit-GEMM convolution / dense kernels on the CDNA4 fp32 matrix core
// (v_mfma_f32_16x16x4_f32: exact f32 fmaf chain, 64 FLOP/clk/SIMD, 157 TF peak).
//
// Layout: activations NHWC, weights [N][K] with K = (kh, kw, ci) contiguous, so both
// MFMA operands are staged global -> LDS as float4 along K and read back with one
// ds_read_b128 per 16x16 tile per 16 k's: lane (row = l&15, q = l>>4) holds
// k = 16*kk + 4*q + j for the j-th of four MFMAs (A and B use the same k assignment,
// so the order of summation inside a 16-k group is a fixed permutation).
#include "kernels.h"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <mutex>
#include <string>
#include <type_traits>

namespace cmoop {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// OPT-IN bf16 matrix-core modes (cfg.gemm_mode / CMOOP_GEMM_MODE; the default is the exact fp32 MFMA):
//  * GEMM_BF16X3 ("bf16x3"): every fp32 operand is split EXACTLY into three bf16 values by truncation
//    (a = a0 + a1 + a2, 8+8+8 mantissa bits) and a product is evaluated as the six bf16 MFMA terms
//    a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 accumulated in fp32 (dropped terms < 2^-24 relative).
//    v_mfma_f32_16x16x32_bf16 runs at 16x the fp32 MFMA rate, so six of them still cost 2.7x less pipe time
//    than the eight exact-fp32 MFMAs they replace.  fp32-accurate (same parity tolerances), not bit-exact.
//  * GEMM_BF16 ("bf16"): operands rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32), one MFMA term,
//    fp32 accumulation -- "bf16 train" (BASELINE configs[4]); the oracle restates the same rounding.
// Both split/round once, at the global->LDS store, into bf16 planes laid out so that an MFMA fragment
// (8 consecutive k of one row) is a single conflict-free ds_read_b128.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (x, y) -> packed bf16 pair per plane, x in the low half.  NP = 3: exact truncation split; NP = 1: RNE rounding
template <int NP>
__device__ __forceinline__ void split_pair(const float x, const float y, unsigned (&w)[NP]) {
    if constexpr (NP == 1) {
        w[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x, y}, bf16x2));
    } else {
        const unsigned ux = __float_as_uint(x), uy = __float_as_uint(y);
        w[0] = __builtin_amdgcn_perm(uy, ux, 0x07060302u);
        const float rx = x - __uint_as_float(ux & 0xFFFF0000u), ry = y - __uint_as_float(uy & 0xFFFF0000u);
        const unsigned vx = __float_as_uint(rx), vy = __float_as_uint(ry);
        w[1] = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
        const float sx = rx - __uint_as_float(vx & 0xFFFF0000u), sy = ry - __uint_as_float(vy & 0xFFFF0000u);
        w[2] = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
    }
}

// the six product terms of one bf16x3 tile update, small terms first: (a2b0, a1b1, a0b2, a1b0, a0b1, a0b0);
// single-plane bf16 uses only the last
__device__ constexpr int X3_TA[6] = {2, 1, 0, 1, 0, 0};
__device__ constexpr int X3_TB[6] = {0, 1, 2, 0, 1, 0};

int gemm_mode_default() {
    static const int v = [] {
        const char* e = std::getenv("CMOOP_GEMM_MODE");
        if (!e) return (int)GEMM_FP32;
        const std::string m(e);
        return m == "bf16x3" ? (int)GEMM_BF16X3 : (m == "bf16" ? (int)GEMM_BF16 : (int)GEMM_FP32);
    }();
    return v;
}
static int resolve_mode(int mode) {
    const int m = mode >= 0 ? mode : gemm_mode_default();
    return (m == GEMM_BF16X3 || m == GEMM_BF16) ? m : GEMM_FP32;
}

struct GeomDev {
    int B, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad_t, pad_l;
    int M, K, cshift, OHW;
    uint32_t ohw_magic, ohw_shift, ow_magic, ow_shift;   // exact n / OHW and n / OW for n < 2^31 (fastdiv)
    int rcp_kw;            // ceil(65536 / KW): tap / KW == (tap * rcp_kw) >> 16 for tap < 64
    const float* zeros;    // 16 zero floats: predicated-off lanes load from here (branch-free gathers)
};
struct EpiDev {
    int bal_L, bal_segmax; // balanced K partition (bal_L > 0): K-chunk units per workgroup, partial-tile slots per tile in the slab
    float* stats;          // non-null: per-M-tile column partials (sum v, sum v^2) of the stored values, [tiles][2][N] (BatchNorm batch statistics)
    const float* bias;
    const float* mask;
    float mask_scale;
    int relu, accumulate, out_stride, OHf, OWf, dropout;
    uint32_t drop_prefix, drop_thr;
    float drop_scale;
};

// CMOOP_PAR_SCALE (default 1): scales how many workgroups a single GEMM launch aims for.  With several
// candidates in flight the streams fill the chip together, so fewer K-splits / row slices (less slab
// traffic, fewer combine passes) can win; 1.0 sizes every launch to fill the chip alone.
static double par_scale() {
    static const double v = [] {
        const char* e = std::getenv("CMOOP_PAR_SCALE");
        const double x = e ? std::atof(e) : 1.0;
        return x > 0.01 && x <= 4.0 ? x : 1.0;
    }();
    return v;
}

// Granlund-Montgomery division by an invariant: q = (umulhi(n, magic) + n) >> shift, exact for n < 2^31
static void fastdiv_init(int d, uint32_t* magic, uint32_t* shift) {
    uint32_t s = 0;
    while ((1u << s) < (uint32_t)d) ++s;
    *shift = s;
    *magic = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << s) - (uint64_t)d)) / (uint64_t)d + 1);
}
__device__ __forceinline__ int fastdiv(int n, uint32_t magic, uint32_t shift) {
    return (int)((__umulhi((uint32_t)n, magic) + (uint32_t)n) >> shift);
}

// one 64-byte page of zeros per device: the source of every predicated-off float4 gather
static const float* zero_page() {
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

static GeomDev to_dev(const ConvGeom& g) {
    GeomDev d;
    d.B = g.B; d.H = g.H; d.W = g.W; d.Cin = g.Cin; d.OH = g.OH; d.OW = g.OW; d.Cout = g.Cout;
    d.KH = g.KH; d.KW = g.KW; d.stride = g.stride; d.pad_t = g.pad_t; d.pad_l = g.pad_l;
    d.M = g.M(); d.K = g.K(); d.cshift = ilog2_exact(g.Cin); d.OHW = g.OH * g.OW;
    d.rcp_kw = (65536 + g.KW - 1) / g.KW;
    fastdiv_init(d.OHW, &d.ohw_magic, &d.ohw_shift);
    fastdiv_init(g.OW, &d.ow_magic, &d.ow_shift);
    d.zeros = zero_page();
    CMOOP_REQUIRE(g.KH * g.KW <= 64, "kernel window too large");
    CMOOP_REQUIRE(d.cshift >= 4, "implicit GEMM needs C_in a power of two >= 16");
    igemm_check_range(g);
    return d;
}

// The buffer descriptors (num_records), the row tables and the per-row voffsets hold BYTE quantities in 32 bits, some
// of them through signed ints ((int)(elements) * 4): every tensor a GEMM launch addresses must stay below 2^29 elements
// (2 GiB), the padding bias of the input descriptor included.  Beyond that the range check of the descriptors would
// silently return zeros, so the launchers refuse instead.
void igemm_check_range(const ConvGeom& g) {
    const int64_t lim = 1ll << 29;
    const int64_t x_bias = ((int64_t)g.pad_t * g.W + g.pad_l) * g.Cin;
    CMOOP_REQUIRE((int64_t)g.B * g.H * g.W * g.Cin + x_bias < lim,
                  "input tensor of a conv layer exceeds 2^29 elements (32-bit byte offsets): lower the batch / eval_batch");
    CMOOP_REQUIRE((int64_t)g.M() * g.Cout < lim,
                  "output tensor of a conv layer exceeds 2^29 elements (32-bit byte offsets): lower the batch / eval_batch");
    CMOOP_REQUIRE((int64_t)g.Cout * g.K() < lim, "conv kernel tensor exceeds 2^29 elements");
}

std::string gemm_kernel_name(int cls, int code) {
    if (cls == 0) {
        const int mode = code / 100000000, c = code % 100000000;
        const int bm = c / 100000, bn = (c / 100) % 1000, bk = c % 100;
        if (mode == GEMM_FP32_HALO)
            return "halo_fwd_kernel<" + std::to_string(bk % 50) + ", " + std::to_string(bm) + ", " + std::to_string(bn) + ", " + (bn >= 64 ? "2, " : "4, ") +
                   (bk >= 50 ? "true>" : "false>");      // the chunk-depth field carries the window size (+ 50: balanced unit partition)
        const int wm = (bm == bn) ? 2 : 4;   // wave layout of each instantiation (launch_igemm_fwd)
        return "igemm_fwd_kernel<" + std::to_string(bm) + ", " + std::to_string(bn) + ", " + std::to_string(bk) + ", " +
               std::to_string(wm) + ", " + std::to_string(mode) + ">";
    }
    // launch_igemm_wgrad's code: mode * 10^7 + (64-row chunks ? 10^6 : 0) + BCO * 1000 + BKI
    const int mode = code / 10000000, mc = (code / 1000000) % 10 ? 64 : 32, c = code % 1000000;
    if (mode == GEMM_FP32_HALO) return "halo_wgrad_kernel<" + std::to_string(c / 1000) + ", " + std::to_string(c % 1000) + ">";
    const std::string tile = std::to_string(c / 1000) + ", " + std::to_string(c % 1000);
    return mode == GEMM_FP32 ? "igemm_wgrad_kernel<" + tile + ", " + std::to_string(mc) + ">"
                             : "igemm_wgrad_bf16_kernel<" + tile + ", " + (mode == GEMM_BF16X3 ? "3" : "1") + ">";
}

// ---------------------------------------------------------------------------
// forward-type kernel: Y[M][N] = im2col(X)[M][K] * Wt[N][K]^T
// block = 256 threads = 4 waves laid out WM x WN over a BM x BN tile; each wave owns
// RT x CT MFMA tiles of 16x16.  Tile shape is picked per layer so that the grid
// fills the 256 CUs (small-spatial deep layers use 64-row tiles).
// ---------------------------------------------------------------------------
template <int BM, int BN, int BK, int WM, int MODE = 0>   // MODE: GEMM_FP32 exact fp32 MFMA, GEMM_BF16X3 / GEMM_BF16 bf16 planes
__global__ __launch_bounds__(256) void igemm_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                        float* __restrict__ Y, GeomDev g, EpiDev e,
                                                        float* __restrict__ slab, int chunks_per_split,
                                                        const uint2* __restrict__ rowtab, int tab_rows) {
    static_assert(MODE == GEMM_FP32 || MODE == GEMM_FP32_DMA || MODE == GEMM_BF16X3 || MODE == GEMM_BF16, "unknown GEMM mode");
    constexpr bool PLANES = MODE == GEMM_BF16X3 || MODE == GEMM_BF16;
    // GEMM_FP32_DMA: exact fp32 MFMA with the operands brought global -> LDS by the LDS-DMA path (buffer_load_dwordx4 ... lds,
    // 16 bytes per lane: new on gfx950) -- no register staging, no ds_write, no vmcnt wait in front of an LDS store.
    constexpr bool DMA = MODE == GEMM_FP32_DMA;
    static_assert(!DMA || (BK == 32 && BM % 32 == 0 && BN % 32 == 0), "LDS-DMA image: 32-deep chunks, whole 8-row wave stripes");
    constexpr bool PIPE = false;   // software-pipelined main loop (see below): measured, no gain (it halves the global-load landing time)
    // (Measured, NOT enabled: four chunks of global loads in flight for the 16-column tiles of the 16-filter layers -- 8 MFMAs
    // per wave and chunk against a ~2 us global round trip.  The four register sets cost 146 VGPRs = 2 workgroups per CU instead
    // of 5, and 16->16 k3 @101x40 forward fell from 42.9 to 31.9 TFLOP/s: occupancy hides that latency better than depth.)
    constexpr bool DEEP4 = false;
    constexpr bool DIST2 = !PLANES && BM <= 64;   // two-chunk-deep global prefetch (two register sets); on 128-row tiles it costs the second workgroup per CU (measured 104 vs 124 TFLOP/s)
    constexpr int NP = MODE == GEMM_BF16X3 ? 3 : 1;
    constexpr int WN = 4 / WM;
    // register-staged image: pitch BK + 8 floats = 2 (mod 4) sixteen-byte slots: conflict-free ds_read_b128 fragments (16-lane
    // groups, 64 banks).  LDS-DMA image: a wave instruction lands 64 lanes x 16 B CONTIGUOUSLY (8 rows x 128 B), so rows
    // are unpadded (pitch 32 floats) and the sixteen-byte slot of a row is XOR-swizzled with the row's low 3 bits on the
    // SOURCE side (lane l fetches global slot (l & 7) ^ (l >> 3)); fragment reads undo it: conflict-free as well.
    constexpr int LDK = DMA ? BK : BK + 8;
    constexpr int TPR = BK / 4;
    constexpr int RPP = 256 / TPR;
    constexpr int APASS = (BM + RPP - 1) / RPP;
    constexpr int BPASS = (BN + RPP - 1) / RPP;
    constexpr int RT = BM / WM / 16, CT = BN / WN / 16;
    static_assert(RT >= 1 && CT >= 1, "tile too small for the wave layout");
    static_assert(!PLANES || BK == 32, "bf16 planes need 32-deep K chunks");
    constexpr int LDH = 48;   // bf16 planes: rows of 32 k + 16 pad = 96 B = 6 slots (conflict-free b128 fragment reads)
    __shared__ __attribute__((aligned(16))) float As[PLANES ? 1 : 2][PLANES ? 4 : BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[PLANES ? 1 : 2][PLANES ? 4 : BN * LDK];
    __shared__ __attribute__((aligned(16))) unsigned short Ah[NP][PLANES ? BM * LDH : 8];
    __shared__ __attribute__((aligned(16))) unsigned short Bh[NP][PLANES ? BN * LDH : 8];

    const int t = threadIdx.x;
    const int lrow = t / TPR, kq = t % TPR;
    const int chunks_total = (g.K + BK - 1) / BK;
    // Balanced K partition (e.bal_L > 0; under-filled grids: the 26x10 / 13x5 layers): the launch's work is the
    // linear sequence of (tile, K chunk) units, tile-major, and workgroup w owns units [w L, (w+1) L) -- every
    // workgroup the SAME number of chunks, whatever the tile count is against the chip's 512 workgroup slots
    // (uniform split-K left 11-49 % of a wave of slots empty: 260 tiles x 7 splits on 512 slots).  A range crosses at
    // most one tile boundary (L <= chunks per tile), so a workgroup runs up to two pieces; each piece writes its raw
    // partial tile into slot (tile, seg) of the slab, seg counting the pieces of that tile in K order, and the
    // combine kernel adds a tile's pieces in that fixed order (deterministic) before the epilogue.
    const int pieces = e.bal_L > 0 ? 2 : 1;
    for (int piece = 0; piece < pieces; ++piece) {
    int mtile, ntile, cbeg, nchunks;
    float* pslab = nullptr;
    if (e.bal_L > 0) {
        const int nt = (g.Cout + BN - 1) / BN;
        const int L = e.bal_L;
        const int U = ((g.M + BM - 1) / BM) * nt * chunks_total;
        const int u0 = blockIdx.x * L, u1 = min(U, u0 + L);
        if (u0 >= u1) break;
        int t0 = u0 / chunks_total;
        int cb = u0 - t0 * chunks_total, ce = min(chunks_total, cb + (u1 - u0));
        if (piece == 1) {
            if (u0 + (ce - cb) >= u1) break;          // the range ended inside its first tile
            t0 += 1; cb = 0; ce = u1 - t0 * chunks_total;
            __syncthreads();                          // the first piece's LDS images are dead only now
        }
        mtile = t0 / nt; ntile = t0 - mtile * nt;
        cbeg = cb; nchunks = ce;
        const int seg = cb == 0 ? 0 : (int)blockIdx.x - (t0 * chunks_total) / L;
        pslab = slab + ((size_t)t0 * e.bal_segmax + seg) * (BM * BN);
    } else {
        // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, so give each
        // XCD a contiguous run of M tiles -- neighbouring tiles share their im2col halo in that XCD's L2.
        mtile = blockIdx.x;
        const int nwg = gridDim.x, xcd = mtile & 7, idx = mtile >> 3;
        const int qn = nwg >> 3, rn = nwg & 7;
        mtile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
        ntile = blockIdx.y;
        // split-K: blockIdx.z owns K chunks [cbeg, nchunks) and writes raw partial sums to its slab
        cbeg = blockIdx.z * chunks_per_split;
        nchunks = min(chunks_total, cbeg + chunks_per_split);
    }
    const int m0 = mtile * BM;
    const int n0 = ntile * BN;

    // per-pass row decode (fixed over the K loop); with the layer's row table (the trainer's forward launches) the fast
    // loader's per-row offset and padding mask are two table words instead of two divisions and ~40 VALU per row
    const bool fast_geom = (g.Cin & (BK - 1)) == 0 && g.KH * g.KW <= 32;
    const bool use_tab = fast_geom && rowtab != nullptr && m0 + BM <= tab_rows;
    int a_ih0[APASS], a_iw0[APASS], a_base[APASS];
    bool a_ok[APASS];
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        int ml = lrow + p * RPP;
        int m = m0 + ml;
        a_ok[p] = ml < BM && m < g.M;
        a_ih0[p] = a_iw0[p] = a_base[p] = 0;
        if (!use_tab) {
            int mm = a_ok[p] ? m : 0;
            int b = fastdiv(mm, g.ohw_magic, g.ohw_shift), r = mm - b * g.OHW;
            int oh = fastdiv(r, g.ow_magic, g.ow_shift), ow = r - oh * g.OW;
            a_ih0[p] = oh * g.stride - g.pad_t;
            a_iw0[p] = ow * g.stride - g.pad_l;
            a_base[p] = b * g.H * g.W;
        }
    }

    // two register sets: the loads of chunk c+2 are issued before chunk c is computed and are
    // written to LDS only after chunk c+1's compute, so a global-load round trip has two MFMA
    // phases to land (the 64-row tiles of the deep layers have only ~1k MFMA cycles per phase)
    f32x4 ra0[APASS], rb0[BPASS], ra1[(DIST2 || DEEP4) ? APASS : 1], rb1[(DIST2 || DEEP4) ? BPASS : 1];
    f32x4 ra2[DEEP4 ? APASS : 1], rb2[DEEP4 ? BPASS : 1], ra3[DEEP4 ? APASS : 1], rb3[DEEP4 ? BPASS : 1];
    // ---- fast operand loader (a K chunk lies inside ONE filter tap: Cin % BK == 0) -----------------------------------
    // Every non-MFMA vector instruction of a wave takes issue cycles from the matrix pipe it shares with its SIMD
    // partner (measured: the un-tuned loader's ~90 VALU per chunk cost 12-15 % of the kernel), so the per-chunk
    // address work is moved off the VALU: buffer loads with a wave-uniform descriptor, a per-row byte offset computed
    // ONCE (voffset), the chunk's tap / channel offset on the scalar unit (soffset), and SAME-padding handled by a
    // precomputed per-row bitmask of invalid taps that turns the voffset out of range (the hardware range check then
    // returns zeros): 2 VALU per A load, 0 per B load.
    const bool fast = fast_geom;
    const uint32_t x_bias = (uint32_t)((g.pad_t * g.W + g.pad_l) << g.cshift);          // floats: makes every row offset >= 0
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X) - x_bias, 0, (int)(((uint32_t)g.B * g.H * g.W << g.cshift) + x_bias) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, g.Cout * g.K * 4, 0x00020000);
    uint32_t a_voff[APASS], a_inv[APASS], b_voff[BPASS];
    if (use_tab) {
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const int ml = lrow + p * RPP;
            const uint2 te = rowtab[m0 + (ml < BM ? ml : 0)];
            a_voff[p] = te.x + 16u * (uint32_t)(DMA ? (kq ^ (lrow & 7)) : kq);
            a_inv[p] = a_ok[p] ? te.y : 0xFFFFFFFFu;
        }
    } else if (fast) {
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            a_voff[p] = ((uint32_t)((a_base[p] + a_ih0[p] * g.W + a_iw0[p]) << g.cshift) + x_bias + 4 * (DMA ? (kq ^ (lrow & 7)) : kq)) * 4u;
            // valid kh / kw form contiguous ranges: [max(0,-ih0), min(KH, H-ih0)) x [max(0,-iw0), min(KW, W-iw0))
            const int hlo = max(0, -a_ih0[p]), hhi = min(g.KH, g.H - a_ih0[p]);
            const int wlo = max(0, -a_iw0[p]), whi = min(g.KW, g.W - a_iw0[p]);
            const uint32_t wmask = (whi > wlo) ? (((1u << whi) - 1u) & ~((1u << wlo) - 1u)) : 0u;
            uint32_t okm = 0;
            for (int kh = 0; kh < g.KH; ++kh)
                if (kh >= hlo && kh < hhi) okm |= wmask << (kh * g.KW);
            a_inv[p] = a_ok[p] ? ~okm : 0xFFFFFFFFu;
        }
    }
    if (fast) {
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            const int nl = lrow + p * RPP, n = n0 + nl;
            b_voff[p] = (nl < BN && n < g.Cout) ? (uint32_t)(n * g.K + 4 * (DMA ? (kq ^ (lrow & 7)) : kq)) * 4u : 0xFFFFFFF0u;
        }
    }
    auto load_chunk_fast = [&](int c, f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {
        // wave-uniform: tap and first channel of the chunk, as a byte offset for the scalar soffset operand
        const int k0 = c * BK;
        const int tap = k0 >> g.cshift, ci0 = k0 & (g.Cin - 1);
        const int kh = (tap * g.rcp_kw) >> 16, kw = tap - kh * g.KW;
        const int a_soff = ((((kh * g.W + kw) << g.cshift) + ci0) * 4);
        const int b_soff = k0 * 4;
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const uint32_t dead = (uint32_t)__builtin_amdgcn_sbfe((int)a_inv[p], tap, 1);     // -1 when this tap is padding for the row
            ra[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(a_voff[p] | dead), a_soff, 0));
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p)
            rb[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)b_voff[p], b_soff, 0));
    };
    // LDS-DMA: chunk c straight into LDS image `buf` (out-of-range lanes land zeros: probed on gfx950, tools/debug/lds_dma_probe.hip)
    auto dma_chunk = [&](int c, int buf) {
        if constexpr (DMA) {
            const int k0 = c * BK;
            const int tap = k0 >> g.cshift, ci0 = k0 & (g.Cin - 1);
            const int kh = (tap * g.rcp_kw) >> 16, kw = tap - kh * g.KW;
            const int a_soff = ((((kh * g.W + kw) << g.cshift) + ci0) * 4);
            const int b_soff = k0 * 4;
            const int wv = __builtin_amdgcn_readfirstlane(t >> 6);          // wave-uniform LDS base (goes to M0)
#pragma unroll
            for (int p = 0; p < APASS; ++p) {
                const uint32_t dead = (uint32_t)__builtin_amdgcn_sbfe((int)a_inv[p], tap, 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)&As[buf][(p * RPP + 8 * wv) * LDK], 16,
                                                         (int)(a_voff[p] | dead), a_soff, 0, 0);
            }
#pragma unroll
            for (int p = 0; p < BPASS; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)&Bs[buf][(p * RPP + 8 * wv) * LDK], 16,
                                                         (int)b_voff[p], b_soff, 0, 0);
        }
    };
    auto load_chunk_slow = [&](int c, f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {
        const int kidx = c * BK + 4 * kq;
        const bool kok = kidx < g.K;
        const int tap = kidx >> g.cshift, ci = kidx & (g.Cin - 1);
        const int kh = (tap * g.rcp_kw) >> 16, kw = tap - kh * g.KW;
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            int ih = a_ih0[p] + kh, iw = a_iw0[p] + kw;
            bool ok = kok && a_ok[p] && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
            const float* src = ok ? X + (((size_t)(a_base[p] + ih * g.W + iw)) << g.cshift) + ci : g.zeros;
            ra[p] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int nl = lrow + p * RPP;
            int n = n0 + nl;
            const float* src = (nl < BN && n < g.Cout && kok) ? Wt + (size_t)n * g.K + kidx : g.zeros;
            rb[p] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    auto load_chunk = [&](int c, f32x4 (&ra)[APASS], f32x4 (&rb)[BPASS]) {
        if (fast) load_chunk_fast(c, ra, rb);
        else load_chunk_slow(c, ra, rb);
    };
    auto store_chunk = [&](int buf, const f32x4 (&ra)[APASS], const f32x4 (&rb)[BPASS]) {
        // (the row guards are compile-time true whenever the passes tile the block exactly: no exec-mask branches then)
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            int ml = lrow + p * RPP;
            if (APASS * RPP == BM || ml < BM) *reinterpret_cast<f32x4*>(&As[buf][ml * LDK + 4 * kq]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int nl = lrow + p * RPP;
            if (BPASS * RPP == BN || nl < BN) *reinterpret_cast<f32x4*>(&Bs[buf][nl * LDK + 4 * kq]) = rb[p];
        }
    };

    // bf16 planes: split (or round) each float4 -- 4 consecutive k of one row -- and store one 8-byte piece per plane
    auto store_split = [&](const f32x4 (&ra)[APASS], const f32x4 (&rb)[BPASS]) {
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            int ml = lrow + p * RPP;
            if (ml < BM) {
                unsigned lo[NP], hi[NP];
                split_pair<NP>(ra[p][0], ra[p][1], lo);
                split_pair<NP>(ra[p][2], ra[p][3], hi);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(&Ah[pl][ml * LDH + 4 * kq]) = make_uint2(lo[pl], hi[pl]);
            }
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            int nl = lrow + p * RPP;
            if (nl < BN) {
                unsigned lo[NP], hi[NP];
                split_pair<NP>(rb[p][0], rb[p][1], lo);
                split_pair<NP>(rb[p][2], rb[p][3], hi);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<uint2*>(&Bh[pl][nl * LDH + 4 * kq]) = make_uint2(lo[pl], hi[pl]);
            }
        }
    };

    const int wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int wrow = (wave / WN) * (BM / WM), wcol = (wave % WN) * (BN / WN);
    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        if constexpr (PLANES) {
            // fragments are ready-made bf16x8: lane (lr, q) reads 16 B = k 8q..8q+7 of its row from each plane
            bf16x8 b[CT][NP];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    b[ct][pl] = *reinterpret_cast<const bf16x8*>(&Bh[pl][(wcol + ct * 16 + lr) * LDH + q * 8]);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                bf16x8 a[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    a[pl] = *reinterpret_cast<const bf16x8*>(&Ah[pl][(wrow + rt * 16 + lr) * LDH + q * 8]);
                __builtin_amdgcn_s_setprio(1);
                // term-major order: consecutive MFMAs hit different accumulators (no dependent back-to-back issue)
#pragma unroll
                for (int term = (NP == 3 ? 0 : 5); term < 6; ++term)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[X3_TA[term] % NP], b[ct][X3_TB[term] % NP], acc[rt][ct], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            (void)buf;
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            f32x4 a[RT], b[CT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                a[rt] = *reinterpret_cast<const f32x4*>(&As[buf][(wrow + rt * 16 + lr) * LDK + (DMA ? (((kk * 4 + q) ^ (lr & 7)) * 4) : kk * 16 + q * 4)]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                b[ct] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wcol + ct * 16 + lr) * LDK + (DMA ? (((kk * 4 + q) ^ (lr & 7)) * 4) : kk * 16 + q * 4)]);
            // raised priority over the MFMA burst: the co-resident wave's address VALU no longer wins
            // issue arbitration against it (+3..6 % measured, same-box A/B)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][j], b[ct][j], acc[rt][ct], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    };

    if constexpr (PLANES) {
        // single LDS image (bf16 planes per operand), next chunk's global loads in flight during the MFMAs
        // (a two-chunk-deep prefetch in two register sets was measured: it drops the kernel to one workgroup per
        // CU and loses 25-35 %)
        load_chunk(cbeg, ra0, rb0);
        for (int c = cbeg; c < nchunks; ++c) {
            store_split(ra0, rb0);
            __syncthreads();
            if (c + 1 < nchunks) load_chunk(c + 1, ra0, rb0);
            compute(0);
            __syncthreads();
        }
    } else if constexpr (DEEP4) {
        // distance-4 prefetch: set (i mod 4) holds chunk i until it is stored to LDS image (i & 1) one iteration before use
        const int n = nchunks - cbeg;
        load_chunk(cbeg, ra0, rb0);
        if (n > 1) load_chunk(cbeg + 1, ra1, rb1);
        if (n > 2) load_chunk(cbeg + 2, ra2, rb2);
        if (n > 3) load_chunk(cbeg + 3, ra3, rb3);
        store_chunk(0, ra0, rb0);
        __syncthreads();
        for (int i = 0; i < n; i += 4) {
            const int c = cbeg + i;
            // chunk i (LDS 0): set 0 is free -> chunk i+4; set 1 (chunk i+1) goes to LDS 1 after the compute
            if (i + 4 < n) load_chunk(c + 4, ra0, rb0);
            compute(0);
            if (i + 1 < n) store_chunk(1, ra1, rb1);
            __syncthreads();
            if (i + 1 >= n) break;
            if (i + 5 < n) load_chunk(c + 5, ra1, rb1);
            compute(1);
            if (i + 2 < n) store_chunk(0, ra2, rb2);
            __syncthreads();
            if (i + 2 >= n) break;
            if (i + 6 < n) load_chunk(c + 6, ra2, rb2);
            compute(0);
            if (i + 3 < n) store_chunk(1, ra3, rb3);
            __syncthreads();
            if (i + 3 >= n) break;
            if (i + 7 < n) load_chunk(c + 7, ra3, rb3);
            compute(1);
            if (i + 4 < n) store_chunk(0, ra0, rb0);
            __syncthreads();
        }
    } else if constexpr (DIST2) {
        // distance-2 prefetch (two register sets): a global-load round trip gets two MFMA phases to land
        load_chunk(cbeg, ra0, rb0);
        store_chunk(0, ra0, rb0);
        if (cbeg + 1 < nchunks) load_chunk(cbeg + 1, ra1, rb1);
        __syncthreads();
        for (int c = cbeg; c < nchunks; c += 2) {
            // even chunk c lives in LDS[0]; set 1 holds chunk c+1; set 0 is free
            if (c + 2 < nchunks) load_chunk(c + 2, ra0, rb0);
            compute(0);
            if (c + 1 < nchunks) store_chunk(1, ra1, rb1);
            __syncthreads();
            if (c + 1 >= nchunks) break;
            // odd chunk c+1 lives in LDS[1]; set 0 holds chunk c+2; set 1 is free
            if (c + 3 < nchunks) load_chunk(c + 3, ra1, rb1);
            compute(1);
            if (c + 2 < nchunks) store_chunk(0, ra0, rb0);
            __syncthreads();
        }
    } else if constexpr (DMA) {
        // operands go global -> LDS directly: chunk c + 1 lands in the other image while chunk c is multiplied; the only
        // wait is vmcnt(0) in front of the barrier that publishes it
        dma_chunk(cbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c = cbeg; c < nchunks; ++c) {
            const int buf = (c - cbeg) & 1;
            if (c + 1 < nchunks) dma_chunk(c + 1, buf ^ 1);
            compute(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if constexpr (BK == 32 && PIPE) {
        // Software-pipelined 128-row tiles: the fragments of 16-k group g+1 are read from LDS while the MFMAs of group
        // g issue (two fragment register sets), and the barrier sits BETWEEN the two groups of a chunk, so every
        // barrier release is followed by a full MFMA burst that covers the next chunk's first fragment reads and the
        // address arithmetic of the next global loads.  MEASURED, NOT ENABLED (PIPE = false): 120-122 vs 123-125 TFLOP/s
        // on 128->128 k5 -- it halves the time the global loads have to land, and the LDS latencies it hides were
        // already covered by the partner wave; asymmetric burst priorities between the co-resident workgroups
        // (to break a suspected lockstep) changed nothing either.
        f32x4 fa[2][RT], fb[2][CT];
        auto read_frags = [&](int buf, int kk, int set) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                fa[set][rt] = *reinterpret_cast<const f32x4*>(&As[buf][(wrow + rt * 16 + lr) * LDK + kk * 16 + q * 4]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                fb[set][ct] = *reinterpret_cast<const f32x4*>(&Bs[buf][(wcol + ct * 16 + lr) * LDK + kk * 16 + q * 4]);
        };
        auto mfma_set = [&](int set) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][rt][j], fb[set][ct][j], acc[rt][ct], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        load_chunk(cbeg, ra0, rb0);
        store_chunk(0, ra0, rb0);
        __syncthreads();
        read_frags(0, 0, 0);
        for (int c = cbeg; c < nchunks; ++c) {
            const int buf = (c - cbeg) & 1;
            const bool more = c + 1 < nchunks;
            if (more) load_chunk(c + 1, ra0, rb0);
            read_frags(buf, 1, 1);
            mfma_set(0);
            if (more) store_chunk(buf ^ 1, ra0, rb0);
            __syncthreads();
            if (more) read_frags(buf ^ 1, 0, 0);
            mfma_set(1);
        }
        __syncthreads();   // the epilogue's statistics reuse the A image
    } else {
        // distance-1 prefetch: one register set keeps the 128-row tiles at two workgroups per CU
        load_chunk(cbeg, ra0, rb0);
        store_chunk(0, ra0, rb0);
        __syncthreads();
        for (int c = cbeg; c < nchunks; ++c) {
            const int buf = (c - cbeg) & 1;
            if (c + 1 < nchunks) load_chunk(c + 1, ra0, rb0);
            compute(buf);
            if (c + 1 < nchunks) store_chunk(buf ^ 1, ra0, rb0);
            __syncthreads();
        }
    }

    // epilogue: C/D map of 16x16x4: col = lane&15, row = 4*(lane>>4) + reg
    const int N = g.Cout;
    if (pslab) {   // balanced partition: this piece's raw partial tile, [BM][BN] in its (tile, seg) slot
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    pslab[(wrow + rt * 16 + q * 4 + r) * BN + wcol + ct * 16 + lr] = acc[rt][ct][r];
        continue;
    }
    if (slab) {   // split-K partial: raw sums, the combine kernel applies the epilogue
        float* Pz = slab + (size_t)blockIdx.z * g.M * N;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wrow + rt * 16 + q * 4 + r;
                if (row >= g.M) continue;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int col = n0 + wcol + ct * 16 + lr;
                    if (col < N) Pz[(size_t)row * N + col] = acc[rt][ct][r];
                }
            }
        continue;
    }
    float csum[CT], csq[CT], bias_v[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        csum[ct] = 0.f; csq[ct] = 0.f;
        const int col = n0 + wcol + ct * 16 + lr;
        bias_v[ct] = (e.bias && col < N) ? e.bias[col] : 0.f;    // once per column, not once per stored element
    }
    // straight-line forms for whole tiles of dense-stored outputs (see halo_fwd_kernel's epilogue): the short-K launches left to this
    // kernel -- the 1x1 skip projections, the small candidates' deep layers -- are mostly epilogue
    const bool whole = m0 + BM <= g.M && (N % BN) == 0 && e.out_stride == 1;
    const uint32_t e_lane = (uint32_t)((m0 + wrow + q * 4) * N + n0 + wcol + lr) * 4u;     // element (rt 0, r 0, ct 0) of this lane, bytes
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, g.M * N * 4, 0x00020000);
    if (whole && !e.dropout && !e.mask && !e.accumulate) {
        auto store_all = [&](auto relu_c, auto stats_c) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int soff = (rt * 16 + r) * N * 4;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        float v = acc[rt][ct][r] + bias_v[ct];
                        if constexpr (decltype(relu_c)::value) v = fmaxf(v, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)(e_lane + ct * 64), soff, 0);
                        if constexpr (decltype(stats_c)::value) { csum[ct] += v; csq[ct] += v * v; }
                    }
                }
        };
        using T1 = std::integral_constant<bool, true>;
        using T0 = std::integral_constant<bool, false>;
        if (e.relu) { if (e.stats) store_all(T1{}, T1{}); else store_all(T1{}, T0{}); }
        else        { if (e.stats) store_all(T0{}, T1{}); else store_all(T0{}, T0{}); }
    } else if (whole && e.mask && !e.dropout && !e.relu && !e.bias && !e.stats) {
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.mask), 0, g.M * N * 4, 0x00020000);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float mk[4][CT], old[4][CT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int soff = (rt * 16 + r) * N * 4;
                    mk[r][ct] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrs, (int)(e_lane + ct * 64), soff, 0));
                    old[r][ct] = e.accumulate ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, (int)(e_lane + ct * 64), soff, 0)) : 0.f;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int soff = (rt * 16 + r) * N * 4;
                    float v = (mk[r][ct] > 0.f) ? acc[rt][ct][r] * e.mask_scale : 0.f;
                    if (e.accumulate) v += old[r][ct];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)(e_lane + ct * 64), soff, 0);
                }
        }
    } else {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wrow + rt * 16 + q * 4 + r;
            if (row >= g.M) continue;
            size_t rbase;
            if (e.out_stride == 1) {
                rbase = (size_t)row * N;
            } else {
                int b = row / g.OHW, rr = row - b * g.OHW;
                int oh = rr / g.OW, ow = rr - oh * g.OW;
                rbase = ((size_t)(b * e.OHf + oh * e.out_stride) * e.OWf + ow * e.out_stride) * N;
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col = n0 + wcol + ct * 16 + lr;
                if (col >= N) continue;
                float v = acc[rt][ct][r] + bias_v[ct];
                if (e.relu) v = fmaxf(v, 0.f);
                const size_t off = rbase + col;
                if (e.dropout) {
                    uint32_t u24 = fmix32(e.drop_prefix ^ (uint32_t)((size_t)row * N + col)) >> 8;
                    v = (u24 >= e.drop_thr) ? v * e.drop_scale : 0.f;
                }
                if (e.mask) v = (e.mask[off] > 0.f) ? v * e.mask_scale : 0.f;
                if (e.accumulate) v += Y[off];
                Y[off] = v;
                csum[ct] += v;
                csq[ct] += v * v;
            }
        }
    }
    }
    if (e.stats) {
        // BatchNorm batch statistics ride along: column sums over this tile's rows in a fixed order (lane's rows ->
        // the four row groups of the wave by shuffle -> the WM waves through LDS), one partial per M tile
        float* red = PLANES ? reinterpret_cast<float*>(&Ah[0][0]) : &As[0][0];   // free after the K loop's last barrier
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            csum[ct] += __shfl_xor(csum[ct], 16, 64);
            csq[ct] += __shfl_xor(csq[ct], 16, 64);
            csum[ct] += __shfl_xor(csum[ct], 32, 64);
            csq[ct] += __shfl_xor(csq[ct], 32, 64);
            if (q == 0) {
                const int c = wcol + ct * 16 + lr;
                red[((wave / WN) * BN + c) * 2] = csum[ct];
                red[((wave / WN) * BN + c) * 2 + 1] = csq[ct];
            }
        }
        __syncthreads();
        if (t < BN && n0 + t < N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += red[(w * BN + t) * 2]; b += red[(w * BN + t) * 2 + 1]; }
            e.stats[((size_t)mtile * 2) * N + n0 + t] = a;
            e.stats[((size_t)mtile * 2 + 1) * N + n0 + t] = b;
        }
    }
    }   // piece loop
}

// ---------------------------------------------------------------------------
// Halo-tiled direct convolution on the fp32 MFMA (stride 1, SAME padding, odd square window): forward-type launches
// (forward, and dgrad with the flip-transposed kernel) of the layers with enough output pixels to fill the chip un-split.
//
// The implicit GEMM above re-gathers a tile's A operand once per filter tap (25x for k5) and pays for it in issue
// slots: per 16-k chunk and wave 2-4 global loads, as many LDS stores, a barrier and the address VALU, against 32-64
// MFMAs.  Here a workgroup stages the INPUT HALO of its 128 consecutive output pixels once per 16-channel chunk
// (~20 KiB instead of 25 x 8 KiB) and then runs all KS*KS taps on it: an A fragment of tap (ky, kx) is the fragment of
// tap (0, 0) at a CONSTANT LDS byte offset (ky * WP + kx) * 96, which goes into the ds_read's immediate field.  The B
// fragments (16 output channels x 16 k per wave instruction) are read straight from global memory in the MFMA lane
// layout, one tap ahead (the weights of a layer are L1 / L2 resident), so the steady state per tap and wave is
// RT ds_read_b128 + CT buffer_load_b128 + 4 RT CT MFMAs with NO barrier, NO LDS store and NO vector address arithmetic;
// the two barriers of the halo refill come once per KS*KS taps.
//
// Tiles are the same flat 128-row M tiles as the implicit GEMM (so the epilogue, the BatchNorm statistics partials and
// the tile order are shared).  A tile may cross image boundaries: rows live in a VIRTUAL tall image, image b at virtual
// rows [b (H + R), b (H + R) + H) with R = KS / 2 zero rows between neighbours -- one gap serves as bottom padding of
// the image above and top padding of the image below, and the tap shift stays a constant offset across the boundary.
// Halo pixels are stored [virtual row][column][16 channels + 8 pad]; the row pitch WP is W + KS - 1 rounded up to a
// multiple of 8 pixels, so the pixel index of consecutive output pixels stays consecutive mod 8 across a row wrap and
// the ds_read_b128 fragments are conflict-free like the implicit GEMM's (pitch 24 floats).
// ---------------------------------------------------------------------------
struct HaloDev {
    int WP, rows_max, VH;                  // halo row pitch (pixels), rows the LDS image holds, H + R
    uint32_t wp_magic, wp_shift, vh_magic, vh_shift;
};
__host__ __device__ constexpr int halo_nst(int bm) { return bm == 256 ? 11 : 9; }
static bool halo_geometry(const ConvGeom& cg, HaloDev* hd, size_t* lds_bytes);
static size_t halo_balanced_slab_floats(const ConvGeom& cg, int* L_out);

template <int KS, int BM, int BN, int WM, bool BAL = false>   // BM x BN output tile (128 x 64; 256 x 32 / 256 x 16 for the narrow layers), WM x (4 / WM) waves
__global__ __launch_bounds__(256, 2) void halo_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                          float* __restrict__ Y, GeomDev g, EpiDev e, HaloDev h,
                                                          float* __restrict__ slab) {
    constexpr int WN = 4 / WM, RT = BM / WM / 16, CT = BN / WN / 16, R = KS / 2, T = KS * KS;
    constexpr int PITCH = 24;      // floats per halo pixel: 16 channels + 8 pad
    constexpr int NST = halo_nst(BM);   // staged float4 per thread and channel chunk (host checks rows_max * WP * 4 <= NST * 256)
    static_assert(RT >= 1 && CT >= 1, "tile too small for the wave layout");
    extern __shared__ __attribute__((aligned(16))) float halo[];

    const int t = threadIdx.x;
    const int ncc = g.Cin >> 4;
    // BAL (under-filled grids: the 26x10 / 13x5 layers at the train batch): the launch's work is the tile-major sequence of
    // (tile, channel chunk, kernel row) units -- KS taps of 16 channels each -- and workgroup w owns units [w L, (w+1) L): the
    // same number for every workgroup whatever the tile count is against the chip's 512 slots.  A range is cut into pieces at
    // tile boundaries; every piece writes its raw partial tile to slot (tile, seg) of the slab, seg counting the tile's pieces
    // in unit order, and splitk_combine_kernel adds them in that order before the epilogue (as for igemm_fwd_kernel).
    const int upt = ncc * KS;                                  // units per tile
    int u = 0, uend = 1;
    if constexpr (BAL) {
        const int U = ((g.M + BM - 1) / BM) * ((g.Cout + BN - 1) / BN) * upt;
        u = blockIdx.x * e.bal_L;
        uend = min(U, u + e.bal_L);
    }
    while (u < uend) {
    int mtile, ntile, c_lo = 0, c_hi = upt;
    float* pslab = nullptr;
    if constexpr (BAL) {
        const int nt = (g.Cout + BN - 1) / BN;
        const int t0 = u / upt;
        c_lo = u - t0 * upt;
        c_hi = min(upt, c_lo + (uend - u));
        mtile = t0 / nt; ntile = t0 - mtile * nt;
        const int seg = c_lo == 0 ? 0 : (int)blockIdx.x - (t0 * upt) / e.bal_L;
        pslab = slab + ((size_t)t0 * e.bal_segmax + seg) * (BM * BN);
        u += c_hi - c_lo;
        __syncthreads();                                       // the previous piece's halo image is dead only now
    } else {
        u = uend;
        mtile = blockIdx.x;
        // XCD-aware tile order, as in igemm_fwd_kernel: neighbouring tiles share halo rows in one XCD's L2
        const int nwg = gridDim.x, xcd = mtile & 7, idx = mtile >> 3;
        const int qn = nwg >> 3, rn = nwg & 7;
        mtile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
        ntile = blockIdx.y;
    }
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int wrow = (wave / WN) * (BM / WM), wcol = (wave % WN) * (BN / WN);

    // output pixel m -> virtual row (b (H + R) + y) and column
    auto vrow = [&](int m, int& x) {
        const int b = fastdiv(m, g.ohw_magic, g.ohw_shift), r = m - b * g.OHW;
        const int y = fastdiv(r, g.ow_magic, g.ow_shift);
        x = r - y * g.OW;
        return b * h.VH + y;
    };
    int xdummy;
    const int vbase = vrow(m0, xdummy);                              // virtual row of the tile's first pixel
    const int vlast = vrow(min(m0 + BM, g.M) - 1, xdummy);
    const int items = (vlast - vbase + 1 + 2 * R) * h.WP * 4;        // float4 slots of this tile's halo (pad columns included)

    // staging descriptors: slot i = t + 256 it -> halo pixel i >> 2 (row-major over [rows][WP]), channels 4 (i & 3) ...
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X), 0, (int)((uint32_t)g.B * g.H * g.W << g.cshift) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt), 0, g.Cout * g.K * 4, 0x00020000);
    // (Measured, not adopted: the image / row decode of the staging slots through a per-row table in LDS -- 313 instead of 367 vector
    // instructions in the tile prologue -- no measurable change on any shape.)
    uint32_t s_voff[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int i = t + 256 * it, pix = i >> 2, j = i & 3;
        const int hy = fastdiv(pix, h.wp_magic, h.wp_shift), hx = pix - hy * h.WP;
        const int v = vbase - R + hy, vc = max(v, 0);
        const int b = fastdiv(vc, h.vh_magic, h.vh_shift), y = vc - b * h.VH;
        const bool ok = i < items && v >= 0 && b < g.B && y < g.H && hx >= R && hx < g.W + R;
        s_voff[it] = ok ? (uint32_t)(((((b * g.H + y) * g.W + hx - R) << g.cshift) + 4 * j) * 4) : 0xFFFFFFFFu;
    }
    float* const st_dst = halo + (t >> 2) * PITCH + (t & 3) * 4;     // + it * 64 * PITCH floats
    f32x4 st[NST];
    auto stage_load = [&](int cc) {
#pragma unroll
        for (int it = 0; it < NST; ++it)
            st[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)s_voff[it], cc * 64, 0));
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int it = 0; it < NST; ++it)
            if (t + 256 * it < items) *reinterpret_cast<f32x4*>(st_dst + it * 64 * PITCH) = st[it];
    };

    // fragment addresses: lane (lr, q) reads channels 4q..4q+3 of output pixel (wrow + 16 rt + lr) shifted by the tap
    const float* a_ptr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        int x;
        const int v = vrow(min(m0 + wrow + rt * 16 + lr, g.M - 1), x);
        a_ptr[rt] = halo + ((v - vbase) * h.WP + x) * PITCH + 4 * q;
    }
    uint32_t b_voff[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = n0 + wcol + ct * 16 + lr;
        b_voff[ct] = col < g.Cout ? (uint32_t)(col * g.K + 4 * q) * 4u : 0xFFFFFFFFu;
    }
    const int wp_f = h.WP * PITCH;             // floats per halo row
    auto load_b = [&](int cc, int tap, f32x4 (&b)[CT]) {
        const int soff = ((tap << g.cshift) + cc * 16) * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            b[ct] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)b_voff[ct], soff, 0));
    };

    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (BAL) {
        // unit loop: unit c = (channel chunk c / KS, kernel row c % KS); the KS taps of a unit are unrolled (their column shift is
        // a ds_read immediate), the kernel row is one address add per unit; B and A fragments run one tap ahead as below
        int cc = c_lo / KS, ky = c_lo - cc * KS;
        f32x4 ac[RT], bc[CT];
        stage_load(cc);
        load_b(cc, ky * KS, bc);
        stage_store();
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ac[rt] = *reinterpret_cast<const f32x4*>(a_ptr[rt] + ky * wp_f);
        for (int c = c_lo; c < c_hi; ++c) {
            const bool last_unit = c + 1 == c_hi, new_cc = ky == KS - 1;
            const int nky = new_cc ? 0 : ky + 1, ncc2 = new_cc ? cc + 1 : cc;
            if (new_cc && !last_unit) stage_load(cc + 1);
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                f32x4 an[RT], bn[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bn[ct] = bc[ct];
                if (kx + 1 < KS) {
                    load_b(cc, ky * KS + kx + 1, bn);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        an[rt] = *reinterpret_cast<const f32x4*>(a_ptr[rt] + ky * wp_f + (kx + 1) * PITCH);
                } else if (!last_unit) {
                    load_b(ncc2, nky * KS, bn);
                    if (!new_cc) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) an[rt] = *reinterpret_cast<const f32x4*>(a_ptr[rt] + nky * wp_f);
                    }
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[rt][j], bc[ct][j], acc[rt][ct], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) bc[ct] = bn[ct];
                if (kx + 1 < KS || (!last_unit && !new_cc)) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) ac[rt] = an[rt];
                } else if (!last_unit) {
                    __syncthreads();           // every wave is past its last read of this chunk's halo
                    stage_store();
                    __syncthreads();
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) ac[rt] = *reinterpret_cast<const f32x4*>(a_ptr[rt]);
                }
            }
            ky = nky; cc = ncc2;
        }
        // this piece's raw partial tile, [BM][BN] in its (tile, seg) slot of the slab
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    pslab[(wrow + rt * 16 + q * 4 + r) * BN + wcol + ct * 16 + lr] = acc[rt][ct][r];
        continue;
    } else {
    f32x4 ac[RT], bc[CT];
    stage_load(0);
    load_b(0, 0, bc);
    stage_store();
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) ac[rt] = *reinterpret_cast<const f32x4*>(a_ptr[rt]);

    for (int cc = 0; cc < ncc; ++cc) {
        const bool more = cc + 1 < ncc;
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
            f32x4 an[RT], bn[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bn[ct] = bc[ct];
            if (tap + 1 < T) {
                load_b(cc, tap + 1, bn);
                const int ky = (tap + 1) / KS, kx = (tap + 1) % KS;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    an[rt] = *reinterpret_cast<const f32x4*>(a_ptr[rt] + ky * wp_f + kx * PITCH);
            } else if (more) {
                load_b(cc + 1, 0, bn);
            }
            if (tap == T - 2 && more) stage_load(cc + 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[rt][j], bc[ct][j], acc[rt][ct], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bc[ct] = bn[ct];
            if (tap + 1 < T) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) ac[rt] = an[rt];
            } else if (more) {
                // (Measured, not adopted: the refill IN FRONT of the last tap's MFMAs -- LDS-counter-only barriers, staged rows
                // fetched five taps ahead -- so that stores, barrier and the next chunk's first fragment reads overlap that burst:
                // no change on any shape (64->64 k3 @101x40 119.5 / 120.6 vs 119.4 / 123.4 TFLOP/s, k5 137.3 vs 138.1): the
                // co-resident workgroups already cover the refill.  Bisect of the isolated launch (64->64 @101x40, 2 020 tiles):
                // one channel chunk costs 0.090 ms (k5) / 0.035 ms (k3) = 147 / 136 TFLOP/s for the main loop alone; the prologue
                // 8-10 us per launch; the epilogue 17-18 us = the 66 MB output written at 3.7 TB/s in bursts, because the
                // workgroups of a round finish together -- 5 % of a k5 launch, 10 % of a k3 launch when it runs alone.)
                __syncthreads();               // every wave is past its last read of this chunk's halo
                stage_store();
                __syncthreads();
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) ac[rt] = *reinterpret_cast<const f32x4*>(a_ptr[rt]);
            }
        }
    }

    }

    // epilogue: identical to igemm_fwd_kernel's un-split path (C/D map of 16x16x4: col = lane&15, row = 4*(lane>>4) + reg)
    const int N = g.Cout;
    float csum[CT], csq[CT], bias_v[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        csum[ct] = 0.f; csq[ct] = 0.f;
        const int col = n0 + wcol + ct * 16 + lr;
        bias_v[ct] = (e.bias && col < N) ? e.bias[col] : 0.f;
    }
    // Two straight-line forms for whole tiles -- the forward launch (bias, ReLU, statistics) and the dgrad launch (ReLU-backward
    // mask, residual accumulate) -- and the general form below for ragged last tiles and everything else.  A per-tile epilogue
    // instruction takes issue time from the co-resident workgroups' MFMAs like any other: the general form spends ~15 vector /
    // scalar instructions per element on guards, option branches and 64-bit addresses and waits for each mask load separately;
    // the straight-line forms address with one per-lane byte offset + a scalar row term + an immediate column term (buffer
    // instructions), batch a 16-row group's loads, and carry no per-element branch.  Same arithmetic in the same order.
    const bool whole = m0 + BM <= g.M && (N % BN) == 0;
    const uint32_t e_lane = (uint32_t)((m0 + wrow + q * 4) * N + n0 + wcol + lr) * 4u;     // element (rt 0, r 0, ct 0) of this lane, bytes
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, g.M * N * 4, 0x00020000);
    if (whole && !e.dropout && !e.mask && !e.accumulate) {
        auto store_all = [&](auto relu_c, auto stats_c) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int soff = (rt * 16 + r) * N * 4;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        float v = acc[rt][ct][r] + bias_v[ct];
                        if constexpr (decltype(relu_c)::value) v = fmaxf(v, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)(e_lane + ct * 64), soff, 0);
                        if constexpr (decltype(stats_c)::value) { csum[ct] += v; csq[ct] += v * v; }
                    }
                }
        };
        using T1 = std::integral_constant<bool, true>;
        using T0 = std::integral_constant<bool, false>;
        if (e.relu) { if (e.stats) store_all(T1{}, T1{}); else store_all(T1{}, T0{}); }
        else        { if (e.stats) store_all(T0{}, T1{}); else store_all(T0{}, T0{}); }
    } else if (whole && e.mask && !e.dropout && !e.relu && !e.bias && !e.stats) {
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.mask), 0, g.M * N * 4, 0x00020000);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float mk[4][CT], old[4][CT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int soff = (rt * 16 + r) * N * 4;
                    mk[r][ct] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(mrs, (int)(e_lane + ct * 64), soff, 0));
                    old[r][ct] = e.accumulate ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yrs, (int)(e_lane + ct * 64), soff, 0)) : 0.f;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int soff = (rt * 16 + r) * N * 4;
                    float v = (mk[r][ct] > 0.f) ? acc[rt][ct][r] * e.mask_scale : 0.f;
                    if (e.accumulate) v += old[r][ct];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)(e_lane + ct * 64), soff, 0);
                }
        }
    } else {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wrow + rt * 16 + q * 4 + r;
            if (row >= g.M) continue;
            const size_t rbase = (size_t)row * N;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col = n0 + wcol + ct * 16 + lr;
                if (col >= N) continue;
                float v = acc[rt][ct][r] + bias_v[ct];
                if (e.relu) v = fmaxf(v, 0.f);
                const size_t off = rbase + col;
                if (e.dropout) {
                    uint32_t u24 = fmix32(e.drop_prefix ^ (uint32_t)((size_t)row * N + col)) >> 8;
                    v = (u24 >= e.drop_thr) ? v * e.drop_scale : 0.f;
                }
                if (e.mask) v = (e.mask[off] > 0.f) ? v * e.mask_scale : 0.f;
                if (e.accumulate) v += Y[off];
                Y[off] = v;
                csum[ct] += v;
                csq[ct] += v * v;
            }
        }
    }
    }
    if (e.stats) {
        __syncthreads();                       // the halo image is dead: reuse it for the cross-wave column sums
        float* red = halo;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            csum[ct] += __shfl_xor(csum[ct], 16, 64);
            csq[ct] += __shfl_xor(csq[ct], 16, 64);
            csum[ct] += __shfl_xor(csum[ct], 32, 64);
            csq[ct] += __shfl_xor(csq[ct], 32, 64);
            if (q == 0) {
                const int c = wcol + ct * 16 + lr;
                red[((wave / WN) * BN + c) * 2] = csum[ct];
                red[((wave / WN) * BN + c) * 2 + 1] = csq[ct];
            }
        }
        __syncthreads();
        if (t < BN && n0 + t < N) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += red[(w * BN + t) * 2]; b += red[(w * BN + t) * 2 + 1]; }
            e.stats[((size_t)mtile * 2) * N + n0 + t] = a;
            e.stats[((size_t)mtile * 2 + 1) * N + n0 + t] = b;
        }
    }
    }   // piece loop
}

// combine the split-K slabs in fixed order and apply the epilogue (VEC = 4 when N % 4 == 0)
struct BalDev { int bm, bn, nt, chunks, L, segmax; };   // balanced K partition: tile shape, N tiles, chunks per tile, units per workgroup

template <int VEC>
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ P, float* __restrict__ Y, GeomDev g,
                                                             EpiDev e, int splits, BalDev bal) {
    const int N = g.Cout;
    const size_t total = (size_t)g.M * N;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < total; i += (size_t)gridDim.x * 256 * VEC) {
        const int row = (int)(i / N), col = (int)(i - (size_t)row * N);
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = 0.f;
        if (bal.L > 0) {   // pieces of this element's tile, in K order (fixed order -> deterministic)
            const int mt = row / bal.bm, ntl = col / bal.bn;
            const int tile = mt * bal.nt + ntl;
            const int first = (tile * bal.chunks) / bal.L;
            const int nseg = ((tile + 1) * bal.chunks + bal.L - 1) / bal.L - first;
            const float* src = P + (size_t)tile * bal.segmax * (bal.bm * bal.bn) + (size_t)(row - mt * bal.bm) * bal.bn + (col - ntl * bal.bn);
            for (int z = 0; z < nseg; ++z) {
                if constexpr (VEC == 4) {
                    const f32x4 p = *reinterpret_cast<const f32x4*>(src + (size_t)z * (bal.bm * bal.bn));
                    v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
                } else {
                    v[0] += src[(size_t)z * (bal.bm * bal.bn)];
                }
            }
        } else
        for (int z = 0; z < splits; ++z) {
            if constexpr (VEC == 4) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(P + (size_t)z * total + i);
                v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
            } else {
                v[0] += P[(size_t)z * total + i];
            }
        }
        size_t off = i;
        if (e.out_stride != 1) {
            int b = fastdiv(row, g.ohw_magic, g.ohw_shift), rr = row - b * g.OHW;
            int oh = fastdiv(rr, g.ow_magic, g.ow_shift), ow = rr - oh * g.OW;
            off = ((size_t)(b * e.OHf + oh * e.out_stride) * e.OWf + ow * e.out_stride) * N + col;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float x = v[j];
            if (e.bias) x += e.bias[col + j];
            if (e.relu) x = fmaxf(x, 0.f);
            if (e.dropout) {
                uint32_t u24 = fmix32(e.drop_prefix ^ (uint32_t)(i + j)) >> 8;
                x = (u24 >= e.drop_thr) ? x * e.drop_scale : 0.f;
            }
            if (e.mask) x = (e.mask[off + j] > 0.f) ? x * e.mask_scale : 0.f;
            if (e.accumulate) x += Y[off + j];
            v[j] = x;
        }
        if constexpr (VEC == 4) *reinterpret_cast<f32x4*>(Y + off) = f32x4{v[0], v[1], v[2], v[3]};
        else Y[off] = v[0];
    }
}

template <int BM, int BN, int BK, int WM, int MODE = 0>
static void launch_fwd_t(const float* X, const float* Wt, float* Y, const GeomDev& g, const EpiDev& e_in, hipStream_t s,
                         const GemmTiming* tm, float* slab, int splits, int balanced_wgs = 0, const uint2* rowtab = nullptr,
                         int tab_rows = 0) {
    EpiDev e = e_in;
    dim3 grid(cdiv(g.M, BM), cdiv(g.Cout, BN), splits);
    const int nchunks = cdiv(g.K, BK);
    int cps = cdiv(nchunks, splits);
    float* sl = splits > 1 ? slab : nullptr;
    BalDev bal{BM, BN, cdiv(g.Cout, BN), nchunks, 0, 0};
    if (balanced_wgs > 0) {   // balanced K partition: every workgroup gets L chunk units (see the kernel)
        const long units = (long)grid.x * grid.y * nchunks;
        bal.L = (int)((units + balanced_wgs - 1) / balanced_wgs);
        bal.segmax = cdiv(nchunks, bal.L) + 1;
        e.bal_L = bal.L; e.bal_segmax = bal.segmax;
        grid = dim3((unsigned)((units + bal.L - 1) / bal.L), 1, 1);
        sl = slab;
        splits = 2;            // any value > 1: take the combine path below
    }
    if (tm && tm->start && tm->ext) {
        hipExtLaunchKernelGGL((igemm_fwd_kernel<BM, BN, BK, WM, MODE>), grid, dim3(256), 0, s, tm->start, tm->stop, 0, X, Wt, Y, g,
                              e, sl, cps, rowtab, tab_rows);
    } else {
        if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->start, s));
        hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, BK, WM, MODE>), grid, dim3(256), 0, s, X, Wt, Y, g, e, sl, cps, rowtab, tab_rows);
        if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->stop, s));
    }
    CMOOP_HIP(hipGetLastError());
    if (splits > 1) {
        const size_t total = (size_t)g.M * g.Cout;
        if (g.Cout % 4 == 0) {
            const unsigned gridc = (unsigned)std::min<size_t>((total / 4 + 255) / 256, 4096);
            hipLaunchKernelGGL(splitk_combine_kernel<4>, dim3(gridc), dim3(256), 0, s, slab, Y, g, e, splits, bal);
        } else {
            const unsigned gridc = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
            hipLaunchKernelGGL(splitk_combine_kernel<1>, dim3(gridc), dim3(256), 0, s, slab, Y, g, e, splits, bal);
        }
        CMOOP_HIP(hipGetLastError());
    }
}

// Pick a multiplier m in [1, max_mult] for `base` workgroups so that base*m fills whole waves of
// `resident` co-resident workgroups (256 CUs x workgroups per CU).  Few-wave grids otherwise pay up to 2x for a
// ragged tail.
static int fill_waves(long base, int max_mult, int resident, double first_thr = 0.85) {
    // the smallest multiplier whose last wave is >= first_thr full; failing that >= 85 %; failing that the best-filled
    // one.  wgrad slices ask for 95 % (a 450-workgroup grid on 512 slots cost the 128->128 k5 wgrad 11 %); forward
    // split-K keeps 85 % because every extra split adds a slab to write and combine.
    for (const double thr : {first_thr, 0.85}) {
        for (int m = 1; m <= max_mult; ++m) {
            const long blocks = base * m;
            const long waves = (blocks + resident - 1) / resident;
            const double u = (double)blocks / (double)(waves * resident);
            if (blocks >= resident / 2 && u >= thr) return m;
        }
    }
    int best = 1;
    double best_u = 0.0;
    for (int m = 1; m <= max_mult; ++m) {
        const long blocks = base * m;
        const long waves = (blocks + resident - 1) / resident;
        const double u = (double)blocks / (double)(waves * resident);
        if (u > best_u) { best_u = u; best = m; }
    }
    return best;
}

// Tile / split choice.  Prefer the big (most efficient) tile; if its grid does not cover the chip
// (deep layers: 13x5 or 26x10 pixels x 64 samples) split the long K axis across blockIdx.z when a
// slab workspace is available, else fall back to 64-row tiles.
// workgroups co-resident on the chip for the 128-row, 32-deep instantiations (LDS-limited: 2 per CU at 64/128 columns, 3 below)
static inline int resident_wgs_128(int bn) { return 256 * (bn >= 64 ? 2 : 3); }

// balanced K partition of an under-filled launch (see igemm_fwd_kernel): slab floats it needs, 0 if not applicable
static size_t balanced_slab_floats(int M, int N, int K, int bk, int bn, int* wgs_out) {
    if (bk != 32) return 0;
    const long tiles = (long)cdiv(M, 128) * cdiv(N, bn);
    // CMOOP_BALANCED_MULT (1, 2, 4): workgroups of a balanced launch in multiples of the co-resident count -- more, shorter
    // workgroups interleave better with the other candidates' kernels, at the price of more partial-tile slots
    static const int mult = [] { const char* v = std::getenv("CMOOP_BALANCED_MULT"); const int m = v ? std::atoi(v) : 1; return (m == 2 || m == 4) ? m : 1; }();
    const int chunks = cdiv(K, 32), W = resident_wgs_128(bn) * mult;
    if (tiles > W || tiles * chunks < 8l * W) return 0;          // at most two pieces per workgroup; at least 8 chunks each
    const int L = (int)((tiles * chunks + W - 1) / W);
    if (wgs_out) *wgs_out = W;
    return (size_t)tiles * (cdiv(chunks, L) + 1) * 128 * bn;
}

static void pick_tile(int M, int N, int K, int bk, size_t ws_floats, int* bm, int* bn, int* splits, int* balanced_wgs) {
    const int bn_small = N <= 16 ? 16 : (N <= 32 ? 32 : 64);
    const int bn_big = N > 64 ? 128 : bn_small;
    *splits = 1;
    *balanced_wgs = 0;
    long blocks = (long)cdiv(M, 128) * cdiv(N, bn_big);
    const long fill = std::max(32l, (long)(384 * par_scale()));
    // (Measured twice, not adopted: 256 x 64 tiles with 16-deep chunks for the 64-column layers at 101x40 -- every wave owns
    // 64 x 64 like the 128 x 128 kernel, two workgroups per CU: forward / dgrad 111.9 / 111.7 vs 122.6 / 122.6 TFLOP/s on
    // 64->64 k5, 95.8 / 97.7 vs 104.9 / 113.9 on k3; 2 155 vs 2 179 evals/h in the job.  Round 2's 32-deep form: 109.6 vs 118.5.)
    if (blocks >= fill) { *bm = 128; *bn = bn_big; return; }
    const int nchunks = cdiv(K, bk);
    // Balanced K partition (default since round 3; CMOOP_BALANCED=0 restores the uniform split): +11..26 % on the 26x10 /
    // 13x5 layer shapes with the launch alone on the chip (256->256 k5 @26x10: 99 -> 127 TFLOP/s; the heaviest bench
    // candidate alone 8.46 -> 7.99 ms per step).  Round 2 measured it 1.3 % SLOWER whole-job with eight candidates in flight
    // and kept it opt-in; re-measured in round 3 on the same box, same minute: 2 173.6 vs 2 167.3 evals/h (+0.3 %, noise
    // level) -- no longer a loss, and the lone / tail case is a clear win.  More, shorter workgroups (CMOOP_BALANCED_MULT
    // 2 / 4: 1 024 / 2 048 instead of 512) lose both alone (8.07 / 8.22 ms) and in the job (2 165.9 / 2 146.3).
    static const bool balanced_on = [] { const char* v = std::getenv("CMOOP_BALANCED"); return !(v && v[0] == '0'); }();
    if (balanced_on) {
        int W = 0;
        const size_t need = balanced_slab_floats(M, N, K, bk, bn_big, &W);
        if (need > 0 && need <= ws_floats) { *bm = 128; *bn = bn_big; *balanced_wgs = W; return; }
    }
    if (nchunks >= (M <= 512 ? 4 : 16) && ws_floats > 0) {
        // dense layers (M = batch rows) are a serial latency chain of K chunks: split them finely
        const int min_chunks = M <= 512 ? 2 : 8;
        int sp = (int)std::min<long>(cdiv((int)(1024 * par_scale()), (int)blocks), nchunks / min_chunks);
        sp = std::min(sp, 32);
        if (sp > 1) sp = fill_waves(blocks, sp, std::max(32, (int)(512 * par_scale())));
        while (sp > 1 && (size_t)sp * M * N > ws_floats) --sp;
        if (sp > 1) {
            const int cps = cdiv(nchunks, sp);
            sp = cdiv(nchunks, cps);          // no empty split
            *bm = 128; *bn = bn_big; *splits = sp;
            return;
        }
    }
    blocks = (long)cdiv(M, 128) * cdiv(N, bn_small);
    if (blocks >= fill) { *bm = 128; *bn = bn_small; return; }
    *bm = 64; *bn = bn_small;
}

size_t igemm_splitk_workspace(const ConvGeom& g) {
    // upper bound of what pick_tile may ask for: splits * M * N with splits <= 768 / blocks + 1
    const int M = g.M(), N = g.Cout;
    const int bn_big = N > 64 ? 128 : (N <= 16 ? 16 : (N <= 32 ? 32 : 64));
    const long blocks = (long)cdiv(M, 128) * cdiv(N, bn_big);
    if (blocks >= 384) return 0;
    const int sp = std::min(32, cdiv(1024, (int)blocks));
    return std::max(std::max((size_t)sp * M * N, balanced_slab_floats(M, N, g.K(), g.Cin % 32 == 0 ? 32 : 16, bn_big, nullptr)),
                    halo_balanced_slab_floats(g, nullptr));
}

// balanced unit partition of the halo kernel on an under-filled grid (see halo_fwd_kernel): units per workgroup for one round
// of 512 workgroups, slab floats it needs; 0 if the geometry does not qualify
static size_t halo_balanced_slab_floats(const ConvGeom& cg, int* L_out) {
    // default since round 3 (CMOOP_HALO_BAL=0 restores the implicit GEMM's balanced K partition).  Isolated forward / dgrad TFLOP/s,
    // halo vs implicit GEMM: 256->256 k5 @26x10 133.3 / 134.1 vs 127.4 / 128.5, 512->512 k5 @13x5 132.2 vs 126.4, 128->256 k5 125.5 vs
    // 119.6, 128->256 k3 @13x5 65.5 vs 48.8, the k3 layers +1..4 %; the pop-40 job 2 282.0 vs 2 258.3 evals/h.
    static const bool on = [] { const char* v = std::getenv("CMOOP_HALO_BAL"); return !(v && v[0] == '0'); }();
    if (!on || cg.Cout % 64 != 0 || !halo_geometry(cg, nullptr, nullptr)) return 0;
    const long tiles = (long)cdiv(cg.M(), 128) * (cg.Cout / 64);
    const int upt = (cg.Cin / 16) * cg.KH;
    const long units = tiles * upt;
    if (tiles >= 1024 || units < 512l * 2 * cg.KH) return 0;          // a filled grid runs un-split; at least two chunks' worth per workgroup
    const int L = (int)((units + 511) / 512);
    if (L_out) *L_out = L;
    return (size_t)tiles * (cdiv(upt, L) + 1) * 128 * 64;
}

// tile / split / operand-path choice of a forward-type launch: pure host arithmetic on the geometry (no HIP call), shared
// by the launcher and by the host-only launch plan the parity-coverage tests read (igemm_fwd_plan)
struct FwdChoice {
    int mode, bm, bn, splits, balanced_wgs, flags;
    bool bk32_tile, use_dma, stats, halo;
    int ks, halo_L;          // halo_L > 0: balanced unit partition of the halo kernel (units per workgroup)
    // the halo kernel carries its window size in the chunk-depth field
    int code() const { return mode * 100000000 + bm * 100000 + bn * 100 + (halo ? ks + (halo_L > 0 ? 50 : 0) : (bk32_tile ? 32 : 16)); }
};

// halo-tiled direct convolution (halo_fwd_kernel): geometry it accepts and the LDS image it needs
static inline int halo_bm(int cout) { return cout >= 64 ? 128 : 256; }   // narrow layers: every wave still owns 64 pixels x 16 / 32 columns
static bool halo_geometry(const ConvGeom& cg, HaloDev* hd, size_t* lds_bytes) {
    const int ks = cg.KH, R = ks / 2;
    if (cg.KH != cg.KW || (ks != 3 && ks != 5) || cg.stride != 1 || cg.OH != cg.H || cg.OW != cg.W) return false;
    if (cg.pad_t != R || cg.pad_l != R || cg.Cin % 16 != 0) return false;
    if (cg.Cout % 64 != 0 && cg.Cout != 32 && cg.Cout != 16) return false;
    const int bm = halo_bm(cg.Cout);
    const int wp = (cg.W + ks - 1 + 7) / 8 * 8;
    // virtual rows a flat bm-pixel tile can span: its pixel rows, plus R gap rows per image boundary it crosses
    const int span = (bm - 2 + cg.W) / cg.W + 1 + ((bm - 1) / (cg.H * cg.W) + 1) * R;
    const int rows = span + 2 * R;
    if (rows * wp * 4 > halo_nst(bm) * 256) return false;
    const size_t lds = (size_t)rows * wp * 24 * sizeof(float);
    if (lds > 64 * 1024) return false;
    if (hd) {
        hd->WP = wp; hd->rows_max = rows; hd->VH = cg.H + R;
        fastdiv_init(wp, &hd->wp_magic, &hd->wp_shift);
        fastdiv_init(cg.H + R, &hd->vh_magic, &hd->vh_shift);
    }
    if (lds_bytes) *lds_bytes = lds;
    return true;
}

// host-only check of the bound above (tests): halo rows the LDS image is sized for and the largest number of rows any tile of
// this geometry really spans (the kernel's own arithmetic: virtual rows of the tile's first and last pixel); 0 / 0 when the
// geometry is not eligible
void halo_rows_bound_and_need(const ConvGeom& cg, int* bound, int* need, int* items_cap) {
    HaloDev hd;
    size_t lds = 0;
    *bound = *need = *items_cap = 0;
    if (!halo_geometry(cg, &hd, &lds)) return;
    const int R = cg.KH / 2, bm = halo_bm(cg.Cout), M = cg.M(), VH = cg.H + R;
    auto vrow = [&](int m) { const int b = m / (cg.H * cg.W), r = m - b * cg.H * cg.W; return b * VH + r / cg.W; };
    int worst = 0;
    for (int m0 = 0; m0 < M; m0 += bm) worst = std::max(worst, vrow(std::min(m0 + bm, M) - 1) - vrow(m0) + 1 + 2 * R);
    *bound = hd.rows_max; *need = worst; *items_cap = halo_nst(bm) * 256 / (hd.WP * 4);
}

static FwdChoice choose_fwd(const ConvGeom& cg, const GemmEpilogue& ep, size_t ws_floats, bool want_stats, bool have_rowtab) {
    FwdChoice c;
    const int M = cg.M(), N = cg.Cout, K = cg.K();
    c.mode = resolve_mode(ep.mode);
    // bf16 planes always use 32-deep chunks (a chunk may span two taps of a 16-channel layer; the per-thread tap decode
    // and the k < K guard handle that)
    const bool bk32 = c.mode != GEMM_FP32 || (cg.Cin % 32 == 0);
    pick_tile(M, N, K, bk32 ? 32 : 16, ws_floats, &c.bm, &c.bn, &c.splits, &c.balanced_wgs);
    // fused column statistics only on un-split launches (split-K partials are raw sums; the caller falls back to the
    // stand-alone reduction there: those are the small 13x5 / 26x10 layers)
    c.stats = want_stats && c.splits == 1 && !c.balanced_wgs && ep.out_stride == 1 && !ep.accumulate;
    // many-wave grids of 128x64 tiles run 3-4 % faster with 16-deep K chunks (half the LDS, 4-5 workgroups
    // per CU instead of 2); single-wave grids prefer the 32-deep chunk (half the barriers)
    // (measured: 16-deep chunks for the 128x128 tile -- three workgroups per CU -- 124 vs 130 TFLOP/s: not used)
    // (measured again with the r2 loader: 32-deep chunks on the >= 1024-tile 128x64 grids 119 vs 123 TFLOP/s: 16 stays)
    c.bk32_tile = c.mode != GEMM_FP32 ||
                  (bk32 && !(c.bm == 128 && c.bn == 64 && c.splits == 1 && !c.balanced_wgs && (long)cdiv(M, 128) * cdiv(N, 64) >= 1024));
    // LDS-DMA operand path: needs the chunk-in-one-tap condition of the fast loader and <= 32 taps.  Measured per tile
    // (isolated, forward / dgrad TFLOP/s, DMA vs register staging): 128x32 (32-column layers @101x40) 101 / 109 vs 90 / 95
    // -> used; 128x128 130 vs 132, 128x64 @51x20 111 vs 116, 64->128 dgrad 116 vs 123 -> not used (the DMA fill rate per
    // wave is the limit once a tile needs four or more 1 KiB fills per operand and chunk).
    static const int dma_env = [] { const char* v = std::getenv("CMOOP_DMA"); return v ? std::atoi(v) : 1; }();
    c.use_dma = dma_env && c.mode == GEMM_FP32 && (cg.Cin % 32 == 0) && cg.KH * cg.KW <= 32 && c.bk32_tile && c.bm == 128 && c.bn == 32;
    if (c.use_dma) c.mode = GEMM_FP32_DMA;      // the instantiation's MODE parameter (rocprofv3 prints it)
    // (Measured, not adopted, round 3: the many-wave 128x64 grids on 32-deep chunks with a SINGLE LDS image -- the trick that
    // helped the <= 64-channel weight gradient: forward 114.5 vs 123.0 TFLOP/s on 64->64 k5 @101x40, dgrad 105.7 vs 114.1 on k3.)
    const int bk = c.bk32_tile ? 32 : 16;
    // halo-tiled direct convolution for the un-split 128-row launches it accepts (default since round 3; CMOOP_HALO=0 restores
    // the implicit GEMM everywhere).  Isolated forward / dgrad TFLOP/s, halo vs implicit GEMM (profiles/r03_halo_kernel_bench.txt):
    // 64->64 k5 @101x40 137.5 vs 122.1, k3 113 / 122 vs 105 / 114; 64->64 k5 @51x20 132.6 vs 117; 128->64 k5 (dgrad of 64->128)
    // 139.2 vs 122.9; 32->32 k5 @101x40 119.4 vs 110 (LDS-DMA tile), @51x20 107 vs 93; 16->16 k5 90.4 vs 66.7, k3 51 vs 46.
    // (32->32 k3 @101x40 lost at first -- 78.7 vs 87.2 on the LDS-DMA tile -- and was excluded until the straight-line epilogue: 96.6 / 99.2 now.)
    // With the straight-line epilogue forms (see the kernel): 64->64 k5 @101x40 143 / 144, k3 133 / 133, 128->128 k5 @51x20 144 / 145,
    // 32->32 k5 129, 16->16 k5 104, 16->16 k3 68; the pop-40 job 2 337 vs 2 289-2 297 evals/h.
    // 128-column layers run as two 64-column workgroups per tile: alone the second round of workgroups has a ragged tail
    // (128->128 k5 @51x20: 117 vs 130; a 128 x 128 halo tile did 136), in the job it is the better form (2 225 vs 2 195 vs
    // 2 180 evals/h for 64-column halo / 128-column halo / implicit GEMM).
    static const bool halo_env = [] { const char* v = std::getenv("CMOOP_HALO"); return !(v && v[0] == '0'); }();
    c.ks = cg.KH;
    c.halo = halo_env && (c.mode == GEMM_FP32 || c.mode == GEMM_FP32_DMA) && c.bm == 128 && c.splits == 1 && !c.balanced_wgs && ep.out_stride == 1 &&
             halo_geometry(cg, nullptr, nullptr);
    c.halo_L = 0;
    if (c.halo) {
        c.mode = GEMM_FP32_HALO; c.use_dma = false;
        c.bm = halo_bm(cg.Cout);
        c.bn = std::min(cg.Cout, 64);
    } else if (halo_env && (c.mode == GEMM_FP32 || c.mode == GEMM_FP32_DMA) && c.bm == 128 && (c.splits > 1 || c.balanced_wgs) && ep.out_stride == 1) {
        int L = 0;
        const size_t need = halo_balanced_slab_floats(cg, &L);
        if (need > 0 && need <= ws_floats) {
            c.halo = true; c.halo_L = L; c.mode = GEMM_FP32_HALO; c.use_dma = false; c.stats = false;
            c.bm = 128; c.bn = 64; c.splits = 1; c.balanced_wgs = 512;
        }
    }
    c.flags = (c.splits > 1 ? GEMM_FLAG_SPLITK : 0) | (c.stats ? GEMM_FLAG_STATS : 0) | (c.balanced_wgs ? GEMM_FLAG_BALANCED : 0) |
              ((!c.halo && have_rowtab && (cg.Cin % bk) == 0 && cg.KH * cg.KW <= 32) ? GEMM_FLAG_ROWTAB : 0);
    return c;
}

int igemm_fwd_plan(const ConvGeom& g, const GemmEpilogue& ep, size_t ws_floats, bool want_stats, bool have_rowtab, int* flags_out) {
    igemm_check_range(g);
    CMOOP_REQUIRE(ilog2_exact(g.Cin) >= 4, "implicit GEMM needs C_in a power of two >= 16");
    const FwdChoice c = choose_fwd(g, ep, ws_floats, want_stats, have_rowtab);
    if (flags_out) *flags_out = c.flags;
    return c.code();
}

int launch_igemm_fwd(const float* X, const float* Wt, float* Y, const ConvGeom& cg, const GemmEpilogue& ep,
                     hipStream_t s, const GemmTiming* tm, float* splitk_ws, size_t splitk_ws_floats, int* stats_blocks,
                     const void* rowtab_v, int tab_rows, int* flags_out) {
    if (stats_blocks) *stats_blocks = 0;
    if (flags_out) *flags_out = 0;
    const uint2* rowtab = static_cast<const uint2*>(rowtab_v);
    GeomDev g = to_dev(cg);
    if (g.M == 0) return 0;
    EpiDev e;
    e.bias = ep.bias; e.mask = ep.mask; e.mask_scale = ep.mask_scale; e.relu = ep.relu;
    e.accumulate = ep.accumulate; e.out_stride = ep.out_stride; e.OHf = ep.OHf; e.OWf = ep.OWf;
    e.dropout = ep.dropout; e.drop_prefix = ep.drop_prefix; e.drop_thr = ep.drop_thr; e.drop_scale = ep.drop_scale;
    if (e.out_stride > 1)
        CMOOP_REQUIRE((int64_t)g.B * e.OHf * e.OWf * g.Cout < (1ll << 29), "scattered output too large");
    const FwdChoice ch = choose_fwd(cg, ep, splitk_ws ? splitk_ws_floats : 0, ep.stats != nullptr && stats_blocks != nullptr, rowtab != nullptr);
    const int mode = ch.mode == GEMM_FP32_DMA ? (int)GEMM_FP32 : ch.mode;
    const int bm = ch.bm, bn = ch.bn, splits = ch.splits, balanced_wgs = ch.balanced_wgs;
    const bool bk32_tile = ch.bk32_tile, use_dma = ch.use_dma;
    e.bal_L = 0; e.bal_segmax = 0;
    e.stats = ch.stats ? ep.stats : nullptr;
    if (e.stats) *stats_blocks = cdiv(g.M, bm);
    if (flags_out) *flags_out = ch.flags;
    if (ch.halo) {
        HaloDev hd;
        size_t lds = 0;
        CMOOP_REQUIRE(halo_geometry(cg, &hd, &lds), "halo kernel chosen for a geometry it does not accept");
        dim3 grid(cdiv(g.M, bm), cdiv(g.Cout, bn));
        BalDev bal{bm, bn, cdiv(g.Cout, bn), (cg.Cin / 16) * cg.KH, ch.halo_L, 0};
        if (ch.halo_L > 0) {
            bal.segmax = cdiv(bal.chunks, bal.L) + 1;
            e.bal_L = bal.L; e.bal_segmax = bal.segmax;
            const long units = (long)grid.x * grid.y * bal.chunks;
            grid = dim3((unsigned)((units + bal.L - 1) / bal.L), 1, 1);
            lds = std::max(lds, (size_t)55 * 1024);                   // one round of two workgroups per CU
        }
        // Workgroups per CU: the kernel's registers allow three, its LDS image usually too.  A grid of one to three rounds
        // is quantised by the slot count -- 1 020 workgroups (128->128 @51x20 as two 64-column halves) are 1.33 rounds of 768
        // slots but 1.99 rounds of 512 -- so the LDS request is padded past a third of the CU's 160 KiB when two per CU
        // fill their last round better (CMOOP_HALO_OCC=2 / 3 forces either).  Isolated: 128->128 k5 @51x20 116 -> 140.8 TFLOP/s,
        // k3 106 -> 124, 64->128 k5 113 -> 133; grids of more than two rounds keep three per CU (64->64 k5 @101x40, 2 020
        // workgroups: 137.5 vs 133.7 with two).
        {
            static const int occ_env = [] { const char* v = std::getenv("CMOOP_HALO_OCC"); return v ? std::atoi(v) : 0; }();
            const long wgs = (long)grid.x * grid.y;
            auto fill = [&](long slots) { return (double)wgs / (double)(((wgs + slots - 1) / slots) * slots); };
            const bool two = occ_env == 2 || (occ_env != 3 && lds <= 53 * 1024 && wgs <= 1536 && fill(512) > fill(768) + 0.05);
            if (two) lds = std::max(lds, (size_t)55 * 1024);
        }
#define CMOOP_HALO_LAUNCH(KS_, BM_, BN_, WM_, BAL_)                                                                                        \
        do {                                                                                                                 \
            if (tm && tm->start && tm->ext) {                                                                                \
                hipExtLaunchKernelGGL((halo_fwd_kernel<KS_, BM_, BN_, WM_, BAL_>), grid, dim3(256), lds, s, tm->start, tm->stop, 0, X, Wt, Y, g, e, hd, splitk_ws); \
            } else {                                                                                                         \
                if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->start, s));                                                \
                hipLaunchKernelGGL((halo_fwd_kernel<KS_, BM_, BN_, WM_, BAL_>), grid, dim3(256), lds, s, X, Wt, Y, g, e, hd, splitk_ws);    \
                if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->stop, s));                                                 \
            }                                                                                                                \
        } while (0)
#define CMOOP_HALO_KS(KS_)                                                                     \
        do {                                                                                   \
            if (ch.halo_L > 0) CMOOP_HALO_LAUNCH(KS_, 128, 64, 2, true);                       \
            else if (bn == 64) CMOOP_HALO_LAUNCH(KS_, 128, 64, 2, false);                      \
            else if (bn == 32) CMOOP_HALO_LAUNCH(KS_, 256, 32, 4, false);                      \
            else CMOOP_HALO_LAUNCH(KS_, 256, 16, 4, false);                                    \
        } while (0)
        if (cg.KH == 5) CMOOP_HALO_KS(5); else CMOOP_HALO_KS(3);
#undef CMOOP_HALO_KS
#undef CMOOP_HALO_LAUNCH
        CMOOP_HIP(hipGetLastError());
        if (ch.halo_L > 0) {     // fixed-order sum of every tile's pieces + the epilogue
            const size_t total = (size_t)g.M * g.Cout;
            const unsigned gridc = (unsigned)std::min<size_t>((total / 4 + 255) / 256, 4096);
            hipLaunchKernelGGL(splitk_combine_kernel<4>, dim3(gridc), dim3(256), 0, s, splitk_ws, Y, g, e, 2, bal);
            CMOOP_HIP(hipGetLastError());
        }
        return ch.code();
    }
#define CMOOP_FWD(BM_, BN_, WM_)                                                              \
    do {                                                                                      \
        if (mode == GEMM_BF16X3) launch_fwd_t<BM_, BN_, 32, WM_, GEMM_BF16X3>(X, Wt, Y, g, e, s, tm, splitk_ws, splits, balanced_wgs, rowtab, tab_rows);   \
        else if (mode == GEMM_BF16) launch_fwd_t<BM_, BN_, 32, WM_, GEMM_BF16>(X, Wt, Y, g, e, s, tm, splitk_ws, splits, balanced_wgs, rowtab, tab_rows);  \
        else if (use_dma && BM_ == 128 && BN_ == 32) launch_fwd_t<BM_, 32, 32, WM_, GEMM_FP32_DMA>(X, Wt, Y, g, e, s, tm, splitk_ws, splits, balanced_wgs, rowtab, tab_rows);   \
        else if (bk32_tile) launch_fwd_t<BM_, BN_, 32, WM_>(X, Wt, Y, g, e, s, tm, splitk_ws, splits, balanced_wgs, rowtab, tab_rows);   \
        else launch_fwd_t<BM_, BN_, 16, WM_>(X, Wt, Y, g, e, s, tm, splitk_ws, splits, 0, rowtab, tab_rows);        \
    } while (0)
    if (bm == 128) {
        if (bn == 128) CMOOP_FWD(128, 128, 2);
        else if (bn == 64) CMOOP_FWD(128, 64, 4);
        else if (bn == 32) CMOOP_FWD(128, 32, 4);
        else CMOOP_FWD(128, 16, 4);
    } else {
        if (bn == 64) CMOOP_FWD(64, 64, 2);
        else if (bn == 32) CMOOP_FWD(64, 32, 4);
        else CMOOP_FWD(64, 16, 4);
    }
#undef CMOOP_FWD
    return ch.code();
}

// ---------------------------------------------------------------------------
// weight-gradient kernel: dWt[N][K] = sum_m dY[m][N] * im2col(X)[m][K]
// The MFMA reduction index is the pixel row m; both operands are read from LDS
// m-major (ds_read_b32, leading dimension == 16 mod 32 -> conflict-free).
// grid = (K tiles of 64, N tiles of BCO, S row-slices); partials P[S][N][K].
// (Measured alternative, not adopted: an XCD-aware 1-D grid that keeps all K tiles of a (co tile, slice) group on one
// XCD cuts the fabric-side reads of the 128->128 k5 launch from 559 MB to 84 MB -- 67 MB algorithmic -- but per-XCD wave
// quantisation (25 K tiles per group on 64 slots) makes it 4-10 % slower alone and 0.6 % slower whole-job: the re-reads
// are served by L2 / Infinity Cache and the kernel is MFMA-bound.)
// (Measured alternative, not adopted: transposing 4x4 blocks in registers on the way into LDS so that fragments are
// ds_read_b128 as in the forward kernel -- conflict-free with pitch 40 + XOR swizzle, coalesced gathers -- ran 8 %
// slower than this m-major image with ds_read_b32 fragments: 95 vs 103 TFLOP/s on 128->128 k5.)
// ---------------------------------------------------------------------------
// Row table of a conv layer (one entry per output pixel row m of the implicit GEMM, padded to a multiple of 32 rows):
//   .x = byte offset of input pixel (b, oh*stride - pad_t, ow*stride - pad_l), channel 0, biased by
//        (pad_t*W + pad_l)*Cin*4 so it is never negative (the buffer descriptor's base is moved back by the same bias)
//   .y = bitmask of the filter taps that fall into the SAME padding for this row (all ones for padding rows >= M)
// It depends on the layer geometry only, so the trainer builds it once per layer; the weight-gradient kernel then
// needs no divisions, no bounds arithmetic and no 64-bit pointer math per gathered element (see load_chunk_fast).
__global__ __launch_bounds__(256) void build_rowtab_kernel(GeomDev g, uint2* __restrict__ tab, int rows_padded) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= rows_padded) return;
    if (m >= g.M) { tab[m] = make_uint2(0u, 0xFFFFFFFFu); return; }
    const int b = fastdiv(m, g.ohw_magic, g.ohw_shift), r = m - b * g.OHW;
    const int oh = fastdiv(r, g.ow_magic, g.ow_shift), ow = r - oh * g.OW;
    const int ih0 = oh * g.stride - g.pad_t, iw0 = ow * g.stride - g.pad_l;
    const int hlo = max(0, -ih0), hhi = min(g.KH, g.H - ih0);
    const int wlo = max(0, -iw0), whi = min(g.KW, g.W - iw0);
    const uint32_t wmask = (whi > wlo) ? (((1u << whi) - 1u) & ~((1u << wlo) - 1u)) : 0u;
    uint32_t okm = 0;
    for (int kh = 0; kh < g.KH; ++kh)
        if (kh >= hlo && kh < hhi) okm |= wmask << (kh * g.KW);
    const uint32_t bias = (uint32_t)((g.pad_t * g.W + g.pad_l) << g.cshift);
    tab[m] = make_uint2(((uint32_t)(((b * g.H + ih0) * g.W + iw0) << g.cshift) + bias) * 4u, ~okm);
}

int rowtab_rows(const ConvGeom& g) { return cdiv(g.M(), 256) * 256; }   // whole 256-row tiles: padding rows are marked all-taps-invalid

void launch_build_rowtab(const ConvGeom& cg, void* tab, hipStream_t s) {
    GeomDev g = to_dev(cg);
    CMOOP_REQUIRE(g.KH * g.KW <= 32, "row table: at most 32 filter taps");
    const int rows = rowtab_rows(cg);
    if (rows == 0) return;
    hipLaunchKernelGGL(build_rowtab_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, g, static_cast<uint2*>(tab), rows);
    CMOOP_HIP(hipGetLastError());
}

template <int BCO, int BKI, int MC = 32>   // MC: pixel rows per chunk; MC = 64 runs a SINGLE LDS image (two barriers per chunk)
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                          float* __restrict__ P, GeomDev g, int rows_per_slice,
                                                          float* __restrict__ Pbias, size_t slab_stride,
                                                          const uint2* __restrict__ rowtab, int tab_rows, int grouped_kt,
                                                          int grouped_ct, int grouped_slices) {
    constexpr int NBUF = MC == 32 ? 2 : 1;
    constexpr int LDX = BKI + 16;                          // == 16 mod 32
    constexpr int LDY = (BCO == 16) ? 16 : BCO + 16;
    constexpr int CT = BCO / 16, KT = BKI / 16;             // co tiles, k tiles of the block
    constexpr int WC = CT >= 4 ? 2 : CT;                    // waves along co; 4 / WC along k (2x2 waves for 64- and 128-wide co tiles)
    constexpr int WK = 4 / WC;
    constexpr int CPW = CT / WC, KPW = KT / WK;             // co tiles / k tiles per wave
    constexpr int TPRX = BKI / 4, RPPX = 256 / TPRX, XPASS = MC / RPPX;
    constexpr int TPRY = BCO / 4, RPPY = 256 / TPRY;
    constexpr int YPASS = (MC + RPPY - 1) / RPPY;
    __shared__ __attribute__((aligned(16))) float Xs[NBUF][MC * LDX];
    __shared__ __attribute__((aligned(16))) float Ys[NBUF][MC * LDY];

    const int t = threadIdx.x;
    // XCD-grouped 1-D grid (grouped_kt > 0): workgroups are dealt round-robin over the 8 XCDs, so linear id L runs on
    // XCD L & 7; all K / co tiles of one row slice are given to ONE XCD (slice = 8 * group + xcd) and the slice count is
    // chosen so that each XCD's share fills whole waves of its 64 workgroup slots.  The dY slice and the input rows a
    // slice gathers are then fetched into one L2 instead of eight (fabric reads of 64->64 k5 @101x40: 1.1 GB -> see DESIGN).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (grouped_kt > 0) {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int per = grouped_kt * grouped_ct;
        const int grp = j / per, r = j - grp * per;
        by = r / grouped_kt; bx = r - by * grouped_kt;
        bz = grp * 8 + xcd;
        if (bz >= grouped_slices) return;
    }
    const int k0 = bx * BKI, co0 = by * BCO;
    const int mbeg = bz * rows_per_slice;
    const int mend = min(g.M, mbeg + rows_per_slice);

    // X gather: this thread owns k index k0 + 4*xq for rows xrow + p*RPPX (fixed over the block's life)
    const int xq = t % TPRX, xrow = t / TPRX;
    const int kidx = k0 + 4 * xq;
    const bool kok = kidx < g.K;
    const int tap = kidx >> g.cshift, ci = kidx & (g.Cin - 1);
    const int kh = (tap * g.rcp_kw) >> 16, kw = tap - kh * g.KW;
    const int yq = t % TPRY, yrow = t / TPRY;
    const bool cok = (co0 + 4 * yq) < g.Cout;   // Cout % 4 may be != 0: guarded per element below

    f32x4 rx[XPASS], ry[YPASS];
    // ---- fast gather (row table present): buffer loads, the row's offset / padding mask from the table (fetched one
    // chunk ahead), this thread's fixed tap offset added on the VALU (3 VALU per gathered float4 instead of ~25), dY
    // through a fixed voffset + scalar row offset (0 VALU).  Rows >= M read zeros through the descriptors' range checks.
    const bool fast = rowtab != nullptr && (g.Cout & 3) == 0;
    const uint32_t x_bias = (uint32_t)((g.pad_t * g.W + g.pad_l) << g.cshift);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X) - x_bias, 0, (int)(((uint32_t)g.B * g.H * g.W << g.cshift) + x_bias) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, g.M * g.Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint2*>(rowtab), 0, fast ? tab_rows * 8 : 0, 0x00020000);
    const uint32_t koff = (uint32_t)((((kh * g.W + kw) << g.cshift) + ci) * 4);
    const uint32_t kdead = kok ? 0u : 0xFFFFFFFFu;
    uint32_t y_voff[YPASS];
#pragma unroll
    for (int p = 0; p < YPASS; ++p) {
        const int rl = yrow + p * RPPY;
        y_voff[p] = (rl < MC && cok) ? (uint32_t)(rl * g.Cout + co0 + 4 * yq) * 4u : 0xFFFFFFF0u;
    }
    uint2 te[XPASS];     // table entries of the chunk whose X loads are issued next
    auto load_tab = [&](int mc) {
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(tr, (xrow + p * RPPX) * 8, mc * 8, 0);
            te[p] = make_uint2(v[0], v[1]);
        }
    };
    auto load_chunk_fast = [&](int mc) {
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
            const uint32_t dead = (uint32_t)__builtin_amdgcn_sbfe((int)te[p].y, tap, 1);
            rx[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)((te[p].x + koff) | dead | kdead), 0, 0));
        }
#pragma unroll
        for (int p = 0; p < YPASS; ++p)
            ry[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, (int)y_voff[p], mc * g.Cout * 4, 0));
    };
    auto load_chunk = [&](int mc) {
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
            const int m = mc + xrow + p * RPPX;
            const int mm = m < mend ? m : 0;
            const int b = fastdiv(mm, g.ohw_magic, g.ohw_shift), r = mm - b * g.OHW;
            const int oh = fastdiv(r, g.ow_magic, g.ow_shift), ow = r - oh * g.OW;
            const int ih = oh * g.stride - g.pad_t + kh, iw = ow * g.stride - g.pad_l + kw;
            const bool ok = kok && m < mend && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
            const float* src = ok ? X + (((size_t)((b * g.H + ih) * g.W + iw)) << g.cshift) + ci : g.zeros;
            rx[p] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int p = 0; p < YPASS; ++p) {
            int rl = yrow + p * RPPY;
            int m = mc + rl;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const bool ok = rl < MC && m < mend && cok;
            if ((g.Cout & 3) == 0) {      // wave-uniform: every conv / hidden dense layer
                const float* src = ok ? dY + (size_t)m * g.Cout + co0 + 4 * yq : g.zeros;
                v = *reinterpret_cast<const f32x4*>(src);
            } else if (ok) {              // output layer (10 / 11 / 35 classes)
                const float* src = dY + (size_t)m * g.Cout + co0 + 4 * yq;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (co0 + 4 * yq + j < g.Cout) v[j] = src[j];
            }
            ry[p] = v;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int p = 0; p < XPASS; ++p) *reinterpret_cast<f32x4*>(&Xs[buf][(xrow + p * RPPX) * LDX + 4 * xq]) = rx[p];
#pragma unroll
        for (int p = 0; p < YPASS; ++p) {
            int rl = yrow + p * RPPY;
            if (rl < MC) *reinterpret_cast<f32x4*>(&Ys[buf][rl * LDY + 4 * yq]) = ry[p];
        }
    };

    const int wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int wave_c = wave % WC, wave_k = wave / WC;
    f32x4 acc[CPW][KPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int j = 0; j < KPW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias gradient (column sums of dY) rides along in the blocks of the first K tile
    const bool do_bias = Pbias != nullptr && bx == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const int nchunks = (mend > mbeg) ? (mend - mbeg + MC - 1) / MC : 0;
    if (nchunks > 0) {
        if (fast) {
            load_tab(mbeg);
            load_chunk_fast(mbeg);
            if (nchunks > 1) load_tab(mbeg + MC);
        } else {
            load_chunk(mbeg);
        }
        store_chunk(0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = NBUF == 2 ? (c & 1) : 0;
        if (do_bias) {
#pragma unroll
            for (int p = 0; p < YPASS; ++p) bsum += ry[p];   // ry still holds chunk c (rows past mend are zero)
        }
        if (c + 1 < nchunks) {
            if (fast) {
                load_chunk_fast(mbeg + (c + 1) * MC);                       // its table entries arrived during chunk c - 1
                if (c + 2 < nchunks) load_tab(mbeg + (c + 2) * MC);
            } else {
                load_chunk(mbeg + (c + 1) * MC);
            }
        }
        // fragments of step st+1 are read while the MFMAs of step st issue (two register sets): with eight
        // 4-row steps per chunk an un-pipelined loop exposes the LDS latency eight times per chunk
        float a[2][CPW], b[2][KPW];
        auto read_frags = [&](int st, int set) {
#pragma unroll
            for (int c2 = 0; c2 < CPW; ++c2) a[set][c2] = Ys[buf][(st * 4 + q) * LDY + (wave_c * CPW + c2) * 16 + lr];
#pragma unroll
            for (int kt = 0; kt < KPW; ++kt) b[set][kt] = Xs[buf][(st * 4 + q) * LDX + (wave_k * KPW + kt) * 16 + lr];
        };
        read_frags(0, 0);
#pragma unroll
        for (int st = 0; st < MC / 4; ++st) {
            if (st + 1 < MC / 4) read_frags(st + 1, (st + 1) & 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int c2 = 0; c2 < CPW; ++c2)
#pragma unroll
                for (int kt = 0; kt < KPW; ++kt)
                    acc[c2][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st & 1][c2], b[st & 1][kt], acc[c2][kt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (NBUF == 2) {
            if (c + 1 < nchunks) store_chunk(buf ^ 1);
            __syncthreads();
        } else {        // single image: every wave must have read chunk c's fragments before chunk c + 1 overwrites them
            __syncthreads();
            if (c + 1 < nchunks) store_chunk(0);
            __syncthreads();
        }
    }

    if (do_bias) {   // fixed-order reduction over the RPPY row lanes that share a column group
        float* red = &Xs[0][0];                       // >= 1024 floats, free after the loop's last barrier
        *reinterpret_cast<f32x4*>(&red[4 * t]) = bsum;
        __syncthreads();
        if (t < BCO && co0 + t < g.Cout) {
            const int q4 = t >> 2, j = t & 3;
            float sacc = 0.f;
            for (int r = 0; r < RPPY; ++r) sacc += red[4 * (r * TPRY + q4) + j];
            Pbias[(size_t)bz * slab_stride + co0 + t] = sacc;
        }
    }
    float* Pout = P + (size_t)bz * slab_stride;
#pragma unroll
    for (int c2 = 0; c2 < CPW; ++c2)
#pragma unroll
        for (int kt = 0; kt < KPW; ++kt) {
            const int kcol = k0 + (wave_k * KPW + kt) * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wave_c * CPW + c2) * 16 + q * 4 + r;
                if (co < g.Cout && kcol < g.K) Pout[(size_t)co * g.K + kcol] = acc[c2][kt][r];
            }
        }
}

// ---------------------------------------------------------------------------
// Halo-tiled weight gradient on the fp32 MFMA (stride 1, SAME, odd square window, W = 40 or 20):
//   dWt[co][ky][kx][ci] = sum over pixels dY[b][y][x][co] * X[b][y + ky - R][x + kx - R][ci]
// The implicit-GEMM form above re-gathers the input once per filter tap and re-reads dY once per 64- or 128-wide K tile
// (13 x for 64->64 k5), with ~110 non-MFMA instructions per 64 MFMAs.  Here a workgroup owns a 64-channel dY tile, a
// 16-channel input tile and ALL KS*KS taps (100 accumulator registers per lane for k5) over a range of image rows, and
// walks it one image row per chunk: the KS input rows a chunk needs live in a ring of KS + 1 rows in LDS, so each new
// chunk stages ONE new input row (W + KS - 1 pixels x 16 channels) and one dY row -- every input row is fetched once per
// workgroup instead of once per tap.  The MFMA reduction index is the pixel: fragments are read pixel-major with
// ds_read_b32 (16 consecutive channels x 4 pixels per wave instruction: conflict-free), the tap's column shift and the
// step's pixel offset are ds_read immediates, the tap's row is a ring-slot base added once per chunk.
// Waves: every wave holds all four 16-channel dY fragments of a step and T / 4 taps (6 of 25, 2 of 9); the last tap is
// split by dY tile over the four waves, so all waves issue the same 4 T/4 + 1 MFMAs per step.
// grid = (Cin / 16, Cout / 64, S row slices); slice z covers image rows [z rps, (z+1) rps) of the flattened (b, y) axis and
// writes its partial sums to P[z] (same slab layout as igemm_wgrad_kernel: the optimiser launch sums them in fixed order).
// ---------------------------------------------------------------------------
template <int KS, int STEPS>   // STEPS = W / 4 MFMA steps per image row
__global__ __launch_bounds__(256, 2) void halo_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                            float* __restrict__ P, GeomDev g, int rows_per_slice,
                                                            float* __restrict__ Pbias, size_t slab_stride) {
    constexpr int R = KS / 2, T = KS * KS, NR = KS + 1, W = STEPS * 4, WPX = W + 2 * R;
    constexpr int TPW = T / 4;                 // whole taps per wave; tap T - 1 is shared
    static_assert(T % 4 == 1, "3x3 and 5x5 windows");
    constexpr int LDY = 80;                    // dY row pitch in LDS: 64 channels + 16 (== 16 mod 64 banks)
    constexpr int XROW = WPX * 16;             // floats per ring row
    constexpr int YP = (W * 16 + 255) / 256;   // dY float4 per thread and row
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Xs = lds;                     // [NR][WPX][16]
    float* const Ys = lds + NR * XROW;         // [2][W][LDY]

    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = lane & 15, q = lane >> 4;
    // (Measured, not adopted: slices on the fastest grid axis, so that the input-channel tiles of one slice -- which read the same dY
    // rows -- land on one XCD's L2: 137.8 vs 137.6 TFLOP/s on 64->64 k5 @101x40, no change on any shape.  The kernel streams: L2 hit
    // rate 0.005, fabric reads 389 MB against 197 MB algorithmic, at 0.86 of the SIMD cycles in MFMA.)
    const int ci0 = blockIdx.x * 16, co0 = blockIdx.y * 64, bz = blockIdx.z;
    const int rows_tot = g.B * g.H;
    const int r0 = min(rows_tot, bz * rows_per_slice), r1 = min(rows_tot, r0 + rows_per_slice);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X), 0, (int)((uint32_t)g.B * g.H * g.W << g.cshift) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY), 0, g.M * g.Cout * 4, 0x00020000);
    // staging roles: thread t < 4 WPX fetches channels 4 (t & 3).. of halo pixel t >> 2 of an input row;
    // dY slot i = t + 256 p is channels 4 (i & 15).. of pixel i >> 4
    const int hx = t >> 2, xj = t & 3;
    const bool x_thread = t < 4 * WPX;
    const uint32_t x_voff = (x_thread && hx >= R && hx < W + R) ? (uint32_t)((((hx - R) << g.cshift) + ci0 + 4 * xj) * 4) : 0xFFFFFFFFu;
    float* const x_dst = Xs + hx * 16 + 4 * xj;
    uint32_t y_voff[YP];
#pragma unroll
    for (int p = 0; p < YP; ++p) {
        const int i = t + 256 * p;
        y_voff[p] = i < W * 16 ? (uint32_t)(((i >> 4) * g.Cout + co0 + 4 * (i & 15)) * 4) : 0xFFFFFFFFu;
    }
    float* const y_dst = Ys + (t >> 4) * LDY + 4 * (t & 15);      // + p * 16 * LDY (+ buffer)
    auto load_x = [&](int b, int yy) {   // input row yy of image b (zeros outside the image)
        const bool ok = yy >= 0 && yy < g.H;
        const int soff = ok ? (int)((uint32_t)((b * g.H + yy) * g.W << g.cshift) * 4u) : 0;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(ok ? x_voff : 0xFFFFFFFFu), soff, 0));
    };
    auto load_y = [&](int b, int y, f32x4 (&ry)[YP]) {
        const int soff = (int)((uint32_t)((b * g.H + y) * g.W) * (uint32_t)g.Cout * 4u);
#pragma unroll
        for (int p = 0; p < YP; ++p)
            ry[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, (int)y_voff[p], soff, 0));
    };
    auto store_y = [&](int buf, const f32x4 (&ry)[YP]) {
#pragma unroll
        for (int p = 0; p < YP; ++p)
            if (t + 256 * p < W * 16) *reinterpret_cast<f32x4*>(y_dst + buf * (W * LDY) + p * 16 * LDY) = ry[p];
    };
    auto slot_of = [&](int yy) { return (yy + NR) % NR; };          // yy >= -R

    // this wave's taps: TPW whole taps starting at wave * TPW, plus dY tile `wave` of tap T - 1
    int tap_ky[TPW + 1], tap_kx[TPW + 1];
#pragma unroll
    for (int i = 0; i <= TPW; ++i) {
        const int tap = i < TPW ? wave * TPW + i : T - 1;
        tap_ky[i] = tap / KS; tap_kx[i] = tap - tap_ky[i] * KS;
    }
    const float* const a_base = Ys + q * LDY + lr;                  // + buf, + 4 st rows, + 16 c2
    const float* const b_base = Xs + q * 16 + lr;                   // + slot row, + (4 st + kx) pixels

    f32x4 acc[4][TPW], accl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
        for (int i = 0; i < TPW; ++i) acc[c2][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = Pbias != nullptr && ci0 == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    int r = r0;
    while (r < r1) {
        // one segment: rows [ys, yend) of image b (a slice may run over several images)
        const int b = r / g.H, ys = r - b * g.H;
        const int yend = min(g.H, ys + (r1 - r));
        f32x4 ry[YP];
        {
            f32x4 px[KS];
#pragma unroll
            for (int d = 0; d < KS; ++d) px[d] = load_x(b, ys - R + d);
            load_y(b, ys, ry);
            __syncthreads();                      // the previous segment's last reads
            if (x_thread) {
#pragma unroll
                for (int d = 0; d < KS; ++d) *reinterpret_cast<f32x4*>(x_dst + slot_of(ys - R + d) * XROW) = px[d];
            }
            store_y(0, ry);
            if (do_bias) {
#pragma unroll
                for (int p = 0; p < YP; ++p) bsum += ry[p];
            }
            __syncthreads();
        }
        for (int y = ys; y < yend; ++y) {
            const int buf = (y - ys) & 1;
            const bool more = y + 1 < yend;
            f32x4 rx = {0.f, 0.f, 0.f, 0.f};
            if (more) {
                rx = load_x(b, y + 1 + R);
                load_y(b, y + 1, ry);
            }
            const float* bp[TPW + 1];
#pragma unroll
            for (int i = 0; i <= TPW; ++i) bp[i] = b_base + slot_of(y + tap_ky[i] - R) * XROW + tap_kx[i] * 16;
            const float* const ap = a_base + buf * (W * LDY);
            const float* const apl = ap + wave * 16;
            float a[2][4], al[2], bt[2][TPW + 1];
            auto read_frags = [&](int st, int set) {
#pragma unroll
                for (int c2 = 0; c2 < 4; ++c2) a[set][c2] = ap[st * 4 * LDY + c2 * 16];
                al[set] = apl[st * 4 * LDY];
#pragma unroll
                for (int i = 0; i <= TPW; ++i) bt[set][i] = bp[i][st * 64];
            };
            read_frags(0, 0);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                if (st + 1 < STEPS) read_frags(st + 1, (st + 1) & 1);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < TPW; ++i)
#pragma unroll
                    for (int c2 = 0; c2 < 4; ++c2)
                        acc[c2][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st & 1][c2], bt[st & 1][i], acc[c2][i], 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_16x16x4f32(al[st & 1], bt[st & 1][TPW], accl, 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            if (more) {
                // the ring slot of row y + 1 + R held row y - R - 1, last read in chunk y - 1; dY buffer buf ^ 1 likewise
                if (x_thread) *reinterpret_cast<f32x4*>(x_dst + slot_of(y + 1 + R) * XROW) = rx;
                store_y(buf ^ 1, ry);
                if (do_bias) {
#pragma unroll
                    for (int p = 0; p < YP; ++p) bsum += ry[p];
                }
            }
            __syncthreads();
        }
        r += yend - ys;
    }

    if (do_bias) {   // fixed-order reduction over the 16 threads that share a channel quad
        float* red = lds;
        __syncthreads();
        *reinterpret_cast<f32x4*>(&red[4 * t]) = bsum;
        __syncthreads();
        if (t < 64) {
            const int q4 = t >> 2, j = t & 3;
            float sacc = 0.f;
            for (int rr = 0; rr < 16; ++rr) sacc += red[4 * (rr * 16 + q4) + j];
            Pbias[(size_t)bz * slab_stride + co0 + t] = sacc;
        }
    }
    float* Pout = P + (size_t)bz * slab_stride;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kcol = ((wave * TPW + i) << g.cshift) + ci0 + lr;
#pragma unroll
        for (int c2 = 0; c2 < 4; ++c2)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                Pout[(size_t)(co0 + c2 * 16 + q * 4 + rr) * g.K + kcol] = acc[c2][i][rr];
    }
    {
        const int kcol = ((T - 1) << g.cshift) + ci0 + lr;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Pout[(size_t)(co0 + wave * 16 + q * 4 + rr) * g.K + kcol] = accl[rr];
    }
}

// ---------------------------------------------------------------------------
// weight-gradient kernel on the bf16 matrix core (GEMM_BF16X3: NP = 3 exact-split planes; GEMM_BF16: NP = 1).
// Same grid / slices / partial layout as igemm_wgrad_kernel.  The reduction index of the MFMA is the pixel row m,
// but both operands arrive m-major (dY[m][co], im2col(X)[m][k]), so the transposition happens on the way into
// LDS: a loader thread owns 8 consecutive rows x 4 columns, splits/rounds them, packs row PAIRS into 32-bit
// words and writes, per plane and column, ONE 16-byte piece = 8 consecutive m of that column -- exactly an MFMA
// fragment, read back with a single ds_read_b128.  Lanes are column-quad-fastest (coalesced gathers); line pitch
// 20 words plus an XOR of the row-group position with (column quad >> 1) & 3 keeps the b128 stores conflict-free
// (8-lane groups, 32 banks); the fragment reads are 2-way (MI355X_MICROARCH.md, LDS table).
// Threads [0, BKI) load X, [BKI, BKI + BCO) load dY (wave-uniform roles for the 64/128 tiles).
// ---------------------------------------------------------------------------
template <int BCO, int BKI, int NP>
__global__ __launch_bounds__(256) void igemm_wgrad_bf16_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                               float* __restrict__ P, GeomDev g, int rows_per_slice,
                                                               float* __restrict__ Pbias, size_t slab_stride) {
    constexpr int MC = 32, PITCH = 20;
    constexpr int CT = BCO / 16, KT = BKI / 16;
    constexpr int WC = CT >= 8 ? 2 : (CT >= 4 ? 4 : CT);
    constexpr int WK = 4 / WC;
    constexpr int CPW = CT / WC, KPW = KT / WK;
    static_assert(BKI + BCO <= 256, "loader roles need BKI + BCO threads");
    __shared__ __attribute__((aligned(16))) unsigned Xh[NP][BKI * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned Yh[NP][BCO * PITCH];

    const int t = threadIdx.x;
    const int k0 = blockIdx.x * BKI, co0 = blockIdx.y * BCO;
    const int mbeg = blockIdx.z * rows_per_slice;
    const int mend = min(g.M, mbeg + rows_per_slice);

    const bool isX = t < BKI, isY = !isX && t < BKI + BCO;
    const int tl = isX ? t : t - BKI;
    // column quad fastest over lanes (coalesced gathers); row group = rows 8rg..8rg+7 of the chunk
    const int cq = tl % ((isX ? BKI : BCO) / 4), rg = tl / ((isX ? BKI : BCO) / 4);
    const int rgs = rg ^ ((cq >> 1) & 3);         // LDS position of the row group: XOR swizzle -> conflict-free b128 stores
    // X role: fixed k index for the block's life
    const int kidx = k0 + 4 * cq;
    const bool kok = isX && kidx < g.K;
    const int tap = kidx >> g.cshift, ci = kidx & (g.Cin - 1);
    const int kh = (tap * g.rcp_kw) >> 16, kw = tap - kh * g.KW;
    const bool cok = isY && (co0 + 4 * cq) < g.Cout;   // launcher guarantees Cout % 4 == 0

    f32x4 r[8];
    auto load_chunk = [&](int mc) {
        if (isX) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = mc + 8 * rg + i;
                const int mm = m < mend ? m : 0;
                const int b = fastdiv(mm, g.ohw_magic, g.ohw_shift), rr = mm - b * g.OHW;
                const int oh = fastdiv(rr, g.ow_magic, g.ow_shift), ow = rr - oh * g.OW;
                const int ih = oh * g.stride - g.pad_t + kh, iw = ow * g.stride - g.pad_l + kw;
                const bool ok = kok && m < mend && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
                const float* src = ok ? X + (((size_t)((b * g.H + ih) * g.W + iw)) << g.cshift) + ci : g.zeros;
                r[i] = *reinterpret_cast<const f32x4*>(src);
            }
        } else if (isY) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = mc + 8 * rg + i;
                const float* src = (cok && m < mend) ? dY + (size_t)m * g.Cout + co0 + 4 * cq : g.zeros;
                r[i] = *reinterpret_cast<const f32x4*>(src);
            }
        }
    };
    auto store_split = [&]() {
        if (!(isX || isY)) return;
        unsigned* base = isX ? &Xh[0][0] : &Yh[0][0];
        const int plane_words = isX ? BKI * PITCH : BCO * PITCH;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned w[4][NP];
#pragma unroll
            for (int u = 0; u < 4; ++u) split_pair<NP>(r[2 * u][j], r[2 * u + 1][j], w[u]);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                *reinterpret_cast<uint4*>(base + pl * plane_words + (4 * cq + j) * PITCH + 4 * rgs) =
                    make_uint4(w[0][pl], w[1][pl], w[2][pl], w[3][pl]);
        }
    };

    const int wave = t >> 6, lane = t & 63, lr = lane & 15, q = lane >> 4;
    const int wave_c = wave % WC, wave_k = wave / WC;
    f32x4 acc[CPW][KPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i)
#pragma unroll
        for (int j = 0; j < KPW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias gradient (exact fp32 column sums of dY) rides along in the blocks of the first K tile
    const bool do_bias = Pbias != nullptr && blockIdx.x == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const int nchunks = (mend > mbeg) ? (mend - mbeg + MC - 1) / MC : 0;
    if (nchunks > 0) load_chunk(mbeg);
    for (int c = 0; c < nchunks; ++c) {
        store_split();
        if (do_bias && isY) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bsum += r[i];     // rows past mend were loaded as zeros
        }
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(mbeg + (c + 1) * MC);
        bf16x8 b[KPW][NP];
#pragma unroll
        for (int kt = 0; kt < KPW; ++kt) {
            const int col = (wave_k * KPW + kt) * 16 + lr;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                b[kt][pl] = *reinterpret_cast<const bf16x8*>(&Xh[pl][col * PITCH + 4 * (q ^ ((col >> 3) & 3))]);
        }
#pragma unroll
        for (int c2 = 0; c2 < CPW; ++c2) {
            bf16x8 a[NP];
            const int colc = (wave_c * CPW + c2) * 16 + lr;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                a[pl] = *reinterpret_cast<const bf16x8*>(&Yh[pl][colc * PITCH + 4 * (q ^ ((colc >> 3) & 3))]);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int term = (NP == 3 ? 0 : 5); term < 6; ++term)
#pragma unroll
                for (int kt = 0; kt < KPW; ++kt)
                    acc[c2][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[X3_TA[term] % NP], b[kt][X3_TB[term] % NP], acc[c2][kt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    }

    if (do_bias) {   // fixed-order reduction over the 4 row groups that share a column quad
        float* red = reinterpret_cast<float*>(&Xh[0][0]);   // >= 4*BCO floats, free after the loop's last barrier
        if (isY) *reinterpret_cast<f32x4*>(&red[4 * (4 * cq + rg)]) = bsum;
        __syncthreads();
        if (t < BCO && co0 + t < g.Cout) {
            const int q4 = t >> 2, j = t & 3;
            float sacc = 0.f;
            for (int rr = 0; rr < 4; ++rr) sacc += red[4 * (4 * q4 + rr) + j];
            Pbias[(size_t)blockIdx.z * slab_stride + co0 + t] = sacc;
        }
    }
    float* Pout = P + (size_t)blockIdx.z * slab_stride;
#pragma unroll
    for (int c2 = 0; c2 < CPW; ++c2)
#pragma unroll
        for (int kt = 0; kt < KPW; ++kt) {
            const int kcol = k0 + (wave_k * KPW + kt) * 16 + lr;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int co = co0 + (wave_c * CPW + c2) * 16 + q * 4 + rr;
                if (co < g.Cout && kcol < g.K) Pout[(size_t)co * g.K + kcol] = acc[c2][kt][rr];
            }
        }
}

// 128-wide K tiles double the MFMA work per barrier but halve the number of tiles: use them only
// when the grid still covers the chip with the row slices available (M / 256)
static inline int wgrad_bco(int N) { return N <= 16 ? 16 : (N <= 32 ? 32 : (N <= 64 ? 64 : 128)); }
static inline int wgrad_bki(int M, int N, int K) {
    if (K < 512) return 64;
    const int bco = wgrad_bco(N);
    const long blocks = (long)cdiv(K, 128) * cdiv(N, bco) * std::max(1, M / 256);
    return blocks >= 1024 ? 128 : 64;
}

// XCD-grouped wgrad grid: OFF by default.  Measured round 2 (after the row-table loader, slice count chosen per XCD so
// that every XCD's share fills whole waves): 96 vs 108 TFLOP/s on 64->64 k5 @101x40, 106 vs 131 on 256->256 @26x10,
// 96 vs 121 on 64->128 @51x20 -- slower on every shape, as in round 1: the fabric-side re-reads (6x algorithmic on the
// 64-column layer) are served by the Infinity Cache and are not what limits the kernel; eight XCDs each hammering one
// slice's lines are.  CMOOP_WGRAD_XCD=1 enables it for experiments.
static bool wgrad_xcd_grouped() {
    static const bool v = [] { const char* e = std::getenv("CMOOP_WGRAD_XCD"); return e && e[0] == '1'; }();
    return v;
}

// halo-tiled weight gradient (halo_wgrad_kernel): geometry it accepts
static bool halo_wgrad_geometry(const ConvGeom& cg) {
    // default since round 3 (CMOOP_HALO_WGRAD=0 restores the implicit GEMM).  Isolated TFLOP/s, halo vs implicit GEMM: 64->64 k5 @101x40
    // 137.4 vs 113, k3 120.6 vs 99; 64->64 k5 @51x20 122 vs 97, k3 107 vs 87; 64->128 k5 129 vs 121, k3 117 vs 104; 128->128 @51x20
    // 131.6 vs 130.6 (k3 equal); the pop-40 job 2 254.6 vs 2 222.3 evals/h.
    static const bool on = [] { const char* v = std::getenv("CMOOP_HALO_WGRAD"); return !(v && v[0] == '0'); }();
    const int ks = cg.KH, R = ks / 2;
    return on && resolve_mode(GEMM_DEFAULT) == GEMM_FP32 && cg.KH == cg.KW && (ks == 3 || ks == 5) && cg.stride == 1 &&
           cg.OH == cg.H && cg.OW == cg.W && cg.pad_t == R && cg.pad_l == R && (cg.W == 40 || cg.W == 20) &&
           cg.Cin % 16 == 0 && cg.Cout % 64 == 0 && (cg.Cin / 16) * (cg.Cout / 64) <= 256;
}
// its slice count: one round of 512 workgroups (two per CU) whenever the slab budget and the row count allow
static int halo_wgrad_slices(const ConvGeom& g) {
    const int tiles = (g.Cin / 16) * (g.Cout / 64), rows = g.B * g.H;
    int S = std::max(1, 512 / tiles);
    const int64_t nk = (int64_t)g.Cout * g.K();
    const int cap = (int)std::max<int64_t>(1, (16ll << 20) / std::max<int64_t>(nk, 1));     // <= ~16M slab floats
    S = std::min(S, cap);
    S = std::min(S, std::max(1, rows / 8));                                                 // >= 8 image rows per slice
    return S;
}

int wgrad_slices(const ConvGeom& g) {
    if (halo_wgrad_geometry(g)) return halo_wgrad_slices(g);
    const int M = g.M(), K = g.K(), N = g.Cout;
    const int bco = wgrad_bco(N);
    const int tiles = cdiv(K, wgrad_bki(M, N, K)) * cdiv(N, bco);
    if (wgrad_xcd_grouped() && resolve_mode(GEMM_DEFAULT) == GEMM_FP32 && M / 256 >= 8) {
        // XCD-grouped grid: S = 8 G, each XCD runs G * tiles workgroups on its 32 CUs: choose the G whose last wave of
        // co-resident workgroups (LDS-limited occupancy) is >= 90 % full, within the slab budget and >= 256 rows per slice
        const int bki = wgrad_bki(M, N, K);
        const int lds_bytes = 2 * 32 * ((bki + 16) + (bco == 16 ? 16 : bco + 16)) * 4;
        const int slots = 32 * std::max(1, std::min(8, (160 * 1024) / lds_bytes));
        const int64_t nk = (int64_t)N * K;
        const int cap = (int)std::max<int64_t>(8, (16ll << 20) / std::max<int64_t>(nk, 1));
        const int maxG = std::max(1, std::min(std::min(cap, M / 256) / 8, 64));
        const int wantG = std::max(1, std::min(maxG, cdiv(2048, tiles * 8)));      // aim at >= ~2048 workgroups chip-wide
        int best = wantG;
        double best_u = -1.0;
        for (int G = wantG; G <= maxG && G <= 2 * wantG + 4; ++G) {
            const long blocks = (long)G * tiles;
            const long waves = (blocks + slots - 1) / slots;
            const double u = (double)blocks / (double)(waves * slots);
            if (u >= 0.90) { best = G; best_u = u; break; }
            if (u > best_u) { best_u = u; best = G; }
        }
        return 8 * best;
    }
    static const int target_wgs = [] { const char* e = std::getenv("CMOOP_WGRAD_WGS"); const int v = e ? std::atoi(e) : 0; return v >= 256 && v <= 8192 ? v : 2048; }();
    int S = cdiv((int)(target_wgs * par_scale()), tiles);
    // cap the slab traffic (S*N*K floats written and read back): at most ~16M floats, but keep >= 8 slices
    const int64_t nk = (int64_t)N * K;
    const int cap = (int)std::max<int64_t>(8, (16ll << 20) / std::max<int64_t>(nk, 1));
    if (S > cap) S = cap;
    const int maxS0 = std::max(1, M / 256);
    if (S > maxS0) S = maxS0;
    // few-wave grids: choose the slice count that fills whole waves of co-resident workgroups
    // (LDS-limited occupancy of each instantiation: 160 KiB / LDS per workgroup)
    const int bki = wgrad_bki(M, N, K);
    const int lds_bytes = 2 * 32 * ((bki + 16) + (bco == 16 ? 16 : bco + 16)) * 4;
    const int resident = std::max(32, (int)(par_scale() * 256 * std::max(1, std::min(8, (160 * 1024) / lds_bytes))));
    if ((long)tiles * S < 4l * resident) S = fill_waves(tiles, S, resident, 0.95);
    const int maxS = std::max(1, M / 256);
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    return S;
}

// 64-row chunks on a SINGLE LDS image for the <= 64-channel weight-gradient tiles (row-table gather only): the same LDS footprint
// as two 32-row images, twice the MFMAs per chunk of fixed loop / wait / barrier overhead (the <64,128> tile runs 64 MFMAs per
// wave and chunk against ~110 other instructions; PMC: 0.70 of the SIMD cycles in MFMA against 0.81 for <128,128>).  Round 3:
// 64->64 k5 @101x40 107.2 -> 113.0 TFLOP/s, k3 95.8 -> 99.6, 32->32 k5 89.6 -> 92.1, 16->16 k3 43.3 -> 47.7; the job 2 179 -> 2 181 / 2 187
// evals/h (two A/B pairs).  CMOOP_WGRAD_MC64=0 restores 32-row chunks everywhere.
static bool wgrad_mc64(bool rowtab_gather, int mode, int bco) {
    static const bool on = [] { const char* v = std::getenv("CMOOP_WGRAD_MC64"); return !(v && v[0] == '0'); }();
    // (the 128-channel tile does NOT gain: <128,128> 130.6 -> 126.8 TFLOP/s on 128->128 k5, 120.6 -> 112.4 on 512->512 k3 -- it already runs
    // 128 MFMAs per chunk and the second barrier costs more than the amortisation buys)
    return on && rowtab_gather && mode == GEMM_FP32 && bco <= 64;
}

// instantiation code + flags of the weight-gradient launch for this geometry with S row slices (host arithmetic only)
int igemm_wgrad_plan(const ConvGeom& g, int S, int mode_req, bool have_rowtab, int* flags_out) {
    igemm_check_range(g);
    const int N = g.Cout, M = g.M();
    const int bco = wgrad_bco(N), bki = wgrad_bki(M, N, g.K());
    int mode = (N % 4 == 0) ? resolve_mode(mode_req) : (int)GEMM_FP32;
    if (mode == GEMM_BF16X3 && bco < 128) mode = GEMM_FP32;
    const bool rt = have_rowtab && g.KH * g.KW <= 32;
    if (mode == GEMM_FP32 && halo_wgrad_geometry(g)) {
        if (flags_out) *flags_out = S > 1 ? GEMM_FLAG_SLABS : 0;
        return GEMM_FP32_HALO * 10000000 + g.KH * 1000 + g.W / 4;
    }
    if (flags_out) *flags_out = ((rt && mode == GEMM_FP32 && (N & 3) == 0) ? GEMM_FLAG_ROWTAB : 0) | (S > 1 ? GEMM_FLAG_SLABS : 0);
    return mode * 10000000 + (wgrad_mc64(rt && (N & 3) == 0, mode, bco) ? 1000000 : 0) + bco * 1000 + bki;
}

int launch_igemm_wgrad(const float* X, const float* dY, float* P, const ConvGeom& cg, int S, hipStream_t s,
                       const GemmTiming* tm, float* Pbias, size_t slab_stride, int mode_req, const void* rowtab, int tab_rows,
                       int* flags_out) {
    if (flags_out) *flags_out = 0;
    GeomDev g = to_dev(cg);
    if (g.M == 0) return 0;
    const size_t stride = slab_stride ? slab_stride : (size_t)g.Cout * g.K;
    int rps = cdiv(g.M, S);
    const int N = g.Cout;
    const int bco = wgrad_bco(N);
    const int bki = wgrad_bki(g.M, g.Cout, g.K);
    // the bf16 loaders read dY as aligned float4: the classifier layer (10 / 11 / 35 columns) stays on the fp32 kernel
    int mode = (N % 4 == 0) ? resolve_mode(mode_req) : (int)GEMM_FP32;
    // bf16x3 pays the split VALU work per loaded element: with fewer than 128 output channels per block there are too
    // few MFMAs per element to hide it (measured 58 vs 73 TFLOP/s on 64->64 k5) -- the exact kernel is the better
    // fp32-accurate choice there.  (GEMM_BF16 must round everywhere to stay consistent with its definition.)
    if (mode == GEMM_BF16X3 && bco < 128) mode = GEMM_FP32;
    if (mode == GEMM_FP32 && halo_wgrad_geometry(cg)) {
        const int ks = cg.KH, steps = cg.W / 4;
        const dim3 hgrid(cg.Cin / 16, cg.Cout / 64, S);
        const int hrps = cdiv(cg.B * cg.H, S);
        size_t lds = ((size_t)(ks + 1) * (cg.W + ks - 1) * 16 + 2 * (size_t)cg.W * 80) * sizeof(float);
        // two workgroups per CU when that fills the last round better (see the forward halo launch)
        {
            const long wgs = (long)hgrid.x * hgrid.y * hgrid.z;
            auto fill = [&](long slots) { return (double)wgs / (double)(((wgs + slots - 1) / slots) * slots); };
            if (wgs <= 1536 && fill(512) > fill(768) + 0.05) lds = std::max(lds, (size_t)55 * 1024);
        }
        if (flags_out) *flags_out = S > 1 ? GEMM_FLAG_SLABS : 0;
#define CMOOP_HWG(KS_, ST_)                                                                                                  \
        do {                                                                                                                 \
            if (tm && tm->start && tm->ext) {                                                                                \
                hipExtLaunchKernelGGL((halo_wgrad_kernel<KS_, ST_>), hgrid, dim3(256), lds, s, tm->start, tm->stop, 0, X, dY, P, g, hrps, Pbias, stride); \
            } else {                                                                                                         \
                if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->start, s));                                                \
                hipLaunchKernelGGL((halo_wgrad_kernel<KS_, ST_>), hgrid, dim3(256), lds, s, X, dY, P, g, hrps, Pbias, stride); \
                if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->stop, s));                                                 \
            }                                                                                                                \
        } while (0)
        if (ks == 5) { if (steps == 10) CMOOP_HWG(5, 10); else CMOOP_HWG(5, 5); }
        else         { if (steps == 10) CMOOP_HWG(3, 10); else CMOOP_HWG(3, 5); }
#undef CMOOP_HWG
        CMOOP_HIP(hipGetLastError());
        return GEMM_FP32_HALO * 10000000 + ks * 1000 + steps;
    }
    dim3 grid(cdiv(g.K, bki), cdiv(N, bco), S);
    int gkt = 0, gct = 0;
    // (Measured, not adopted, round 3: row slices on the FASTEST grid axis, so that with round-robin dispatch every K tile of a
    // slice runs on XCD s % 8 and co-resident K tiles share the slice's dY / input rows in one L2 -- PMC shows 36 % L2 hit
    // rate and 1.1 GB of fabric reads per launch on 64->64 k5 @101x40 with K tiles fastest.  95.2 vs 107.2 TFLOP/s there,
    // 89.1 vs 99.2 @51x20, no change on 128->128: like the XCD-grouped grid, concentrating a slice's lines on one L2 is slower
    // than spreading them over eight L2s and the Infinity Cache.)
    if (wgrad_xcd_grouped() && mode == GEMM_FP32 && S >= 8) {
        gkt = (int)grid.x; gct = (int)grid.y;
        grid = dim3((unsigned)(cdiv(S, 8) * 8 * gkt * gct), 1, 1);
    }
    // the row table must cover every row a chunk can touch (rows are consumed 32 at a time)
    const uint2* rt = (rowtab && tab_rows >= cdiv(g.M, 32) * 32 && g.KH * g.KW <= 32) ? static_cast<const uint2*>(rowtab) : nullptr;
    if (flags_out) *flags_out = ((rt && mode == GEMM_FP32 && (g.Cout & 3) == 0) ? GEMM_FLAG_ROWTAB : 0) | (S > 1 ? GEMM_FLAG_SLABS : 0);
#define CMOOP_WGK(KERNEL)                                                                                       \
    do {                                                                                                       \
        if (tm && tm->start && tm->ext) {                                                                      \
            hipExtLaunchKernelGGL(KERNEL, grid, dim3(256), 0, s, tm->start, tm->stop, 0, X, dY, P, g, rps, Pbias, stride); \
        } else {                                                                                               \
            if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->start, s));                                      \
            hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, s, X, dY, P, g, rps, Pbias, stride);                 \
            if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->stop, s));                                       \
        }                                                                                                      \
    } while (0)
#define CMOOP_WGF(KERNEL)                                                                                       \
    do {                                                                                                       \
        if (tm && tm->start && tm->ext) {                                                                      \
            hipExtLaunchKernelGGL(KERNEL, grid, dim3(256), 0, s, tm->start, tm->stop, 0, X, dY, P, g, rps, Pbias, stride, rt, tab_rows, gkt, gct, S); \
        } else {                                                                                               \
            if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->start, s));                                      \
            hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, s, X, dY, P, g, rps, Pbias, stride, rt, tab_rows, gkt, gct, S);   \
            if (tm && tm->start) CMOOP_HIP(hipEventRecord(tm->stop, s));                                       \
        }                                                                                                      \
    } while (0)
    const bool use_mc64 = wgrad_mc64(rt != nullptr && (N & 3) == 0, mode, bco);     // (tab_rows covers whole 256-row tiles: rowtab_rows)
    rps = use_mc64 ? cdiv(rps, 64) * 64 : cdiv(rps, 32) * 32;      // slices are whole chunks of the kernel's chunk depth
#define CMOOP_WG2(BCO_, BKI_)                                                          \
    do {                                                                               \
        if (mode == GEMM_BF16X3) CMOOP_WGK((igemm_wgrad_bf16_kernel<BCO_, BKI_, 3>));  \
        else if (mode == GEMM_BF16) CMOOP_WGK((igemm_wgrad_bf16_kernel<BCO_, BKI_, 1>)); \
        else if (use_mc64 && BCO_ <= 64) CMOOP_WGF((igemm_wgrad_kernel<(BCO_ <= 64 ? BCO_ : 64), BKI_, 64>)); \
        else CMOOP_WGF((igemm_wgrad_kernel<BCO_, BKI_>));                              \
    } while (0)
#define CMOOP_WG(BCO_)                    \
    do {                                  \
        if (bki == 128) CMOOP_WG2(BCO_, 128); \
        else CMOOP_WG2(BCO_, 64);         \
    } while (0)
    if (bco == 16) CMOOP_WG(16);
    else if (bco == 32) CMOOP_WG(32);
    else if (bco == 64) CMOOP_WG(64);
    else CMOOP_WG(128);
#undef CMOOP_WG
#undef CMOOP_WG2
#undef CMOOP_WGF
#undef CMOOP_WGK
    CMOOP_HIP(hipGetLastError());
    return mode * 10000000 + (use_mc64 ? 1000000 : 0) + bco * 1000 + bki;
}

// out[i] = sum_s P[s][i], fixed summation order.  Vector form: each thread owns one float4 column group,
// 4 slice lanes x 8 independent accumulators keep 8 loads of 16 B in flight per thread (512 first-layer slabs: 16 rounds).
template <int VEC>
__global__ __launch_bounds__(256) void reduce_slices_kernel(const float* __restrict__ P, float* __restrict__ out, int S,
                                                            int64_t n, int64_t stride) {
    __shared__ float red[4][64 * VEC];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * 64 + e) * VEC;
    float acc[8][VEC];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[u][j] = 0.f;
    if (i < n) {
        int s = sl;
        for (; s + 28 < S; s += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* src = P + (size_t)(s + 4 * u) * stride + i;
                if constexpr (VEC == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src);
                    acc[u][0] += v[0]; acc[u][1] += v[1]; acc[u][2] += v[2]; acc[u][3] += v[3];
                } else {
                    acc[u][0] += src[0];
                }
            }
        }
        for (; s < S; s += 4) {
            const float* src = P + (size_t)s * stride + i;
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[0][j] += src[j];
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        red[sl][e * VEC + j] = ((acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j])) + ((acc[4][j] + acc[5][j]) + (acc[6][j] + acc[7][j]));
    __syncthreads();
    if (sl == 0 && i < n) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            out[i + j] = ((red[0][e * VEC + j] + red[1][e * VEC + j]) + red[2][e * VEC + j]) + red[3][e * VEC + j];
    }
}

void launch_reduce_slices(const float* P, float* out, int S, int64_t n, hipStream_t s, int64_t stride) {
    if (n == 0) return;
    if (!stride) stride = n;
    const bool vec = (n % 4 == 0) && (stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    if (vec) hipLaunchKernelGGL(reduce_slices_kernel<4>, dim3((unsigned)cdiv64(n / 4, 64)), dim3(256), 0, s, P, out, S, n, stride);
    else hipLaunchKernelGGL(reduce_slices_kernel<1>, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, P, out, S, n, stride);
    CMOOP_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* __restrict__ W, float* __restrict__ Wd, int Cout,
                                                             int KH, int KW, int Cin) {
    // thread per output element of Wd[ci][kh'][kw'][co]; co fastest -> coalesced stores
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t n = (int64_t)Cout * KH * KW * Cin;
    if (i >= n) return;
    int co = (int)(i % Cout);
    int64_t r = i / Cout;
    int kw = (int)(r % KW); r /= KW;
    int kh = (int)(r % KH);
    int ci = (int)(r / KH);
    Wd[i] = W[(((size_t)co * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * Cin + ci];
}

// every conv layer of a net in ONE launch: blockIdx.y = layer (table row), blockIdx.x strides over its 32 x 32 (co, ci)
// tiles per filter tap.  The tile goes through LDS so that BOTH sides move whole 128-byte rows: reads run along ci
// (contiguous in W[co][kh][kw][ci]), writes along co (contiguous in Wd[ci][kh'][kw'][co]).  The element-per-thread form
// this replaces read 4-byte words strided by a whole filter (145 us per step on the 13.6 M-parameter candidate, 1.7 % of
// its step: profiles/r03_lone_heaviest).
__global__ __launch_bounds__(256) void flip_transpose_all_kernel(const float* __restrict__ params, float* __restrict__ wd_all,
                                                                 const FlipEntry* __restrict__ table) {
    __shared__ float tile[32][33];
    const FlipEntry e = table[blockIdx.y];
    const float* W = params + e.w_off;
    float* Wd = wd_all + e.wd_off;
    const int taps = e.KH * e.KW, cot = (e.Cout + 31) >> 5, cit = (e.Cin + 31) >> 5;
    const int tiles = taps * cot * cit;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int tap = t / (cot * cit), r = t - tap * (cot * cit);
        const int co0 = (r / cit) << 5, ci0 = (r - (r / cit) * cit) << 5;
        const int kh = tap / e.KW, kw = tap - kh * e.KW;
        const int ftap = (e.KH - 1 - kh) * e.KW + (e.KW - 1 - kw);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int co = co0 + ty + 8 * p, ci = ci0 + tx;
            tile[ty + 8 * p][tx] = (co < e.Cout && ci < e.Cin) ? W[((size_t)co * taps + tap) * e.Cin + ci] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ci = ci0 + ty + 8 * p, co = co0 + tx;
            if (ci < e.Cin && co < e.Cout) Wd[((size_t)ci * taps + ftap) * e.Cout + co] = tile[tx][ty + 8 * p];
        }
        __syncthreads();
    }
}

void launch_flip_transpose_all(const float* params, float* wd_all, const FlipEntry* table_dev, int layers, int64_t max_elems,
                               hipStream_t s) {
    if (layers <= 0 || max_elems <= 0) return;
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64(max_elems, 1024), 2048));   // one workgroup per 32 x 32 tile of the largest layer
    hipLaunchKernelGGL(flip_transpose_all_kernel, dim3(gx, (unsigned)layers), dim3(256), 0, s, params, wd_all, table_dev);
    CMOOP_HIP(hipGetLastError());
}

void launch_flip_transpose(const float* W, float* Wd, int Cout, int KH, int KW, int Cin, hipStream_t s) {
    int64_t n = (int64_t)Cout * KH * KW * Cin;
    hipLaunchKernelGGL(flip_transpose_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, W, Wd, Cout, KH, KW, Cin);
    CMOOP_HIP(hipGetLastError());
}

}  // namespace cmoop
