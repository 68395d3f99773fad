// Shared host/device helpers for libcmoop_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace cmoop {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define CMOOP_HIP(expr)                                                                    \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            throw ::cmoop::Error(std::string(#expr) + ": " + hipGetErrorString(e__) +      \
                                 " (" __FILE__ ":" + std::to_string(__LINE__) + ")");      \
    } while (0)

#define CMOOP_REQUIRE(cond, msg)                                                           \
    do {                                                                                   \
        if (!(cond)) throw ::cmoop::Error(std::string("cmoop: ") + (msg));                 \
    } while (0)

// ---------------------------------------------------------------------------
// Counter-based RNG, bit-exact twin of oracle/rng.py (murmur3 fmix32 chain).
// Streams: 0x1000+tensor (init), 0x2000+fc layer (dropout), 0x3000 (shuffle).
// ---------------------------------------------------------------------------
constexpr uint32_t STREAM_INIT = 0x1000u, STREAM_DROPOUT = 0x2000u, STREAM_SHUFFLE = 0x3000u;

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t rng_prefix(uint32_t seed, uint32_t stream, uint32_t ctr) {
    return fmix32(fmix32(fmix32(seed + 0x9E3779B9u) ^ stream) ^ ctr);
}
__host__ __device__ __forceinline__ uint32_t rng_u32(uint32_t seed, uint32_t stream, uint32_t ctr, uint32_t idx) {
    return fmix32(rng_prefix(seed, stream, ctr) ^ idx);
}

static inline int ilog2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return ((1 << s) == v) ? s : -1;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace cmoop
