// Shared host/device helpers of libppo2hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/ppo2hip.h"

namespace ppo2hip {

void set_error(const char *fmt, ...);

// Returns from the enclosing function with the hip error code when a runtime call fails.
#define PPO2HIP_CHECK_HIP(expr)                                                              \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            ::ppo2hip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return (int)e__;                                                                 \
        }                                                                                    \
    } while (0)

#define PPO2HIP_REQUIRE(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            ::ppo2hip::set_error(__VA_ARGS__);     \
            return PPO2HIP_EINVAL;                 \
        }                                          \
    } while (0)

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

constexpr int kWave = 64;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Bump allocator over a caller-provided workspace (256-byte aligned carve-outs).
struct Carver {
    char *base;
    size_t off, cap;
    bool ok;
    Carver(void *p, size_t bytes) : base((char *)p), off(0), cap(bytes), ok(true) {}
    template <class T>
    T *take(size_t n) {
        size_t o = align_up(off, 256);
        size_t end = o + n * sizeof(T);
        if (end > cap && base) ok = false;
        off = end;
        return base ? (T *)(base + o) : nullptr;
    }
    size_t used() const { return align_up(off, 256); }
};

// map flat env-major sample s -> element offset in a time-major [T, E] array (T == 0: identity)
__host__ __device__ inline int64_t tm_offset(int64_t s, int T, int E) {
    if (T == 0) return s;
    int64_t env = s / T;
    int64_t t = s - env * T;
    return t * (int64_t)E + env;
}

// Sample numbers that arrive from a caller's index list are checked where they are used: one outside [0, limit) raises
// bit `bit` of the device's fault word (read and cleared by ppo2hip_take_faults) and is replaced by sample 0, so that a
// stale list is reported instead of faulting the GPU or gathering garbage silently.
constexpr unsigned kFaultIndexRange = 1u, kFaultComposeRange = 2u, kFaultGridSync = 4u;
__device__ inline int64_t checked_sample(int64_t smp, int64_t limit, unsigned *fault, unsigned bit = kFaultIndexRange) {
    if ((uint64_t)smp >= (uint64_t)limit) {
        atomicOr(fault, bit);
        return 0;
    }
    return smp;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in every thread.
template <class T>
__device__ inline T block_sum(T v, T *smem /* >= 17 elements */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        T s = 0;
        for (int i = 0; i < nw; i++) s += smem[i];
        smem[16] = s;
    }
    __syncthreads();
    return smem[16];
}

}  // namespace ppo2hip
