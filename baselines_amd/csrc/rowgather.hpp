// Reading minibatch rows straight out of the rollout arrays (what `arr[mbinds]` means in
// baselines/ppo2/ppo2.py:117) inside a consumer kernel: buffer-descriptor loads and the sample -> row map.
#pragma once
#include <cstdint>

#include "common.hpp"

namespace ppo2hip {

// 32-bit byte offsets (one register per address instead of two) and hardware range checking: a lane that reads past
// the last row gets 0 instead of a fault, so the loads need no per-lane guards.  Arrays must stay under 4 GB.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_f32(rsrc_t r, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ int buf_i32(rsrc_t r, unsigned byte_off) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0);
}

// Minibatch position i -> sample number (idx[i], or i itself without an index list) -> rollout row.  Sample numbers are
// env-major (s = env * T + t, what sf01 produces); the arrays are time-major (row = t * E + env); T = 0 means the arrays
// are flat already.  env = s / T runs as a multiplication (Granlund-Montgomery round-up form, s < 2^31):
//   t = mulhi(magic, s), env = (t + ((s - t) >> sh1)) >> sh2,  magic = floor(2^32 (2^c - T) / T) + 1, c = ceil(log2 T).
struct RowMap {
    const int32_t *idx;
    int64_t B;
    int T, E;
    unsigned magic;
    int sh1, sh2;
    unsigned nsamples;       // sample numbers a list may name: T E, or the row count of flat arrays
    unsigned *fault;         // the device's fault word (checked_sample)
    // branch-free on purpose: a branch in the middle of a software-pipelined loop makes the compiler drain its loads there
    __device__ __forceinline__ int row_of_sample(int smp) const {
        const unsigned n = (unsigned)smp, t = __umulhi(magic, n);
        const int env = (int)((t + ((n - t) >> sh1)) >> sh2);
        const int row = (smp - env * T) * E + env;
        return T == 0 ? smp : row;
    }
    // sample number of minibatch position i (pos_ok: i is inside the caller's range); r_idx covers idx[0 .. B) (size 0
    // without a list: the load returns 0 and i itself is taken)
    __device__ __forceinline__ int sample_at(rsrc_t r_idx, int64_t i, bool pos_ok) const {
        int listed = buf_i32(r_idx, (unsigned)(pos_ok ? i : B) * 4u);
        if ((unsigned)listed >= nsamples) {      // a stale list: reported, not followed (never taken on a good one)
            atomicOr(fault, kFaultIndexRange);
            listed = 0;
        }
        const int self = pos_ok ? (int)i : 0;
        return idx ? listed : self;
    }
};
inline RowMap make_row_map(const int32_t *idx, int64_t B, int T, int E, int64_t nsamples, unsigned *fault) {
    RowMap m;
    m.idx = idx; m.B = B; m.T = T; m.E = E;
    m.nsamples = (unsigned)(nsamples < 0x7fffffffll ? nsamples : 0x7fffffffll); m.fault = fault;
    const unsigned t = T > 0 ? (unsigned)T : 1u;
    int sh = 0;
    while ((1ull << sh) < t) sh++;
    m.magic = (unsigned)((((1ull << sh) - t) << 32) / t + 1);
    m.sh1 = sh < 1 ? sh : 1;
    m.sh2 = sh > 1 ? sh - 1 : 0;
    return m;
}

// whether every rollout array of an MLP policy can be read through 32-bit buffer offsets
inline bool rollout_fits_buffer_loads(const ppo2hip_policy_desc &d, const ppo2hip_rollout &ro, int64_t B) {
    int64_t widest = d.ob_dim;
    if (d.ac_dim > widest) widest = d.ac_dim;
    if (d.nvalue > widest) widest = d.nvalue;
    if (widest < 2) widest = 2;
    const int64_t limit = (int64_t(1) << 32) - 4096;
    return ro.n > 0 && ro.n * widest * 4 < limit && B * 4 < limit;
}

}  // namespace ppo2hip
