// Host-side minibatch permutation generator, bit-exact with NumPy's legacy RandomState:
// MT19937 (init_genrand seeding) and the Fisher-Yates pass of RandomState.shuffle whose
// bounded draws mask a 32-bit output to the next power of two minus one and reject values
// above the bound.  This is what np.random.seed / np.random.shuffle do at
// baselines/ppo2/ppo2.py:113 and :130 (seeded through baselines/common/misc_util.py:69-76).
//
// The swap targets of a block of positions are drawn first and the swaps applied second,
// with the target cache lines prefetched: the draw order (and therefore the stream and the
// result) is unchanged, the pass just stops stalling on one random access per element.
#include <cstdint>
#include <cstring>

#include "common.hpp"

struct ppo2hip_rng {
    uint32_t mt[624];
    int pos;
};

namespace {

constexpr int N = 624, M = 397;

void seed_state(ppo2hip_rng *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < N; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = N;
}

void regenerate(ppo2hip_rng *s) {
    uint32_t *mt = s->mt;
    int k = 0;
    auto twist = [](uint32_t u, uint32_t v) {
        uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
        return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
    };
    for (; k < N - M; k++) mt[k] = mt[k + M] ^ twist(mt[k], mt[k + 1]);
    for (; k < N - 1; k++) mt[k] = mt[k + (M - N)] ^ twist(mt[k], mt[k + 1]);
    mt[N - 1] = mt[M - 1] ^ twist(mt[N - 1], mt[0]);
    s->pos = 0;
}

inline uint32_t next_u32(ppo2hip_rng *s) {
    if (s->pos >= N) regenerate(s);
    uint32_t y = s->mt[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

inline uint32_t bounded(ppo2hip_rng *s, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    while ((v = next_u32(s) & mask) > max) {
    }
    return v;
}

template <class I>
int shuffle_impl(ppo2hip_rng *rng, I *x, int64_t n) {
    if (!rng || (!x && n > 0)) {
        ppo2hip::set_error("shuffle: null pointer");
        return PPO2HIP_EINVAL;
    }
    if (n < 0 || n > 0xffffffffll) {
        ppo2hip::set_error("shuffle: n=%lld outside [0, 2^32)", (long long)n);
        return PPO2HIP_EINVAL;
    }
    constexpr int BLK = 2048, AHEAD = 24;
    uint32_t js[BLK];
    int64_t i = n - 1;
    while (i >= 1) {
        const int cnt = (int)(i < BLK ? i : BLK);
        for (int k = 0; k < cnt; k++) js[k] = bounded(rng, (uint32_t)(i - k));
        for (int k = 0; k < cnt; k++) {
            if (k + AHEAD < cnt) __builtin_prefetch(&x[js[k + AHEAD]], 1, 1);
            const int64_t a = i - k;
            const I tmp = x[a];
            x[a] = x[js[k]];
            x[js[k]] = tmp;
        }
        i -= cnt;
    }
    return PPO2HIP_OK;
}

}  // namespace

extern "C" ppo2hip_rng *ppo2hip_rng_create(uint32_t seed) {
    ppo2hip_rng *r = new ppo2hip_rng;
    seed_state(r, seed);
    return r;
}
extern "C" void ppo2hip_rng_destroy(ppo2hip_rng *rng) { delete rng; }
extern "C" void ppo2hip_rng_seed(ppo2hip_rng *rng, uint32_t seed) {
    if (rng) seed_state(rng, seed);
}
extern "C" int ppo2hip_rng_set_state(ppo2hip_rng *rng, const uint32_t *key, int pos) {
    if (!rng || !key || pos < 0 || pos > N) {
        ppo2hip::set_error("rng_set_state: bad argument (pos=%d)", pos);
        return PPO2HIP_EINVAL;
    }
    std::memcpy(rng->mt, key, sizeof(rng->mt));
    rng->pos = pos;
    return PPO2HIP_OK;
}
extern "C" int ppo2hip_rng_get_state(const ppo2hip_rng *rng, uint32_t *key, int *pos) {
    if (!rng || !key || !pos) {
        ppo2hip::set_error("rng_get_state: null pointer");
        return PPO2HIP_EINVAL;
    }
    std::memcpy(key, rng->mt, sizeof(rng->mt));
    *pos = rng->pos;
    return PPO2HIP_OK;
}
extern "C" int ppo2hip_shuffle_i64(ppo2hip_rng *rng, int64_t *inds, int64_t n) { return shuffle_impl(rng, inds, n); }
extern "C" int ppo2hip_shuffle_i32(ppo2hip_rng *rng, int32_t *inds, int64_t n) { return shuffle_impl(rng, inds, n); }
