/* Plain-C restatement of the integer and scan parts of the PPO2 hot path.
 *
 * TEST INFRASTRUCTURE ONLY (checker for the HIP path and CPU baseline leg of
 * bench.py); the product never links or loads this file.
 *
 *  oracle_shuffle_epochs : np.random.seed(s) + np.random.shuffle, as called at
 *                          baselines/ppo2/ppo2.py:110-113 (legacy RandomState:
 *                          MT19937 init_genrand, Fisher-Yates with masked
 *                          rejection draws).
 *  oracle_gae            : baselines/ppo2/common_functions.py:308-328
 *                          (f32 in, f64 carry, f32 store, returns = adv + value in f32).
 *  oracle_discount       : baselines/a2c/utils.py:154-160 with the bootstrap rule
 *                          of baselines/a2c/a2c.py:119-126.
 *
 * Build: gcc -O2 -shared -fPIC -o oracle/_build/liboracle.so oracle/scan_perm_c.c
 */
#include <stdint.h>
#include <stddef.h>

#define MT_N 624
#define MT_M 397

typedef struct { uint32_t mt[MT_N]; int pos; } mt_state;

static void mt_seed(mt_state *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < MT_N; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->pos = MT_N;
}

static void mt_regen(mt_state *s) {
    uint32_t *mt = s->mt;
    for (int k = 0; k < MT_N; k++) {
        uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % MT_N] & 0x7fffffffu);
        mt[k] = mt[(k + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->pos = 0;
}

static uint32_t mt_next(mt_state *s) {
    if (s->pos >= MT_N) mt_regen(s);
    uint32_t y = s->mt[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static uint32_t mt_interval(mt_state *s, uint32_t max) {
    if (max == 0) return 0;
    uint32_t mask = max, v;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    while ((v = (mt_next(s) & mask)) > max) {}
    return v;
}

/* inds_out: [nepochs][n] int64; epoch e is the in-place shuffle of epoch e-1 */
void oracle_shuffle_epochs(uint32_t seed, int64_t n, int nepochs, int64_t *inds_out) {
    mt_state s;
    mt_seed(&s, seed);
    for (int64_t i = 0; i < n; i++) inds_out[i] = i;
    for (int e = 0; e < nepochs; e++) {
        int64_t *x = inds_out + (size_t)e * n;
        if (e > 0) for (int64_t i = 0; i < n; i++) x[i] = x[i - n];
        for (int64_t i = n - 1; i > 0; i--) {
            int64_t j = mt_interval(&s, (uint32_t)i);
            int64_t t = x[i]; x[i] = x[j]; x[j] = t;
        }
    }
}

void oracle_gae(const float *rewards, const float *values, const uint8_t *dones,
                const float *last_values, const uint8_t *last_dones, const double *gamma,
                double lam, int T, int E, int R, float *advs, float *returns) {
    for (int e = 0; e < E; e++)
        for (int r = 0; r < R; r++) {
            double last = 0.0;
            for (int t = T - 1; t >= 0; t--) {
                size_t i = ((size_t)t * E + e) * R + r;
                double nnt, nv;
                if (t == T - 1) { nnt = 1.0 - (double)(last_dones[e] != 0); nv = last_values[(size_t)e * R + r]; }
                else { nnt = 1.0 - (double)(dones[(size_t)(t + 1) * E + e] != 0); nv = values[i + (size_t)E * R]; }
                double delta = (double)rewards[i] + gamma[r] * nv * nnt - (double)values[i];
                last = delta + gamma[r] * lam * nnt * last;
                advs[i] = (float)last;
                returns[i] = advs[i] + values[i];
            }
        }
}

/* rewards/dones/out: env-major [E][T]; dones = done flag AFTER each step */
void oracle_discount(const float *rewards, const uint8_t *dones, const float *last_values,
                     double gamma, int E, int T, float *out) {
    for (int e = 0; e < E; e++) {
        const float *rw = rewards + (size_t)e * T;
        const uint8_t *dn = dones + (size_t)e * T;
        double r = 0.0;
        if (dn[T - 1] == 0) r = (double)last_values[e] + gamma * 0.0 * (1.0 - 0.0);
        for (int t = T - 1; t >= 0; t--) {
            r = (double)rw[t] + gamma * r * (1.0 - (double)(dn[t] != 0));
            out[(size_t)e * T + t] = (float)r;
        }
    }
}
