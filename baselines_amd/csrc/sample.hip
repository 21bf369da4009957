// The act half of a rollout step on the device: action sampling with its neglogp, and the fork's
// done-bootstrap of the rewards.  `a0 = pd.sample(); neglogp0 = pd.neglogp(a0)`
// (baselines/ppo2/policies.py:168-170, baselines/a2c/policies.py:137-139) with
// DiagGaussianPd.sample = mean + std * N(0,1) (baselines/common/distributions.py:211-212) and
// CategoricalPd.sample = argmax(logits - log(-log(U))) (:164-166).
//
// Noise: Philox4x32-10 (Salmon et al., SC'11), one call per (row, group of four elements), keyed by the
// caller's seed, counter = (row, group, step).  No state lives on the device: a (seed, step) pair names
// the noise of a whole rollout step, whatever the launch geometry.
#include "kernels.hpp"

namespace ppo2hip {

struct Philox {
    uint32_t k0, k1;
    __device__ inline void run(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            c0 = hi1 ^ c1 ^ a;
            c1 = lo1;
            c2 = hi0 ^ c3 ^ b;
            c3 = lo0;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

// U(0,1), open at both ends, 24 bits
__device__ inline float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f; }

// four uniforms -> four normals (Box-Muller on the pairs (0,1) and (2,3): cosine branch first)
__device__ inline void normals4(const uint32_t x[4], float z[4]) {
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const float r = sqrtf(-2.f * logf(u01(x[2 * p])));
        float sn, cs;
        sincospif(2.f * u01(x[2 * p + 1]), &sn, &cs);
        z[2 * p] = r * cs;
        z[2 * p + 1] = r * sn;
    }
}

constexpr int kMaxSampleDim = 64;

// one thread per row
__global__ void sample_actions_kernel(int dist, int A, const float *__restrict__ pi, const float *__restrict__ logstd,
                                      int64_t B, Philox gen, uint32_t step_lo, uint32_t step_hi, int deterministic,
                                      void *__restrict__ actions, float *__restrict__ neglogp,
                                      float *__restrict__ noise_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const float *row = pi + i * A;
    if (dist == PPO2HIP_DIST_GAUSSIAN) {
        float *act = (float *)actions + i * A;
        float ss = 0.f, sl = 0.f;
        for (int g = 0; g < A; g += 4) {
            uint32_t x[4];
            float z[4] = {0.f, 0.f, 0.f, 0.f};
            if (!deterministic) {
                gen.run((uint32_t)i, (uint32_t)(g >> 2), step_lo, step_hi, x);
                normals4(x, z);
            }
            for (int k = g; k < g + 4 && k < A; k++) {
                const float ls = logstd[k], sd = expf(ls), m = row[k];
                const float a = deterministic ? m : m + sd * z[k - g];
                const float zq = (a - m) / sd;        // neglogp is evaluated on the stored action (:203)
                ss += zq * zq;
                sl += ls;
                act[k] = a;
                if (noise_out) noise_out[i * A + k] = z[k - g];
            }
        }
        neglogp[i] = 0.5f * ss + kHalfLog2Pi * (float)A + sl;
    } else {
        float mx = row[0];
        for (int k = 1; k < A; k++) mx = fmaxf(mx, row[k]);
        float z0 = 0.f, best = 0.f;
        int arg = 0;
        for (int g = 0; g < A; g += 4) {
            uint32_t x[4] = {0u, 0u, 0u, 0u};
            if (!deterministic) gen.run((uint32_t)i, (uint32_t)(g >> 2), step_lo, step_hi, x);
            for (int k = g; k < g + 4 && k < A; k++) {
                const float l = row[k];
                z0 += expf(l - mx);
                const float u = u01(x[k - g]);
                const float score = deterministic ? l : l - logf(-logf(u));
                if (k == 0 || score > best) { best = score; arg = k; }   // first maximum, as tf.argmax
                if (noise_out) noise_out[i * A + k] = deterministic ? 0.f : u;
            }
        }
        ((int32_t *)actions)[i] = arg;
        neglogp[i] = logf(z0) - (row[arg] - mx);
    }
}

// baselines/ppo2/common_functions.py:265-276, one thread per env (the branch looks at channel 0 before the
// row changes, so a row is one thread's work); float64 product, float32 store
__global__ void done_bootstrap_kernel(float *__restrict__ rewards, const uint8_t *__restrict__ dones,
                                      const float *__restrict__ v_reset, GammaVec gamma, int64_t E, int R) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || !dones[e]) return;
    float *rw = rewards + e * R;
    const float *v = v_reset + e * R;
    const int from = rw[0] > 0.f ? 0 : 2;
    for (int r = from; r < R; r++) rw[r] = (float)__dadd_rn((double)rw[r], __dmul_rn(gamma.g[r], (double)v[r]));   // built without fma contraction: NumPy rounds the product
}

int sample_rows(int dist, int ac_dim, const float *pi, const float *logstd, int64_t B, uint64_t seed, uint64_t step,
                int deterministic, void *actions, float *neglogp, float *noise_out, hipStream_t s) {
    if (B <= 0) return 0;
    Philox gen{(uint32_t)seed, (uint32_t)(seed >> 32)};
    hipLaunchKernelGGL(sample_actions_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, s, dist, ac_dim, pi, logstd, B,
                       gen, (uint32_t)step, (uint32_t)(step >> 32), deterministic, actions, neglogp, noise_out);
    return check_launch("sample_actions_kernel");
}

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" int ppo2hip_done_bootstrap(float *rewards, const uint8_t *dones, const float *v_reset, const double *gamma,
                                      int64_t E, int R, void *stream) {
    PPO2HIP_REQUIRE(E >= 0 && R >= 1 && R <= kMaxR, "done_bootstrap: bad shape E=%lld R=%d", (long long)E, R);
    if (E == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(rewards && dones && v_reset && gamma, "done_bootstrap: null pointer");
    GammaVec gv;
    for (int i = 0; i < kMaxR; i++) gv.g[i] = i < R ? gamma[i] : 0.0;
    hipLaunchKernelGGL(done_bootstrap_kernel, dim3((unsigned)ceil_div(E, 256)), dim3(256), 0, (hipStream_t)stream,
                       rewards, dones, v_reset, gv, E, R);
    return check_launch("done_bootstrap_kernel");
}
