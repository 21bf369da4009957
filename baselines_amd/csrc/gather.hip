// Minibatch gather: what `arr[mbinds]` does for the seven arrays at
// baselines/ppo2/ppo2.py:117, read straight out of the time-major rollout (the sf01
// flatten never materialises), fused with the advantage normalisation of
// baselines/ppo2/common_functions.py:67.
#include <mutex>

#include "kernels.hpp"

namespace ppo2hip {

static __device__ unsigned g_fault_bits[4];
static unsigned *g_fault_ptr[64];
static std::mutex g_fault_mu;

unsigned *fault_word() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        set_error("fault word: no current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_fault_mu);
    if (!g_fault_ptr[dev]) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fault_bits)) != hipSuccess || !p) {
            set_error("fault word: no device address for the symbol");
            return nullptr;
        }
        g_fault_ptr[dev] = (unsigned *)p;
    }
    return g_fault_ptr[dev];
}

// Compute units of the current device (cached per device): what a launch whose workgroups wait for each other may count
// on being resident together.  0 when the query fails.
int device_cu_count() {
    static int cached[64];
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cached[dev] = n;
    }
    return cached[dev];
}

// one 4-byte (or 1-byte) word of one observation row per loop trip
template <class W>
__global__ void gather_rows_kernel(const W *__restrict__ src, const int32_t *__restrict__ idx, int64_t B,
                                   int64_t row_words, int T, int E, W *__restrict__ dst, int64_t limit, unsigned *fault) {
    const int64_t total = B * row_words;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = g / row_words, w = g - i * row_words;
        const int64_t smp = idx ? checked_sample(idx[i], limit, fault) : i;
        dst[g] = src[tm_offset(smp, T, E) * row_words + w];
    }
}

// 16-byte words for big, aligned rows (image observations)
__global__ void gather_rows16_kernel(const uint4 *__restrict__ src, const int32_t *__restrict__ idx, int64_t B,
                                     int64_t row_words, int T, int E, uint4 *__restrict__ dst, int64_t limit, unsigned *fault) {
    const int64_t i = blockIdx.x;
    const int64_t smp = idx ? checked_sample(idx[i], limit, fault) : i;
    const uint4 *s = src + tm_offset(smp, T, E) * row_words;
    uint4 *d = dst + i * row_words;
    for (int64_t w = threadIdx.x; w < row_words; w += blockDim.x) d[w] = s[w];
}

struct SmallGather {
    const float *returns, *values, *neglogpacs;
    const void *advs;
    const uint8_t *masks;
    const void *actions;
    int adv_f64, T, E, nvalue, ac_words;
    const double *stats;
    float *o_returns, *o_values, *o_neglogpacs, *o_advs;
    uint8_t *o_masks;
    uint32_t *o_actions;
    int32_t *o_rows;      // optional: the rollout row of every minibatch sample
};

__global__ void gather_small_kernel(SmallGather g, const int32_t *__restrict__ idx, int64_t B, int64_t limit, unsigned *fault) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int64_t smp = idx ? checked_sample(idx[i], limit, fault) : i;
    const int64_t r = tm_offset(smp, g.T, g.E);
    for (int k = 0; k < g.nvalue; k++) {
        g.o_returns[i * g.nvalue + k] = g.returns[r * g.nvalue + k];
        g.o_values[i * g.nvalue + k] = g.values[r * g.nvalue + k];
    }
    g.o_neglogpacs[i] = g.neglogpacs[r];
    if (g.o_rows) g.o_rows[i] = (int32_t)r;
    double a = g.adv_f64 ? ((const double *)g.advs)[r] : (double)((const float *)g.advs)[r];
    if (g.stats) {
        // (a - mean) / (std + 1e-8), population std
        const double n = g.stats[2];
        const double mean = g.stats[0] / n;
        double var = g.stats[1] / n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        a = (a - mean) / (sqrt(var) + 1e-8);
    }
    g.o_advs[i] = (float)a;
    if (g.o_masks) g.o_masks[i] = g.masks[r];
    const uint32_t *act = (const uint32_t *)g.actions;
    for (int k = 0; k < g.ac_words; k++) g.o_actions[i * g.ac_words + k] = act[r * g.ac_words + k];
}

__global__ void gather_states_kernel(const float *__restrict__ states, const int32_t *__restrict__ envs,
                                     int nenv_mb, int width, float *__restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nenv_mb * width) return;
    const int e = (int)(g / width), w = (int)(g - (int64_t)e * width);
    out[g] = states[(int64_t)envs[e] * width + w];
}

int gather_minibatch(const GatherArgs &g, hipStream_t s) {
    const ppo2hip_rollout *ro = g.ro;
    const int64_t B = g.B;
    if (B <= 0) return 0;
    unsigned *fault = fault_word();
    if (!fault) return PPO2HIP_EINVAL;
    const int64_t limit = rollout_samples(*ro);
    int rc = 0;
    // observations (unless the consumer reads them where they lie)
    if (!g.obs_rows) {
    const bool a16 = (g.ob_bytes % 16 == 0) && ((uintptr_t)ro->obs % 16 == 0) && ((uintptr_t)g.obs % 16 == 0) && g.ob_bytes >= 1024;
    if (a16) {
        hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)B), dim3(256), 0, s, (const uint4 *)ro->obs, g.idx, B,
                           g.ob_bytes / 16, ro->T, ro->E, (uint4 *)g.obs, limit, fault);
    } else if (g.ob_bytes % 4 == 0) {
        const int64_t total = B * (g.ob_bytes / 4);
        const int64_t nb = ceil_div(total, 256) < 65536 ? ceil_div(total, 256) : 65536;
        hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3((unsigned)nb), dim3(256), 0, s, (const uint32_t *)ro->obs,
                           g.idx, B, g.ob_bytes / 4, ro->T, ro->E, (uint32_t *)g.obs, limit, fault);
    } else {
        const int64_t total = B * g.ob_bytes;
        const int64_t nb = ceil_div(total, 256) < 65536 ? ceil_div(total, 256) : 65536;
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, dim3((unsigned)nb), dim3(256), 0, s, (const uint8_t *)ro->obs,
                           g.idx, B, g.ob_bytes, ro->T, ro->E, (uint8_t *)g.obs, limit, fault);
    }
    rc = check_launch("gather_rows");
    if (rc) return rc;
    }
    SmallGather sg;
    sg.returns = ro->returns; sg.values = ro->values; sg.neglogpacs = ro->neglogpacs; sg.advs = ro->advs;
    sg.masks = ro->masks; sg.actions = ro->actions; sg.adv_f64 = ro->adv_f64; sg.T = ro->T; sg.E = ro->E;
    sg.nvalue = g.nvalue; sg.ac_words = (int)(g.ac_bytes / 4); sg.stats = g.stats;
    sg.o_returns = g.returns; sg.o_values = g.values; sg.o_neglogpacs = g.neglogpacs; sg.o_advs = g.advs;
    sg.o_masks = (g.masks && ro->masks) ? g.masks : nullptr; sg.o_actions = (uint32_t *)g.actions;
    sg.o_rows = g.obs_rows;
    hipLaunchKernelGGL(gather_small_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, s, sg, g.idx, B, limit, fault);
    return check_launch("gather_small");
}

int gather_states(const float *states, const int32_t *envs, int nenv_mb, int width, float *out, hipStream_t s) {
    const int64_t total = (int64_t)nenv_mb * width;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(gather_states_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, states, envs,
                       nenv_mb, width, out);
    return check_launch("gather_states");
}

}  // namespace ppo2hip

// ---------------------------------------------------------------------------------------
// Composition of permutations: the running minibatch order after one more epoch's shuffle
// (inds_e = inds_{e-1}[sigma_e], the in-place np.random.shuffle of baselines/ppo2/ppo2.py:113
// restated as a gather).
// ---------------------------------------------------------------------------------------
namespace ppo2hip {
// word 1 of the fault array receives what word 0 held; word 0 is cleared by the same atomic, so a bit raised by a kernel
// on another stream lands either in this read or in the next one, never in neither
__global__ void take_faults_kernel(unsigned *words) {
    if (threadIdx.x == 0 && blockIdx.x == 0) words[1] = atomicExch(&words[0], 0u);
}
// the same exchange with the result stored straight into page-locked host memory (no copy call on the host's side)
__global__ void post_faults_kernel(unsigned *words, unsigned *host_bits) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        __hip_atomic_store(host_bits, atomicExch(&words[0], 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void compose_kernel(const int32_t *__restrict__ prev, const int32_t *__restrict__ sigma,
                               int32_t *__restrict__ out, int64_t n, unsigned *fault) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    out[k] = prev[checked_sample(sigma[k], n, fault, kFaultComposeRange)];
}
}  // namespace ppo2hip

extern "C" int ppo2hip_compose_i32(const int32_t *prev, const int32_t *sigma, int32_t *out, int64_t n, void *stream) {
    using namespace ppo2hip;
    PPO2HIP_REQUIRE(n >= 0 && n <= 0x7fffffffll, "compose: bad n=%lld", (long long)n);
    if (n == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(prev && sigma && out && out != prev && out != sigma, "compose: null or aliased pointer");
    unsigned *fault = fault_word();
    if (!fault) return PPO2HIP_EINVAL;
    hipLaunchKernelGGL(compose_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, prev, sigma,
                       out, n, fault);
    return check_launch("compose_kernel");
}

extern "C" int ppo2hip_take_faults(uint32_t *bits, void *stream) {
    using namespace ppo2hip;
    PPO2HIP_REQUIRE(bits != nullptr, "take_faults: null pointer");
    unsigned *fault = fault_word();
    if (!fault) return PPO2HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    uint32_t host = 0;
    hipLaunchKernelGGL(take_faults_kernel, dim3(1), dim3(64), 0, s, fault);
    int rc = check_launch("take_faults_kernel");
    if (rc) return rc;
    PPO2HIP_CHECK_HIP(hipMemcpyAsync(&host, fault + 1, sizeof(host), hipMemcpyDeviceToHost, s));
    PPO2HIP_CHECK_HIP(hipStreamSynchronize(s));
    *bits = host;
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_post_faults(uint32_t *host_bits, void *stream) {
    using namespace ppo2hip;
    PPO2HIP_REQUIRE(host_bits != nullptr, "post_faults: null pointer");
    unsigned *fault = fault_word();
    if (!fault) return PPO2HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(post_faults_kernel, dim3(1), dim3(64), 0, s, fault, host_bits);
    return check_launch("post_faults_kernel");
}
