// Global-norm clip + Adam over the flat parameter arena: one read of (theta, g, m, v), one
// write of (theta, m, v) — 28 B per parameter, plus a 4 B/param norm pass when clipping.
//   clip : tf.clip_by_global_norm, g * c / max(||g||, c)   (baselines/ppo2/common_functions.py:59-60,
//          live use baselines/a2c/a2c.py:40-42)
//   Adam : baselines/common/mpi_adam.py:30-35 with epsilon 1e-5 (common_functions.py:51-52).  As there, 1 - beta is
//          formed in double and then rounded to float32 (1.f - 0.999f would be off by 4.7e-5 relative)
#include <mutex>

#include "kernels.hpp"

namespace ppo2hip {

constexpr int kNormBlocks = 256;
constexpr int kScratchHeader = 64;   // bytes: [0] float global norm, [1] float applied scale

__global__ void sumsq_partial_kernel(const float *__restrict__ g, int64_t P, float gscale, double *__restrict__ partial) {
    __shared__ double red[17];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = (double)(g[i] * gscale);
        s += x * x;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// npartial == 0 with clipping on: the arena is small, so every block sums the squares of the
// WHOLE gradient itself (same order in every block, hence the same norm) and no separate
// norm launch is needed.  Loads are issued in batches so that a batch costs one round trip.
__global__ __launch_bounds__(1024) void adam_kernel(float *__restrict__ theta, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, int64_t P, float a, float beta1, float beta2, float omb1, float omb2, float eps,
                            float gscale, float max_norm, const double *__restrict__ partial, int npartial,
                            float *__restrict__ header) {
    __shared__ float scale_s;
    __shared__ double red[17];
    const int tid = threadIdx.x, nt = blockDim.x;
    // this block's own elements: in flight while the norm is being summed
    constexpr int U = 4;
    const int64_t first = (int64_t)blockIdx.x * nt * U;
    float gi[U], mi[U], vi[U], th[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t i = first + (int64_t)u * nt + tid;
        const bool ok = i < P;
        gi[u] = ok ? g[i] : 0.f;
        mi[u] = ok ? m[i] : 0.f;
        vi[u] = ok ? v[i] : 0.f;
        th[u] = ok ? theta[i] : 0.f;
    }
    double own = 0.0;
    if (max_norm > 0.f && npartial == 0) {
        constexpr int W = 16;
        for (int64_t base = 0; base < P; base += (int64_t)nt * W) {
            float x[W];
#pragma unroll
            for (int w = 0; w < W; w++) {
                const int64_t i = base + (int64_t)w * nt + tid;
                x[w] = i < P ? g[i] * gscale : 0.f;
            }
#pragma unroll
            for (int w = 0; w < W; w++) own += (double)x[w] * (double)x[w];
        }
        own = block_sum(own, red);
    }
    if (tid == 0) {
        float scale = gscale;
        if (max_norm > 0.f) {
            double ss = own;
            for (int b = 0; b < npartial; b++) ss += partial[b];
            const float norm = (float)sqrt(ss);
            scale = gscale * (max_norm / fmaxf(norm, max_norm));
            if (blockIdx.x == 0) header[0] = norm;
        }
        if (blockIdx.x == 0) header[1] = scale;
        scale_s = scale;
    }
    __syncthreads();
    const float scale = scale_s;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t i = first + (int64_t)u * nt + tid;
        if (i < P) {
            const float gs = gi[u] * scale;
            const float m1 = beta1 * mi[u] + omb1 * gs;
            const float v1 = beta2 * vi[u] + omb2 * (gs * gs);
            m[i] = m1;
            v[i] = v1;
            theta[i] = th[u] + (-a) * m1 / (sqrtf(v1) + eps);
        }
    }
}

}  // namespace ppo2hip

using namespace ppo2hip;

constexpr size_t kScratchBytes = kScratchHeader + kNormBlocks * sizeof(double);

// Scratch of ppo2hip_adam_step_static: the norm header and the per-block partial sums of squares, held by every
// GPU the library runs on until the process ends.  In host code this name only stands for the symbol; the
// address the kernels write through is looked up per device in static_scratch().
static __device__ __align__(256) unsigned char scratch[PPO2HIP_ADAM_STATIC_SCRATCH_BYTES];
static_assert(kScratchBytes <= PPO2HIP_ADAM_STATIC_SCRATCH_BYTES, "static scratch smaller than the kernels' layout");

constexpr int kMaxDevices = 64;
static void *g_static_ptr[kMaxDevices];          // device address of `scratch` on each GPU, once looked up
static std::mutex g_static_mu;

// address of this device's copy of `scratch` (symbol lookup on the first call per device; no allocation)
static int static_scratch(void **out) {
    int dev = -1;
    PPO2HIP_CHECK_HIP(hipGetDevice(&dev));
    PPO2HIP_REQUIRE(dev >= 0 && dev < kMaxDevices, "adam_step_static: device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_static_mu);
    if (!g_static_ptr[dev]) {
        void *p = nullptr;
        PPO2HIP_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        PPO2HIP_REQUIRE(p != nullptr, "adam_step_static: no device address for the scratch symbol");
        g_static_ptr[dev] = p;
    }
    *out = g_static_ptr[dev];
    return PPO2HIP_OK;
}

extern "C" size_t ppo2hip_adam_scratch_bytes(int64_t P) {
    (void)P;
    return kScratchBytes;
}

extern "C" int ppo2hip_adam_step_static(float *params, const float *grads, float *m, float *v, int64_t P, double lr,
                                        int64_t t, double beta1, double beta2, double epsilon, double grad_scale,
                                        double max_grad_norm, float *norm_out, void *stream) {
    PPO2HIP_REQUIRE(P >= 0 && P <= PPO2HIP_ADAM_STATIC_MAX_PARAMS,
                    "adam_step_static: P=%lld beyond the supported maximum of %lld parameters", (long long)P,
                    (long long)PPO2HIP_ADAM_STATIC_MAX_PARAMS);
    void *sp = nullptr;
    int rc = static_scratch(&sp);
    if (rc) return rc;
    // every call rewrites the header and the partial sums it reads, so nothing of the previous call's contents is used
    rc = ppo2hip_adam_step(params, grads, m, v, P, lr, t, beta1, beta2, epsilon, grad_scale, max_grad_norm, sp, stream);
    if (rc || !norm_out || P == 0) return rc;
    PPO2HIP_CHECK_HIP(hipMemcpyAsync(norm_out, sp, 2 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_adam_step(float *params, const float *grads, float *m, float *v, int64_t P, double lr,
                                 int64_t t, double beta1, double beta2, double epsilon, double grad_scale,
                                 double max_grad_norm, void *ws, void *stream) {
    PPO2HIP_REQUIRE(P >= 0 && t >= 1, "adam_step: bad P=%lld t=%lld", (long long)P, (long long)t);
    if (P == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(params && grads && m && v && ws, "adam_step: null pointer");
    hipStream_t s = (hipStream_t)stream;
    float *header = (float *)ws;
    double *partial = (double *)((char *)ws + kScratchHeader);
    int npartial = 0;
    if (max_grad_norm > 0.0 && P > 65536) {
        npartial = (int)(ceil_div(P, 1024) < kNormBlocks ? ceil_div(P, 1024) : kNormBlocks);
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3(npartial), dim3(256), 0, s, grads, P, (float)grad_scale, partial);
        int rc = check_launch("sumsq_partial_kernel");
        if (rc) return rc;
    }
    const double a = lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
    // small arenas (norm summed inside every block): 1024 threads, so that the norm pass is one batch of loads
    const int threads = npartial == 0 && max_grad_norm > 0.0 ? 1024 : 256;
    const int64_t nb = ceil_div(P, (int64_t)threads * 4);      // 4 elements per thread
    PPO2HIP_REQUIRE(nb <= 0x7fffffffll, "adam_step: arena too large");
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(threads), 0, s, params, grads, m, v, P, (float)a,
                       (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)epsilon, (float)grad_scale, (float)max_grad_norm, partial,
                       npartial, header);
    return check_launch("adam_kernel");
}
