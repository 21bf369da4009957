// Advantage post-processing: the fork's 5-channel scalarisation and the sums behind the
// per-minibatch normalisation.  All reductions are fixed-shape (per-block partials, then
// a single ordered pass), so results are reproducible run to run.
#include "common.hpp"
#include "kernels.hpp"

namespace ppo2hip {

constexpr int kAdvsBlocks = 512;   // partial-sum rows of compute_advs pass 1
constexpr int kStatChunks = 128;   // partial-sum chunks per minibatch of adv_stats
struct RVec { double v[kMaxR]; };

// pass 1: sum over rows of the float32 unit vectors dir_v / (|dir_v| + 1e-8)
// (baselines/ppo2/common_functions.py:361-367), one partial row per block
__global__ void advs_dir_partial_kernel(const float *__restrict__ values, RVec max_return, int64_t N,
                                        int R, double *__restrict__ partial /* [gridDim.x][kMaxR] */) {
    __shared__ double red[17];
    double acc[kMaxR];
#pragma unroll
    for (int k = 0; k < kMaxR; k++) acc[k] = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (int64_t)gridDim.x * blockDim.x) {
        double d[kMaxR];
        double ss = 0.0;
#pragma unroll
        for (int k = 0; k < kMaxR; k++)
            if (k < R) {
                d[k] = max_return.v[k] - (double)values[t * R + k];
                ss += d[k] * d[k];
            }
        const double len = sqrt(ss) + 1e-8;
#pragma unroll
        for (int k = 0; k < kMaxR; k++)
            if (k < R) acc[k] += (double)(float)(d[k] / len);   // stored as float32 (:364,:366)
    }
#pragma unroll
    for (int k = 0; k < kMaxR; k++) {
        const double s = block_sum(acc[k], red);
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * kMaxR + k] = s;
    }
}

// pass 2: advs[t] = sum_k(advs_ori[t,k] * L) / (L + 1e-8) with L = |mean_dir| (:368,:392,:395)
__global__ void advs_scalar_kernel(const float *__restrict__ advs_ori, const double *__restrict__ partial,
                                   int npartial, int64_t N, int R, double *__restrict__ out) {
    __shared__ float Ls;
    if (threadIdx.x == 0) {
        float ss = 0.f;
        for (int k = 0; k < R; k++) {
            double s = 0.0;
            for (int b = 0; b < npartial; b++) s += partial[(int64_t)b * kMaxR + k];
            const float m = (float)(s / (double)N);          // float32 mean (:367)
            ss = __fadd_rn(ss, __fmul_rn(m, m));
        }
        Ls = sqrtf(ss);
    }
    __syncthreads();
    const float L = Ls;
    const double denom = (double)L + 1e-8;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < R; k++) s = __fadd_rn(s, __fmul_rn(advs_ori[t * R + k], L));
        out[t] = (double)s / denom;
    }
}

// Partial sums of a and a^2 over chunk c of minibatch j (grid = kStatChunks x nmb).
template <class A>
__global__ void adv_stats_partial_kernel(const A *__restrict__ advs, const int32_t *__restrict__ idx,
                                         int64_t B, int T, int E, double *__restrict__ partial, int64_t limit,
                                         unsigned *fault) {
    __shared__ double red[17];
    const int j = blockIdx.y;
    const int64_t per = (B + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = lo + per < B ? lo + per : B;
    double s = 0.0, ss = 0.0;
    constexpr int U = 4;      // rows per thread and round: index loads first, then the gathers they address
    for (int64_t base = lo; base < hi; base += (int64_t)blockDim.x * U) {
        int64_t smp[U];
        double a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + (int64_t)u * blockDim.x + threadIdx.x;
            const int64_t row = (int64_t)j * B + (i < hi ? i : lo);
            smp[u] = idx ? checked_sample(idx[row], limit, fault) : row;
        }
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = (double)advs[tm_offset(smp[u], T, E)];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + (int64_t)u * blockDim.x + threadIdx.x;
            if (i < hi) {
                s += a[u];
                ss += a[u] * a[u];
            }
        }
    }
    s = block_sum(s, red);
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
        partial[((int64_t)j * gridDim.x + blockIdx.x) * 2 + 0] = s;
        partial[((int64_t)j * gridDim.x + blockIdx.x) * 2 + 1] = ss;
    }
}

// one wave per minibatch: lanes take chunks lane, lane + 64, ..., then a fixed-order wave sum
__global__ void adv_stats_final_kernel(const double *__restrict__ partial, int nchunks, int64_t B, int nmb,
                                       double *__restrict__ stats) {
    const int j = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, ss = 0.0;
    for (int c = lane; c < nchunks; c += 64) {
        s += partial[((int64_t)j * nchunks + c) * 2 + 0];
        ss += partial[((int64_t)j * nchunks + c) * 2 + 1];
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) {
        stats[j * 3 + 0] = s;
        stats[j * 3 + 1] = ss;
        stats[j * 3 + 2] = (double)B;
    }
}

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" size_t ppo2hip_compute_advs_workspace_bytes(int64_t N, int R) {
    (void)N; (void)R;
    return (size_t)kAdvsBlocks * kMaxR * sizeof(double);
}

extern "C" int ppo2hip_compute_advs(const float *values, const float *advs_ori, const double *max_return,
                                    int64_t N, int R, void *workspace, size_t workspace_bytes, double *out,
                                    void *stream) {
    PPO2HIP_REQUIRE(N >= 0 && R >= 1 && R <= kMaxR, "compute_advs: bad shape N=%lld R=%d", (long long)N, R);
    if (N == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(values && advs_ori && max_return && out && workspace, "compute_advs: null pointer");
    if (workspace_bytes < ppo2hip_compute_advs_workspace_bytes(N, R)) {
        set_error("compute_advs: workspace too small");
        return PPO2HIP_ENOSPACE;
    }
    RVec mr;
    for (int k = 0; k < kMaxR; k++) mr.v[k] = k < R ? max_return[k] : 0.0;
    const int nblk = (int)(ceil_div(N, 256) < kAdvsBlocks ? ceil_div(N, 256) : kAdvsBlocks);
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)workspace;
    hipLaunchKernelGGL(advs_dir_partial_kernel, dim3(nblk), dim3(256), 0, s, values, mr, N, R, partial);
    int rc = check_launch("advs_dir_partial_kernel");
    if (rc) return rc;
    const int nblk2 = (int)(ceil_div(N, 256) < 2048 ? ceil_div(N, 256) : 2048);
    hipLaunchKernelGGL(advs_scalar_kernel, dim3(nblk2), dim3(256), 0, s, advs_ori, partial, nblk, N, R, out);
    return check_launch("advs_scalar_kernel");
}

extern "C" size_t ppo2hip_adv_stats_workspace_bytes(int nmb) {
    return (size_t)(nmb > 0 ? nmb : 1) * kStatChunks * 2 * sizeof(double);
}

extern "C" int ppo2hip_adv_stats(const void *advs, int adv_f64, const int32_t *idx, int64_t B, int nmb,
                                 int T, int E, void *workspace, double *stats, void *stream) {
    PPO2HIP_REQUIRE(B >= 1 && nmb >= 1 && nmb <= 65535, "adv_stats: bad shape B=%lld nmb=%d", (long long)B, nmb);
    PPO2HIP_REQUIRE(advs && stats && workspace, "adv_stats: null pointer");
    PPO2HIP_REQUIRE((T == 0) || (T > 0 && E > 0), "adv_stats: bad rollout shape T=%d E=%d", T, E);
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)workspace;
    int nchunks = (int)(ceil_div(B, 1024) < kStatChunks ? ceil_div(B, 1024) : kStatChunks);
    dim3 grid(nchunks, nmb);
    unsigned *fault = fault_word();
    if (!fault) return PPO2HIP_EINVAL;
    // time-major arrays say how many samples there are; for flat rows (T = 0) the caller's list is taken as it is
    const int64_t limit = T > 0 ? (int64_t)T * E : (int64_t)1 << 62;
    if (adv_f64)
        hipLaunchKernelGGL(adv_stats_partial_kernel<double>, grid, dim3(256), 0, s, (const double *)advs, idx, B, T, E, partial, limit, fault);
    else
        hipLaunchKernelGGL(adv_stats_partial_kernel<float>, grid, dim3(256), 0, s, (const float *)advs, idx, B, T, E, partial, limit, fault);
    int rc = check_launch("adv_stats_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(adv_stats_final_kernel, dim3((unsigned)nmb), dim3(64), 0, s, partial, nchunks, B, nmb, stats);
    return check_launch("adv_stats_final_kernel");
}
