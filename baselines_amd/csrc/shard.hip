// Env-sharded data parallelism with ONE global permutation (every rank draws the same minibatch order over all ranks'
// samples, baselines/ppo2/ppo2.py:110-116, and works on the entries whose env it owns): the ownership filter and the
// advantage statistics of a rank's share, both over the device-resident global index list.
//
// A rank owns a contiguous block of envs, and the env-major flat sample number is env * nsteps + t, so "owned" is a
// contiguous range [lo, hi) of sample numbers and the local sample number is s - lo.
#include "kernels.hpp"

namespace ppo2hip {

constexpr int kFilterThreads = 256, kFilterPer = 8, kFilterChunk = kFilterThreads * kFilterPer;

// pass 1: how many entries of chunk c of minibatch j this rank owns
__global__ __launch_bounds__(kFilterThreads) void filter_count_kernel(const int32_t *__restrict__ idx, int64_t B, int32_t lo,
                                                                      int32_t hi, int32_t *__restrict__ chunk_counts) {
    __shared__ int red[17];
    const int j = blockIdx.y, c = blockIdx.x;
    const int64_t base = (int64_t)j * B, first = (int64_t)c * kFilterChunk + (int64_t)threadIdx.x * kFilterPer;
    int n = 0;
#pragma unroll
    for (int u = 0; u < kFilterPer; u++) {
        const int64_t i = first + u;
        if (i < B) {
            const int32_t s = idx[base + i];
            n += (s >= lo && s < hi) ? 1 : 0;
        }
    }
    n = block_sum(n, red);
    if (threadIdx.x == 0) chunk_counts[(int64_t)j * gridDim.x + c] = n;
}

// pass 2: the owned entries of minibatch j, re-based to the rank's own sample numbers, in list order, at
// out[j * out_stride ...]; up to 64 slots behind them are set to sample 0 (a consumer that rounds a count up to whole
// tiles then reads valid sample numbers); counts[j] = how many there are
__global__ __launch_bounds__(kFilterThreads) void filter_scatter_kernel(const int32_t *__restrict__ idx, int64_t B, int32_t lo,
                                                                        int32_t hi, const int32_t *__restrict__ chunk_counts,
                                                                        int32_t *__restrict__ out, int64_t out_stride,
                                                                        int32_t *__restrict__ counts, unsigned *fault) {
    __shared__ int red[17];
    __shared__ int wave_off[kFilterThreads / 64];
    const int j = blockIdx.y, c = blockIdx.x, nchunks = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // entries owned in the chunks before this one, and in all of them (a block sum of at most a few thousand counts)
    int before = 0, total = 0;
    for (int k = threadIdx.x; k < nchunks; k += kFilterThreads) {
        const int v = chunk_counts[(int64_t)j * nchunks + k];
        total += v;
        before += k < c ? v : 0;
    }
    before = block_sum(before, red);
    total = block_sum(total, red);
    const int64_t base = (int64_t)j * B, first = (int64_t)c * kFilterChunk + (int64_t)threadIdx.x * kFilterPer;
    int32_t smp[kFilterPer];
    unsigned mine = 0;          // bit u: entry u of this thread's run is owned
#pragma unroll
    for (int u = 0; u < kFilterPer; u++) {
        const int64_t i = first + u;
        smp[u] = i < B ? idx[base + i] : -1;
        mine |= (smp[u] >= lo && smp[u] < hi) ? 1u << u : 0u;
    }
    const int n = __popc(mine);
    // exclusive prefix of n over the block: within the wave by shuffles, across waves through LDS
    int incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_off[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += wave_off[w];
    const int64_t pos = (int64_t)before + woff + incl - n;
    int32_t *o = out + (int64_t)j * out_stride;
    int k = 0;
#pragma unroll
    for (int u = 0; u < kFilterPer; u++)
        if (mine >> u & 1u) {
            if (pos + k < out_stride) o[pos + k] = smp[u] - lo;
            k++;
        }
    if (c == 0 && threadIdx.x == 0) {
        counts[j] = total;
        if (total > out_stride) atomicOr(fault, kFaultIndexRange);      // the caller's list slots cannot hold the share
    }
    // padding, dealt over the minibatch's blocks
    const int64_t pad_end = out_stride < (int64_t)total + 64 ? out_stride : (int64_t)total + 64;
    for (int64_t k = (int64_t)total + (int64_t)c * kFilterThreads + threadIdx.x; k < pad_end; k += (int64_t)nchunks * kFilterThreads)
        o[k] = 0;
}

// {sum a, sum a^2, count} over this rank's share of chunk c of minibatch j of the GLOBAL list
template <class A>
__global__ void adv_stats_owned_partial_kernel(const A *__restrict__ advs, const int32_t *__restrict__ idx, int64_t B, int T, int E,
                                               int32_t lo, int32_t hi, double *__restrict__ partial) {
    __shared__ double red[17];
    const int j = blockIdx.y;
    const int64_t per = (B + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per;
    const int64_t b1 = b0 + per < B ? b0 + per : B;
    double s = 0.0, ss = 0.0, n = 0.0;
    constexpr int U = 4;      // rows per thread and round: index loads first, then the gathers they address
    for (int64_t base = b0; base < b1; base += (int64_t)blockDim.x * U) {
        int32_t smp[U];
        bool own[U];
        double a[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + (int64_t)u * blockDim.x + threadIdx.x;
            smp[u] = i < b1 ? idx[(int64_t)j * B + i] : -1;
            own[u] = smp[u] >= lo && smp[u] < hi;
        }
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = own[u] ? (double)advs[tm_offset((int64_t)(smp[u] - lo), T, E)] : 0.0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (own[u]) {
                s += a[u];
                ss += a[u] * a[u];
                n += 1.0;
            }
        }
    }
    s = block_sum(s, red);
    ss = block_sum(ss, red);
    n = block_sum(n, red);
    if (threadIdx.x == 0) {
        double *p = partial + ((int64_t)j * gridDim.x + blockIdx.x) * 3;
        p[0] = s; p[1] = ss; p[2] = n;
    }
}

// one wave per minibatch: lanes take chunks lane, lane + 64, ..., then a fixed-order wave sum
__global__ void adv_stats_owned_final_kernel(const double *__restrict__ partial, int nchunks, double *__restrict__ stats) {
    const int j = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, ss = 0.0, n = 0.0;
    for (int c = lane; c < nchunks; c += 64) {
        const double *p = partial + ((int64_t)j * nchunks + c) * 3;
        s += p[0]; ss += p[1]; n += p[2];
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    n = wave_sum(n);
    if (lane == 0) {
        stats[j * 3 + 0] = s;
        stats[j * 3 + 1] = ss;
        stats[j * 3 + 2] = n;
    }
}

constexpr int kOwnedStatChunks = 128;

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" size_t ppo2hip_filter_owned_workspace_bytes(int64_t nbatch_train, int nmb) {
    if (nbatch_train < 1 || nmb < 1) return 0;
    return (size_t)ceil_div(nbatch_train, kFilterChunk) * (size_t)nmb * sizeof(int32_t);
}

extern "C" int ppo2hip_filter_owned_i32(const int32_t *idx, int64_t nbatch_train, int nmb, int32_t lo, int32_t hi,
                                        int32_t *out, int64_t out_stride, int32_t *counts, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    PPO2HIP_REQUIRE(nbatch_train >= 1 && nmb >= 1 && nmb <= 65535 && out_stride >= 1 && lo >= 0 && hi >= lo,
                    "filter_owned: bad shape B=%lld nmb=%d stride=%lld range [%d, %d)", (long long)nbatch_train, nmb,
                    (long long)out_stride, lo, hi);
    PPO2HIP_REQUIRE(idx && out && counts && workspace, "filter_owned: null pointer");
    PPO2HIP_REQUIRE((int64_t)nmb * nbatch_train <= 0x7fffffffll, "filter_owned: list too long");
    if (workspace_bytes < ppo2hip_filter_owned_workspace_bytes(nbatch_train, nmb)) {
        set_error("filter_owned: workspace too small");
        return PPO2HIP_ENOSPACE;
    }
    unsigned *fault = fault_word();
    if (!fault) return PPO2HIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div(nbatch_train, kFilterChunk), (unsigned)nmb);
    hipLaunchKernelGGL(filter_count_kernel, grid, dim3(kFilterThreads), 0, s, idx, nbatch_train, lo, hi, (int32_t *)workspace);
    int rc = check_launch("filter_count_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(filter_scatter_kernel, grid, dim3(kFilterThreads), 0, s, idx, nbatch_train, lo, hi,
                       (const int32_t *)workspace, out, out_stride, counts, fault);
    return check_launch("filter_scatter_kernel");
}

extern "C" size_t ppo2hip_adv_stats_owned_workspace_bytes(int nmb) {
    return (size_t)(nmb > 0 ? nmb : 1) * kOwnedStatChunks * 3 * sizeof(double);
}

extern "C" int ppo2hip_adv_stats_owned(const void *advs, int adv_f64, const int32_t *idx, int64_t nbatch_train, int nmb,
                                       int32_t lo, int32_t hi, int T, int E, void *workspace, double *stats, void *stream) {
    PPO2HIP_REQUIRE(nbatch_train >= 1 && nmb >= 1 && nmb <= 65535, "adv_stats_owned: bad shape B=%lld nmb=%d",
                    (long long)nbatch_train, nmb);
    PPO2HIP_REQUIRE(advs && idx && stats && workspace, "adv_stats_owned: null pointer");
    PPO2HIP_REQUIRE(T > 0 && E > 0 && lo >= 0 && hi >= lo && (int64_t)hi - lo <= (int64_t)T * E,
                    "adv_stats_owned: the owned range [%d, %d) must fit the rank's %d x %d rollout", lo, hi, T, E);
    hipStream_t s = (hipStream_t)stream;
    double *partial = (double *)workspace;
    const int nchunks = (int)(ceil_div(nbatch_train, 1024) < kOwnedStatChunks ? ceil_div(nbatch_train, 1024) : kOwnedStatChunks);
    const dim3 grid(nchunks, nmb);
    if (adv_f64)
        hipLaunchKernelGGL(adv_stats_owned_partial_kernel<double>, grid, dim3(256), 0, s, (const double *)advs, idx, nbatch_train,
                           T, E, lo, hi, partial);
    else
        hipLaunchKernelGGL(adv_stats_owned_partial_kernel<float>, grid, dim3(256), 0, s, (const float *)advs, idx, nbatch_train,
                           T, E, lo, hi, partial);
    int rc = check_launch("adv_stats_owned_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(adv_stats_owned_final_kernel, dim3((unsigned)nmb), dim3(64), 0, s, partial, nchunks, stats);
    return check_launch("adv_stats_owned_final_kernel");
}
