// Per-step filters of the vectorised env wrappers, kept on the device next to the rollout:
// RunningMeanStd (baselines/common/running_mean_std.py:9-29), the observation / return filters of
// VecNormalize (baselines/common/vec_env/vec_normalize.py:19-40) and the VecFrameStack roll
// (baselines/common/vec_env/vec_frame_stack.py:19-26).  All of it is HBM-bound streaming work;
// the statistics are float64 throughout, as the reference keeps them.
#include "kernels.hpp"

namespace ppo2hip {

constexpr int kRmsThreads = 256;
constexpr int kRmsMaxSlabs = 1024;

struct RmsGrid {
    int cw;        // columns per block
    int rpi;       // rows a block covers per trip
    int ctiles;    // column tiles
    int nslabs;    // row slabs
};

static RmsGrid rms_grid(int64_t n, int64_t d) {
    RmsGrid g;
    g.cw = (int)(d < kRmsThreads ? d : kRmsThreads);
    g.rpi = kRmsThreads / g.cw;
    g.ctiles = (int)ceil_div(d, g.cw);
    int64_t want = ceil_div(n, (int64_t)g.rpi * 8);            // at least eight trips per slab
    int64_t cap = kRmsMaxSlabs / g.ctiles;
    if (cap < 1) cap = 1;
    g.nslabs = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    return g;
}

// Sums of (x - K) and (x - K)^2 per column over one slab of rows, K = the running mean: the shift keeps
// the one-pass variance free of cancellation once the running mean has settled near the data.
template <class X>
__global__ __launch_bounds__(kRmsThreads) void rms_moments_kernel(const X *__restrict__ x, int64_t n, int64_t d,
                                                                  const double *__restrict__ shift, int cw, int rpi,
                                                                  double *__restrict__ partial /* [nslabs][2][d] */) {
    __shared__ double red[2][kRmsThreads];
    const int rl = threadIdx.x / cw, cl = threadIdx.x - rl * cw;
    const int64_t c = (int64_t)blockIdx.x * cw + cl;
    const bool live = rl < rpi && c < d;
    const int64_t rows_per_slab = (n + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
    const int64_t r1 = r0 + rows_per_slab < n ? r0 + rows_per_slab : n;
    double s1 = 0.0, s2 = 0.0;
    if (live) {
        const double k = shift[c];
        int64_t r = r0 + rl;
        for (; r + 3 * (int64_t)rpi < r1; r += 4 * (int64_t)rpi) {
            X v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = x[(r + (int64_t)u * rpi) * d + c];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double t = (double)v[u] - k;
                s1 += t;
                s2 += t * t;
            }
        }
        for (; r < r1; r += rpi) {
            const double t = (double)x[r * d + c] - k;
            s1 += t;
            s2 += t * t;
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < cw && c < d) {
        double a = 0.0, b = 0.0;
        for (int j = 0; j < rpi; j++) {
            a += red[0][j * cw + threadIdx.x];
            b += red[1][j * cw + threadIdx.x];
        }
        double *p = partial + (int64_t)blockIdx.y * 2 * d;
        p[c] = a;
        p[d + c] = b;
    }
}

// One wave per column: fold the slabs, then RunningMeanStd.update_from_moments
// (baselines/common/running_mean_std.py:15-29) term by term.  state = mean[d], var[d], count[d]
// (the scalar count is kept once per column so that no column reads what another one writes).
__global__ __launch_bounds__(64) void rms_merge_kernel(const double *__restrict__ partial, int nslabs, int64_t n,
                                                       int64_t d, double *__restrict__ state) {
    const int64_t c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int s = threadIdx.x; s < nslabs; s += 64) {
        a += partial[(int64_t)s * 2 * d + c];
        b += partial[(int64_t)s * 2 * d + d + c];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x != 0) return;
    const double mean = state[c], var = state[d + c], count = state[2 * d + c];
    const double batch_count = (double)n;
    const double m1 = a / batch_count;
    const double batch_mean = mean + m1;                       // the shift was the running mean
    double batch_var = b / batch_count - m1 * m1;
    batch_var = batch_var > 0.0 ? batch_var : 0.0;
    const double delta = batch_mean - mean;
    const double tot_count = count + batch_count;
    const double new_mean = mean + delta * batch_count / tot_count;
    const double m_a = var * count;
    const double m_b = batch_var * batch_count;
    const double M2 = m_a + m_b + delta * delta * count * batch_count / (count + batch_count);
    state[c] = new_mean;
    state[d + c] = M2 / (count + batch_count);
    state[2 * d + c] = batch_count + count;
}

// clip((x - mean) / sqrt(var + eps), -clip, clip); mean == nullptr leaves x uncentred (the reward filter)
template <class X, class Y>
__global__ void normalize_clip_kernel(const X *__restrict__ x, int64_t total, int64_t d, const double *__restrict__ mean,
                                      const double *__restrict__ var, double eps, double clip, Y *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = i % d;
        double v = (double)x[i];
        if (mean) v -= mean[c];
        v = v / sqrt(var[c] + eps);
        v = v < -clip ? -clip : (v > clip ? clip : v);
        out[i] = (Y)v;
    }
}

// self.ret = self.ret * self.gamma + rews (vec_normalize.py:27)
template <class X>
__global__ void discounted_return_kernel(const X *__restrict__ rews, int64_t E, double gamma, double *__restrict__ ret) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < E) ret[i] = ret[i] * gamma + (double)rews[i];
}

// One thread per pixel.  np.roll(stackedobs, shift=-1, axis=-1) moves the last axis by ONE ELEMENT
// (vec_frame_stack.py:21), which is one frame for single-channel frames only; for c > 1 the reference
// still shifts by one element before it overwrites the last c places, and so does this kernel.
// Envs whose episode just ended start from zeros (:22-24).
template <class W>
__global__ void frame_stack_kernel(W *__restrict__ stacked, const W *__restrict__ obs, const uint8_t *__restrict__ news,
                                   int64_t E, int64_t pixels, int c, int nstack) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= E * pixels) return;
    const int64_t e = p / pixels;
    const bool fresh = news && news[e] != 0;
    W *row = stacked + p * (int64_t)c * nstack;
    const W *o = obs + p * c;
    const int keep = c * (nstack - 1);
    for (int k = 0; k < keep; k++) row[k] = fresh ? (W)0 : row[k + 1];
    for (int k = 0; k < c; k++) row[keep + k] = o[k];
}

// Plain column sums and sums of squares (x cast to float64 first), plus the row count: the vector the MPI running
// statistics all-reduce (baselines/common/mpi_running_mean_std.py:37-42).  One wave per column folds the slabs.
__global__ __launch_bounds__(64) void moments_fold_kernel(const double *__restrict__ partial, int nslabs, int64_t n,
                                                          int64_t d, double *__restrict__ out /* [2d + 1] */) {
    const int64_t c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int s = threadIdx.x; s < nslabs; s += 64) {
        a += partial[(int64_t)s * 2 * d + c];
        b += partial[(int64_t)s * 2 * d + d + c];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x != 0) return;
    out[c] = a;
    out[d + c] = b;
    if (c == 0) out[2 * d] = (double)n;
}

__global__ void add_f64_kernel(double *__restrict__ dst, const double *__restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

static int rms_update_impl(const void *x, int x_f64, int64_t n, int64_t d, double *state, void *workspace,
                           size_t workspace_bytes, hipStream_t s) {
    const RmsGrid g = rms_grid(n, d);
    Carver cv(workspace, workspace_bytes);
    double *partial = cv.take<double>((size_t)g.nslabs * 2 * d);
    if (!cv.ok || !workspace) {
        set_error("rms_update: workspace too small (%zu bytes given, %zu needed)", workspace_bytes, cv.used());
        return PPO2HIP_EINVAL;
    }
    const dim3 grid((unsigned)g.ctiles, (unsigned)g.nslabs);
    if (x_f64)
        hipLaunchKernelGGL(rms_moments_kernel<double>, grid, dim3(kRmsThreads), 0, s, (const double *)x, n, d, state, g.cw,
                           g.rpi, partial);
    else
        hipLaunchKernelGGL(rms_moments_kernel<float>, grid, dim3(kRmsThreads), 0, s, (const float *)x, n, d, state, g.cw,
                           g.rpi, partial);
    int rc = check_launch("rms_moments_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(rms_merge_kernel, dim3((unsigned)d), dim3(64), 0, s, partial, g.nslabs, n, d, state);
    return check_launch("rms_merge_kernel");
}

static int normalize_impl(const void *x, int x_f64, int64_t n, int64_t d, const double *mean, const double *var,
                          double eps, double clip, void *out, int out_f64, hipStream_t s) {
    const int64_t total = n * d;
    const int64_t nb = ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192;
    const dim3 grid((unsigned)nb), block(256);
    if (x_f64 && out_f64)
        hipLaunchKernelGGL((normalize_clip_kernel<double, double>), grid, block, 0, s, (const double *)x, total, d, mean, var,
                           eps, clip, (double *)out);
    else if (x_f64)
        hipLaunchKernelGGL((normalize_clip_kernel<double, float>), grid, block, 0, s, (const double *)x, total, d, mean, var,
                           eps, clip, (float *)out);
    else if (out_f64)
        hipLaunchKernelGGL((normalize_clip_kernel<float, double>), grid, block, 0, s, (const float *)x, total, d, mean, var,
                           eps, clip, (double *)out);
    else
        hipLaunchKernelGGL((normalize_clip_kernel<float, float>), grid, block, 0, s, (const float *)x, total, d, mean, var,
                           eps, clip, (float *)out);
    return check_launch("normalize_clip_kernel");
}

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" size_t ppo2hip_rms_workspace_bytes(int64_t n, int64_t d) {
    if (n <= 0 || d <= 0) return 256;
    const RmsGrid g = rms_grid(n, d);
    return align_up((size_t)g.nslabs * 2 * (size_t)d * sizeof(double), 256);
}

extern "C" int ppo2hip_rms_update(const void *x, int x_f64, int64_t n, int64_t d, double *state, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    PPO2HIP_REQUIRE(n >= 0 && d >= 1 && d <= 0x7fffffffll, "rms_update: bad shape n=%lld d=%lld", (long long)n, (long long)d);
    // np.mean over an empty batch is NaN in the reference; refuse instead of poisoning the statistics
    PPO2HIP_REQUIRE(n > 0, "rms_update: empty batch");
    PPO2HIP_REQUIRE(x && state, "rms_update: null pointer");
    return rms_update_impl(x, x_f64, n, d, state, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ppo2hip_column_moments(const void *x, int x_f64, int64_t n, int64_t d, const double *zeros, double *out,
                                      void *workspace, size_t workspace_bytes, void *stream) {
    PPO2HIP_REQUIRE(n >= 1 && d >= 1 && d <= 0x7fffffffll, "column_moments: bad shape n=%lld d=%lld", (long long)n, (long long)d);
    PPO2HIP_REQUIRE(x && zeros && out, "column_moments: null pointer");
    const RmsGrid g = rms_grid(n, d);
    Carver cv(workspace, workspace_bytes);
    double *partial = cv.take<double>((size_t)g.nslabs * 2 * d);
    if (!cv.ok || !workspace) {
        set_error("column_moments: workspace too small (%zu bytes given, %zu needed)", workspace_bytes, cv.used());
        return PPO2HIP_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)g.ctiles, (unsigned)g.nslabs);
    if (x_f64)
        hipLaunchKernelGGL(rms_moments_kernel<double>, grid, dim3(kRmsThreads), 0, s, (const double *)x, n, d, zeros, g.cw,
                           g.rpi, partial);
    else
        hipLaunchKernelGGL(rms_moments_kernel<float>, grid, dim3(kRmsThreads), 0, s, (const float *)x, n, d, zeros, g.cw,
                           g.rpi, partial);
    int rc = check_launch("rms_moments_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(moments_fold_kernel, dim3((unsigned)d), dim3(64), 0, s, partial, g.nslabs, n, d, out);
    return check_launch("moments_fold_kernel");
}

extern "C" int ppo2hip_add_f64(double *dst, const double *src, int64_t n, void *stream) {
    PPO2HIP_REQUIRE(n >= 0, "add_f64: bad n");
    if (n == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(dst && src, "add_f64: null pointer");
    hipLaunchKernelGGL(add_f64_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    return check_launch("add_f64_kernel");
}

extern "C" int ppo2hip_normalize_clip(const void *x, int x_f64, int64_t n, int64_t d, const double *mean,
                                      const double *var, double epsilon, double clip, void *out, int out_f64,
                                      void *stream) {
    PPO2HIP_REQUIRE(n >= 0 && d >= 1, "normalize_clip: bad shape n=%lld d=%lld", (long long)n, (long long)d);
    PPO2HIP_REQUIRE(clip >= 0.0, "normalize_clip: negative clip");
    if (n == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(x && var && out, "normalize_clip: null pointer");
    return normalize_impl(x, x_f64, n, d, mean, var, epsilon, clip, out, out_f64, (hipStream_t)stream);
}

extern "C" int ppo2hip_vecnorm_returns(const void *rews, int rews_f64, int64_t E, double gamma, double *ret,
                                       double *state, double epsilon, double clip, void *out, int out_f64,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    PPO2HIP_REQUIRE(E >= 1, "vecnorm_returns: bad E=%lld", (long long)E);
    PPO2HIP_REQUIRE(rews && ret && out, "vecnorm_returns: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (rews_f64)
        hipLaunchKernelGGL(discounted_return_kernel<double>, dim3((unsigned)ceil_div(E, 256)), dim3(256), 0, s,
                           (const double *)rews, E, gamma, ret);
    else
        hipLaunchKernelGGL(discounted_return_kernel<float>, dim3((unsigned)ceil_div(E, 256)), dim3(256), 0, s,
                           (const float *)rews, E, gamma, ret);
    int rc = check_launch("discounted_return_kernel");
    if (rc) return rc;
    if (!state) {
        // ret=False: VecNormalize keeps self.ret but leaves the rewards alone (vec_normalize.py:29)
        PPO2HIP_REQUIRE(!out_f64 == !rews_f64, "vecnorm_returns: without statistics out has the rewards' dtype");
        if (out != rews)
            PPO2HIP_CHECK_HIP(hipMemcpyAsync(out, rews, (size_t)E * (rews_f64 ? 8 : 4), hipMemcpyDeviceToDevice, s));
        return PPO2HIP_OK;
    }
    rc = rms_update_impl(ret, 1, E, 1, state, workspace, workspace_bytes, s);
    if (rc) return rc;
    return normalize_impl(rews, rews_f64, E, 1, nullptr, state + 1, epsilon, clip, out, out_f64, s);
}

extern "C" int ppo2hip_frame_stack(void *stacked, const void *obs, const uint8_t *news, int64_t E, int64_t pixels,
                                   int channels, int nstack, int elem_bytes, void *stream) {
    PPO2HIP_REQUIRE(E >= 0 && pixels >= 0 && channels >= 1 && nstack >= 1, "frame_stack: bad shape");
    PPO2HIP_REQUIRE(elem_bytes == 1 || elem_bytes == 4, "frame_stack: elements of 1 or 4 bytes, got %d", elem_bytes);
    const int64_t total = E * pixels;
    if (total == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(stacked && obs, "frame_stack: null pointer");
    const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
    if (elem_bytes == 1)
        hipLaunchKernelGGL(frame_stack_kernel<uint8_t>, grid, block, 0, (hipStream_t)stream, (uint8_t *)stacked,
                           (const uint8_t *)obs, news, E, pixels, channels, nstack);
    else
        hipLaunchKernelGGL(frame_stack_kernel<uint32_t>, grid, block, 0, (hipStream_t)stream, (uint32_t *)stacked,
                           (const uint32_t *)obs, news, E, pixels, channels, nstack);
    return check_launch("frame_stack_kernel");
}
