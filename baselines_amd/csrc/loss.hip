// Distribution terms and the PPO loss graph with its hand-derived backward.
//   distributions : baselines/common/distributions.py:202-210 (DiagGaussian), :141-163 (Categorical)
//   loss graph    : baselines/ppo2/common_functions.py:26-44 (fork) and :30-33,:56 (upstream)
// One lane per minibatch row; the five reported scalars and the logstd gradient are summed
// through fixed-shape per-block partials in float64 (reproducible).  tf.maximum sends the
// gradient to its first argument on ties, tf.clip_by_value passes it on the closed interval.
#include "kernels.hpp"

namespace ppo2hip {

constexpr int kLossBlock = 256;
constexpr int kMaxAc = 64;
constexpr int kPartialStride = 8 + kMaxAc;   // doubles per block: 5 loss sums (+3 pad) + dlogstd

int loss_partial_blocks(int64_t B) {
    int64_t n = ceil_div(B, kLossBlock);
    return (int)(n < 1024 ? n : 1024);
}

struct LossDev {
    int dist, ac_dim, nvalue, upstream;
    int64_t B;
    const float *pi, *logstd, *vf;
    const void *actions;
    const float *advs, *returns, *oldv, *oldnlp;
    float cliprange, ent_coef, vf_coef;
    float inv_count;
    float *dpi, *dvf;
    double *partial;
};

__device__ inline float clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__global__ __launch_bounds__(kLossBlock) void ppo_loss_kernel(LossDev a) {
    __shared__ double red[17];
    __shared__ double wred[kLossBlock / 64][kMaxAc];
    double s_pg = 0, s_vf = 0, s_ent = 0, s_kl = 0, s_cf = 0;
    const int A = a.ac_dim;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per-lane logstd-gradient accumulators live in LDS-free registers only for small A;
    // they are reduced per wave after each grid-stride trip instead (A wave sums per trip)
    double dls_wave[1];
    (void)dls_wave;
    for (int k = lane; k < kMaxAc; k += 64) wred[wave][k] = 0.0;
    __syncthreads();

    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t start = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // every lane of a wave runs the same number of trips so the wave reductions stay converged
    const int64_t trips = (a.B + stride - 1) / stride;
    for (int64_t trip = 0; trip < trips; trip++) {
        const int64_t i = start + trip * stride;
        const bool live = i < a.B;
        float dnlp = 0.f, nlp = 0.f, adv = 0.f, oldnlp = 0.f;
        if (live) {
            adv = a.advs[i];
            oldnlp = a.oldnlp[i];
        }
        if (a.dist == PPO2HIP_DIST_GAUSSIAN) {
            const float *act = (const float *)a.actions;
            if (live) {
                float ss = 0.f, sl = 0.f;
                for (int k = 0; k < A; k++) {
                    const float ls = a.logstd[k];
                    const float zq = (act[i * A + k] - a.pi[i * A + k]) / expf(ls);
                    ss += zq * zq;
                    sl += ls;
                }
                nlp = 0.5f * ss + kHalfLog2Pi * (float)A + sl;
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float pg2 = -adv * clipf(ratio, 1.0f - a.cliprange, 1.0f + a.cliprange);
                s_pg += (double)fmaxf(pg1, pg2);
                dnlp = (pg1 >= pg2) ? adv * ratio * a.inv_count : 0.f;
                const float d = nlp - oldnlp;
                s_kl += 0.5 * (double)(d * d);
                s_cf += (fabsf(ratio - 1.0f) > a.cliprange) ? 1.0 : 0.0;
                s_ent += (double)(sl + (kHalfLog2Pi + 0.5f) * (float)A);
            }
            for (int k = 0; k < A; k++) {
                float g = 0.f;
                if (live) {
                    const float sd = expf(a.logstd[k]);
                    const float zq = (act[i * A + k] - a.pi[i * A + k]) / sd;
                    a.dpi[i * A + k] = dnlp * (-zq / sd);
                    g = dnlp * (1.f - zq * zq) - a.ent_coef * a.inv_count;
                }
                const double w = wave_sum((double)g);
                if (lane == 0) wred[wave][k] += w;
            }
        } else {
            const int32_t *act = (const int32_t *)a.actions;
            if (live) {
                const float *l = a.pi + i * A;
                float mx = l[0];
                for (int k = 1; k < A; k++) mx = fmaxf(mx, l[k]);
                float z0 = 0.f;
                for (int k = 0; k < A; k++) z0 += expf(l[k] - mx);
                const float logz = logf(z0);
                const int ai = act[i];
                nlp = logz - (l[ai] - mx);
                float H = 0.f;
                for (int k = 0; k < A; k++) {
                    const float a0 = l[k] - mx;
                    H += expf(a0) / z0 * (logz - a0);
                }
                const float ratio = expf(oldnlp - nlp);
                const float pg1 = -adv * ratio;
                const float pg2 = -adv * clipf(ratio, 1.0f - a.cliprange, 1.0f + a.cliprange);
                s_pg += (double)fmaxf(pg1, pg2);
                dnlp = (pg1 >= pg2) ? adv * ratio * a.inv_count : 0.f;
                const float d = nlp - oldnlp;
                s_kl += 0.5 * (double)(d * d);
                s_cf += (fabsf(ratio - 1.0f) > a.cliprange) ? 1.0 : 0.0;
                s_ent += (double)H;
                const float ec = a.ent_coef * a.inv_count;
                for (int k = 0; k < A; k++) {
                    const float a0 = l[k] - mx;
                    const float p = expf(a0) / z0;
                    // d nlp / d l_k = p - onehot ;  dH / d l_k = -p (log p + H)
                    a.dpi[i * A + k] = dnlp * (p - (k == ai ? 1.f : 0.f)) + ec * p * ((a0 - logz) + H);
                }
            }
        }
        if (live) {
            const int R = a.nvalue;
            const float sc = a.inv_count / (float)R;
            for (int k = 0; k < R; k++) {
                const float v = a.vf[i * R + k], ret = a.returns[i * R + k];
                const float e1 = v - ret;
                if (a.upstream) {
                    const float ov = a.oldv[i * R + k];
                    const float dv = v - ov;
                    const float vc = ov + clipf(dv, -a.cliprange, a.cliprange);
                    const float e2 = vc - ret;
                    const float l1 = e1 * e1, l2 = e2 * e2;
                    s_vf += 0.5 * (double)fmaxf(l1, l2) / (double)R;
                    float g;
                    if (l1 >= l2) g = e1;
                    else g = (dv >= -a.cliprange && dv <= a.cliprange) ? e2 : 0.f;
                    a.dvf[i * R + k] = a.vf_coef * sc * g;     // d(.5 * max) = (v - R) or (vc - R) * dvc/dv
                } else {
                    s_vf += (double)(e1 * e1) / (double)R;
                    a.dvf[i * R + k] = 2.f * e1 * sc;
                }
            }
        }
    }
    double *out = a.partial + (int64_t)blockIdx.x * kPartialStride;
    double v;
    v = block_sum(s_pg, red);  if (threadIdx.x == 0) out[0] = v;
    v = block_sum(s_vf, red);  if (threadIdx.x == 0) out[1] = v;
    v = block_sum(s_ent, red); if (threadIdx.x == 0) out[2] = v;
    v = block_sum(s_kl, red);  if (threadIdx.x == 0) out[3] = v;
    v = block_sum(s_cf, red);  if (threadIdx.x == 0) out[4] = v;
    __syncthreads();
    if (a.dist == PPO2HIP_DIST_GAUSSIAN)
        for (int k = threadIdx.x; k < A; k += blockDim.x) {
            double t = 0.0;
            for (int w = 0; w < kLossBlock / 64; w++) t += wred[w][k];
            out[8 + k] = t;
        }
}

// one wave per output scalar: lanes stride over the per-block partials, then a wave sum
__global__ __launch_bounds__(1024) void ppo_loss_final_kernel(const double *__restrict__ partial, int nblocks, int A,
                                                              int gaussian, double inv_count, float *__restrict__ losses,
                                                              float *__restrict__ dlogstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nout = 8 + (gaussian ? A : 0);
    for (int j = wave; j < nout; j += 16) {
        double s = 0.0;
        if (j < 5 || j >= 8)
            for (int b = lane; b < nblocks; b += 64) s += partial[(int64_t)b * kPartialStride + j];
        s = wave_sum(s);
        if (lane == 0) {
            if (j < 8) losses[j] = (float)(s * inv_count);
            else if (dlogstd) dlogstd[j - 8] = (float)s;
        }
    }
}

int loss_final(const double *partial, int nblocks, int A_gaussian, double inv_count, float *losses, float *dlogstd,
               hipStream_t s) {
    hipLaunchKernelGGL(ppo_loss_final_kernel, dim3(1), dim3(1024), 0, s, partial, nblocks, A_gaussian,
                       A_gaussian > 0 ? 1 : 0, inv_count, losses, dlogstd);
    return check_launch("ppo_loss_final_kernel");
}

int ppo_loss_backward(const LossArgs &x, hipStream_t s) {
    if (x.ac_dim < 1 || x.ac_dim > kMaxAc) {
        set_error("ppo_loss: ac_dim=%d outside [1, %d]", x.ac_dim, kMaxAc);
        return PPO2HIP_EUNSUPPORTED;
    }
    LossDev a;
    a.dist = x.dist; a.ac_dim = x.ac_dim; a.nvalue = x.nvalue; a.upstream = x.cfg.upstream; a.B = x.B;
    a.pi = x.pi; a.logstd = x.logstd; a.vf = x.vf; a.actions = x.actions; a.advs = x.advs;
    a.returns = x.returns; a.oldv = x.oldv; a.oldnlp = x.oldnlp; a.cliprange = x.cliprange;
    a.ent_coef = x.cfg.ent_coef; a.vf_coef = x.cfg.upstream ? x.cfg.vf_coef : 1.0f;
    a.inv_count = (float)x.inv_count; a.dpi = x.dpi; a.dvf = x.dvf; a.partial = x.partial;
    const int nb = loss_partial_blocks(x.B);
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(nb), dim3(kLossBlock), 0, s, a);
    int rc = check_launch("ppo_loss_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(ppo_loss_final_kernel, dim3(1), dim3(1024), 0, s, x.partial, nb, x.ac_dim,
                       x.dist == PPO2HIP_DIST_GAUSSIAN ? 1 : 0, x.inv_count, x.losses, x.dlogstd);
    return check_launch("ppo_loss_final_kernel");
}

__global__ void neglogp_kernel(int dist, int A, const float *__restrict__ pi, const float *__restrict__ logstd,
                               const void *__restrict__ actions, int64_t B, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    if (dist == PPO2HIP_DIST_GAUSSIAN) {
        const float *act = (const float *)actions;
        float ss = 0.f, sl = 0.f;
        for (int k = 0; k < A; k++) {
            const float ls = logstd[k];
            const float zq = (act[i * A + k] - pi[i * A + k]) / expf(ls);
            ss += zq * zq;
            sl += ls;
        }
        out[i] = 0.5f * ss + kHalfLog2Pi * (float)A + sl;
    } else {
        const int32_t *act = (const int32_t *)actions;
        const float *l = pi + i * A;
        float mx = l[0];
        for (int k = 1; k < A; k++) mx = fmaxf(mx, l[k]);
        float z0 = 0.f;
        for (int k = 0; k < A; k++) z0 += expf(l[k] - mx);
        out[i] = logf(z0) - (l[act[i]] - mx);
    }
}

int neglogp_rows(int dist, int ac_dim, const float *pi, const float *logstd, const void *actions, int64_t B,
                 float *out, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(neglogp_kernel, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, s, dist, ac_dim, pi, logstd,
                       actions, B, out);
    return check_launch("neglogp_kernel");
}

// out[c] = scale * sum over rows of rows[r * stride + c], c < ncols <= 16: the mean of the per-minibatch loss rows
// (np.mean(mblossvals, axis=0), baselines/ppo2/ppo2.py:139), float64 sums in a fixed order.
__global__ __launch_bounds__(256) void row_mean_kernel(const float *__restrict__ rows, int nrows, int stride, int ncols,
                                                       double scale, float *__restrict__ out) {
    __shared__ double red[16][4];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;          // 16 row groups x 16 columns
    double s = 0.0;
    if (c < ncols)
        for (int r = g; r < nrows; r += 16) s += (double)rows[(int64_t)r * stride + c];
    // fold the 16 row groups: first within a wave (4 groups per wave), then across the 4 waves
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if ((threadIdx.x & 63) < 16) red[c][threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < ncols) out[threadIdx.x] = (float)(scale * ((red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3])));
}

}  // namespace ppo2hip

extern "C" int ppo2hip_row_mean(const float *rows, int nrows, int stride, int ncols, double scale, float *out, void *stream) {
    using namespace ppo2hip;
    PPO2HIP_REQUIRE(nrows >= 1 && ncols >= 1 && ncols <= 16 && stride >= ncols, "row_mean: bad shape");
    PPO2HIP_REQUIRE(rows && out, "row_mean: null pointer");
    hipLaunchKernelGGL(row_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows, nrows, stride, ncols, scale, out);
    return check_launch("row_mean_kernel");
}
