// Fused minibatch gradient of the two-trunk tanh MLP policy (hidden widths 32 or 64):
// gather -> advantage normalisation -> forward -> distribution + PPO loss -> backward ->
// weight-gradient accumulation, in ONE launch.  What TensorFlow builds for
// baselines/a2c/policies.py:116-146 (and the 64-wide case of baselines/ppo2/policies.py:132-181)
// under the loss of baselines/ppo2/common_functions.py:26-63.
//
// Mapping to CDNA4
//   * one wave64 owns a tile of 32 minibatch rows and one trunk (workgroups alternate between
//     the actor and the value trunk); the grid is persistent and strides over the tiles.
//   * every contraction is v_mfma_f32_32x32x2_f32 (exact fp32 fma chain) with the 32 samples on
//     the lane/column axis: a layer's 32x32 accumulator (units in registers, samples on lanes)
//     is directly the B operand of the next layer and of the input-gradient products, so the
//     forward/backward chain never leaves registers.  The k order of such a product is the
//     accumulator's register order: register r, lane half h <-> unit rho(r) + 4h.
//   * weights (both orientations) sit in LDS for the whole launch and feed the A operand with
//     conflict-free 32-float runs; the weight-gradient products contract over SAMPLES, so their
//     operands are the activations transposed through a per-wave LDS tile (row stride 33).
//   * weight gradients accumulate in registers across all of a wave's tiles, are summed over
//     the workgroup's four waves in LDS and leave as one slab per workgroup; a fixed-order
//     reduction over slabs follows (no float atomics: run-to-run reproducible).
#include <atomic>
#include <chrono>

#include "gemm_core.hpp"
#include "kernels.hpp"

namespace ppo2hip {

constexpr int kFusedThreads = 256;
constexpr int kLdT = 33;                 // row stride of the transposed tiles (bank-conflict-free both ways)
constexpr int kLossStride = 8 + 64;      // doubles per workgroup partial row: as ppo_loss_final_kernel expects

struct TrunkOffsets {
    int64_t w1, w2, wh;                  // offsets of (w, b) pairs in the parameter/gradient arena
};

struct FusedArgs {
    // rollout + minibatch selection
    const float *obs;
    const void *actions;
    const float *returns, *values, *neglogpacs;
    const void *advs;
    int adv_f64, T, E;
    const int32_t *idx;
    int64_t B;
    const double *stats;
    int64_t nsamples;        // sample numbers the index list may name
    unsigned *fault;         // the device's fault word
    // network
    const float *params;
    TrunkOffsets actor, value;
    int64_t logstd_off;
    int D, A, R, dist, upstream;
    float cliprange, ent_coef, vf_coef, inv_count;
    // outputs
    float *slabs;            // [2 trunks][nwg_per_trunk][slab_elems]
    int64_t slab_elems;
    double *loss_partial;    // [gridDim.x][kLossStride]
};

__device__ inline int rho(int r) { return (r & 3) + 8 * (r >> 2); }


template <int H1B, int H2B>
struct FusedLayout {
    static constexpr int H1 = 32 * H1B, H2 = 32 * H2B;
    // weights, per workgroup (floats)
    static constexpr int W1 = 0;                          // [32][H1]   rows: D inputs, bias row, zeros
    static constexpr int W2 = W1 + 32 * H1;               // [H1][H2]
    static constexpr int W2T = W2 + H1 * H2;              // [H2][H1]
    static constexpr int B2 = W2T + H1 * H2;              // [H2]
    static constexpr int WH = B2 + H2;                    // [H2][32]   columns >= O zero
    static constexpr int WHT = WH + H2 * 32;              // [32][H2]
    static constexpr int BH = WHT + 32 * H2;              // [32]
    static constexpr int WEIGHTS = BH + 32;
    // per-wave scratch (floats)
    static constexpr int XS = 0;                          // [32 samples][33]
    static constexpr int TA = XS + 32 * kLdT;             // [H1][33]
    static constexpr int TB = TA + H1 * kLdT;             // [H2][33]
    static constexpr int TC = TB + H2 * kLdT;             // [32][33]
    static constexpr int WAVE = TC + 32 * kLdT;
    static constexpr int TOTAL = WEIGHTS + 4 * WAVE;
    // slab image of one trunk's gradient (floats): [w1 (D+1 rows) | w2 | b2 | wh [H2][32] | bh [32]]
    static constexpr int S_W1 = 0;
    static constexpr int S_W2 = 32 * H1;
    static constexpr int S_B2 = S_W2 + H1 * H2;
    static constexpr int S_WH = S_B2 + H2;
    static constexpr int S_BH = S_WH + H2 * 32;
    static constexpr int SLAB = S_BH + 32;
};

template <int H1B, int H2B>
__global__ __launch_bounds__(kFusedThreads, 1) void mlp_fused_kernel(FusedArgs g) {
    using L = FusedLayout<H1B, H2B>;
    constexpr int H1 = L::H1, H2 = L::H2;
    __shared__ float lds[L::TOTAL];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const bool is_value = (blockIdx.x & 1) != 0;
    const int nwg_trunk = gridDim.x >> 1;
    const int wg_in_trunk = blockIdx.x >> 1;
    const TrunkOffsets off = is_value ? g.value : g.actor;
    const int D = g.D;
    const int O = is_value ? g.R : g.A;
    const bool gaussian = g.dist == PPO2HIP_DIST_GAUSSIAN;
    const bool wide_out = (!is_value) && (!gaussian);     // categorical logits: up to 32 live rows
    const int NRO = wide_out ? 16 : 4;                    // accumulator registers that can hold live outputs

    // ---- stage this trunk's weights into LDS --------------------------------------------
    {
        const float *P = g.params;
        float *w1 = lds + L::W1;
        for (int e = tid; e < 32 * H1; e += kFusedThreads) {
            const int k = e / H1, j = e - k * H1;
            float v = 0.f;
            if (k < D) v = P[off.w1 + (int64_t)k * H1 + j];
            else if (k == D) v = P[off.w1 + (int64_t)D * H1 + j];     // bias row follows the matrix
            w1[e] = v;
        }
        for (int e = tid; e < H1 * H2; e += kFusedThreads) {
            const int k = e / H2, j = e - k * H2;
            const float v = P[off.w2 + e];
            lds[L::W2 + e] = v;
            lds[L::W2T + j * H1 + k] = v;
        }
        for (int e = tid; e < H2; e += kFusedThreads) lds[L::B2 + e] = P[off.w2 + (int64_t)H1 * H2 + e];
        for (int e = tid; e < H2 * 32; e += kFusedThreads) {
            const int k = e >> 5, j = e & 31;
            const float v = j < O ? P[off.wh + (int64_t)k * O + j] : 0.f;
            lds[L::WH + e] = v;
            lds[L::WHT + j * H2 + k] = v;
        }
        for (int e = tid; e < 32; e += kFusedThreads) lds[L::BH + e] = e < O ? P[off.wh + (int64_t)H2 * O + e] : 0.f;
        for (int e = tid; e < 4 * L::WAVE; e += kFusedThreads) lds[L::WEIGHTS + e] = 0.f;
    }
    __syncthreads();

    float *const xs = lds + L::WEIGHTS + wave * L::WAVE + L::XS;
    float *const ta = lds + L::WEIGHTS + wave * L::WAVE + L::TA;
    float *const tb = lds + L::WEIGHTS + wave * L::WAVE + L::TB;
    float *const tc = lds + L::WEIGHTS + wave * L::WAVE + L::TC;
    const float *const w1s = lds + L::W1;
    const float *const w2s = lds + L::W2;
    const float *const w2ts = lds + L::W2T;
    const float *const b2s = lds + L::B2;
    const float *const whs = lds + L::WH;
    const float *const whts = lds + L::WHT;
    const float *const bhs = lds + L::BH;

    // advantage normalisation constants (baselines/ppo2/common_functions.py:67)
    double adv_mean = 0.0, adv_rstd = 1.0;
    if (g.stats) {
        const double n = g.stats[2];
        adv_mean = g.stats[0] / n;
        double var = g.stats[1] / n - adv_mean * adv_mean;
        var = var > 0.0 ? var : 0.0;
        adv_rstd = 1.0 / (sqrt(var) + 1e-8);
    }
    float inv_std[4] = {0.f, 0.f, 0.f, 0.f};
    float sum_logstd = 0.f;
    if (!is_value && gaussian) {
        for (int k = 0; k < g.A; k++) sum_logstd += g.params[g.logstd_off + k];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = r + 4 * lh;
            if (k < g.A) inv_std[r] = expf(-g.params[g.logstd_off + k]);
        }
    }

    // ---- gradient accumulators (registers, whole launch) -----------------------------------
    f32x16 dW1[H1B], dW2[H1B][H2B], dWh[H2B];
    float db2[H2B][16], dbh[16], dls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 16; r++) {
        dbh[r] = 0.f;
#pragma unroll
        for (int a = 0; a < H1B; a++) {
            dW1[a][r] = 0.f;
#pragma unroll
            for (int b = 0; b < H2B; b++) dW2[a][b][r] = 0.f;
        }
#pragma unroll
        for (int b = 0; b < H2B; b++) {
            dWh[b][r] = 0.f;
            db2[b][r] = 0.f;
        }
    }
    double s_pg = 0, s_vf = 0, s_ent = 0, s_kl = 0, s_cf = 0;

    const int64_t ntiles = (g.B + 31) >> 5;
    const int64_t wave_global = (int64_t)wg_in_trunk * 4 + wave;
    const int64_t wave_stride = (int64_t)nwg_trunk * 4;
    const float one_lo = lh == 0 ? 1.f : 0.f;

    for (int64_t tile = wave_global; tile < ntiles; tile += wave_stride) {
        const int64_t i = tile * 32 + li;
        const bool live = i < g.B;
        const int64_t smp = live ? (g.idx ? checked_sample(g.idx[i], g.nsamples, g.fault) : i) : 0;
        const int64_t row = tm_offset(smp, g.T, g.E);

        // ---- gather the 32 observation rows into the x tile (+ ones feature at column D) ----
        for (int e = lane; e < 32 * D; e += 64) {
            const int s = e / D, c = e - s * D;
            const int64_t srow = __shfl(row, s, 64);
            xs[s * kLdT + c] = g.obs[srow * D + c];
        }
        if (lh == 0) xs[li * kLdT + D] = 1.f;
        __builtin_amdgcn_wave_barrier();

        // ---- forward ---------------------------------------------------------------------
        f32x16 h1[H1B], h2[H2B], out;
#pragma unroll
        for (int mb = 0; mb < H1B; mb++)
#pragma unroll
            for (int r = 0; r < 16; r++) h1[mb][r] = 0.f;
        const int k1steps = (D + 2) >> 1;
        for (int q = 0; q < k1steps; q++) {
            const float b = xs[li * kLdT + 2 * q + lh];
#pragma unroll
            for (int mb = 0; mb < H1B; mb++)
                h1[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1s[(2 * q + lh) * H1 + 32 * mb + li], b, h1[mb], 0, 0, 0);
        }
#pragma unroll
        for (int mb = 0; mb < H1B; mb++)
#pragma unroll
            for (int r = 0; r < 16; r++) h1[mb][r] = fast_tanh(h1[mb][r]);

#pragma unroll
        for (int m = 0; m < H2B; m++) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int mb = 0; mb < H1B; mb++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w2s[(32 * mb + rho(r) + 4 * lh) * H2 + 32 * m + li], h1[mb][r], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b2s[32 * m + li] * one_lo, one_lo, acc, 0, 0, 0);   // + bias
#pragma unroll
            for (int r = 0; r < 16; r++) h2[m][r] = fast_tanh(acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) out[r] = 0.f;
#pragma unroll
        for (int m = 0; m < H2B; m++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                out = __builtin_amdgcn_mfma_f32_32x32x2f32(whs[(32 * m + rho(r) + 4 * lh) * 32 + li], h2[m][r], out, 0, 0, 0);
        out = __builtin_amdgcn_mfma_f32_32x32x2f32(bhs[li] * one_lo, one_lo, out, 0, 0, 0);

        // transposed copies of the activations for the weight-gradient products
#pragma unroll
        for (int mb = 0; mb < H1B; mb++)
#pragma unroll
            for (int r = 0; r < 16; r++) ta[(32 * mb + rho(r) + 4 * lh) * kLdT + li] = h1[mb][r];
#pragma unroll
        for (int m = 0; m < H2B; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) tb[(32 * m + rho(r) + 4 * lh) * kLdT + li] = h2[m][r];

        // ---- loss and d(loss)/d(out) --------------------------------------------------------
        f32x16 dout;
#pragma unroll
        for (int r = 0; r < 16; r++) dout[r] = 0.f;
        if (!is_value) {
            float adv = 0.f, oldnlp = 0.f;
            if (live) {
                const double a = g.adv_f64 ? ((const double *)g.advs)[row] : (double)((const float *)g.advs)[row];
                adv = (float)((a - adv_mean) * adv_rstd);
                oldnlp = g.neglogpacs[row];
            }
            float nlp, entropy_row;
            float zq[4] = {0.f, 0.f, 0.f, 0.f};
            float pr[16], hterm[16];
            int act_i = 0;
            if (gaussian) {
                const float *act = (const float *)g.actions + row * g.A;
                float ss = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = r + 4 * lh;
                    if (k < g.A) {
                        zq[r] = (act[k] - out[r]) * inv_std[r];
                        ss += zq[r] * zq[r];
                    }
                }
                ss += __shfl_xor(ss, 32, 64);
                nlp = 0.5f * ss + 0.91893853320467274178f * (float)g.A + sum_logstd;
                entropy_row = sum_logstd + (0.91893853320467274178f + 0.5f) * (float)g.A;
            } else {
                act_i = ((const int32_t *)g.actions)[row];
                float mx = -3.0e38f;
#pragma unroll
                for (int r = 0; r < 16; r++)
                    if (rho(r) + 4 * lh < g.A) mx = fmaxf(mx, out[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float z0 = 0.f, la = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int k = rho(r) + 4 * lh;
                    pr[r] = k < g.A ? expf(out[r] - mx) : 0.f;
                    z0 += pr[r];
                    if (k == act_i) la = out[r] - mx;
                }
                z0 += __shfl_xor(z0, 32, 64);
                la += __shfl_xor(la, 32, 64);
                const float logz = logf(z0);
                nlp = logz - la;
                float H = 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int k = rho(r) + 4 * lh;
                    pr[r] = pr[r] / z0;
                    hterm[r] = k < g.A ? (out[r] - mx) - logz : 0.f;     // log p_k
                    H -= pr[r] * hterm[r];
                }
                H += __shfl_xor(H, 32, 64);
                entropy_row = H;
            }
            const float ratio = expf(oldnlp - nlp);
            const float pg1 = -adv * ratio;
            const float pg2 = -adv * fminf(fmaxf(ratio, 1.0f - g.cliprange), 1.0f + g.cliprange);
            const float dnlp = (live && pg1 >= pg2) ? adv * ratio * g.inv_count : 0.f;
            if (live && lh == 0) {
                s_pg += (double)fmaxf(pg1, pg2);
                const float d = nlp - oldnlp;
                s_kl += 0.5 * (double)(d * d);
                s_cf += (fabsf(ratio - 1.0f) > g.cliprange) ? 1.0 : 0.0;
                s_ent += (double)entropy_row;
            }
            if (gaussian) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = r + 4 * lh;
                    if (k < g.A) {
                        dout[r] = dnlp * (-zq[r] * inv_std[r]);
                        if (live) dls[r] += dnlp * (1.f - zq[r] * zq[r]) - g.ent_coef * g.inv_count;
                    }
                }
            } else {
                const float ec = live ? g.ent_coef * g.inv_count : 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int k = rho(r) + 4 * lh;
                    if (k < g.A) dout[r] = dnlp * (pr[r] - (k == act_i ? 1.f : 0.f)) + ec * pr[r] * (hterm[r] + entropy_row);
                }
            }
        } else {
            const float sc = g.inv_count / (float)g.R;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = r + 4 * lh;
                if (k < g.R && live) {
                    const float v = out[r], ret = g.returns[row * g.R + k];
                    const float e1 = v - ret;
                    if (g.upstream) {
                        const float ov = g.values[row * g.R + k];
                        const float dv = v - ov;
                        const float vc = ov + fminf(fmaxf(dv, -g.cliprange), g.cliprange);
                        const float e2 = vc - ret;
                        const float l1 = e1 * e1, l2 = e2 * e2;
                        s_vf += 0.5 * (double)fmaxf(l1, l2) / (double)g.R;
                        const float gg = (l1 >= l2) ? e1 : ((dv >= -g.cliprange && dv <= g.cliprange) ? e2 : 0.f);
                        dout[r] = g.vf_coef * sc * gg;
                    } else {
                        s_vf += (double)(e1 * e1) / (double)g.R;
                        dout[r] = 2.f * e1 * sc;
                    }
                }
            }
        }
        // bias gradient of the head + transposed copy of dout
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (r < NRO) {
                dbh[r] += dout[r];
                tc[(rho(r) + 4 * lh) * kLdT + li] = dout[r];
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- head: weight gradient, then input gradient ----------------------------------------
        for (int q = 0; q < 16; q++) {
            const float b = tc[li * kLdT + 2 * q + lh];
#pragma unroll
            for (int m = 0; m < H2B; m++)
                dWh[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(tb[(32 * m + li) * kLdT + 2 * q + lh], b, dWh[m], 0, 0, 0);
        }
        f32x16 dz2[H2B];
#pragma unroll
        for (int m = 0; m < H2B; m++) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            if (wide_out) {
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(whts[(rho(r) + 4 * lh) * H2 + 32 * m + li], dout[r], acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(whts[(rho(r) + 4 * lh) * H2 + 32 * m + li], dout[r], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                dz2[m][r] = acc[r] * (1.f - h2[m][r] * h2[m][r]);
                db2[m][r] += dz2[m][r];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < H2B; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) tb[(32 * m + rho(r) + 4 * lh) * kLdT + li] = dz2[m][r];
        __builtin_amdgcn_wave_barrier();

        // ---- layer 2: weight gradient (h1^T dz2), then input gradient -----------------------------
        for (int q = 0; q < 16; q++) {
            float a[H1B], b[H2B];
#pragma unroll
            for (int mi = 0; mi < H1B; mi++) a[mi] = ta[(32 * mi + li) * kLdT + 2 * q + lh];
#pragma unroll
            for (int mj = 0; mj < H2B; mj++) b[mj] = tb[(32 * mj + li) * kLdT + 2 * q + lh];
#pragma unroll
            for (int mi = 0; mi < H1B; mi++)
#pragma unroll
                for (int mj = 0; mj < H2B; mj++)
                    dW2[mi][mj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[mj], dW2[mi][mj], 0, 0, 0);
        }
        f32x16 dz1[H1B];
#pragma unroll
        for (int mb = 0; mb < H1B; mb++) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
            for (int m = 0; m < H2B; m++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w2ts[(32 * m + rho(r) + 4 * lh) * H1 + 32 * mb + li], dz2[m][r], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) dz1[mb][r] = acc[r] * (1.f - h1[mb][r] * h1[mb][r]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int mb = 0; mb < H1B; mb++)
#pragma unroll
            for (int r = 0; r < 16; r++) ta[(32 * mb + rho(r) + 4 * lh) * kLdT + li] = dz1[mb][r];
        __builtin_amdgcn_wave_barrier();

        // ---- layer 1: weight (+ bias, through the ones feature) gradient ----------------------------
        for (int q = 0; q < 16; q++) {
            const float a = xs[(2 * q + lh) * kLdT + li];
#pragma unroll
            for (int mj = 0; mj < H1B; mj++)
                dW1[mj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ta[(32 * mj + li) * kLdT + 2 * q + lh], dW1[mj], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- workgroup reduction of the per-wave gradient images, then one slab per workgroup --------
    // weights and tiles are dead now: the whole LDS array holds the four per-wave images
    __syncthreads();
    float *img = lds + wave * L::SLAB;
#pragma unroll
    for (int mj = 0; mj < H1B; mj++)
#pragma unroll
        for (int r = 0; r < 16; r++) img[L::S_W1 + (rho(r) + 4 * lh) * H1 + 32 * mj + li] = dW1[mj][r];
#pragma unroll
    for (int mi = 0; mi < H1B; mi++)
#pragma unroll
        for (int mj = 0; mj < H2B; mj++)
#pragma unroll
            for (int r = 0; r < 16; r++) img[L::S_W2 + (32 * mi + rho(r) + 4 * lh) * H2 + 32 * mj + li] = dW2[mi][mj][r];
#pragma unroll
    for (int mi = 0; mi < H2B; mi++)
#pragma unroll
        for (int r = 0; r < 16; r++) img[L::S_WH + (32 * mi + rho(r) + 4 * lh) * 32 + li] = dWh[mi][r];
    // bias gradients: sum over the 32 sample lanes of each half
#pragma unroll
    for (int m = 0; m < H2B; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = db2[m][r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (li == 0) img[L::S_B2 + 32 * m + rho(r) + 4 * lh] = v;
        }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float v = dbh[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (li == 0) img[L::S_BH + rho(r) + 4 * lh] = v;
    }
    __syncthreads();
    float *slab = g.slabs + ((int64_t)(is_value ? 1 : 0) * nwg_trunk + wg_in_trunk) * g.slab_elems;
    for (int e = tid; e < L::SLAB; e += kFusedThreads) {
        const float *p = lds + e;
        slab[e] = (p[0] + p[L::SLAB]) + (p[2 * L::SLAB] + p[3 * L::SLAB]);
    }

    // ---- loss partial row of this workgroup ---------------------------------------------------------
    __shared__ double red[17];
    double *lp = g.loss_partial + (int64_t)blockIdx.x * kLossStride;
    double v;
    v = block_sum(s_pg, red);  if (tid == 0) lp[0] = v;
    v = block_sum(s_vf, red);  if (tid == 0) lp[1] = v;
    v = block_sum(s_ent, red); if (tid == 0) lp[2] = v;
    v = block_sum(s_kl, red);  if (tid == 0) lp[3] = v;
    v = block_sum(s_cf, red);  if (tid == 0) lp[4] = v;
    for (int k = 0; k < 8; k++) {
        // dlogstd[k]: register k & 3 of lane half k >> 2
        const double mine = (!is_value && gaussian && lh == (k >> 2)) ? (double)dls[k & 3] : 0.0;
        v = block_sum(mine, red);
        if (tid == 0) lp[8 + k] = v;
    }
}

// One launch finishes a minibatch gradient: block ranges [0, nseg) sum the trunk matrices out
// of the per-workgroup slabs, [nseg, nseg + nhead) the head matrices (32-column image rows ->
// O-column arena rows), and the last block turns the loss partial rows into the five reported
// scalars and the logstd gradient.  Every sum runs in a fixed order.
struct SegTable {
    int n;
    int64_t src[8], dst[8], size[8];
    int trunk[8];
};
struct FinalizeArgs {
    const float *slabs;
    int nslabs[2];           // gradient images of the policy trunk, then of the value trunk (laid out in that order)
    int64_t slab_elems;
    SegTable t;
    int nseg_blocks, nhead_blocks;
    int s_wh, s_bh, s_logstd, H2, A, R, head_cols;
    int64_t a_dst, v_dst;
    float *grads;
    const double *loss_partial;
    int nloss_rows, A_gaussian;
    double inv_count;
    float *losses, *dlogstd;
};

constexpr int kFusedAdamMaxBlocks = 256;      // one per CU: resident together whatever else the chip is doing
size_t fused_adam_extra_bytes() { return kFusedAdamMaxBlocks * 2 * sizeof(unsigned long long); }

constexpr int kFinGroups = 16;   // slab groups per block: 1024 threads, every lane a handful of independent loads

// ADAM: the blocks that sum gradient elements then take the optimiser step on those same elements.  The global norm
// needs every block's sum of squares, and the hand-over is one hop: a block publishes its sum as two 64-bit words
// {launch tag, half of the double}, written with agent-scope stores, and every block reads all the words with
// agent-scope loads until each carries this launch's tag - no arrival counter, no wait for the store to drain, no
// cache write-back or invalidate.  All blocks of the launch are resident at once (at most 2 x 1024 threads of them per
// CU are asked for, far fewer than the chip holds); the wait is bounded, and a lane that gives up raises the fault
// word's grid-sync bit instead of spinning on.  The parameters, m and v of a block's elements are loaded before the
// wait.  Tags come from a process-wide counter, so what an earlier launch left in the words never matches.
// (the sequence starts at a clock-derived value and advances by an odd constant: consecutive launches differ in many bits,
// 2^32 launches pass before a tag returns, and whatever a fresh workspace holds - zeros, small integers, float bit
// patterns - matches a tag only with the probability of a 32-bit collision)
static unsigned next_adam_tag() {
    static std::atomic<unsigned> counter{(unsigned)std::chrono::steady_clock::now().time_since_epoch().count() * 2654435761u};
    unsigned t = counter.fetch_add(0x9e3779b9u, std::memory_order_relaxed);
    if (t == 0) t = counter.fetch_add(0x9e3779b9u, std::memory_order_relaxed);
    return t;
}

template <bool ADAM>
__global__ __launch_bounds__(64 * kFinGroups) void fused_finalize_kernel(FinalizeArgs f, AdamFuse ad) {
    __shared__ float part[kFinGroups][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int b = blockIdx.x;
    if (b < f.nseg_blocks + f.nhead_blocks) {
        const float *p = nullptr;
        float *out = nullptr;
        int nz = 0;
        if (b < f.nseg_blocks) {
            int seg = 0;
            int64_t base = 0;
            for (; seg < f.t.n; seg++) {
                const int64_t padded = (f.t.size[seg] + 63) / 64 * 64;
                if ((int64_t)b * 64 < base + padded) break;
                base += padded;
            }
            const int64_t e = (int64_t)b * 64 + lane - base;
            if (seg < f.t.n && e < f.t.size[seg]) {
                p = f.slabs + (int64_t)(f.t.trunk[seg] ? f.nslabs[0] : 0) * f.slab_elems + f.t.src[seg] + e;
                nz = f.nslabs[f.t.trunk[seg]];
                out = f.grads + f.t.dst[seg] + e;
            }
        } else {
            // head matrices (image rows of head_cols columns -> O-column arena rows, bias row last), then
            // for the policy trunk the logstd gradient when the kernel summed it inside the bias image
            const int hb = b - f.nseg_blocks;
            const int nls = f.s_logstd >= 0 ? f.A_gaussian : 0;
            const int na = (f.H2 + 1) * f.A, nv = (f.H2 + 1) * f.R;
            const int nba = (na + nls + 63) / 64;
            const int trunk = hb >= nba ? 1 : 0;
            const int e = (hb - (trunk ? nba : 0)) * 64 + lane;
            const int O = trunk ? f.R : f.A;
            if (e < (trunk ? nv : na)) {
                const int k = e / O, o = e - k * O;
                const int src = k < f.H2 ? f.s_wh + k * f.head_cols + o : f.s_bh + o;
                p = f.slabs + (int64_t)(trunk ? f.nslabs[0] : 0) * f.slab_elems + src;
                nz = f.nslabs[trunk];
                out = f.grads + (trunk ? f.v_dst : f.a_dst) + e;
            } else if (!trunk && e < na + nls) {
                p = f.slabs + f.s_logstd + (e - na);
                nz = f.nslabs[0];
                out = f.dlogstd + (e - na);
            }
        }
        float th = 0.f, mi = 0.f, vi = 0.f;
        int64_t pidx = -1;
        if (ADAM && grp == 0 && out) {
            pidx = out - f.grads;
            th = ad.theta[pidx]; mi = ad.m[pidx]; vi = ad.v[pidx];
        }
        // this thread's slabs grp, grp + 16, ...: all loads in flight before the first add (a loop of load-then-add runs one
        // memory round trip per slab), summed in slab order
        float s = 0.f;
        if (p) {
            constexpr int U = 10;
            for (int z0 = grp; z0 < nz; z0 += kFinGroups * U) {
                float x[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int z = z0 + u * kFinGroups;
                    x[u] = z < nz ? p[(int64_t)z * f.slab_elems] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; u++) s += x[u];
            }
        }
        part[grp][lane] = s;
        __syncthreads();
        float t = 0.f;
        if (grp == 0 && out) {
#pragma unroll
            for (int q = 0; q < kFinGroups; q++) t += part[q][lane];
            *out = t;
        }
        if (ADAM && grp == 0) {           // wave 0 of every gradient block
            const int nb = f.nseg_blocks + f.nhead_blocks;
            float scale = ad.gscale;
            if (ad.max_norm > 0.f) {
                const double x = (double)(t * ad.gscale);
                const double mine = wave_sum(x * x);
                if (lane == 0) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(mine), tg = (unsigned long long)ad.tag << 32;
                    __hip_atomic_store(&ad.words[2 * b], tg | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ad.words[2 * b + 1], tg | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // every block sums the same values in the same order, so all of them apply the same scale
                // a lane's words (blocks lane, lane + 64, ...) are requested together, so a round of polling is one round trip
                constexpr int kMine = kFusedAdamMaxBlocks / 64;
                unsigned long long hi[kMine], lo[kMine];
                int spins = 0;
                if (ad.max_spins < 0) atomicOr(ad.fault, kFaultGridSync);     // debugging aid: behave as if a peer never arrived
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < kMine; u++) {
                        const int q = lane + 64 * u;
                        hi[u] = lo[u] = (unsigned long long)ad.tag << 32;
                        if (q < nb) {
                            hi[u] = __hip_atomic_load(&ad.words[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            lo[u] = __hip_atomic_load(&ad.words[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kMine; u++) ok = ok && (unsigned)(hi[u] >> 32) == ad.tag && (unsigned)(lo[u] >> 32) == ad.tag;
                    if (ok) break;
                    if (++spins > ad.max_spins) {   // never expected; leave rather than hang the device
                        atomicOr(ad.fault, kFaultGridSync);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                double ss = 0.0;
#pragma unroll
                for (int u = 0; u < kMine; u++)
                    if (lane + 64 * u < nb) ss += __longlong_as_double((long long)((hi[u] << 32) | (lo[u] & 0xffffffffull)));
                ss = wave_sum(ss);
                const float norm = (float)sqrt(ss);
                scale = ad.gscale * (ad.max_norm / fmaxf(norm, ad.max_norm));
                if (b == 0 && lane == 0) ad.header[0] = norm;
            }
            if (b == 0 && lane == 0) ad.header[1] = scale;
            if (pidx >= 0) {
                const float gs = t * scale;
                const float m1 = ad.beta1 * mi + ad.omb1 * gs;
                const float v1 = ad.beta2 * vi + ad.omb2 * (gs * gs);
                ad.m[pidx] = m1;
                ad.v[pidx] = v1;
                ad.theta[pidx] = th + (-ad.a) * m1 / (sqrtf(v1) + ad.eps);
            }
        }
    } else {
        // loss scalars (+ the logstd gradient when it travels in the loss partial rows): one wave per output
        const int nout = 8 + (f.s_logstd < 0 ? f.A_gaussian : 0);
        for (int j = grp; j < nout; j += kFinGroups) {
            double s = 0.0;
            if (j < 5 || j >= 8)
                for (int r = lane; r < f.nloss_rows; r += 64) s += f.loss_partial[(int64_t)r * kLossStride + j];
            s = wave_sum(s);
            if (lane == 0) {
                if (j < 8) f.losses[j] = (float)(s * f.inv_count);
                else if (f.dlogstd) f.dlogstd[j - 8] = (float)s;
            }
        }
    }
}

int fused_finalize(const FusedImage &im, const float *slabs, int nslabs_policy, int nslabs_value, const double *lpart, int nloss_rows,
                   const ppo2hip_policy_desc &d, const FusedOffsets &fo, double inv_count, float *grads, float *losses,
                   hipStream_t s, const AdamFuse *adam) {
    const int H1 = d.hidden1, H2 = d.hidden2, D = d.ob_dim, A = d.ac_dim, R = d.nvalue;
    FinalizeArgs fa;
    fa.t.n = 0;
    auto seg = [&](int trunk, int64_t src, int64_t dst, int64_t size) {
        SegTable &t = fa.t;
        t.trunk[t.n] = trunk; t.src[t.n] = src; t.dst[t.n] = dst; t.size[t.n] = size; t.n++;
    };
    for (int trunk = 0; trunk < 2; trunk++) {
        const int64_t w1 = trunk ? fo.v_w1 : fo.a_w1, w2 = trunk ? fo.v_w2 : fo.a_w2;
        seg(trunk, im.s_w1, w1, (int64_t)(D + 1) * H1);            // w1 rows + bias row, contiguous in the arena
        seg(trunk, im.s_w2, w2, (int64_t)H1 * H2 + H2);            // w2 then b2 (adjacent in image and arena)
    }
    int64_t blocks = 0;
    for (int i = 0; i < fa.t.n; i++) blocks += (fa.t.size[i] + 63) / 64;
    const bool gauss = d.dist == PPO2HIP_DIST_GAUSSIAN;
    fa.slabs = slabs; fa.nslabs[0] = nslabs_policy; fa.nslabs[1] = nslabs_value; fa.slab_elems = im.slab_elems;
    fa.nseg_blocks = (int)blocks;
    fa.nhead_blocks = ((H2 + 1) * A + (gauss && im.s_logstd >= 0 ? A : 0) + 63) / 64 + ((H2 + 1) * R + 63) / 64;
    fa.s_wh = im.s_wh; fa.s_bh = im.s_bh; fa.s_logstd = im.s_logstd; fa.H2 = H2; fa.A = A; fa.R = R; fa.head_cols = im.head_cols;
    fa.a_dst = fo.a_wh; fa.v_dst = fo.v_wh;
    fa.grads = grads; fa.loss_partial = lpart; fa.nloss_rows = nloss_rows; fa.A_gaussian = gauss ? A : 0;
    fa.inv_count = inv_count; fa.losses = losses; fa.dlogstd = gauss ? grads + fo.logstd : nullptr;
    if (adam) {
        if (fa.nseg_blocks + fa.nhead_blocks > kFusedAdamMaxBlocks || (gauss && im.s_logstd < 0)) {
            set_error("fused_finalize: optimiser step cannot be folded in for this shape");
            return PPO2HIP_EUNSUPPORTED;
        }
        AdamFuse af = *adam;
        af.tag = next_adam_tag();
        // polls before a waiting lane gives up and raises the grid-sync fault bit; PPO2HIP_GRID_SYNC_SPINS overrides the
        // limit (negative: every launch raises the bit, which is how the tests exercise the reporting path)
        static const int spins = getenv("PPO2HIP_GRID_SYNC_SPINS") ? atoi(getenv("PPO2HIP_GRID_SYNC_SPINS")) : (1 << 20);
        af.max_spins = spins;
        af.fault = fault_word();
        if (!af.fault) return PPO2HIP_EINVAL;
        hipLaunchKernelGGL(fused_finalize_kernel<true>, dim3(fa.nseg_blocks + fa.nhead_blocks + 1), dim3(64 * kFinGroups), 0, s, fa, af);
    } else {
        hipLaunchKernelGGL(fused_finalize_kernel<false>, dim3(fa.nseg_blocks + fa.nhead_blocks + 1), dim3(64 * kFinGroups), 0, s, fa, AdamFuse{});
    }
    return check_launch("fused_finalize_kernel");
}

int fused_loss_finalize(const double *lpart, int nrows, int A_gaussian, double inv_count, float *losses, float *dlogstd,
                        hipStream_t s) {
    FinalizeArgs fa{};
    fa.t.n = 0; fa.nseg_blocks = 0; fa.nhead_blocks = 0;
    fa.s_logstd = -1;                       // the logstd gradient travels in the partial rows
    fa.loss_partial = lpart; fa.nloss_rows = nrows; fa.A_gaussian = A_gaussian; fa.inv_count = inv_count;
    fa.losses = losses; fa.dlogstd = A_gaussian ? dlogstd : nullptr;
    hipLaunchKernelGGL(fused_finalize_kernel<false>, dim3(1), dim3(64 * kFinGroups), 0, s, fa, AdamFuse{});
    return check_launch("fused_finalize_kernel (losses)");
}

bool mlp_fused_supported(const ppo2hip_policy_desc &d) {
    if (d.kind != PPO2HIP_POLICY_MLP) return false;
    if (d.hidden1 != 64 || d.hidden2 != 64) return false;
    if (d.ob_dim < 1 || d.ob_dim > 31) return false;
    if (d.nvalue > 8) return false;
    if (d.dist == PPO2HIP_DIST_GAUSSIAN) return d.ac_dim <= 8;
    return d.ac_dim <= 32;
}

int mlp_fused_grid() { return 256; }   // one resident workgroup per CU (LDS-limited), 128 per trunk

size_t mlp_fused_workspace_bytes() {
    using L = FusedLayout<2, 2>;
    return align_up((size_t)mlp_fused_grid() * L::SLAB * sizeof(float), 256) +
           align_up((size_t)mlp_fused_grid() * kLossStride * sizeof(double), 256);
}

int mlp_fused_grad(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const float *params,
                   const FusedOffsets &fo, const ppo2hip_rollout &ro, const int32_t *idx, int64_t B,
                   const double *stats, float cliprange, double inv_count, float *grads, float *losses, void *ws,
                   size_t ws_bytes, hipStream_t s) {
    using L = FusedLayout<2, 2>;
    static_assert(4 * L::SLAB <= L::TOTAL, "LDS must hold the four per-wave gradient images");
    static_assert(L::TOTAL * sizeof(float) <= 160 * 1024, "LDS budget");
    if (ws_bytes < mlp_fused_workspace_bytes()) {
        set_error("mlp_fused_grad: workspace too small");
        return PPO2HIP_ENOSPACE;
    }
    const int grid = mlp_fused_grid();
    float *slabs = (float *)ws;
    double *lpart = (double *)((char *)ws + align_up((size_t)grid * L::SLAB * sizeof(float), 256));
    FusedArgs g;
    g.obs = (const float *)ro.obs; g.actions = ro.actions; g.returns = ro.returns; g.values = ro.values;
    g.neglogpacs = ro.neglogpacs; g.advs = ro.advs; g.adv_f64 = ro.adv_f64; g.T = ro.T; g.E = ro.E;
    g.idx = idx; g.B = B; g.stats = stats; g.params = params;
    g.nsamples = rollout_samples(ro); g.fault = fault_word();
    if (!g.fault) return PPO2HIP_EINVAL;
    g.actor = TrunkOffsets{fo.a_w1, fo.a_w2, fo.a_wh};
    g.value = TrunkOffsets{fo.v_w1, fo.v_w2, fo.v_wh};
    g.logstd_off = fo.logstd; g.D = d.ob_dim; g.A = d.ac_dim; g.R = d.nvalue; g.dist = d.dist;
    g.upstream = lc.upstream; g.cliprange = cliprange; g.ent_coef = lc.ent_coef;
    g.vf_coef = lc.upstream ? lc.vf_coef : 1.0f; g.inv_count = (float)inv_count;
    g.slabs = slabs; g.slab_elems = L::SLAB; g.loss_partial = lpart;
    hipLaunchKernelGGL((mlp_fused_kernel<2, 2>), dim3(grid), dim3(kFusedThreads), 0, s, g);
    int rc = check_launch("mlp_fused_kernel");
    if (rc) return rc;

    FusedImage im;
    im.slab_elems = L::SLAB; im.s_w1 = L::S_W1; im.s_w2 = L::S_W2; im.s_wh = L::S_WH; im.s_bh = L::S_BH; im.head_cols = 32; im.s_logstd = -1;
    return fused_finalize(im, slabs, grid / 2, grid / 2, lpart, grid, d, fo, inv_count, grads, losses, s);
}

}  // namespace ppo2hip
