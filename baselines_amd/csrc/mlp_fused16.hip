// Fused minibatch gradient of the two-trunk tanh MLP policy, 64-64 hidden units, on
// 16-sample tiles: gather -> advantage normalisation -> forward -> distribution + PPO loss ->
// backward -> weight-gradient accumulation in ONE launch.  Same mathematics as
// mlp_fused.hip (baselines/a2c/policies.py:116-146 under the loss of
// baselines/ppo2/common_functions.py:26-63), re-tiled so that TWO waves share every SIMD.
//
// Why 16-sample tiles: a wave's per-tile state halves (activations 16 registers per layer,
// transposed LDS tiles 4.3 KB), which brings a wave under 256 registers and its LDS scratch
// to 12 KB, so a 512-thread workgroup (8 waves) fits one CU: while one wave of a SIMD waits on
// LDS or runs tanh / loss VALU work, the other keeps the matrix pipe busy.
//
// Mapping (v_mfma_f32_16x16x4_f32, exact fp32 fma chain; lane l: li = l & 15, lk = l >> 4):
//   operands  A[i = li][k = lk], B[k = lk][j = li];  result  D[row = 4 lk + r][col = li], r = 0..3.
//   Samples sit on the column (li) axis, units on rows: a layer's accumulator is directly the B
//   operand of the next product; its register r supplies unit 4 lk + r, so the A operand for
//   that k-step is the weight row of unit 16 mb + 4 lk + r.  Weight-gradient products contract
//   over samples and take both operands from per-wave transposed LDS tiles (row stride 17).
//   LDS row strides are chosen so the two 16-lane halves of a ds_read_b32 group fall on
//   disjoint banks (stride = 16 mod 32 where rows are 1 apart, = 4 mod 8 where they are 4 apart).
#include "gemm_core.hpp"
#include "kernels.hpp"

namespace ppo2hip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace f16 {
constexpr int kThreads = 512;
constexpr int kWaves = 8;
constexpr int H = 64;             // both hidden widths
constexpr int HB = H / 16;        // 16-unit blocks
constexpr int LDW1 = 80, LDW2 = 68, LDWH = 20, LDWHT = 68, LDX = 34, LDT = 17;
// weights (floats)
constexpr int W1 = 0;                          // [32][80]  rows: D inputs, bias row, zeros
constexpr int W2 = W1 + 32 * LDW1;             // [64][68]  row = input unit
constexpr int W2T = W2 + H * LDW2;             // [64][68]  row = output unit
constexpr int B2 = W2T + H * LDW2;             // [64]
constexpr int WH = B2 + H;                     // [64][20]  row = input unit, 16 output columns
constexpr int WHT = WH + H * LDWH;             // [16][68]  row = output
constexpr int BH = WHT + 16 * LDWHT;           // [16]
constexpr int WEIGHTS = BH + 16;
// per-wave scratch (floats)
constexpr int XS = 0;                          // [16 samples][34]
constexpr int TA = XS + 16 * LDX;              // [64 units][17]
constexpr int TB = TA + H * LDT;               // [64][17]
constexpr int TC = TB + H * LDT;               // [16 outputs][17]
constexpr int WAVE = TC + 16 * LDT;
constexpr int TOTAL = WEIGHTS + kWaves * WAVE;
// gradient image of one trunk (floats): [w1 32x64 | w2 64x64 | b2 64 | wh 64x16 | bh 16]
constexpr int S_W1 = 0;
constexpr int S_W2 = 32 * H;
constexpr int S_B2 = S_W2 + H * H;
constexpr int S_WH = S_B2 + H;
constexpr int S_BH = S_WH + H * 16;
constexpr int SLAB = S_BH + 16;
static_assert(4 * SLAB <= TOTAL, "LDS must hold four gradient images for the cross-wave reduction");
static_assert(TOTAL * 4 + 256 <= 160 * 1024, "LDS budget");
constexpr int kLossStride = 8 + 64;
}  // namespace f16

struct Fused16Args {
    const float *obs;
    const void *actions;
    const float *returns, *values, *neglogpacs;
    const void *advs;
    int adv_f64, T, E;
    const int32_t *idx;
    int64_t B;
    const double *stats;
    const float *params;
    int64_t a_w1, a_w2, a_wh, v_w1, v_w2, v_wh, logstd_off;
    int D, A, R, dist, upstream;
    float cliprange, ent_coef, vf_coef, inv_count;
    float *slabs;            // [2 trunks][nwg_per_trunk][SLAB]
    double *loss_partial;    // [gridDim.x][kLossStride]
};

__device__ inline float fast_tanh16(float x) {
    // |error| <= ~1e-7 absolute over the whole range (v_exp_f32 / v_rcp_f32 are 1 ulp)
    const float t = __expf(2.0f * x);
    return 1.0f - 2.0f * __frcp_rn(t + 1.0f);
}

__global__ __launch_bounds__(f16::kThreads, 2) void mlp_fused16_kernel(Fused16Args g) {
    using namespace f16;
    __shared__ float lds[TOTAL];
    __shared__ double red[17];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const bool is_value = (blockIdx.x & 1) != 0;
    const int nwg_trunk = gridDim.x >> 1;
    const int wg_in_trunk = blockIdx.x >> 1;
    const int64_t o_w1 = is_value ? g.v_w1 : g.a_w1, o_w2 = is_value ? g.v_w2 : g.a_w2, o_wh = is_value ? g.v_wh : g.a_wh;
    const int D = g.D;
    const int O = is_value ? g.R : g.A;
    const bool gaussian = g.dist == PPO2HIP_DIST_GAUSSIAN;

    // ---- stage this trunk's weights (both orientations where a transposed read is needed) ----
    {
        const float *P = g.params;
        for (int e = tid; e < 32 * LDW1; e += kThreads) {
            const int k = e / LDW1, j = e - k * LDW1;
            float v = 0.f;
            if (j < H && k <= D) v = P[o_w1 + (int64_t)k * H + j];      // rows 0..D-1 weights, row D = bias
            lds[W1 + e] = v;
        }
        for (int e = tid; e < H * H; e += kThreads) {
            const int k = e >> 6, j = e & 63;
            const float v = P[o_w2 + e];
            lds[W2 + k * LDW2 + j] = v;
            lds[W2T + j * LDW2 + k] = v;
        }
        for (int e = tid; e < H; e += kThreads) lds[B2 + e] = P[o_w2 + (int64_t)H * H + e];
        for (int e = tid; e < H * 16; e += kThreads) {
            const int k = e >> 4, j = e & 15;
            const float v = j < O ? P[o_wh + (int64_t)k * O + j] : 0.f;
            lds[WH + k * LDWH + j] = v;
            lds[WHT + j * LDWHT + k] = v;
        }
        for (int e = tid; e < 16; e += kThreads) lds[BH + e] = e < O ? P[o_wh + (int64_t)H * O + e] : 0.f;
        for (int e = tid; e < kWaves * WAVE; e += kThreads) lds[WEIGHTS + e] = 0.f;
    }
    __syncthreads();

    float *const xs = lds + WEIGHTS + wave * WAVE + XS;
    float *const ta = lds + WEIGHTS + wave * WAVE + TA;
    float *const tb = lds + WEIGHTS + wave * WAVE + TB;
    float *const tc = lds + WEIGHTS + wave * WAVE + TC;
    const float *const w1s = lds + W1;
    const float *const w2s = lds + W2;
    const float *const w2ts = lds + W2T;
    const float *const b2s = lds + B2;
    const float *const whs = lds + WH;
    const float *const whts = lds + WHT;
    const float *const bhs = lds + BH;

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
            const int k = 4 * lk + r;
            if (k < g.A) inv_std[r] = expf(-g.params[g.logstd_off + k]);
        }
    }

    // ---- gradient accumulators (registers, whole launch) ------------------------------------
    f32x4 dW1[2][HB], dW2[HB][HB], dWh[HB];
    float db2[HB], dbh = 0.f, dls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < HB; a++) {
        db2[a] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            dWh[a][r] = 0.f;
            dW1[0][a][r] = 0.f;
            dW1[1][a][r] = 0.f;
#pragma unroll
            for (int b = 0; b < HB; b++) dW2[a][b][r] = 0.f;
        }
    }
    double s_pg = 0, s_vf = 0, s_ent = 0, s_kl = 0, s_cf = 0;

    const int64_t ntiles = (g.B + 15) >> 4;
    const int64_t wave_global = (int64_t)wg_in_trunk * kWaves + wave;
    const int64_t wave_stride = (int64_t)nwg_trunk * kWaves;
    const float one0 = lk == 0 ? 1.f : 0.f;
    const int kq1 = (D + 4) >> 2;              // k-steps over the D inputs + the ones feature
    const bool wide_in = D + 1 > 16;           // second 16-row block of dW1 in use

    for (int64_t tile = wave_global; tile < ntiles; tile += wave_stride) {
        const int64_t i = tile * 16 + li;
        const bool live = i < g.B;
        const int64_t smp = live ? (g.idx ? (int64_t)g.idx[i] : i) : 0;
        const int64_t row = tm_offset(smp, g.T, g.E);

        // ---- gather the 16 observation rows into the x tile (+ ones feature at column D) ----
        for (int e = lane; e < 16 * D; e += 64) {
            const int s = e / D, c = e - s * D;
            const int64_t srow = __shfl(row, s, 64);
            xs[s * LDX + c] = g.obs[srow * D + c];
        }
        if (lk == 0) xs[li * LDX + D] = 1.f;
        __builtin_amdgcn_wave_barrier();

        // ---- forward ---------------------------------------------------------------------
        f32x4 h1[HB], h2[HB], out;
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) h1[mb][r] = 0.f;
        for (int q = 0; q < kq1; q++) {
            const float b = xs[li * LDX + 4 * q + lk];
#pragma unroll
            for (int mb = 0; mb < HB; mb++) h1[mb] = MFMA16(w1s[(4 * q + lk) * LDW1 + 16 * mb + li], b, h1[mb]);
        }
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) h1[mb][r] = fast_tanh16(h1[mb][r]);

#pragma unroll
        for (int m = 0; m < HB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) h2[m][r] = 0.f;
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int m = 0; m < HB; m++)
                    h2[m] = MFMA16(w2s[(16 * mb + 4 * lk + r) * LDW2 + 16 * m + li], h1[mb][r], h2[m]);
#pragma unroll
        for (int m = 0; m < HB; m++) h2[m] = MFMA16(b2s[16 * m + li] * one0, one0, h2[m]);        // + bias
#pragma unroll
        for (int m = 0; m < HB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) h2[m][r] = fast_tanh16(h2[m][r]);

#pragma unroll
        for (int r = 0; r < 4; r++) out[r] = 0.f;
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) out = MFMA16(whs[(16 * mb + 4 * lk + r) * LDWH + li], h2[mb][r], out);
        out = MFMA16(bhs[li] * one0, one0, out);

        // transposed copies of the activations for the weight-gradient products
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                ta[(16 * mb + 4 * lk + r) * LDT + li] = h1[mb][r];
                tb[(16 * mb + 4 * lk + r) * LDT + li] = h2[mb][r];
            }

        // ---- loss and d(loss)/d(out): output row 4 lk + r of sample li -------------------------
        f32x4 dout;
#pragma unroll
        for (int r = 0; r < 4; r++) dout[r] = 0.f;
        if (!is_value) {
            float adv = 0.f, oldnlp = 0.f;
            if (live) {
                const double a = g.adv_f64 ? ((const double *)g.advs)[row] : (double)((const float *)g.advs)[row];
                adv = (float)((a - adv_mean) * adv_rstd);
                oldnlp = g.neglogpacs[row];
            }
            float nlp, entropy_row;
            float zq[4] = {0.f, 0.f, 0.f, 0.f}, pr[4] = {0.f, 0.f, 0.f, 0.f}, lp[4] = {0.f, 0.f, 0.f, 0.f};
            int act_i = 0;
            if (gaussian) {
                const float *act = (const float *)g.actions + row * g.A;
                float ss = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * lk + r;
                    if (k < g.A) {
                        zq[r] = (act[k] - out[r]) * inv_std[r];
                        ss += zq[r] * zq[r];
                    }
                }
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                nlp = 0.5f * ss + 0.91893853320467274178f * (float)g.A + sum_logstd;
                entropy_row = sum_logstd + (0.91893853320467274178f + 0.5f) * (float)g.A;
            } else {
                act_i = ((const int32_t *)g.actions)[row];
                float mx = -3.0e38f;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (4 * lk + r < g.A) mx = fmaxf(mx, out[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float z0 = 0.f, la = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * lk + r;
                    pr[r] = k < g.A ? expf(out[r] - mx) : 0.f;
                    z0 += pr[r];
                    if (k == act_i) la = out[r] - mx;
                }
                z0 += __shfl_xor(z0, 16, 64);
                z0 += __shfl_xor(z0, 32, 64);
                la += __shfl_xor(la, 16, 64);
                la += __shfl_xor(la, 32, 64);
                const float logz = logf(z0);
                nlp = logz - la;
                float Hs = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * lk + r;
                    pr[r] = pr[r] / z0;
                    lp[r] = k < g.A ? (out[r] - mx) - logz : 0.f;      // log p_k
                    Hs -= pr[r] * lp[r];
                }
                Hs += __shfl_xor(Hs, 16, 64);
                Hs += __shfl_xor(Hs, 32, 64);
                entropy_row = Hs;
            }
            const float ratio = expf(oldnlp - nlp);
            const float pg1 = -adv * ratio;
            const float pg2 = -adv * fminf(fmaxf(ratio, 1.0f - g.cliprange), 1.0f + g.cliprange);
            const float dnlp = (live && pg1 >= pg2) ? adv * ratio * g.inv_count : 0.f;
            if (live && lk == 0) {
                s_pg += (double)fmaxf(pg1, pg2);
                const float d = nlp - oldnlp;
                s_kl += 0.5 * (double)(d * d);
                s_cf += (fabsf(ratio - 1.0f) > g.cliprange) ? 1.0 : 0.0;
                s_ent += (double)entropy_row;
            }
            if (gaussian) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * lk + r;
                    if (k < g.A) {
                        dout[r] = dnlp * (-zq[r] * inv_std[r]);
                        if (live) dls[r] += dnlp * (1.f - zq[r] * zq[r]) - g.ent_coef * g.inv_count;
                    }
                }
            } else {
                const float ec = live ? g.ent_coef * g.inv_count : 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = 4 * lk + r;
                    if (k < g.A) dout[r] = dnlp * (pr[r] - (k == act_i ? 1.f : 0.f)) + ec * pr[r] * (lp[r] + entropy_row);
                }
            }
        } else {
            const float sc = g.inv_count / (float)g.R;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int k = 4 * lk + r;
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
#pragma unroll
        for (int r = 0; r < 4; r++) tc[(4 * lk + r) * LDT + li] = dout[r];
        __builtin_amdgcn_wave_barrier();

        // ---- head: weight gradient (h2^T dout; bias gradient = row sums of dout^T) -------------------
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float b = tc[li * LDT + 4 * q + lk];
            dbh += b;
#pragma unroll
            for (int mi = 0; mi < HB; mi++) dWh[mi] = MFMA16(tb[(16 * mi + li) * LDT + 4 * q + lk], b, dWh[mi]);
        }
        // ---- head: input gradient, through tanh' ----------------------------------------------------
        f32x4 dz2[HB];
#pragma unroll
        for (int m = 0; m < HB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) dz2[m][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int m = 0; m < HB; m++) dz2[m] = MFMA16(whts[(4 * lk + r) * LDWHT + 16 * m + li], dout[r], dz2[m]);
#pragma unroll
        for (int m = 0; m < HB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) dz2[m][r] *= 1.f - h2[m][r] * h2[m][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < HB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) tb[(16 * m + 4 * lk + r) * LDT + li] = dz2[m][r];
        __builtin_amdgcn_wave_barrier();

        // ---- layer 2: weight gradient (h1^T dz2), bias gradient from the same operand reads ---------
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float a[HB], b[HB];
#pragma unroll
            for (int mi = 0; mi < HB; mi++) a[mi] = ta[(16 * mi + li) * LDT + 4 * q + lk];
#pragma unroll
            for (int mj = 0; mj < HB; mj++) {
                b[mj] = tb[(16 * mj + li) * LDT + 4 * q + lk];
                db2[mj] += b[mj];
            }
#pragma unroll
            for (int mi = 0; mi < HB; mi++)
#pragma unroll
                for (int mj = 0; mj < HB; mj++) dW2[mi][mj] = MFMA16(a[mi], b[mj], dW2[mi][mj]);
        }
        // ---- layer 2: input gradient, through tanh' ----------------------------------------------------
        f32x4 dz1[HB];
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) dz1[mb][r] = 0.f;
#pragma unroll
        for (int m = 0; m < HB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int mb = 0; mb < HB; mb++)
                    dz1[mb] = MFMA16(w2ts[(16 * m + 4 * lk + r) * LDW2 + 16 * mb + li], dz2[m][r], dz1[mb]);
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) dz1[mb][r] *= 1.f - h1[mb][r] * h1[mb][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) ta[(16 * mb + 4 * lk + r) * LDT + li] = dz1[mb][r];
        __builtin_amdgcn_wave_barrier();

        // ---- layer 1: weight (+ bias, through the ones feature) gradient --------------------------------
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float b[HB];
#pragma unroll
            for (int mj = 0; mj < HB; mj++) b[mj] = ta[(16 * mj + li) * LDT + 4 * q + lk];
            const float a0 = xs[(4 * q + lk) * LDX + li];
#pragma unroll
            for (int mj = 0; mj < HB; mj++) dW1[0][mj] = MFMA16(a0, b[mj], dW1[0][mj]);
            if (wide_in) {
                const float a1 = xs[(4 * q + lk) * LDX + 16 + li];
#pragma unroll
                for (int mj = 0; mj < HB; mj++) dW1[1][mj] = MFMA16(a1, b[mj], dW1[1][mj]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- cross-wave reduction of the gradient images (two rounds of four), one slab per workgroup ------
    auto put = [&](float *img, bool add) {
        auto st = [&](int off, float v) { img[off] = add ? img[off] + v : v; };
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int mj = 0; mj < HB; mj++)
#pragma unroll
                for (int r = 0; r < 4; r++) st(S_W1 + (16 * f + 4 * lk + r) * H + 16 * mj + li, dW1[f][mj][r]);
#pragma unroll
        for (int mi = 0; mi < HB; mi++)
#pragma unroll
            for (int mj = 0; mj < HB; mj++)
#pragma unroll
                for (int r = 0; r < 4; r++) st(S_W2 + (16 * mi + 4 * lk + r) * H + 16 * mj + li, dW2[mi][mj][r]);
#pragma unroll
        for (int mi = 0; mi < HB; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++) st(S_WH + (16 * mi + 4 * lk + r) * 16 + li, dWh[mi][r]);
#pragma unroll
        for (int mj = 0; mj < HB; mj++) {
            float v = db2[mj];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lk == 0) st(S_B2 + 16 * mj + li, v);
        }
        float v = dbh;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lk == 0) st(S_BH + li, v);
    };
    __syncthreads();                                   // weights and tiles are dead: LDS holds images now
    if (wave >= 4) put(lds + (wave - 4) * SLAB, false);
    __syncthreads();
    if (wave < 4) put(lds + wave * SLAB, true);
    __syncthreads();
    float *slab = g.slabs + ((int64_t)(is_value ? 1 : 0) * nwg_trunk + wg_in_trunk) * SLAB;
    for (int e = tid; e < SLAB; e += kThreads) {
        const float *p = lds + e;
        slab[e] = (p[0] + p[SLAB]) + (p[2 * SLAB] + p[3 * SLAB]);
    }

    // ---- loss partial row of this workgroup ---------------------------------------------------------
    double *lpr = g.loss_partial + (int64_t)blockIdx.x * kLossStride;
    double v;
    v = block_sum(s_pg, red);  if (tid == 0) lpr[0] = v;
    v = block_sum(s_vf, red);  if (tid == 0) lpr[1] = v;
    v = block_sum(s_ent, red); if (tid == 0) lpr[2] = v;
    v = block_sum(s_kl, red);  if (tid == 0) lpr[3] = v;
    v = block_sum(s_cf, red);  if (tid == 0) lpr[4] = v;
    for (int k = 0; k < 16; k++) {
        // dlogstd[k]: register k & 3 of lane quarter k >> 2
        const double mine = (!is_value && gaussian && lk == (k >> 2)) ? (double)dls[k & 3] : 0.0;
        v = block_sum(mine, red);
        if (tid == 0) lpr[8 + k] = v;
    }
}

bool mlp_fused16_supported(const ppo2hip_policy_desc &d) {
    if (d.kind != PPO2HIP_POLICY_MLP) return false;
    if (d.hidden1 != 64 || d.hidden2 != 64) return false;
    if (d.ob_dim < 1 || d.ob_dim > 31) return false;
    if (d.nvalue > 8) return false;
    return d.ac_dim <= 16;
}

int mlp_fused16_grid() { return 256; }   // one 512-thread workgroup per CU (LDS-limited), 128 per trunk

size_t mlp_fused16_workspace_bytes() {
    return align_up((size_t)mlp_fused16_grid() * f16::SLAB * sizeof(float), 256) +
           align_up((size_t)mlp_fused16_grid() * f16::kLossStride * sizeof(double), 256);
}

int mlp_fused16_grad(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const float *params,
                     const FusedOffsets &fo, const ppo2hip_rollout &ro, const int32_t *idx, int64_t B,
                     const double *stats, float cliprange, double inv_count, float *grads, float *losses, void *ws,
                     size_t ws_bytes, hipStream_t s) {
    using namespace f16;
    if (ws_bytes < mlp_fused16_workspace_bytes()) {
        set_error("mlp_fused16_grad: workspace too small");
        return PPO2HIP_ENOSPACE;
    }
    const int grid = mlp_fused16_grid();
    float *slabs = (float *)ws;
    double *lpart = (double *)((char *)ws + align_up((size_t)grid * SLAB * sizeof(float), 256));
    Fused16Args g;
    g.obs = (const float *)ro.obs; g.actions = ro.actions; g.returns = ro.returns; g.values = ro.values;
    g.neglogpacs = ro.neglogpacs; g.advs = ro.advs; g.adv_f64 = ro.adv_f64; g.T = ro.T; g.E = ro.E;
    g.idx = idx; g.B = B; g.stats = stats; g.params = params;
    g.a_w1 = fo.a_w1; g.a_w2 = fo.a_w2; g.a_wh = fo.a_wh; g.v_w1 = fo.v_w1; g.v_w2 = fo.v_w2; g.v_wh = fo.v_wh;
    g.logstd_off = fo.logstd; g.D = d.ob_dim; g.A = d.ac_dim; g.R = d.nvalue; g.dist = d.dist;
    g.upstream = lc.upstream; g.cliprange = cliprange; g.ent_coef = lc.ent_coef;
    g.vf_coef = lc.upstream ? lc.vf_coef : 1.0f; g.inv_count = (float)inv_count;
    g.slabs = slabs; g.loss_partial = lpart;
    hipLaunchKernelGGL(mlp_fused16_kernel, dim3(grid), dim3(kThreads), 0, s, g);
    int rc = check_launch("mlp_fused16_kernel");
    if (rc) return rc;
    FusedImage im;
    im.slab_elems = SLAB; im.s_w1 = S_W1; im.s_w2 = S_W2; im.s_wh = S_WH; im.s_bh = S_BH; im.head_cols = 16;
    return fused_finalize(im, slabs, grid / 2, lpart, grid, d, fo, inv_count, grads, losses, s);
}

}  // namespace ppo2hip
