// Forward, loss and input-gradient chain of the fork's own network — two separate 256-128 tanh trunks, R value
// channels (baselines/ppo2/policies.py:139-151 under the loss of baselines/ppo2/common_functions.py:26-54) — in ONE
// launch on 16-sample tiles, for minibatch rows that gather_minibatch has laid out contiguously.
//
// What is fused and what is not.  A trunk's 256 x 128 weight-gradient accumulator is 32768 floats: it fits no wave's
// registers, so the three weight-gradient products stay on the GEMM (fc_wgrad_slabs), fed by what this kernel
// writes: h1, h2, d(loss)/d(head), dz2, dz1.  Everything else — layer 1, layer 2, head, distribution and loss
// arithmetic, and the two input-gradient products — happens between one read of a sample's row and those writes,
// with the activations in registers: per sample and trunk 1.6 KB are written and later read once, where the
// layer-by-layer route moves 5.2 KB through HBM in 13 launches.
//
// Mapping (v_mfma_f32_16x16x4_f32; lane l: li = l & 15 the sample, lk = l >> 4):
//   operands A[i = li][k = lk], B[k = lk][j = li]; result D[row = 4 lk + r][col = li].
//   Units sit on rows, samples on columns, so a layer's accumulator (register r of block m = unit 16 m + 4 lk + r) is
//   the B operand of the next product as it stands; the A operand of that k-step is the weight row of that unit.
// W2 is held ONCE in LDS (128 KB), row-major [in][out] with the out index XOR-ed by
//   f(in) = (in & 15) | (((in ^ (in >> 2)) & 1) << 4),
// which makes both its uses free of bank conflicts within a half-wave: the forward product reads a row `in` per lane
// quarter with lanes along out, the backward product a column segment per lane quarter with lanes along in.
#include "gemm_core.hpp"
#include "kernels.hpp"
#include "rowgather.hpp"

namespace ppo2hip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef short bf16x8v __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// 32 x 32 x 16 product of bf16 operands (lane l: row / column l & 31, k = the eight values of half l >> 5), float32 accumulate:
// result register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
#define MFMA16Q(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, (a)), __builtin_bit_cast(bf16x8v, (b)), (c), 0, 0, 0)

namespace f256 {
constexpr int kThreads = 512, kWaves = 8;
constexpr int H1 = 256, H2 = 128, HB1 = H1 / 16, HB2 = H2 / 16;
constexpr int KQ = 5;                          // layer-1 k-steps: ob_dim + 1 <= 20
constexpr int W2S = 0;                         // [256 in][128 out ^ f(in)]
constexpr int W1LD = H1 + 16;                  // row stride of W1S: the four lane quarters of a read fall on distinct banks
constexpr int W1S = W2S + H1 * H2;             // [20 rows][W1LD]: rows 0..D-1 weights, row D bias, rest zero
constexpr int WHS = W1S + 4 * KQ * W1LD;       // [128 units][16 outputs]
constexpr int B2S = WHS + H2 * 16;             // [128]
constexpr int BHS = B2S + H2;                  // [16]
constexpr int TOTAL = BHS + 16;
static_assert(TOTAL * 4 + 1024 <= 160 * 1024, "LDS budget");
constexpr int kLossStride = 8 + 64;            // the partial-row format of the loss block of fused_finalize_kernel
}  // namespace f256

struct Fused256Args {
    // the rollout arrays as they stand (time-major with map.T / map.E set, or flat rows with map.T = 0), n rows each
    const float *obs;        // [n, D]
    const void *actions;     // [n, A] float (Gaussian) or [n] int32 (Categorical)
    const float *returns, *values, *neglogpacs;      // [n, R], [n, R], [n]
    const void *advs;        // [n] float or double, normalised here with stats (null: taken as they are)
    int adv_f64;
    int64_t n;
    RowMap map;              // minibatch position -> rollout row
    const double *stats;     // {sum, sum of squares, count} of the minibatch's advantages, or null
    float *xg;               // [B, D] out: the observation rows in minibatch order, for the weight-gradient launch (or null)
    int64_t B;
    const float *params;
    int64_t a_w1, a_w2, a_wh, v_w1, v_w2, v_wh, logstd_off;
    int D, A, R, dist, upstream;
    float cliprange, ent_coef, vf_coef, inv_count;
    float *H1[2], *H2[2], *DZ1[2], *DZ2[2], *DOUT[2];    // per trunk (0 policy, 1 value): [B,256] [B,128] [B,256] [B,128] [B,O]
    double *loss_partial;    // [gridDim.x][kLossStride]
};

__global__ __launch_bounds__(f256::kThreads, 2) void mlp_fused256_kernel(Fused256Args g) {
    using namespace f256;
    __shared__ __attribute__((aligned(16))) float lds[TOTAL];
    __shared__ double red[kWaves * 24];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int trunk = blockIdx.x & 1;
    const bool is_value = trunk != 0;
    const int nwg_trunk = gridDim.x >> 1, wg_in_trunk = blockIdx.x >> 1;
    const int64_t o_w1 = is_value ? g.v_w1 : g.a_w1, o_w2 = is_value ? g.v_w2 : g.a_w2, o_wh = is_value ? g.v_wh : g.a_wh;
    const int D = g.D, O = is_value ? g.R : g.A;
    const bool gaussian = g.dist == PPO2HIP_DIST_GAUSSIAN;
    const float *P = g.params;

    // ---- stage the trunk's weights ---------------------------------------------------------------------------
    for (int e = tid; e < H1 * H2; e += kThreads) {
        const int in = e >> 7, out = e & 127;
        const int f = (in & 15) | (((in ^ (in >> 2)) & 1) << 4);
        lds[W2S + in * H2 + (out ^ f)] = P[o_w2 + e];
    }
    for (int e = tid; e < 4 * KQ * H1; e += kThreads) {
        const int k = e >> 8;
        lds[W1S + k * W1LD + (e & 255)] = k <= D ? P[o_w1 + e] : 0.f;      // (w, b) are adjacent in the arena: row D is the bias
    }
    for (int e = tid; e < H2 * 16; e += kThreads) {
        const int u = e >> 4, o = e & 15;
        lds[WHS + e] = o < O ? P[o_wh + (int64_t)u * O + o] : 0.f;
    }
    if (tid < H2) lds[B2S + tid] = P[o_w2 + (int64_t)H1 * H2 + tid];
    if (tid < 16) lds[BHS + tid] = tid < O ? P[o_wh + (int64_t)H2 * O + tid] : 0.f;
    __syncthreads();

    // Lane-constant byte addresses of the operand reads; every read below is one of these plus a compile-time offset
    // that fits the 16-bit offset field of ds_read (the bases are made opaque once per tile, or the compiler hoists
    // one address register per read out of the tile loop and spills them).
    //   W2 forward, k-step (mb, r): in = 16 mb + 4 lk + r, out = 16 m + li -> in*128 + 32 (m >> 1) + ((li ^ f) ^ (m & 1 ? 16 : 0))
    //   W2 backward, k-step (m, r): in = 16 mb + li, out = 16 m + 4 lk + r -> same form with the roles exchanged
    int a_fw[4], a_bw[4];
    {
        const int F = li | (((li ^ (li >> 2)) & 1) << 4);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int inl = 4 * lk + r;
            const int f = inl | (((r ^ lk) & 1) << 4);
            a_fw[r] = 4 * (W2S + inl * H2 + (li ^ f));
            a_bw[r] = 4 * (W2S + li * H2 + ((4 * lk + r) ^ F));
        }
    }
    int a_l1 = 4 * (W1S + lk * W1LD + li);         // + 4 (4 q W1LD + 16 mb)
    int a_hf = 4 * (WHS + lk * 64 + li);           // + 4 (256 m + 16 r): unit 16 m + 4 lk + r, output li
    int a_hb = 4 * (WHS + li * 16 + 4 * lk);       // + 4 (256 m): unit 16 m + li, outputs 4 lk .. 4 lk + 3
    int a_b = 16 * lk;                             // + 4 (B2S + 16 m) or + 4 BHS
    const char *const ldsb = (const char *)lds;
    auto ldf = [ldsb](int base, int off) { return *(const float *)(ldsb + base + off); };
    auto ldf4 = [ldsb](int base, int off) { return *(const f32x4 *)(ldsb + base + off); };

    float inv_std[4] = {0.f, 0.f, 0.f, 0.f};
    float sum_logstd = 0.f;
    if (!is_value && gaussian) {
        for (int k = 0; k < g.A; k++) sum_logstd += P[g.logstd_off + k];
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (4 * lk + r < g.A) inv_std[r] = expf(-P[g.logstd_off + 4 * lk + r]);
    }
    const float sc = g.inv_count / (float)g.R, rR = 1.f / (float)g.R;
    float s_pg = 0.f, s_vf = 0.f, s_ent = 0.f, s_kl = 0.f, s_cf = 0.f;
    float dls[4] = {0.f, 0.f, 0.f, 0.f};          // d(loss)/d(logstd_o), o = 4 lk + r, summed over this lane's samples

    float *const gH1 = g.H1[trunk], *const gH2 = g.H2[trunk], *const gDZ1 = g.DZ1[trunk], *const gDZ2 = g.DZ2[trunk];
    float *const gDOUT = g.DOUT[trunk];
    const int64_t ntiles = (g.B + 15) >> 4;
    const rsrc_t r_obs = make_rsrc(g.obs, g.n * D * 4);
    const rsrc_t r_act = make_rsrc(g.actions, gaussian ? g.n * g.A * 4 : g.n * 4);
    const rsrc_t r_ret = make_rsrc(g.returns, g.n * g.R * 4), r_val = make_rsrc(g.values, g.n * g.R * 4);
    const rsrc_t r_nlp = make_rsrc(g.neglogpacs, g.n * 4), r_adv = make_rsrc(g.advs, g.n * (g.adv_f64 ? 8 : 4));
    const rsrc_t r_idx = make_rsrc(g.map.idx, g.map.idx ? g.B * 4 : 0);
    // rollout row of this lane's sample in a tile (positions past the end read index 0: a valid row whose results are dropped)
    auto tile_row = [&](int64_t tile) -> int {
        const int64_t i = tile * 16 + li;
        return g.map.row_of_sample(g.map.sample_at(r_idx, i, i < g.B));
    };
    // advantage normalisation constants (baselines/ppo2/common_functions.py:67)
    double adv_mean = 0.0, adv_rstd = 1.0;
    if (g.stats) {
        const double cnt = g.stats[2];
        adv_mean = g.stats[0] / cnt;
        double var = g.stats[1] / cnt - adv_mean * adv_mean;
        var = var > 0.0 ? var : 0.0;
        adv_rstd = 1.0 / (sqrt(var) + 1e-8);
    }
    const int64_t wave_global = (int64_t)wg_in_trunk * kWaves + wave, wave_stride = (int64_t)nwg_trunk * kWaves;

    // Two dependent round trips stand before a tile's first product (index, then the row it names): the observation
    // registers of the next tile are loaded while this one is worked on, and the index of the tile after that with them.
    auto load_obs = [&](int grow, float (&x)[KQ]) {
#pragma unroll
        for (int q = 0; q < KQ; q++) x[q] = buf_f32(r_obs, (unsigned)grow * (unsigned)(D * 4) + 4u * (unsigned)(lk + 4 * q));   // past the row's end: the next row, or 0
    };
    int grow_next = tile_row(wave_global), grow_after = tile_row(wave_global + wave_stride);
    float xn[KQ];
    load_obs(grow_next, xn);
    for (int64_t tile = wave_global; tile < ntiles; tile += wave_stride) {
        const int grow = grow_next;
#pragma unroll
        for (int r = 0; r < 4; r++) asm volatile("" : "+v"(a_fw[r]), "+v"(a_bw[r]));
        asm volatile("" : "+v"(a_l1), "+v"(a_hf), "+v"(a_hb), "+v"(a_b));
        const int64_t srow = tile * 16 + li;
        const bool live = srow < g.B;
        const int64_t row = live ? srow : g.B - 1;

        // ---- inputs: this lane's features lk + 4 q of its sample (the ones feature at index D), loss operands ----
        float xr[KQ];
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            const int k = lk + 4 * q;
            xr[q] = k < D ? xn[q] : (k == D ? 1.f : 0.f);
            if (g.xg && !is_value && live && k < D) g.xg[srow * D + k] = xn[q];
        }
        // ---- layer 1 ----------------------------------------------------------------------------------------
        f32x4 h1[HB1];
#pragma unroll
        for (int mb = 0; mb < HB1; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) h1[mb][r] = 0.f;
        // Every product below runs as a software pipeline over its k-steps: the weight operands of step s + 1 are
        // read from LDS while the matrix cores work on step s (a wave has one other wave on its SIMD to hide behind).
        {
            float w[2][HB1];
#pragma unroll
            for (int mb = 0; mb < HB1; mb++) w[0][mb] = ldf(a_l1, 4 * (16 * mb));
#pragma unroll
            for (int q = 0; q < KQ; q++) {
                if (q + 1 < KQ) {
#pragma unroll
                    for (int mb = 0; mb < HB1; mb++) w[(q + 1) & 1][mb] = ldf(a_l1, 4 * (4 * (q + 1) * W1LD + 16 * mb));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < HB1; mb++) h1[mb] = MFMA16(w[q & 1][mb], xr[q], h1[mb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        grow_next = grow_after;
        load_obs(grow_next, xn);
        grow_after = tile_row(tile + 2 * wave_stride);
#pragma unroll
        for (int mb = 0; mb < HB1; mb++) {
#pragma unroll
            for (int r = 0; r < 4; r++) h1[mb][r] = fast_tanh(h1[mb][r]);
            if (live) *(f32x4 *)(gH1 + srow * H1 + 16 * mb + 4 * lk) = h1[mb];
        }

        // ---- layer 2: the bias seeds the accumulators -----------------------------------------------------------
        f32x4 h2[HB2];
#pragma unroll
        for (int m = 0; m < HB2; m++) h2[m] = ldf4(a_b, 4 * (B2S + 16 * m));
        {
            float w[2][HB2];
            auto fetch = [&](int s, float (&o)[HB2]) {      // k-step s = (mb, r): unit 16 mb + 4 lk + r against all 128 outputs
                const int mb = s >> 2, r = s & 3;
                // W2 rows of 16 unit blocks span 128 KB: the upper half goes through a second base
                const int base = a_fw[r] + (mb >= 8 ? 65536 : 0), off = 4 * ((mb & 7) * 16 * H2);
#pragma unroll
                for (int m = 0; m < HB2; m++) o[m] = ldf((m & 1) ? base ^ 64 : base, off + 4 * 32 * (m >> 1));
            };
            fetch(0, w[0]);
#pragma unroll
            for (int s = 0; s < 4 * HB1; s++) {
                if (s + 1 < 4 * HB1) fetch(s + 1, w[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < HB2; m++) h2[m] = MFMA16(w[s & 1][m], h1[s >> 2][s & 3], h2[m]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int m = 0; m < HB2; m++) {
#pragma unroll
            for (int r = 0; r < 4; r++) h2[m][r] = fast_tanh(h2[m][r]);
            if (live) *(f32x4 *)(gH2 + srow * H2 + 16 * m + 4 * lk) = h2[m];
        }

        // ---- loss operands of this lane's outputs ----------------------------------------------------------------
        float adv = 0.f, oldnlp = 0.f, act[4] = {0.f, 0.f, 0.f, 0.f}, ret[4] = {0.f, 0.f, 0.f, 0.f}, oldv[4] = {0.f, 0.f, 0.f, 0.f};
        int act_i = 0;
        if (!is_value) {
            double a;
            if (g.adv_f64) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(r_adv, (int)((unsigned)grow * 8u), 0, 0);
                a = __longlong_as_double((long long)(((unsigned long long)v[1] << 32) | v[0]));
            } else {
                a = (double)buf_f32(r_adv, (unsigned)grow * 4u);
            }
            adv = (float)((a - adv_mean) * adv_rstd);
            oldnlp = buf_f32(r_nlp, (unsigned)grow * 4u);
            if (gaussian) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (4 * lk + r < g.A) act[r] = buf_f32(r_act, ((unsigned)grow * (unsigned)g.A + 4u * lk + r) * 4u);
            } else {
                act_i = buf_i32(r_act, (unsigned)grow * 4u);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * lk + r < g.R) {
                    ret[r] = buf_f32(r_ret, ((unsigned)grow * (unsigned)g.R + 4u * lk + r) * 4u);
                    oldv[r] = buf_f32(r_val, ((unsigned)grow * (unsigned)g.R + 4u * lk + r) * 4u);
                }
        }

        // ---- head: register r of lane quarter lk = output 4 lk + r ---------------------------------------------
        f32x4 out = ldf4(a_b, 4 * BHS);
        {
            f32x4 part[3];          // four chains instead of one of 32 dependent products
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) part[c][r] = 0.f;
#pragma unroll
            for (int m = 0; m < HB2; m++) {
                out = MFMA16(ldf(a_hf, 4 * (256 * m)), h2[m][0], out);
#pragma unroll
                for (int r = 1; r < 4; r++) part[r - 1] = MFMA16(ldf(a_hf, 4 * (256 * m + 16 * r)), h2[m][r], part[r - 1]);
            }
            out = (out + part[0]) + (part[1] + part[2]);
        }

        // ---- loss and d(loss)/d(out) ---------------------------------------------------------------------------
        f32x4 dout;
#pragma unroll
        for (int r = 0; r < 4; r++) dout[r] = 0.f;
        auto quarters_sum = [](float v) {
            v += __shfl_xor(v, 16, 64);
            return v + __shfl_xor(v, 32, 64);
        };
        if (!is_value) {
            float nlp, entropy_row;
            float zq[4] = {0.f, 0.f, 0.f, 0.f}, pr[4] = {0.f, 0.f, 0.f, 0.f}, lp[4] = {0.f, 0.f, 0.f, 0.f};
            if (gaussian) {
                float ss = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (4 * lk + r < g.A) {
                        zq[r] = (act[r] - out[r]) * inv_std[r];
                        ss += zq[r] * zq[r];
                    }
                ss = quarters_sum(ss);
                nlp = 0.5f * ss + kHalfLog2Pi * (float)g.A + sum_logstd;
                entropy_row = sum_logstd + (kHalfLog2Pi + 0.5f) * (float)g.A;
            } else {
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
                z0 = quarters_sum(z0);
                la = quarters_sum(la);
                const float logz = logf(z0);
                nlp = logz - la;
                float Hs = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pr[r] = pr[r] / z0;
                    lp[r] = 4 * lk + r < g.A ? (out[r] - mx) - logz : 0.f;
                    Hs -= pr[r] * lp[r];
                }
                entropy_row = quarters_sum(Hs);
            }
            const float ratio = expf(oldnlp - nlp);
            const float pg1 = -adv * ratio;
            const float pg2 = -adv * fminf(fmaxf(ratio, 1.0f - g.cliprange), 1.0f + g.cliprange);
            const float dnlp = (live && pg1 >= pg2) ? adv * ratio * g.inv_count : 0.f;
            if (live && lk == 0) {
                s_pg += fmaxf(pg1, pg2);
                const float d = nlp - oldnlp;
                s_kl += 0.5f * (d * d);
                s_cf += (fabsf(ratio - 1.0f) > g.cliprange) ? 1.f : 0.f;
                s_ent += entropy_row;
            }
            if (gaussian) {
                const float dent = live ? g.ent_coef * g.inv_count : 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (4 * lk + r < g.A) {
                        dout[r] = dnlp * (-zq[r] * inv_std[r]);
                        dls[r] += dnlp * (1.f - zq[r] * zq[r]) - dent;
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
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * lk + r < g.R && live) {
                    const float v = out[r];
                    const float e1 = v - ret[r];
                    if (g.upstream) {
                        const float dv = v - oldv[r];
                        const float vc = oldv[r] + fminf(fmaxf(dv, -g.cliprange), g.cliprange);
                        const float e2 = vc - ret[r];
                        const float l1 = e1 * e1, l2 = e2 * e2;
                        s_vf += 0.5f * fmaxf(l1, l2) * rR;
                        const float gg = (l1 >= l2) ? e1 : ((dv >= -g.cliprange && dv <= g.cliprange) ? e2 : 0.f);
                        dout[r] = g.vf_coef * sc * gg;
                    } else {
                        s_vf += (e1 * e1) * rR;
                        dout[r] = 2.f * e1 * sc;
                    }
                }
        }
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * lk + r < O) gDOUT[srow * O + 4 * lk + r] = dout[r];
        }

        // ---- head input gradient: k-step r carries outputs 4 lk + r ---------------------------------------------
        f32x4 dz2[HB2];
#pragma unroll
        for (int m = 0; m < HB2; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) dz2[m][r] = 0.f;
        {
            f32x4 wh[HB2];
#pragma unroll
            for (int m = 0; m < HB2; m++) wh[m] = ldf4(a_hb, 4 * 256 * m);
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (r < O) {
#pragma unroll
                    for (int m = 0; m < HB2; m++) dz2[m] = MFMA16(wh[m][r], dout[r], dz2[m]);
                }
        }
#pragma unroll
        for (int m = 0; m < HB2; m++) {
            dz2[m] = dz2[m] - dz2[m] * (h2[m] * h2[m]);
            if (live) *(f32x4 *)(gDZ2 + srow * H2 + 16 * m + 4 * lk) = dz2[m];
        }

        // ---- layer 2 input gradient, eight unit blocks at a time --------------------------------------------------
        // h1 is not held in registers from layer 2 to here (64 of the 256 a lane has): the lane reads back the very
        // addresses it wrote above, through a pointer the compiler cannot see through (it would forward the stores).
        int64_t h1_back = row * H1 + 4 * lk;
        asm volatile("" : "+v"(h1_back));
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x4 dz1[HB1 / 2], h1b[HB1 / 2];
#pragma unroll
            for (int j = 0; j < HB1 / 2; j++) h1b[j] = *(const f32x4 *)(gH1 + h1_back + 16 * (8 * half + j));
#pragma unroll
            for (int j = 0; j < HB1 / 2; j++)
#pragma unroll
                for (int r = 0; r < 4; r++) dz1[j][r] = 0.f;
            float w[2][HB1 / 2];
            auto fetch = [&](int s, float (&o)[HB1 / 2]) {  // k-step s = (m, r): output 16 m + 4 lk + r against this half's units
                const int m = s >> 2, r = s & 3;
                const int base = ((m & 1) ? a_bw[r] ^ 64 : a_bw[r]) + 65536 * half, off = 4 * 32 * (m >> 1);
#pragma unroll
                for (int j = 0; j < HB1 / 2; j++) o[j] = ldf(base, off + 4 * (j * 16 * H2));
            };
            fetch(0, w[0]);
#pragma unroll
            for (int s = 0; s < 4 * HB2; s++) {
                if (s + 1 < 4 * HB2) fetch(s + 1, w[(s + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HB1 / 2; j++) dz1[j] = MFMA16(w[s & 1][j], dz2[s >> 2][s & 3], dz1[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < HB1 / 2; j++) {
                const int mb = 8 * half + j;
                dz1[j] = dz1[j] - dz1[j] * (h1b[j] * h1b[j]);
                if (live) *(f32x4 *)(gDZ1 + srow * H1 + 16 * mb + 4 * lk) = dz1[j];
            }
        }
    }

    // ---- loss partial row of this workgroup: five sums, then the logstd gradients at 8 + o ------------------------
    {
        double vals[5 + 16];
        vals[0] = (double)s_pg; vals[1] = (double)s_vf; vals[2] = (double)s_ent; vals[3] = (double)s_kl; vals[4] = (double)s_cf;
#pragma unroll
        for (int o = 0; o < 16; o++) vals[5 + o] = (lk == (o >> 2)) ? (double)dls[o & 3] : 0.0;
#pragma unroll
        for (int k = 0; k < 21; k++) {
            const double v = wave_sum(vals[k]);
            if (lane == 0) red[wave * 24 + k] = v;
        }
        __syncthreads();
        if (tid < 21) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < kWaves; w++) v += red[w * 24 + tid];
            g.loss_partial[(int64_t)blockIdx.x * kLossStride + (tid < 5 ? tid : 8 + (tid - 5))] = v;
        }
        if (tid >= 5 && tid < 8) g.loss_partial[(int64_t)blockIdx.x * kLossStride + tid] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// All weight and bias gradients of a trunk from what the chain kernel wrote, each activation row read once.
//
// A workgroup owns a run of samples and, over its eight waves, one full set of accumulators:
//   dW2 = h1^T dz2 (256 x 128)   wave (wr, wc) = (wave >> 1, wave & 1) holds rows 64 wr.., columns 64 wc..: 4 x 4 blocks
//   dW1 = [x 1]^T dz1 (32 x 256) wave w holds unit blocks 2 w, 2 w + 1 against both feature blocks; the ones feature at
//                                index D makes row D the bias gradient, as the arena lays (w, b) out
//   dWh = h2^T dout (128 x 16)   wave w holds unit block w
//   db2, dbh                     a product with the 16 x 4 operand "row 0 is all ones, the rest zero": row 0 of the result is
//                                the column sum; wave w holds db2's column block w, wave 0 dbh
// Samples run along k, four per MFMA step, sixteen per slab.  The six operand tiles of a slab (58 KB) are double-buffered
// in LDS: while the matrix cores work on slab i, slab i + 1 travels HBM -> registers -> the other buffer, one barrier
// per slab.  Row pitches are 16 mod 64 floats so the four sample rows of an operand read fall on distinct banks.
// Each workgroup writes its sums to a slab of its own; a second launch adds the slabs in a fixed order.
namespace w256 {
constexpr int kThreads = 512, kWaves = 8, KS = 16;
constexpr int P1 = 272, P2 = 144, PX = 48, PO = 16;
constexpr int oH1 = 0, oDZ1 = oH1 + KS * P1, oH2 = oDZ1 + KS * P1, oDZ2 = oH2 + KS * P2, oX = oDZ2 + KS * P2, oDO = oX + KS * PX;
constexpr int STAGE = oDO + KS * PO;
static_assert(2 * STAGE * 4 <= 160 * 1024, "LDS budget");
// a workgroup's slab: [32 x 256] dW1 rows (feature, the ones feature among them) | [257 x 128] dW2, db2 | [129 x 16] dWh, dbh
constexpr int S1 = 0, S2 = S1 + 32 * 256, S3 = S2 + 257 * 128, SLAB = S3 + 129 * 16;
constexpr int kSlabsPerTrunk = 128;
}  // namespace w256

struct Wgrad256Args {
    const float *X;          // [B, D] observation rows in minibatch order
    const float *H1[2], *H2[2], *DZ1[2], *DZ2[2], *DOUT[2];
    int64_t B, rows_per_wg;
    int D, A, R;
    float *slabs;            // [2][kSlabsPerTrunk][SLAB]
};

__global__ __launch_bounds__(w256::kThreads, 2) void mlp_wgrad256_kernel(Wgrad256Args g) {
    using namespace w256;
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int trunk = blockIdx.x & 1, wg = blockIdx.x >> 1;
    const int64_t s_beg = (int64_t)wg * g.rows_per_wg;
    if (s_beg >= g.B) return;                                  // the reduction knows how many slabs were written
    const int64_t s_end = s_beg + g.rows_per_wg < g.B ? s_beg + g.rows_per_wg : g.B;
    const int D = g.D, O = trunk ? g.R : g.A;
    const float *const gH1 = g.H1[trunk], *const gH2 = g.H2[trunk], *const gDZ1 = g.DZ1[trunk], *const gDZ2 = g.DZ2[trunk];
    const float *const gDO = g.DOUT[trunk], *const gX = g.X;

    // ---- HBM -> registers -> LDS: what this thread carries of a slab ---------------------------------------------------
    const int r64 = tid >> 6, c64 = tid & 63;        // H1 / DZ1: rows r64 and r64 + 8, float4 column c64
    const int r32 = tid >> 5, c32 = tid & 31;        // H2 / DZ2: row r32, float4 column c32;  X: row r32, feature c32
    const int r16 = (tid >> 4) & 15, c16 = tid & 15; // DOUT: row r16, output c16 (threads 0..255)
    f32x4 pH1[2], pDZ1[2], pH2, pDZ2;
    float pX, pDO;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int64_t s0) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int64_t r = s0 + r64 + 8 * e;
            const bool ok = r < s_end;
            pH1[e] = ok ? *(const f32x4 *)(gH1 + r * 256 + 4 * c64) : zero4;
            pDZ1[e] = ok ? *(const f32x4 *)(gDZ1 + r * 256 + 4 * c64) : zero4;
        }
        const int64_t r = s0 + r32;
        const bool ok = r < s_end;
        pH2 = ok ? *(const f32x4 *)(gH2 + r * 128 + 4 * c32) : zero4;
        pDZ2 = ok ? *(const f32x4 *)(gDZ2 + r * 128 + 4 * c32) : zero4;
        pX = ok ? (c32 < D ? gX[r * D + c32] : (c32 == D ? 1.f : 0.f)) : 0.f;
        const int64_t ro = s0 + r16;
        pDO = (tid < 256 && ro < s_end && c16 < O) ? gDO[ro * O + c16] : 0.f;
    };
    auto stash = [&](float *st) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
            *(f32x4 *)(st + oH1 + (r64 + 8 * e) * P1 + 4 * c64) = pH1[e];
            *(f32x4 *)(st + oDZ1 + (r64 + 8 * e) * P1 + 4 * c64) = pDZ1[e];
        }
        *(f32x4 *)(st + oH2 + r32 * P2 + 4 * c32) = pH2;
        *(f32x4 *)(st + oDZ2 + r32 * P2 + 4 * c32) = pDZ2;
        st[oX + r32 * PX + c32] = pX;
        if (c32 < PX - 32) st[oX + r32 * PX + 32 + c32] = 0.f;
        if (tid < 256) st[oDO + r16 * PO + c16] = pDO;
    };

    // dW2 runs on the bf16 matrix pipe as exact piece products (cut_run8 of gemm_core.hpp: three bf16 pieces per value, six
    // products per multiply-add): a slab's 16 samples are the k of one v_mfma_f32_32x32x16_bf16, so a wave's 64 x 64
    // share of dW2 is 2 x 2 blocks of 6 instructions per slab that leave the vector ALUs free for all but 8 of their 32
    // cycles, where the float32 form took 64 instructions of 32 cycles that block them.  The other sums are too small
    // for the 32 x 32 shape and stay float32.
    f32x16v acc2[2][2];
    f32x4 acc1[2][2], acch, accb2, accbh;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[a][b][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int c = 0; c < 2; c++) acc1[f][c][r] = 0.f;
        acch[r] = 0.f; accb2[r] = 0.f; accbh[r] = 0.f;
    }
    const int wr = wave >> 1, wc = wave & 1;
    const float ones_row = li == 0 ? 1.f : 0.f;
    // lane bases (floats) of the operand reads inside a stage; + (4 kk) rows per k-step
    // dW2 operands: lane (l32, hh) takes column 64 wr + 32 a + l32 of the h1 tile (64 wc + 32 b + l32 of the dz2 tile) and
    // the slab's sample rows 2 hh + {0, 1, 4, 5, 8, 9, 12, 13} (any split of the sixteen over the lane halves serves; both
    // operands use the same one, and with pitches of 16 mod 64 floats rows two apart put the halves on different banks)
    const int l32 = lane & 31, hh = lane >> 5;
    const int bA2 = oH1 + 2 * hh * P1 + 64 * wr + l32, bB2 = oDZ2 + 2 * hh * P2 + 64 * wc + l32;
    const int bA1 = oX + lk * PX + li, bB1 = oDZ1 + lk * P1 + 32 * wave + li;
    const int bAh = oH2 + lk * P2 + 16 * wave + li, bBh = oDO + lk * PO + li, bBb = oDZ2 + lk * P2 + 16 * wave + li;

    // operand registers of one k-step (four sample rows): 4 + 4 for dW2, 2 + 2 for dW1, head unit block, dout, dz2 block
    struct Ops { float a1[2], b1[2], ah, bh, bb; };
    auto read_ops = [&](const float *st, int kk, Ops &o) {
#pragma unroll
        for (int f = 0; f < 2; f++) o.a1[f] = st[bA1 + 4 * kk * PX + 16 * f];
#pragma unroll
        for (int c = 0; c < 2; c++) o.b1[c] = st[bB1 + 4 * kk * P1 + 16 * c];
        o.ah = st[bAh + 4 * kk * P2]; o.bh = st[bBh + 4 * kk * PO]; o.bb = st[bBb + 4 * kk * P2];
    };
    // the slab's dW2 products: A blocks cut first, then one B block at a time
    auto multiply2 = [&](const float *st) {
        uint4 ap[2][3];
#pragma unroll
        for (int a = 0; a < 2; a++) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = st[bA2 + ((j & 1) + 4 * (j >> 1)) * P1 + 32 * a];
            cut_run8(v, ap[a]);
        }
#pragma unroll
        for (int b = 0; b < 2; b++) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = st[bB2 + ((j & 1) + 4 * (j >> 1)) * P2 + 32 * b];
            uint4 bp[3];
            cut_run8(v, bp);
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};      // small products first
#pragma unroll
            for (int t = 0; t < 6; t++)
#pragma unroll
                for (int a = 0; a < 2; a++) acc2[a][b] = MFMA16Q(ap[a][pa[t]], bp[pb[t]], acc2[a][b]);
        }
    };
    auto multiply = [&](const Ops &o) {
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int c = 0; c < 2; c++) acc1[f][c] = MFMA16(o.a1[f], o.b1[c], acc1[f][c]);
        acch = MFMA16(o.ah, o.bh, acch);
        accb2 = MFMA16(ones_row, o.bb, accb2);
        if (wave == 0) accbh = MFMA16(ones_row, o.bh, accbh);
    };

    // Slab i is multiplied out of one buffer while slab i + 1 goes from registers into the other in the middle of the
    // iteration (nobody reads that buffer since the last barrier) and slab i + 2 leaves HBM for the registers right
    // after; the operand registers of k-step kk + 1 are read while the matrix cores work on k-step kk.
    fetch(s_beg);
    stash(lds);
    if (s_beg + KS < s_end) fetch(s_beg + KS);
    __syncthreads();
    int buf = 0;
    for (int64_t s0 = s_beg; s0 < s_end; s0 += KS) {
        const float *st = lds + buf * STAGE;
        Ops o[2];
        read_ops(st, 0, o[0]);
        multiply2(st);
#pragma unroll
        for (int kk = 0; kk < KS / 4; kk++) {
            if (kk + 1 < KS / 4) read_ops(st, kk + 1, o[(kk + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            multiply(o[kk & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 1 && s0 + KS < s_end) {
                stash(lds + (buf ^ 1) * STAGE);
                if (s0 + 2 * KS < s_end) fetch(s0 + 2 * KS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        buf ^= 1;
    }

    // ---- this workgroup's slab: result register r of lane (li, lk) = row 4 lk + r, column li of its block ---------------
    float *slab = g.slabs + ((int64_t)trunk * kSlabsPerTrunk + wg) * SLAB;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                slab[S2 + (64 * wr + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hh) * 128 + 64 * wc + 32 * b + l32] = acc2[a][b][r];
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int f = 0; f < 2; f++)
#pragma unroll
            for (int c = 0; c < 2; c++) slab[S1 + (16 * f + 4 * lk + r) * 256 + 32 * wave + 16 * c + li] = acc1[f][c][r];
        slab[S3 + (16 * wave + 4 * lk + r) * 16 + li] = acch[r];
    }
    if (lk == 0) {
        slab[S2 + 256 * 128 + 16 * wave + li] = accb2[0];
        if (wave == 0) slab[S3 + 128 * 16 + li] = accbh[0];
    }
}

// grads[...] = sum over a trunk's slabs, in slab order: 64 consecutive elements x 8 slab groups per workgroup
struct Wgrad256Reduce {
    const float *slabs;
    float *grads;
    int64_t o_w1[2], o_w2[2], o_wh[2];
    int nslabs, D, O[2];
};
__global__ __launch_bounds__(512) void mlp_wgrad256_reduce_kernel(Wgrad256Reduce g) {
    using namespace w256;
    __shared__ float part[8][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, trunk = blockIdx.y;
    const int O = g.O[trunk], n1 = (g.D + 1) * 256, n2 = 257 * 128, n3 = 129 * O;
    const int e = blockIdx.x * 64 + lane;
    int so = -1;
    int64_t dst = 0;
    if (e < n1) {
        so = S1 + e; dst = g.o_w1[trunk] + e;
    } else if (e < n1 + n2) {
        so = S2 + (e - n1); dst = g.o_w2[trunk] + (e - n1);
    } else if (e < n1 + n2 + n3) {
        const int q = e - n1 - n2;
        so = S3 + (q / O) * 16 + q % O; dst = g.o_wh[trunk] + q;
    }
    float s = 0.f;
    if (so >= 0) {
        const float *base = g.slabs + (int64_t)trunk * kSlabsPerTrunk * SLAB + so;
        for (int z0 = grp; z0 < g.nslabs; z0 += 32) {
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) x[u] = z0 + 8 * u < g.nslabs ? base[(int64_t)(z0 + 8 * u) * SLAB] : 0.f;
            s += (x[0] + x[1]) + (x[2] + x[3]);
        }
    }
    part[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && so >= 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q++) t += part[q][lane];
        g.grads[dst] = t;
    }
}

size_t mlp_fused256_wgrad_bytes() { return align_up((size_t)2 * w256::kSlabsPerTrunk * w256::SLAB * sizeof(float), 256); }

int mlp_fused256_wgrad(const ppo2hip_policy_desc &d, const FusedOffsets &fo, const Fused256Buffers &b, int64_t B, float *grads,
                       hipStream_t s) {
    using namespace w256;
    Wgrad256Args g;
    // observation rows in minibatch order: the chain launch's gathered copy, or the rows as they stand when the caller
    // gathered them (random 68-byte rows cost this kernel's slab pipeline 80 us per minibatch when it gathered them itself)
    g.X = b.xg ? b.xg : b.obs;
    for (int t = 0; t < 2; t++) { g.H1[t] = b.H1[t]; g.H2[t] = b.H2[t]; g.DZ1[t] = b.DZ1[t]; g.DZ2[t] = b.DZ2[t]; g.DOUT[t] = b.DOUT[t]; }
    g.B = B;
    g.rows_per_wg = (int64_t)ceil_div(ceil_div(B > 0 ? B : 1, (int64_t)kSlabsPerTrunk), (int64_t)KS) * KS;
    g.D = d.ob_dim; g.A = d.ac_dim; g.R = d.nvalue; g.slabs = b.wslabs;
    hipLaunchKernelGGL(mlp_wgrad256_kernel, dim3(2 * kSlabsPerTrunk), dim3(kThreads), 0, s, g);
    int rc = check_launch("mlp_wgrad256_kernel");
    if (rc) return rc;
    Wgrad256Reduce r;
    r.slabs = b.wslabs; r.grads = grads;
    r.o_w1[0] = fo.a_w1; r.o_w2[0] = fo.a_w2; r.o_wh[0] = fo.a_wh; r.o_w1[1] = fo.v_w1; r.o_w2[1] = fo.v_w2; r.o_wh[1] = fo.v_wh;
    r.nslabs = (int)ceil_div(B > 0 ? B : 1, g.rows_per_wg); r.D = d.ob_dim; r.O[0] = d.ac_dim; r.O[1] = d.nvalue;
    const int omax = d.ac_dim > d.nvalue ? d.ac_dim : d.nvalue;
    const int nmax = (d.ob_dim + 1) * 256 + 257 * 128 + 129 * omax;
    hipLaunchKernelGGL(mlp_wgrad256_reduce_kernel, dim3((unsigned)ceil_div(nmax, 64), 2), dim3(512), 0, s, r);
    return check_launch("mlp_wgrad256_reduce_kernel");
}

bool mlp_fused256_supported(const ppo2hip_policy_desc &d) {
    if (d.kind != PPO2HIP_POLICY_MLP || d.hidden1 != f256::H1 || d.hidden2 != f256::H2) return false;
    return d.ob_dim >= 1 && d.ob_dim + 1 <= 4 * f256::KQ && d.ac_dim <= 16 && d.nvalue <= 8;
}

int mlp_fused256_grid() { return 256; }

size_t mlp_fused256_loss_bytes() { return align_up((size_t)mlp_fused256_grid() * f256::kLossStride * sizeof(double), 256); }

int mlp_fused256_chain(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const float *params, const FusedOffsets &fo,
                       const Fused256Buffers &b, int64_t B, float cliprange, double inv_count, float *losses, float *dlogstd,
                       hipStream_t s) {
    Fused256Args g;
    g.obs = b.obs; g.actions = b.actions; g.returns = b.returns; g.values = b.values; g.neglogpacs = b.neglogpacs;
    g.advs = b.advs; g.adv_f64 = b.adv_f64; g.n = b.n; g.map = make_row_map(b.idx, B, b.T, b.E, b.T > 0 ? (int64_t)b.T * b.E : b.n, fault_word()); g.stats = b.stats;
    if (!g.map.fault) return PPO2HIP_EINVAL;
    g.xg = b.xg; g.B = B; g.params = params;
    g.a_w1 = fo.a_w1; g.a_w2 = fo.a_w2; g.a_wh = fo.a_wh; g.v_w1 = fo.v_w1; g.v_w2 = fo.v_w2; g.v_wh = fo.v_wh; g.logstd_off = fo.logstd;
    g.D = d.ob_dim; g.A = d.ac_dim; g.R = d.nvalue; g.dist = d.dist; g.upstream = lc.upstream;
    g.cliprange = cliprange; g.ent_coef = lc.ent_coef; g.vf_coef = lc.upstream ? lc.vf_coef : 1.0f; g.inv_count = (float)inv_count;
    for (int t = 0; t < 2; t++) { g.H1[t] = b.H1[t]; g.H2[t] = b.H2[t]; g.DZ1[t] = b.DZ1[t]; g.DZ2[t] = b.DZ2[t]; g.DOUT[t] = b.DOUT[t]; }
    g.loss_partial = b.loss_partial;
    const int grid = mlp_fused256_grid();
    hipLaunchKernelGGL(mlp_fused256_kernel, dim3(grid), dim3(f256::kThreads), 0, s, g);
    int rc = check_launch("mlp_fused256_kernel");
    if (rc) return rc;
    return fused_loss_finalize(b.loss_partial, grid, d.dist == PPO2HIP_DIST_GAUSSIAN ? d.ac_dim : 0, inv_count, losses, dlogstd, s);
}

}  // namespace ppo2hip
