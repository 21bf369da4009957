// Reverse scans over the time-major rollout: GAE/returns, A2C discounted returns,
// and the sf01 flatten.  HBM-bound: one lane per (env, channel) column, time walked
// backwards, every access coalesced along the env axis.
#include "kernels.hpp"

// The float64 carry of the scans must round exactly as NumPy's separate multiply and add
// do, so no operation in this file may be contracted into an fma: the Makefile builds this
// file with -ffp-contract=off (a file-scope pragma is not honoured for device code here).

namespace ppo2hip {


// Arithmetic of baselines/ppo2/common_functions.py:325-328 in the reference's evaluation
// order, with explicit round-to-nearest double ops so the compiler cannot contract
// them into FMAs: the float64 carry is then bit-identical to NumPy's.
template <int UNROLL>
__global__ void gae_kernel(const float *__restrict__ rewards, const float *__restrict__ values,
                           const uint8_t *__restrict__ dones, const float *__restrict__ last_values,
                           const uint8_t *__restrict__ last_dones, GammaVec gv, double lam, int T,
                           int E, int R, float *__restrict__ advs, float *__restrict__ returns,
                           float *__restrict__ advs_ori) {
    const int64_t C = (int64_t)E * R;
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int env = (int)(c / R);
    const int r = (int)(c - (int64_t)env * R);
    const double gamma = gv.g[r];
    const double gl = __dmul_rn(gamma, lam);

    double nextvalue = (double)last_values[c];
    double nnt = 1.0 - (double)(last_dones[env] != 0);
    double last = 0.0;

    // UNROLL steps per batch: their loads are independent and in flight together (one memory round
    // trip per batch), the carry chain is arithmetic only.  The last batch is padded with clamped,
    // unused loads instead of falling back to one round trip per step.
    for (int t = T - 1; t >= 0; t -= UNROLL) {
        float rw[UNROLL], vl[UNROLL];
        uint8_t dn[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int tt = t - u >= 0 ? t - u : 0;
            const int64_t off = (int64_t)tt * C + c;
            rw[u] = rewards[off];
            vl[u] = values[off];
            dn[u] = dones[(int64_t)tt * E + env];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (t - u < 0) break;
            const int64_t off = (int64_t)(t - u) * C + c;
            const double v = (double)vl[u];
            double delta = __dadd_rn((double)rw[u], __dmul_rn(__dmul_rn(gamma, nextvalue), nnt));
            delta = __dsub_rn(delta, v);
            last = __dadd_rn(delta, __dmul_rn(__dmul_rn(gl, nnt), last));
            const float a = (float)last;
            const float ret = __fadd_rn(a, vl[u]);
            if (advs) advs[off] = a;
            returns[off] = ret;
            if (advs_ori) advs_ori[off] = __fsub_rn(ret, vl[u]);
            nextvalue = v;
            nnt = 1.0 - (double)(dn[u] != 0);   // dones[t] masks the step from t-1 to t
        }
    }
}

// The bandwidth regime (64 K columns and more): a lane owns TWO adjacent columns — 8-byte loads and stores, a quarter
// of the chip's wave slots still filled at 65536 columns — and keeps two batches of kGaeVecU steps in flight: the loads
// of batch k + 1 are issued before batch k's arithmetic, so a wave always has 16-32 steps (3 loads each) under way.
// The rollout is read once and written once, so every access is marked non-temporal.  Same operations in the same
// order as gae_kernel for each column: the bits are the same.
// R1: one value channel — the two columns are two envs and their done flags one 16-bit load.
typedef float gae_f32x2 __attribute__((ext_vector_type(2)));
constexpr int kGaeVecU = 16, kGaeVecBlock = 64;

template <bool R1>
__global__ __launch_bounds__(kGaeVecBlock) void gae_vec2_kernel(
    const float *__restrict__ rewards, const float *__restrict__ values, const uint8_t *__restrict__ dones,
    const float *__restrict__ last_values, const uint8_t *__restrict__ last_dones, GammaVec gv, double lam, int T, int E,
    int R, float *__restrict__ advs, float *__restrict__ returns, float *__restrict__ advs_ori) {
    constexpr int U = kGaeVecU;
    using F = gae_f32x2;
    const int64_t CV = (int64_t)E * R / 2;                  // column pairs per row
    const int64_t cv = (int64_t)blockIdx.x * kGaeVecBlock + threadIdx.x;
    if (cv >= CV) return;
    const F *rw_p = (const F *)rewards + cv;
    const F *vl_p = (const F *)values + cv;
    F *ad_p = advs ? (F *)advs + cv : nullptr;
    F *rt_p = (F *)returns + cv;
    F *ao_p = advs_ori ? (F *)advs_ori + cv : nullptr;
    int env[2];
    double gamma[2], gl[2], nextvalue[2], nnt[2], last[2];
    {
        const F lv = ((const F *)last_values)[cv];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int64_t c = 2 * cv + i;
            env[i] = (int)(c / R);
            gamma[i] = gv.g[R1 ? 0 : (int)(c - (int64_t)env[i] * R)];
            gl[i] = __dmul_rn(gamma[i], lam);
            nextvalue[i] = (double)lv[i];
            nnt[i] = 1.0 - (double)(last_dones[env[i]] != 0);
            last[i] = 0.0;
        }
    }
    F rwA[U], vlA[U], rwB[U], vlB[U];
    uint32_t dnA[U], dnB[U];               // the pair's two done flags: bits 0-7 and 8-15
    auto fetch = [&](int t, F (&rw)[U], F (&vl)[U], uint32_t (&dn)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int tt = t - u >= 0 ? t - u : 0;       // a last batch that runs past step 0 repeats step 0's loads, unused
            const int64_t off = (int64_t)tt * CV;
            rw[u] = __builtin_nontemporal_load(rw_p + off);
            vl[u] = __builtin_nontemporal_load(vl_p + off);
            if constexpr (R1) {
                dn[u] = __builtin_nontemporal_load((const uint16_t *)dones + off + cv);
            } else {
                const uint8_t *dr = dones + (int64_t)tt * E;
                dn[u] = (uint32_t)dr[env[0]] | ((uint32_t)dr[env[1]] << 8);
            }
        }
    };
    auto compute = [&](int t, F (&rw)[U], F (&vl)[U], uint32_t (&dn)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (t - u < 0) break;
            const int64_t off = (int64_t)(t - u) * CV;
            F a2, r2, o2;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const float vf = vl[u][i];
                const double v = (double)vf;
                double delta = __dadd_rn((double)rw[u][i], __dmul_rn(__dmul_rn(gamma[i], nextvalue[i]), nnt[i]));
                delta = __dsub_rn(delta, v);
                last[i] = __dadd_rn(delta, __dmul_rn(__dmul_rn(gl[i], nnt[i]), last[i]));
                const float a = (float)last[i];
                const float ret = __fadd_rn(a, vf);
                a2[i] = a;
                r2[i] = ret;
                o2[i] = __fsub_rn(ret, vf);
                nextvalue[i] = v;
                nnt[i] = 1.0 - (double)(((dn[u] >> (8 * i)) & 0xffu) != 0);   // dones[t] masks the step from t-1 to t
            }
            if (ad_p) __builtin_nontemporal_store(a2, ad_p + off);
            __builtin_nontemporal_store(r2, rt_p + off);
            if (ao_p) __builtin_nontemporal_store(o2, ao_p + off);
        }
    };
    fetch(T - 1, rwA, vlA, dnA);
    for (int t = T - 1; t >= 0; t -= 2 * U) {
        if (t - U >= 0) fetch(t - U, rwB, vlB, dnB);
        compute(t, rwA, vlA, dnA);
        if (t - U < 0) break;
        if (t - 2 * U >= 0) fetch(t - 2 * U, rwA, vlA, dnA);
        compute(t - U, rwB, vlB, dnB);
    }
}

// Same scan for rollouts with few columns (E * R of a few thousand), where one lane per column is bound
// by the latency of its own loads and by the length of the per-step instruction sequence.  A 256-thread
// block takes 16 columns and walks the rollout in 64-step chunks:
//   all threads  : chunk loads (the next chunk's are in flight while this one is processed), then for
//                  every element delta_t = (r_t + (gamma V_{t+1}) nt_{t+1}) - V_t and decay_t = (gamma lam) nt_{t+1}
//                  — neither depends on the carry;
//   16 lanes     : the carry chain alone, A_t = delta_t + decay_t A_{t+1}: one multiply and one add per step;
//   all threads  : Ret = A + V, Ret - V, and the stores.
// Every value goes through the operations of the step-by-step loop in the same order, so the bits are the same.
constexpr int kGaeCols = 16, kGaeChunk = 64, kGaeThreads = 256;

__global__ __launch_bounds__(kGaeThreads) void gae_tiled_kernel(
    const float *__restrict__ rewards, const float *__restrict__ values, const uint8_t *__restrict__ dones,
    const float *__restrict__ last_values, const uint8_t *__restrict__ last_dones, GammaVec gv, double lam, int T,
    int E, int R, float *__restrict__ advs, float *__restrict__ returns, float *__restrict__ advs_ori) {
    constexpr int N = kGaeChunk * kGaeCols, PER = N / kGaeThreads;
    static_assert(PER == 4, "four elements per thread and chunk");
    __shared__ double s_delta[N], s_decay[N];
    __shared__ float s_vl[N + kGaeCols];               // s_vl / s_nn row 0 = carry-in (V and nonterminal of the step after
    __shared__ double s_nn[N + kGaeCols];              // the chunk's first), rows 1.. = the chunk
    const int64_t C = (int64_t)E * R;
    const int64_t c0 = (int64_t)blockIdx.x * kGaeCols;
    const int tid = threadIdx.x;
    const int nchunks = (T + kGaeChunk - 1) / kGaeChunk;

    // chunk k covers steps t0, t0 - 1, ... with t0 = T - 1 - k * chunk.  Two register sets: the loads of chunk
    // k + 2 are issued while chunk k is processed, so they have a whole chunk's time to arrive.
    auto fetch = [&](int k, float (&rw)[PER], float (&vl)[PER], uint8_t (&dn)[PER]) {
        const int t0 = T - 1 - k * kGaeChunk;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int e = tid + i * kGaeThreads, tt = e / kGaeCols, j = e % kGaeCols;
            const int t = t0 - tt >= 0 ? t0 - tt : 0;
            const int64_t c = c0 + j < C ? c0 + j : C - 1;
            rw[i] = rewards[(int64_t)t * C + c];
            vl[i] = values[(int64_t)t * C + c];
            dn[i] = dones[(int64_t)t * E + c / R];
        }
    };

    // per-column constants of the elements this thread handles (column j = tid % 16 for all of them)
    const int jcol = tid % kGaeCols;
    const int64_t ccol = c0 + jcol < C ? c0 + jcol : C - 1;
    const double gamma = gv.g[(int)(ccol % R)];
    const double gl = __dmul_rn(gamma, lam);
    double last = 0.0;                         // carry of column tid (scanner lanes)
    const bool scanner = tid < kGaeCols && c0 + tid < C;
    if (tid < kGaeCols) {
        s_vl[tid] = last_values[ccol];
        s_nn[tid] = 1.0 - (double)(last_dones[ccol / R] != 0);
    }
    auto chunk = [&](int k, float (&rw)[PER], float (&vl)[PER], uint8_t (&dn)[PER]) {
        const int t0 = T - 1 - k * kGaeChunk;
        const int steps = t0 + 1 < kGaeChunk ? t0 + 1 : kGaeChunk;
        // raw values and nonterminal factors of the chunk, one row below their step (row 0 is the carry-in)
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int e = tid + i * kGaeThreads;
            s_vl[kGaeCols + e] = vl[i];
            s_nn[kGaeCols + e] = 1.0 - (double)(dn[i] != 0);      // dones[t] masks the step from t-1 to t
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int e = tid + i * kGaeThreads;
            const double nv = (double)s_vl[e], nt = s_nn[e];      // the step after this one: one row up
            s_delta[e] = __dsub_rn(__dadd_rn((double)rw[i], __dmul_rn(__dmul_rn(gamma, nv), nt)), (double)vl[i]);
            s_decay[e] = __dmul_rn(gl, nt);
        }
        const float vkeep[PER] = {vl[0], vl[1], vl[2], vl[3]};
        if (k + 2 < nchunks) fetch(k + 2, rw, vl, dn);            // this set is free again: its raw values are in LDS / vkeep
        __syncthreads();
        if (scanner) {
            // the scanning lanes do nothing but the chain: the carry, still in double, goes back into the
            // delta slot; rounding to float32 and the rest happen in the parallel phase below
            constexpr int U = 8;
            int tb = 0;
            for (; tb + U <= steps; tb += U) {
                double d[U], q[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    d[u] = s_delta[(tb + u) * kGaeCols + tid];
                    q[u] = s_decay[(tb + u) * kGaeCols + tid];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    last = __dadd_rn(d[u], __dmul_rn(q[u], last));
                    s_delta[(tb + u) * kGaeCols + tid] = last;
                }
            }
            for (; tb < steps; tb++) {
                last = __dadd_rn(s_delta[tb * kGaeCols + tid], __dmul_rn(s_decay[tb * kGaeCols + tid], last));
                s_delta[tb * kGaeCols + tid] = last;
            }
        }
        if (tid < kGaeCols) {       // carry-in of the next chunk = this chunk's last step
            s_vl[tid] = s_vl[kGaeCols + (steps - 1) * kGaeCols + tid];
            s_nn[tid] = s_nn[kGaeCols + (steps - 1) * kGaeCols + tid];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int e = tid + i * kGaeThreads, tt = e / kGaeCols, j = e % kGaeCols;
            if (t0 - tt >= 0 && c0 + j < C) {
                const int64_t off = (int64_t)(t0 - tt) * C + c0 + j;
                const float a = (float)s_delta[e];
                const float ret = __fadd_rn(a, vkeep[i]);
                if (advs) advs[off] = a;
                returns[off] = ret;
                if (advs_ori) advs_ori[off] = __fsub_rn(ret, vkeep[i]);
            }
        }
    };
    float rwA[PER], vlA[PER], rwB[PER], vlB[PER];
    uint8_t dnA[PER], dnB[PER];
    fetch(0, rwA, vlA, dnA);
    if (nchunks > 1) fetch(1, rwB, vlB, dnB);
    for (int k = 0; k < nchunks; k += 2) {
        chunk(k, rwA, vlA, dnA);
        if (k + 1 < nchunks) chunk(k + 1, rwB, vlB, dnB);
    }
}

// r = reward + gamma*r*(1-done), baselines/a2c/utils.py:158; the scan starts from the
// bootstrap value when the env's last done flag is clear (baselines/a2c/a2c.py:122-123).
template <int UNROLL>
__global__ void discount_kernel(const float *__restrict__ rewards, const uint8_t *__restrict__ dones,
                                const float *__restrict__ last_values, double gamma, int T, int E,
                                float *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double r = 0.0;
    if (last_values && dones[(int64_t)(T - 1) * E + e] == 0) r = (double)last_values[e];
    int t = T - 1;
    for (; t - (UNROLL - 1) >= 0; t -= UNROLL) {
        float rw[UNROLL];
        uint8_t dn[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            rw[u] = rewards[(int64_t)(t - u) * E + e];
            dn[u] = dones[(int64_t)(t - u) * E + e];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            r = __dadd_rn((double)rw[u], __dmul_rn(__dmul_rn(gamma, r), 1.0 - (double)(dn[u] != 0)));
            out[(int64_t)(t - u) * E + e] = (float)r;
        }
    }
    for (; t >= 0; t--) {
        const int64_t off = (int64_t)t * E + e;
        r = __dadd_rn((double)rewards[off], __dmul_rn(__dmul_rn(gamma, r), 1.0 - (double)(dones[off] != 0)));
        out[off] = (float)r;
    }
}

// Y[t] = X[t] + gamma*Y[t+1]*(1 - starts[t+1]) down one column: scipy's lfilter([1],[1,-gamma]) on the
// reversed input (baselines/common/math_util.py:23, float64 out, no flags) and discount_with_boundaries
// (baselines/common/math_util.py:63-73, where Y has X's dtype, so a float32 Y carries the ROUNDED value).
template <class X, class Y, int UNROLL>
__global__ void discount_axis0_kernel(const X *__restrict__ x, const uint8_t *__restrict__ starts, double gamma,
                                      int T, int64_t C, Y *__restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double y = 0.0;
    auto step = [&](int t, X xv) {
        // the carry is zero at t = T-1, so starts[T] is never needed
        const double keep = (starts && t + 1 < T) ? 1.0 - (double)(starts[t + 1] != 0) : 1.0;
        const Y stored = (Y)__dadd_rn((double)xv, __dmul_rn(__dmul_rn(gamma, y), keep));
        out[(int64_t)t * C + c] = stored;
        y = (double)stored;
    };
    int t = T - 1;
    for (; t - (UNROLL - 1) >= 0; t -= UNROLL) {
        X xv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) xv[u] = x[(int64_t)(t - u) * C + c];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) step(t - u, xv[u]);
    }
    for (; t >= 0; t--) step(t, x[(int64_t)t * C + c]);
}

// [T, E, row] -> [E, T, row].  Large rows: one workgroup streams one row.
template <class W>
__global__ void sf01_bigrow_kernel(const W *__restrict__ src, int T, int E, int64_t row_words,
                                   W *__restrict__ dst) {
    const int64_t rowid = blockIdx.x;  // over T*E source rows
    const int t = (int)(rowid / E), e = (int)(rowid - (int64_t)t * E);
    const W *s = src + rowid * row_words;
    W *d = dst + ((int64_t)e * T + t) * row_words;
    for (int64_t i = threadIdx.x; i < row_words; i += blockDim.x) d[i] = s[i];
}

// Small rows: a 16x16 tile of rows goes through LDS so both the reads (along env)
// and the writes (along time) move contiguous segments.
template <class W>
__global__ void sf01_tile_kernel(const W *__restrict__ src, int T, int E, int row_words,
                                 W *__restrict__ dst) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    W *tile = (W *)smem_raw;  // [16 e][16 t][row_words]
    const int t0 = blockIdx.y * 16, e0 = blockIdx.x * 16;
    const int seg = 16 * row_words;
    for (int i = threadIdx.x; i < 16 * seg; i += blockDim.x) {
        const int tl = i / seg, rem = i - tl * seg;
        const int el = rem / row_words, w = rem - el * row_words;
        const int t = t0 + tl, e = e0 + el;
        if (t < T && e < E) tile[(el * 16 + tl) * row_words + w] = src[((int64_t)t * E + e) * row_words + w];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * seg; i += blockDim.x) {
        const int el = i / seg, rem = i - el * seg;
        const int tl = rem / row_words, w = rem - tl * row_words;
        const int t = t0 + tl, e = e0 + el;
        if (t < T && e < E) dst[((int64_t)e * T + t) * row_words + w] = tile[(el * 16 + tl) * row_words + w];
    }
}

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" int ppo2hip_gae_f32(const float *rewards, const float *values, const uint8_t *dones,
                               const float *last_values, const uint8_t *last_dones,
                               const double *gamma, double lam, int T, int E, int R, float *advs,
                               float *returns, float *advs_ori, void *stream) {
    PPO2HIP_REQUIRE(T >= 0 && E >= 0 && R >= 1 && R <= kMaxR, "gae: bad shape T=%d E=%d R=%d (R <= %d)", T, E, R, kMaxR);
    if (T == 0 || E == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(rewards && values && dones && last_values && last_dones && gamma && returns, "gae: null pointer");
    GammaVec gv;
    for (int i = 0; i < kMaxR; i++) gv.g[i] = i < R ? gamma[i] : 0.0;
    const int64_t C = (int64_t)E * R;
    hipStream_t s = (hipStream_t)stream;
    auto aligned = [](const void *p, uintptr_t a) { return ((uintptr_t)p & (a - 1)) == 0; };
    // one lane per column (or column pair) fills the chip from about 64 K columns on; below that, blocks that stream
    // chunks through LDS keep far more loads in flight per column
    if (C < 65536 && T >= 32) {
        hipLaunchKernelGGL(gae_tiled_kernel, dim3((unsigned)ceil_div(C, kGaeCols)), dim3(kGaeThreads), 0, s,
                           rewards, values, dones, last_values, last_dones, gv, lam, T, E, R, advs, returns, advs_ori);
    } else if (C >= 65536 && C % 2 == 0 && aligned(rewards, 8) && aligned(values, 8) && aligned(last_values, 8) &&
               aligned(advs, 8) && aligned(returns, 8) && aligned(advs_ori, 8) && (R > 1 || aligned(dones, 2))) {
        const dim3 grid((unsigned)ceil_div(C / 2, kGaeVecBlock));
        if (R == 1)
            hipLaunchKernelGGL(gae_vec2_kernel<true>, grid, dim3(kGaeVecBlock), 0, s, rewards, values, dones, last_values,
                               last_dones, gv, lam, T, E, R, advs, returns, advs_ori);
        else
            hipLaunchKernelGGL(gae_vec2_kernel<false>, grid, dim3(kGaeVecBlock), 0, s, rewards, values, dones, last_values,
                               last_dones, gv, lam, T, E, R, advs, returns, advs_ori);
    } else {
        const int block = C <= 65536 ? 64 : 256;   // few columns: spread them over more CUs
        hipLaunchKernelGGL(gae_kernel<16>, dim3((unsigned)ceil_div(C, block)), dim3(block), 0, s, rewards, values, dones,
                           last_values, last_dones, gv, lam, T, E, R, advs, returns, advs_ori);
    }
    return check_launch("gae_kernel");
}

extern "C" int ppo2hip_discount_with_dones_f32(const float *rewards, const uint8_t *dones,
                                               const float *last_values, double gamma, int T, int E,
                                               float *out, void *stream) {
    PPO2HIP_REQUIRE(T >= 0 && E >= 0, "discount_with_dones: bad shape T=%d E=%d", T, E);
    if (T == 0 || E == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(rewards && dones && out, "discount_with_dones: null pointer");
    const int block = E <= 65536 ? 64 : 256;
    hipLaunchKernelGGL(discount_kernel<8>, dim3((unsigned)ceil_div(E, block)), dim3(block), 0,
                       (hipStream_t)stream, rewards, dones, last_values, gamma, T, E, out);
    return check_launch("discount_kernel");
}

extern "C" int ppo2hip_discount(const void *x, int x_f64, const uint8_t *starts, double gamma, int T, int64_t C,
                                void *out, int out_f64, void *stream) {
    PPO2HIP_REQUIRE(T >= 0 && C >= 0, "discount: bad shape T=%d C=%lld", T, (long long)C);
    PPO2HIP_REQUIRE(out_f64 || !x_f64, "discount: float64 input needs float64 output");
    if (T == 0 || C == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(x && out, "discount: null pointer");
    const int block = C <= 65536 ? 64 : 256;
    const dim3 grid((unsigned)ceil_div(C, block));
    hipStream_t s = (hipStream_t)stream;
    if (x_f64)
        hipLaunchKernelGGL((discount_axis0_kernel<double, double, 8>), grid, dim3(block), 0, s, (const double *)x, starts,
                           gamma, T, C, (double *)out);
    else if (out_f64)
        hipLaunchKernelGGL((discount_axis0_kernel<float, double, 8>), grid, dim3(block), 0, s, (const float *)x, starts,
                           gamma, T, C, (double *)out);
    else
        hipLaunchKernelGGL((discount_axis0_kernel<float, float, 8>), grid, dim3(block), 0, s, (const float *)x, starts,
                           gamma, T, C, (float *)out);
    return check_launch("discount_axis0_kernel");
}

extern "C" int ppo2hip_sf01(const void *src, int T, int E, int64_t row_bytes, void *dst, void *stream) {
    PPO2HIP_REQUIRE(T >= 0 && E >= 0 && row_bytes > 0, "sf01: bad shape T=%d E=%d row_bytes=%lld", T, E, (long long)row_bytes);
    if (T == 0 || E == 0) return PPO2HIP_OK;
    PPO2HIP_REQUIRE(src && dst, "sf01: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool w4 = (row_bytes % 4 == 0) && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0);
    if (row_bytes >= 256) {
        const int64_t rows = (int64_t)T * E;
        PPO2HIP_REQUIRE(rows < (1ll << 31), "sf01: too many rows");
        if (w4) hipLaunchKernelGGL(sf01_bigrow_kernel<uint32_t>, dim3((unsigned)rows), dim3(256), 0, s,
                                   (const uint32_t *)src, T, E, row_bytes / 4, (uint32_t *)dst);
        else hipLaunchKernelGGL(sf01_bigrow_kernel<uint8_t>, dim3((unsigned)rows), dim3(256), 0, s,
                                (const uint8_t *)src, T, E, row_bytes, (uint8_t *)dst);
    } else {
        dim3 grid((unsigned)ceil_div(E, 16), (unsigned)ceil_div(T, 16));
        const size_t lds = 256 * (size_t)row_bytes;
        if (w4) hipLaunchKernelGGL(sf01_tile_kernel<uint32_t>, grid, dim3(256), lds, s, (const uint32_t *)src,
                                   T, E, (int)(row_bytes / 4), (uint32_t *)dst);
        else hipLaunchKernelGGL(sf01_tile_kernel<uint8_t>, grid, dim3(256), lds, s, (const uint8_t *)src, T,
                                E, (int)row_bytes, (uint8_t *)dst);
    }
    return check_launch("sf01");
}
