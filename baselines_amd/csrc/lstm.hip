// Element-wise halves of the LSTM cells of baselines/a2c/utils.py:86-147 (gates in the order
// i, f, o, u; state [c | h]; c and h zeroed where the done-before-step mask is set).  The
// matrix products (x wx for all steps at once, h wh per step, and their gradients) run on the
// MFMA GEMM; these kernels do the masking, the gate non-linearities and their backward.
// Rows are env-major: row(e, t) = e * T + t (batch_to_seq, a2c/utils.py:70-75).
#include <atomic>
#include <cstdlib>

#include "kernels.hpp"

namespace ppo2hip {

__device__ inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// cm = c_prev * (1 - m), hm = h_prev * (1 - m) for step t of every env
__global__ void lstm_mask_kernel(const float *__restrict__ c_prev, const float *__restrict__ h_prev,
                                 int64_t prev_stride, const uint8_t *__restrict__ masks, int64_t mask_stride,
                                 int nenv, int nh, float *__restrict__ cm, float *__restrict__ hm, int64_t out_stride) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nenv * nh) return;
    const int e = (int)(g / nh), j = (int)(g - (int64_t)e * nh);
    const float keep = masks[(int64_t)e * mask_stride] ? 0.f : 1.f;
    cm[(int64_t)e * out_stride + j] = c_prev[(int64_t)e * prev_stride + j] * keep;
    hm[(int64_t)e * out_stride + j] = h_prev[(int64_t)e * prev_stride + j] * keep;
}

// z rows hold x wx + b on entry (plus nparts partial products of hm wh in zparts, summed here in order) and the
// activated gates (i, f, o, u) on exit.  With mask_next the kernel also prepares the next step's masked states.
__global__ void lstm_gate_kernel(float *__restrict__ z, int64_t z_stride, const float *__restrict__ zparts, int nparts,
                                 const float *__restrict__ cm, int64_t s_stride, int nenv, int nh, float *__restrict__ c,
                                 float *__restrict__ h, float *__restrict__ state_out /* [nenv, 2nh] or null */,
                                 const uint8_t *__restrict__ mask_next, int64_t mask_stride, float *__restrict__ cm_next,
                                 float *__restrict__ hm_next) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nenv * nh) return;
    const int e = (int)(g / nh), j = (int)(g - (int64_t)e * nh);
    float *zr = z + (int64_t)e * z_stride;
    float zv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) zv[q] = zr[q * nh + j];
    const int64_t part_stride = (int64_t)nenv * 4 * nh;
    for (int p = 0; p < nparts; p++) {
        const float *zp = zparts + p * part_stride + (int64_t)e * 4 * nh;
#pragma unroll
        for (int q = 0; q < 4; q++) zv[q] += zp[q * nh + j];
    }
    const float i = sigmoidf(zv[0]), f = sigmoidf(zv[1]), o = sigmoidf(zv[2]), u = tanhf(zv[3]);
    const float cn = f * cm[(int64_t)e * s_stride + j] + i * u;
    const float hn = o * tanhf(cn);
    zr[j] = i; zr[nh + j] = f; zr[2 * nh + j] = o; zr[3 * nh + j] = u;
    c[(int64_t)e * s_stride + j] = cn;
    h[(int64_t)e * s_stride + j] = hn;
    if (state_out) {
        state_out[(int64_t)e * 2 * nh + j] = cn;
        state_out[(int64_t)e * 2 * nh + nh + j] = hn;
    }
    if (mask_next) {
        const float keep = mask_next[(int64_t)e * mask_stride] ? 0.f : 1.f;
        cm_next[(int64_t)e * s_stride + j] = cn * keep;
        hm_next[(int64_t)e * s_stride + j] = hn * keep;
    }
}

// Backward of one step.  dh = dH_head[t] + dhm_next * (1 - m_{t+1}), dc_in = dcm_next * (1 - m_{t+1}), where
// dhm_next is the sum, in order, of the nparts partial products of dz_{t+1} wh^T.
// Writes dz over the gate rows and the new (unmasked) dcm carry.
__global__ void lstm_gate_bwd_kernel(float *__restrict__ gates, int64_t z_stride, const float *__restrict__ c,
                                     const float *__restrict__ cm, const float *__restrict__ dh_head,
                                     int64_t s_stride, const float *__restrict__ dhm_parts, int nparts,
                                     float *__restrict__ dcm_carry, const uint8_t *__restrict__ mask_next,
                                     int64_t mask_stride, int nenv, int nh) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nenv * nh) return;
    const int e = (int)(g / nh), j = (int)(g - (int64_t)e * nh);
    float keep = 0.f;   // no later step: no carry
    if (mask_next) keep = mask_next[(int64_t)e * mask_stride] ? 0.f : 1.f;
    const int64_t sr = (int64_t)e * s_stride + j, cr = (int64_t)e * nh + j;
    float *zr = gates + (int64_t)e * z_stride;
    const float i = zr[j], f = zr[nh + j], o = zr[2 * nh + j], u = zr[3 * nh + j];
    float dhm = 0.f;
    if (mask_next)
        for (int p = 0; p < nparts; p++) dhm += dhm_parts[(int64_t)p * nenv * nh + cr];
    const float dh = dh_head[sr] + dhm * keep;
    const float tc = tanhf(c[sr]);
    const float dc = (mask_next ? dcm_carry[cr] * keep : 0.f) + dh * o * (1.f - tc * tc);
    zr[j] = dc * u * i * (1.f - i);
    zr[nh + j] = dc * cm[sr] * f * (1.f - f);
    zr[2 * nh + j] = dh * tc * o * (1.f - o);
    zr[3 * nh + j] = dc * i * (1.f - u * u);
    dcm_carry[cr] = dc * f;
}

// ---------------------------------------------------------------------------------------
// One launch per step and direction for the plain cell: the step's product and its element-wise half together.
//
// A workgroup (4 waves) owns a tile of 16 envs x 16 hidden units.  The product's k range is split over the four
// waves, each lane quarter owning a run of consecutive k (the order of k inside an MFMA step is free as long as both
// operands agree, and runs of consecutive k load as float4); the four partial tiles meet in LDS and are added in
// wave order.  Thread (env, unit) of the tile then does the cell arithmetic for its element.
//   forward : z[env, g nh + unit] += sum_k hm[env, k] wh[k, g nh + unit] for the four gates g, then the gates
//   backward: dhm[env, unit] = sum_k dz_{t+1}[env, k] wh[unit, k] (k over the 4 nh gate columns), then the gate
//             gradients of step t, which overwrite the step's gate rows
// Operands go from L2 straight into registers (a step's whole operand set is 1.1 MB): all loads of a wave are in
// flight before its first product.  nh must be a multiple of 64.
// ---------------------------------------------------------------------------------------
typedef float lf32x4 __attribute__((ext_vector_type(4)));
#define LSTM_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NH>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(float *__restrict__ z, int64_t z_stride, const float *__restrict__ wh,
                                                            const float *__restrict__ hm, const float *__restrict__ cm,
                                                            int64_t s_stride, int nenv, float *__restrict__ c,
                                                            float *__restrict__ h, float *__restrict__ state_out,
                                                            const uint8_t *__restrict__ mask_next, int64_t mask_stride,
                                                            float *__restrict__ cm_next, float *__restrict__ hm_next) {
    constexpr int KL = NH / 16;           // consecutive k per lane quarter (k range of a wave: NH / 4)
    __shared__ float part[4][4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int u0 = blockIdx.x * 16, e0 = blockIdx.y * 16;
    const int env_a = e0 + li < nenv ? e0 + li : nenv - 1;          // operand row of this lane (clamped; dropped on output)
    const int kb = wave * (NH / 4) + lk * KL;
    lf32x4 a4[KL / 4];
    float b[KL][4];
#pragma unroll
    for (int q = 0; q < KL / 4; q++) a4[q] = *(const lf32x4 *)(hm + (int64_t)env_a * s_stride + kb + 4 * q);
#pragma unroll
    for (int s = 0; s < KL; s++)
#pragma unroll
        for (int g = 0; g < 4; g++) b[s][g] = wh[(int64_t)(kb + s) * (4 * NH) + g * NH + u0 + li];
    __builtin_amdgcn_sched_barrier(0);        // every load of the wave is in flight before its first product
    lf32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; g++) acc[g] = lf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KL; s++)
#pragma unroll
        for (int g = 0; g < 4; g++) acc[g] = LSTM_MFMA16(a4[s >> 2][s & 3], b[s][g], acc[g]);
    // result register r of lane (li, lk): env 4 lk + r, unit li
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int r = 0; r < 4; r++) part[wave][g][4 * lk + r][li] = acc[g][r];
    __syncthreads();
    const int el = tid >> 4, ul = tid & 15, e = e0 + el, j = u0 + ul;
    if (e >= nenv) return;
    float *zr = z + (int64_t)e * z_stride;
    float zv[4];
#pragma unroll
    for (int g = 0; g < 4; g++)
        zv[g] = zr[g * NH + j] + ((part[0][g][el][ul] + part[1][g][el][ul]) + (part[2][g][el][ul] + part[3][g][el][ul]));
    const float i = sigmoidf(zv[0]), f = sigmoidf(zv[1]), o = sigmoidf(zv[2]), u = tanhf(zv[3]);
    const int64_t sr = (int64_t)e * s_stride + j;
    const float cn = f * cm[sr] + i * u;
    const float hn = o * tanhf(cn);
    zr[j] = i; zr[NH + j] = f; zr[2 * NH + j] = o; zr[3 * NH + j] = u;
    c[sr] = cn;
    h[sr] = hn;
    if (state_out) {
        state_out[(int64_t)e * 2 * NH + j] = cn;
        state_out[(int64_t)e * 2 * NH + NH + j] = hn;
    }
    if (mask_next) {
        const float keep = mask_next[(int64_t)e * mask_stride] ? 0.f : 1.f;
        cm_next[sr] = cn * keep;
        hm_next[sr] = hn * keep;
    }
}

// dz_next: the gate-gradient rows of step t + 1 (null at the last step: no carry); gates: step t's rows, in: activated
// gates, out: gate gradients; dcm_carry as in lstm_gate_bwd_kernel.
template <int NH>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(float *__restrict__ gates, const float *__restrict__ dz_next,
                                                            int64_t z_stride, const float *__restrict__ wh,
                                                            const float *__restrict__ c, const float *__restrict__ cm,
                                                            const float *__restrict__ dh_head, int64_t s_stride,
                                                            float *__restrict__ dcm_carry, const uint8_t *__restrict__ mask_next,
                                                            int64_t mask_stride, int nenv) {
    constexpr int KL = NH / 4;            // consecutive k per lane quarter (k = gate column, 4 NH of them; NH per wave)
    __shared__ float part[4][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int u0 = blockIdx.x * 16, e0 = blockIdx.y * 16;
    if (dz_next) {
        const int env_a = e0 + li < nenv ? e0 + li : nenv - 1;
        const int kb = wave * NH + lk * KL;
        const float *ap = dz_next + (int64_t)env_a * z_stride + kb, *bp = wh + (int64_t)(u0 + li) * (4 * NH) + kb;
        lf32x4 acc = lf32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int CH = KL / 4 < 8 ? KL / 4 : 8;       // float4 pairs in flight per round
#pragma unroll
        for (int q0 = 0; q0 < KL / 4; q0 += CH) {
            lf32x4 a4[CH], b4[CH];
#pragma unroll
            for (int q = 0; q < CH; q++) {
                a4[q] = *(const lf32x4 *)(ap + 4 * (q0 + q));
                b4[q] = *(const lf32x4 *)(bp + 4 * (q0 + q));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < CH; q++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc = LSTM_MFMA16(a4[q][r], b4[q][r], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) part[wave][4 * lk + r][li] = acc[r];
    }
    __syncthreads();
    const int el = tid >> 4, ul = tid & 15, e = e0 + el, j = u0 + ul;
    if (e >= nenv) return;
    float keep = 0.f, dhm = 0.f;
    if (dz_next) {
        keep = mask_next[(int64_t)e * mask_stride] ? 0.f : 1.f;
        dhm = (part[0][el][ul] + part[1][el][ul]) + (part[2][el][ul] + part[3][el][ul]);
    }
    const int64_t sr = (int64_t)e * s_stride + j, cr = (int64_t)e * NH + j;
    float *zr = gates + (int64_t)e * z_stride;
    const float i = zr[j], f = zr[NH + j], o = zr[2 * NH + j], u = zr[3 * NH + j];
    const float dh = dh_head[sr] + dhm * keep;
    const float tc = tanhf(c[sr]);
    const float dc = (dz_next ? dcm_carry[cr] * keep : 0.f) + dh * o * (1.f - tc * tc);
    zr[j] = dc * u * i * (1.f - i);
    zr[NH + j] = dc * cm[sr] * f * (1.f - f);
    zr[2 * NH + j] = dh * tc * o * (1.f - o);
    zr[3 * NH + j] = dc * i * (1.f - u * u);
    dcm_carry[cr] = dc * f;
}

template <int NH>
static int launch_step_fwd(float *z, int64_t z_stride, const float *wh, const float *hm, const float *cm, int64_t s_stride,
                           int nenv, float *c, float *h, float *state_out, const uint8_t *mask_next, int64_t mask_stride,
                           float *cm_next, float *hm_next, hipStream_t s) {
    hipLaunchKernelGGL(lstm_step_fwd_kernel<NH>, dim3(NH / 16, (unsigned)ceil_div(nenv, 16)), dim3(256), 0, s, z, z_stride, wh, hm,
                       cm, s_stride, nenv, c, h, state_out, mask_next, mask_stride, cm_next, hm_next);
    return check_launch("lstm_step_fwd_kernel");
}
template <int NH>
static int launch_step_bwd(float *gates, const float *dz_next, int64_t z_stride, const float *wh, const float *c,
                           const float *cm, const float *dh_head, int64_t s_stride, float *dcm_carry,
                           const uint8_t *mask_next, int64_t mask_stride, int nenv, hipStream_t s) {
    hipLaunchKernelGGL(lstm_step_bwd_kernel<NH>, dim3(NH / 16, (unsigned)ceil_div(nenv, 16)), dim3(256), 0, s, gates, dz_next,
                       z_stride, wh, c, cm, dh_head, s_stride, dcm_carry, mask_next, mask_stride, nenv);
    return check_launch("lstm_step_bwd_kernel");
}

// ---------------------------------------------------------------------------------------
// All T steps of the forward recurrence in ONE launch.  A step's kernel above runs 5 us and waits 2-4 us for the step
// before it to drain; here a workgroup keeps its tile (16 envs x 16 units) for the whole sequence: its slice of wh stays
// in registers (loaded once, not once per step), its cell state c stays in the owning thread's register, and only h
// travels — the sixteen workgroups of an env group each publish their 16 x 16 piece of the masked h_t and read the whole
// row block back; env groups never wait for each other.
// The hand-over is one hop: a value travels as a 64-bit word {float bits, tag} written and read with agent-scope atomic
// accesses (they go past the per-XCD caches), tag = launch tag + step, so a reader knows from the word itself whether
// it holds this step's value and simply reads again until all its sixteen words do — no arrival counter, no store
// drain, no second round trip.  Two word buffers alternate by step parity: nobody can write step t + 2 before every
// workgroup of the group has published t + 1, i.e. has finished reading t.  The buffers are zeroed before the launch
// and tags never are zero.  Every wait is bounded: a lane that gives up raises the library's fault word and carries on
// with what it has.  All (NH / 16) x ceil(nenv / 16) workgroups must be resident together: at most one per CU is
// asked for.
// ---------------------------------------------------------------------------------------
struct LstmSeqFwd {
    float *z;                 // [nenv][T][4 NH] gate pre-activations x wx + b in, activated gates out
    const float *wh;
    float *hm, *cm;           // [nenv][T][NH] masked previous states; row t = 0 filled by the caller
    float *c, *h;             // [nenv][T][NH]
    float *state_out;         // [nenv][2 NH] or null
    const uint8_t *masks;     // [nenv][T]
    unsigned long long *xchg; // [2][nenv][NH] words, zeroed by the caller
    unsigned *fault;
    unsigned tag0;            // word tags of this launch: tag0 + t, never 0
    int nenv, T;
};

template <int NH>
__global__ __launch_bounds__(256) void lstm_seq_fwd_kernel(LstmSeqFwd a) {
    constexpr int KL = NH / 16;
    __shared__ float part[4][4][16][17];
    __shared__ float hs[16][NH + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int u0 = blockIdx.x * 16, e0 = blockIdx.y * 16;
    const int kb = wave * (NH / 4) + lk * KL;
    const int64_t s_stride = (int64_t)a.T * NH, z_stride = (int64_t)a.T * 4 * NH;
    float b[KL][4];
#pragma unroll
    for (int s = 0; s < KL; s++)
#pragma unroll
        for (int g = 0; g < 4; g++) b[s][g] = a.wh[(int64_t)(kb + s) * (4 * NH) + g * NH + u0 + li];
    const int el = tid >> 4, ul = tid & 15, e = e0 + el, j = u0 + ul;
    const bool live = e < a.nenv;
    const int ee = live ? e : a.nenv - 1;
    const int64_t sr0 = (int64_t)(live ? e : 0) * s_stride + j;
    float c_prev = live ? a.cm[sr0] : 0.f;             // step 0's masked state; afterwards this thread's own
    bool gave_up = false;
    for (int t = 0; t < a.T; t++) {
        // what the step needs besides h goes out first and arrives while h is waited for
        float *zr = a.z + (int64_t)(live ? e : 0) * z_stride + (int64_t)t * 4 * NH;
        float zx[4];
#pragma unroll
        for (int g = 0; g < 4; g++) zx[g] = zr[g * NH + j];
        const float keep = (t + 1 < a.T && live && a.masks[(int64_t)e * a.T + t + 1]) ? 0.f : 1.f;
        // hm_t of the group's 16 envs, all NH units: thread (el, ul) fetches units ul + 16 p of env el — sixteen lanes
        // read one 128-byte run — and the block goes through LDS into operand order
        if (t == 0) {
            const float *hp = a.hm + (int64_t)ee * s_stride + ul;
#pragma unroll
            for (int p = 0; p < KL; p++) hs[el][ul + 16 * p] = hp[16 * p];
        } else {
            unsigned long long *xp = a.xchg + ((int64_t)(t & 1) * a.nenv + ee) * NH + ul;
            const unsigned want = a.tag0 + (unsigned)t;
            unsigned long long w[KL];
            int tries = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int p = 0; p < KL; p++) w[p] = __hip_atomic_load(xp + 16 * p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int p = 0; p < KL; p++) ok = ok && (unsigned)(w[p] >> 32) == want;
                if (ok || gave_up) break;
                if (++tries > (1 << 18)) {
                    atomicOr(a.fault, kFaultGridSync);
                    gave_up = true;
                    break;
                }
            }
#pragma unroll
            for (int p = 0; p < KL; p++) hs[el][ul + 16 * p] = __uint_as_float((unsigned)w[p]);
        }
        __syncthreads();
        float av[KL];
#pragma unroll
        for (int q = 0; q < KL; q++) av[q] = hs[li][kb + q];
        lf32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; g++) acc[g] = lf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KL; s++)
#pragma unroll
            for (int g = 0; g < 4; g++) acc[g] = LSTM_MFMA16(av[s], b[s][g], acc[g]);
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int r = 0; r < 4; r++) part[wave][g][4 * lk + r][li] = acc[g][r];
        __syncthreads();
        if (live) {
            float zv[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                zv[g] = zx[g] + ((part[0][g][el][ul] + part[1][g][el][ul]) + (part[2][g][el][ul] + part[3][g][el][ul]));
            const float i = sigmoidf(zv[0]), f = sigmoidf(zv[1]), o = sigmoidf(zv[2]), u = tanhf(zv[3]);
            const int64_t sr = sr0 + (int64_t)t * NH;
            const float cn = f * c_prev + i * u;
            const float hn = o * tanhf(cn);
            if (t + 1 < a.T) {
                // the travelling value first: everything else of this step can trail behind it
                const float hk = hn * keep;
                const unsigned long long word = ((unsigned long long)(a.tag0 + (unsigned)(t + 1)) << 32) | __float_as_uint(hk);
                __hip_atomic_store(a.xchg + ((int64_t)((t + 1) & 1) * a.nenv + e) * NH + j, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c_prev = cn * keep;
                a.cm[sr + NH] = c_prev;
                a.hm[sr + NH] = hk;
            } else if (a.state_out) {
                a.state_out[(int64_t)e * 2 * NH + j] = cn;
                a.state_out[(int64_t)e * 2 * NH + NH + j] = hn;
            }
            zr[j] = i; zr[NH + j] = f; zr[2 * NH + j] = o; zr[3 * NH + j] = u;
            a.c[sr] = cn;
            a.h[sr] = hn;
        }
        __syncthreads();          // the partial tiles are free for the next step
    }
}

template <int NH>
static int launch_seq_fwd(const LstmSeqFwd &a, hipStream_t s) {
    hipLaunchKernelGGL(lstm_seq_fwd_kernel<NH>, dim3(NH / 16, (unsigned)ceil_div(a.nenv, 16)), dim3(256), 0, s, a);
    return check_launch("lstm_seq_fwd_kernel");
}

// the whole-sequence kernels need all their workgroups resident at once: one per CU of THIS device is what they may
// count on (a partitioned or smaller GPU takes the per-step launches instead)
bool lstm_seq_ok(int nenv, int nh) {
    static const bool off = getenv("PPO2HIP_LSTM_SEQ_OFF") != nullptr;
    const int cus = device_cu_count();
    return !off && (nh == 64 || nh == 128 || nh == 256 || nh == 512) &&
           (int64_t)(nh / 16) * ceil_div(nenv, 16) <= (cus < 256 ? cus : 256);
}
size_t lstm_seq_xchg_bytes(int nenv, int nh) { return (size_t)2 * (nh / 16) * nenv * nh * sizeof(unsigned long long); }

// tags of a launch: T + 1 consecutive values, never 0, different from launch to launch
static unsigned next_seq_tag(int T) {
    static std::atomic<unsigned> g_tag{1u};
    unsigned t = g_tag.fetch_add((unsigned)T + 1u);
    if (t > 0xffffffffu - 2u * ((unsigned)T + 1u)) { g_tag.store(1u); t = g_tag.fetch_add((unsigned)T + 1u); }
    return t;
}

int lstm_seq_fwd(float *z, const float *wh, float *hm, float *cm, int nenv, int T, int nh, float *c, float *h, float *state_out,
                 const uint8_t *masks, void *xchg, hipStream_t s) {
    if (nenv <= 0 || T <= 0) return 0;
    unsigned *fault = fault_word();
    if (!fault) return PPO2HIP_EINVAL;
    PPO2HIP_CHECK_HIP(hipMemsetAsync(xchg, 0, (size_t)2 * nenv * nh * sizeof(unsigned long long), s));
    LstmSeqFwd a{z, wh, hm, cm, c, h, state_out, masks, (unsigned long long *)xchg, fault, next_seq_tag(T), nenv, T};
    switch (nh) {
        case 64: return launch_seq_fwd<64>(a, s);
        case 128: return launch_seq_fwd<128>(a, s);
        case 256: return launch_seq_fwd<256>(a, s);
        case 512: return launch_seq_fwd<512>(a, s);
        default: break;
    }
    set_error("lstm_seq_fwd: no instance for nlstm=%d", nh);
    return PPO2HIP_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------------------
// All T steps of the backward recurrence in one launch, same hand-over as the forward.  What travels is not the gate
// gradients of a step (16 envs x 4 NH per group) but what each workgroup's own 64 gate columns contribute to
// dhm = dz wh^T for ALL NH units — a 16 x NH tile per workgroup, computed as soon as its gate gradients exist, from its
// slice of wh in registers — so a workgroup reads 16 producers x (16 envs x 16 units) words per step, the same volume
// as the forward, and adds them in producer order.  dcm stays in the owning thread's register.
// ---------------------------------------------------------------------------------------
struct LstmSeqBwd {
    float *gates;             // [nenv][T][4 NH] activated gates in, gate gradients out
    const float *wh;
    const float *c, *cm;      // [nenv][T][NH]
    const float *dh_head;     // [nenv][T][NH]
    const uint8_t *masks;     // [nenv][T]
    unsigned long long *xchg; // [2][NH / 16][nenv][NH] words, zeroed by the caller
    unsigned *fault;
    unsigned tag0;
    int nenv, T;
};

template <int NH>
__global__ __launch_bounds__(256) void lstm_seq_bwd_kernel(LstmSeqBwd a) {
    constexpr int NUB = NH / 16;          // workgroups per env group = producers of a sum
    constexpr int NBW = NH / 64;          // 16-unit blocks of the partial product per wave
    __shared__ float dzs[16][68];         // this workgroup's gate gradients of the step, [env][gate 16 + unit]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int ub = blockIdx.x, u0 = ub * 16, e0 = blockIdx.y * 16;
    const int64_t s_stride = (int64_t)a.T * NH, z_stride = (int64_t)a.T * 4 * NH;
    // B operand of the partial product: B[k][n] = wh[unit n][column of local k], local k = gate 16 + unit
    float bw[16][NBW];
#pragma unroll
    for (int s = 0; s < 16; s++) {
        const int kl = 4 * s + lk, col = (kl >> 4) * NH + u0 + (kl & 15);
#pragma unroll
        for (int nb = 0; nb < NBW; nb++) bw[s][nb] = a.wh[(int64_t)(16 * (NBW * wave + nb) + li) * (4 * NH) + col];
    }
    const int el = tid >> 4, ul = tid & 15, e = e0 + el, j = u0 + ul;
    const bool live = e < a.nenv;
    const int ee = live ? e : a.nenv - 1;          // rows past the last env re-read that env's inputs (same env group) and store nothing
    float dcm = 0.f;
    bool gave_up = false;
    for (int t = a.T - 1; t >= 0; t--) {
        const bool last = t == a.T - 1;
        const int64_t sr = (int64_t)ee * s_stride + (int64_t)t * NH + j;
        float *zr = a.gates + (int64_t)ee * z_stride + (int64_t)t * 4 * NH;
        // the step's own inputs first; they arrive while the sums are waited for
        const float gi = zr[j], gf = zr[NH + j], go = zr[2 * NH + j], gu = zr[3 * NH + j];
        const float cv = a.c[sr], cmv = a.cm[sr], dhh = a.dh_head[sr];
        const float keep = (!last && a.masks[(int64_t)ee * a.T + t + 1]) ? 0.f : 1.f;
        float dhm = 0.f;
        if (!last && live) {          // a row past the last env publishes zeros and needs no sum: it never waits
            unsigned long long *xp = a.xchg + (((int64_t)(t & 1) * NUB) * a.nenv + ee) * NH + j;
            const unsigned want = a.tag0 + (unsigned)t;
            unsigned long long w[NUB];
            int tries = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int p = 0; p < NUB; p++)
                    w[p] = __hip_atomic_load(xp + (int64_t)p * a.nenv * NH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int p = 0; p < NUB; p++) ok = ok && (unsigned)(w[p] >> 32) == want;
                if (ok || gave_up) break;
                if (++tries > (1 << 18)) {
                    atomicOr(a.fault, kFaultGridSync);
                    gave_up = true;
                    break;
                }
            }
#pragma unroll
            for (int p = 0; p < NUB; p++) dhm += __uint_as_float((unsigned)w[p]);
        }
        const float dh = dhh + dhm * keep;
        const float tc = tanhf(cv);
        const float dc = (last ? 0.f : dcm * keep) + dh * go * (1.f - tc * tc);
        const float d0 = dc * gu * gi * (1.f - gi), d1 = dc * cmv * gf * (1.f - gf), d2 = dh * tc * go * (1.f - go),
                    d3 = dc * gi * (1.f - gu * gu);
        dcm = dc * gf;
        if (t > 0) {
            dzs[el][ul] = live ? d0 : 0.f; dzs[el][16 + ul] = live ? d1 : 0.f;
            dzs[el][32 + ul] = live ? d2 : 0.f; dzs[el][48 + ul] = live ? d3 : 0.f;
        }
        if (live) { zr[j] = d0; zr[NH + j] = d1; zr[2 * NH + j] = d2; zr[3 * NH + j] = d3; }
        if (t > 0) {
            __syncthreads();
            // what these 64 gate columns add to dhm of step t - 1, for all NH units: out before anything else
            lf32x4 acc[NBW];
#pragma unroll
            for (int nb = 0; nb < NBW; nb++) acc[nb] = lf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; s++) {
                const float av = dzs[li][4 * s + lk];
#pragma unroll
                for (int nb = 0; nb < NBW; nb++) acc[nb] = LSTM_MFMA16(av, bw[s][nb], acc[nb]);
            }
            const unsigned long long tag = (unsigned long long)(a.tag0 + (unsigned)(t - 1)) << 32;
            unsigned long long *op = a.xchg + (((int64_t)((t - 1) & 1) * NUB + ub) * a.nenv) * NH;
#pragma unroll
            for (int nb = 0; nb < NBW; nb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int env = e0 + 4 * lk + r;
                    if (env < a.nenv)
                        __hip_atomic_store(op + (int64_t)env * NH + 16 * (NBW * wave + nb) + li, tag | __float_as_uint(acc[nb][r]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            __syncthreads();          // the gradient tile is free for the next step
        }
    }
}

template <int NH>
static int launch_seq_bwd(const LstmSeqBwd &a, hipStream_t s) {
    hipLaunchKernelGGL(lstm_seq_bwd_kernel<NH>, dim3(NH / 16, (unsigned)ceil_div(a.nenv, 16)), dim3(256), 0, s, a);
    return check_launch("lstm_seq_bwd_kernel");
}

int lstm_seq_bwd(float *gates, const float *wh, const float *c, const float *cm, const float *dh_head, const uint8_t *masks,
                 int nenv, int T, int nh, void *xchg, hipStream_t s) {
    if (nenv <= 0 || T <= 0) return 0;
    unsigned *fault = fault_word();
    if (!fault) return PPO2HIP_EINVAL;
    PPO2HIP_CHECK_HIP(hipMemsetAsync(xchg, 0, lstm_seq_xchg_bytes(nenv, nh), s));
    LstmSeqBwd a{gates, wh, c, cm, dh_head, masks, (unsigned long long *)xchg, fault, next_seq_tag(T), nenv, T};
    switch (nh) {
        case 64: return launch_seq_bwd<64>(a, s);
        case 128: return launch_seq_bwd<128>(a, s);
        case 256: return launch_seq_bwd<256>(a, s);
        case 512: return launch_seq_bwd<512>(a, s);
        default: break;
    }
    set_error("lstm_seq_bwd: no instance for nlstm=%d", nh);
    return PPO2HIP_EUNSUPPORTED;
}

// widths with a compiled instance (the loops over k are unrolled per width)
#define LSTM_FOR_WIDTH(nh, CALL)                 \
    switch (nh) {                                \
        case 64: return CALL(64);                \
        case 128: return CALL(128);              \
        case 256: return CALL(256);              \
        case 512: return CALL(512);              \
        default: break;                          \
    }

bool lstm_step_fused_width(int nh) { return nh == 64 || nh == 128 || nh == 256 || nh == 512; }

int lstm_step_fwd(float *z, int64_t z_stride, const float *wh, const float *hm, const float *cm, int64_t s_stride, int nenv,
                  int nh, float *c, float *h, float *state_out, const uint8_t *mask_next, int64_t mask_stride, float *cm_next,
                  float *hm_next, hipStream_t s) {
    if (nenv <= 0) return 0;
#define CALL(N) launch_step_fwd<N>(z, z_stride, wh, hm, cm, s_stride, nenv, c, h, state_out, mask_next, mask_stride, cm_next, hm_next, s)
    LSTM_FOR_WIDTH(nh, CALL)
#undef CALL
    set_error("lstm_step_fwd: no instance for nlstm=%d", nh);
    return PPO2HIP_EUNSUPPORTED;
}

int lstm_step_bwd(float *gates, const float *dz_next, int64_t z_stride, const float *wh, const float *c, const float *cm,
                  const float *dh_head, int64_t s_stride, float *dcm_carry, const uint8_t *mask_next, int64_t mask_stride,
                  int nenv, int nh, hipStream_t s) {
    if (nenv <= 0) return 0;
#define CALL(N) launch_step_bwd<N>(gates, dz_next, z_stride, wh, c, cm, dh_head, s_stride, dcm_carry, mask_next, mask_stride, nenv, s)
    LSTM_FOR_WIDTH(nh, CALL)
#undef CALL
    set_error("lstm_step_bwd: no instance for nlstm=%d", nh);
    return PPO2HIP_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------------------
// Layer-normalised cell (baselines/a2c/utils.py:110-147):
//   z = LN(x wx; gx, bx) + LN(hm wh; gh, bh) + b,   c = f cm + i u,   h = o tanh(LN(c; gc, bc)),
//   LN(x; g, b) = (x - mean) / sqrt(var + 1e-5) * g + b over the row, var = mean((x - mean)^2).
// One 256-thread block per row; thread `tid` owns hidden units tid, tid + 256, ... (up to 4, so
// nh <= 1024) and for each of them the four gate columns.  Row sums run in double.
// ---------------------------------------------------------------------------------------
constexpr int kLnThreads = 256;
constexpr int kLnUnits = 4;          // hidden units per thread
constexpr float kLnEps = 1e-5f;

// xhat = (x - mean) * rstd in place, rstd[row] saved; rows are `stride` floats apart
__global__ __launch_bounds__(kLnThreads) void ln_rows_kernel(float *__restrict__ x, int64_t stride, int n,
                                                             float *__restrict__ rstd) {
    __shared__ double red[17];
    float *row = x + (int64_t)blockIdx.x * stride;
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += kLnThreads) s += (double)row[j];
    const float mean = (float)(block_sum(s, red) / n);
    double q = 0.0;
    for (int j = threadIdx.x; j < n; j += kLnThreads) {
        const float d = row[j] - mean;
        q += (double)d * (double)d;
    }
    const float r = 1.0f / sqrtf((float)(block_sum(q, red) / n) + kLnEps);
    for (int j = threadIdx.x; j < n; j += kLnThreads) row[j] = (row[j] - mean) * r;
    if (threadIdx.x == 0) rstd[blockIdx.x] = r;
}

struct LnParams {
    const float *gx, *bx, *gh, *bh, *b, *gc, *bc;
};

// Step t of every env (block = env).  zx: normalised x wx rows; zh: raw hm wh rows on entry, normalised on
// exit; gates out (activated i, f, o, u); c, cn (normalised c), h.  Row of env e is e * stride-of-that-array.
__global__ __launch_bounds__(kLnThreads) void lnlstm_step_fwd_kernel(
    const float *__restrict__ zx, float *__restrict__ zh, int64_t z_stride, float *__restrict__ gates, LnParams P,
    const float *__restrict__ cm, int64_t s_stride, int nh, float *__restrict__ c, float *__restrict__ cn,
    float *__restrict__ h, float *__restrict__ rstd_h, float *__restrict__ rstd_c, int64_t r_stride,
    float *__restrict__ state_out) {
    __shared__ double red[17];
    const int e = blockIdx.x, tid = threadIdx.x, n4 = 4 * nh;
    const float *zxr = zx + (int64_t)e * z_stride;
    float *zhr = zh + (int64_t)e * z_stride, *gr = gates + (int64_t)e * z_stride;
    float v[kLnUnits][4];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kLnUnits; k++) {
        const int j = tid + k * kLnThreads;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            v[k][g] = j < nh ? zhr[g * nh + j] : 0.f;
            s += (double)v[k][g];
        }
    }
    const float mean = (float)(block_sum(s, red) / n4);
    double q = 0.0;
#pragma unroll
    for (int k = 0; k < kLnUnits; k++)
#pragma unroll
        for (int g = 0; g < 4; g++)
            if (tid + k * kLnThreads < nh) {
                const float d = v[k][g] - mean;
                q += (double)d * (double)d;
            }
    const float rh = 1.0f / sqrtf((float)(block_sum(q, red) / n4) + kLnEps);
    float cv[kLnUnits], ov[kLnUnits];
    double sc = 0.0;
#pragma unroll
    for (int k = 0; k < kLnUnits; k++) {
        const int j = tid + k * kLnThreads;
        cv[k] = 0.f; ov[k] = 0.f;
        if (j < nh) {
            float z[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = g * nh + j;
                const float xh = (v[k][g] - mean) * rh;
                zhr[col] = xh;
                z[g] = (zxr[col] * P.gx[col] + P.bx[col]) + (xh * P.gh[col] + P.bh[col]) + P.b[col];
            }
            const float i = sigmoidf(z[0]), f = sigmoidf(z[1]), o = sigmoidf(z[2]), u = tanhf(z[3]);
            gr[j] = i; gr[nh + j] = f; gr[2 * nh + j] = o; gr[3 * nh + j] = u;
            cv[k] = f * cm[(int64_t)e * s_stride + j] + i * u;
            ov[k] = o;
            sc += (double)cv[k];
        }
    }
    const float mc = (float)(block_sum(sc, red) / nh);
    double qc = 0.0;
#pragma unroll
    for (int k = 0; k < kLnUnits; k++)
        if (tid + k * kLnThreads < nh) {
            const float d = cv[k] - mc;
            qc += (double)d * (double)d;
        }
    const float rc = 1.0f / sqrtf((float)(block_sum(qc, red) / nh) + kLnEps);
#pragma unroll
    for (int k = 0; k < kLnUnits; k++) {
        const int j = tid + k * kLnThreads;
        if (j < nh) {
            const float ch = (cv[k] - mc) * rc;
            const float hn = ov[k] * tanhf(ch * P.gc[j] + P.bc[j]);
            const int64_t sr = (int64_t)e * s_stride + j;
            c[sr] = cv[k];
            cn[sr] = ch;
            h[sr] = hn;
            if (state_out) {
                state_out[(int64_t)e * 2 * nh + j] = cv[k];
                state_out[(int64_t)e * 2 * nh + nh + j] = hn;
            }
        }
    }
    if (tid == 0) {
        rstd_h[(int64_t)e * r_stride] = rh;
        rstd_c[(int64_t)e * r_stride] = rc;
    }
}

// Backward of step t (block = env).  gates in: activated i, f, o, u; out: d(loss)/dz.  dl out: d(loss)/d(LN(c)).
// dzx / dzh out: gradients w.r.t. the raw products x wx / hm wh (through their layer norms).
__global__ __launch_bounds__(kLnThreads) void lnlstm_step_bwd_kernel(
    float *__restrict__ gates, const float *__restrict__ zx, const float *__restrict__ zh, float *__restrict__ dzx,
    float *__restrict__ dzh, int64_t z_stride, LnParams P, const float *__restrict__ cm, const float *__restrict__ cn,
    const float *__restrict__ dh_head, float *__restrict__ dl, int64_t s_stride, const float *__restrict__ rstd_x,
    const float *__restrict__ rstd_h, const float *__restrict__ rstd_c, int64_t r_stride,
    const float *__restrict__ dhm_next, float *__restrict__ dcm_carry, const uint8_t *__restrict__ mask_next,
    int64_t mask_stride, int nh) {
    __shared__ double red[17];
    const int e = blockIdx.x, tid = threadIdx.x, n4 = 4 * nh;
    float keep = 0.f;   // no later step: no carry
    if (mask_next) keep = mask_next[(int64_t)e * mask_stride] ? 0.f : 1.f;
    float *gr = gates + (int64_t)e * z_stride;
    const float *zxr = zx + (int64_t)e * z_stride, *zhr = zh + (int64_t)e * z_stride;
    const float rx = rstd_x[(int64_t)e * r_stride], rh = rstd_h[(int64_t)e * r_stride], rc = rstd_c[(int64_t)e * r_stride];
    float g4[kLnUnits][4], dch[kLnUnits], chv[kLnUnits], dov[kLnUnits];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < kLnUnits; k++) {
        const int j = tid + k * kLnThreads;
        dch[k] = 0.f; chv[k] = 0.f; dov[k] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; g++) g4[k][g] = 0.f;
        if (j < nh) {
            const int64_t sr = (int64_t)e * s_stride + j, cr = (int64_t)e * nh + j;
#pragma unroll
            for (int g = 0; g < 4; g++) g4[k][g] = gr[g * nh + j];
            const float dh = dh_head[sr] + (mask_next ? dhm_next[cr] * keep : 0.f);
            const float ch = cn[sr];
            const float tl = tanhf(ch * P.gc[j] + P.bc[j]);
            dov[k] = dh * tl;
            const float dlnc = dh * g4[k][2] * (1.f - tl * tl);
            dl[sr] = dlnc;
            chv[k] = ch;
            dch[k] = dlnc * P.gc[j];
            s1 += (double)dch[k];
            s2 += (double)dch[k] * (double)ch;
        }
    }
    const float m1 = (float)(block_sum(s1, red) / nh), m2 = (float)(block_sum(s2, red) / nh);
    float dz[kLnUnits][4];
    double ax1 = 0.0, ax2 = 0.0, ah1 = 0.0, ah2 = 0.0;
    float xx[kLnUnits][4], xh[kLnUnits][4];
#pragma unroll
    for (int k = 0; k < kLnUnits; k++) {
        const int j = tid + k * kLnThreads;
#pragma unroll
        for (int g = 0; g < 4; g++) { dz[k][g] = 0.f; xx[k][g] = 0.f; xh[k][g] = 0.f; }
        if (j < nh) {
            const int64_t sr = (int64_t)e * s_stride + j, cr = (int64_t)e * nh + j;
            const float i = g4[k][0], f = g4[k][1], o = g4[k][2], u = g4[k][3];
            const float dc = (mask_next ? dcm_carry[cr] * keep : 0.f) + rc * (dch[k] - m1 - chv[k] * m2);
            dz[k][0] = dc * u * i * (1.f - i);
            dz[k][1] = dc * cm[sr] * f * (1.f - f);
            dz[k][2] = dov[k] * o * (1.f - o);
            dz[k][3] = dc * i * (1.f - u * u);
            dcm_carry[cr] = dc * f;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = g * nh + j;
                gr[col] = dz[k][g];
                xx[k][g] = zxr[col];
                xh[k][g] = zhr[col];
                const float dx = dz[k][g] * P.gx[col], dhh = dz[k][g] * P.gh[col];
                ax1 += (double)dx; ax2 += (double)dx * (double)xx[k][g];
                ah1 += (double)dhh; ah2 += (double)dhh * (double)xh[k][g];
            }
        }
    }
    const float mx1 = (float)(block_sum(ax1, red) / n4), mx2 = (float)(block_sum(ax2, red) / n4);
    const float mh1 = (float)(block_sum(ah1, red) / n4), mh2 = (float)(block_sum(ah2, red) / n4);
    float *dxr = dzx + (int64_t)e * z_stride, *dhr = dzh + (int64_t)e * z_stride;
#pragma unroll
    for (int k = 0; k < kLnUnits; k++) {
        const int j = tid + k * kLnThreads;
        if (j < nh) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = g * nh + j;
                dxr[col] = rx * (dz[k][g] * P.gx[col] - mx1 - xx[k][g] * mx2);
                dhr[col] = rh * (dz[k][g] * P.gh[col] - mh1 - xh[k][g] * mh2);
            }
        }
    }
}

// out_prod[j] = sum_r a[r, j] * b[r, j] (b null: plain column sums) over `rows` rows, fixed order;
// the gain / bias gradients of the layer norms.  One thread per column, rows read in batches.
__global__ void ln_colsum_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t rows, int n,
                                 float *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    constexpr int U = 8;
    for (int64_t r0 = 0; r0 < rows; r0 += U) {
        float x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t r = r0 + u < rows ? r0 + u : rows - 1;
            x[u] = a[r * n + j];
            y[u] = b ? b[r * n + j] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            if (r0 + u < rows) s += (double)x[u] * (double)y[u];
    }
    out[j] = (float)s;
}

int ln_rows(float *x, int64_t stride, int64_t rows, int n, float *rstd, hipStream_t s) {
    if (rows == 0) return 0;
    hipLaunchKernelGGL(ln_rows_kernel, dim3((unsigned)rows), dim3(kLnThreads), 0, s, x, stride, n, rstd);
    return check_launch("ln_rows_kernel");
}

int lnlstm_step_fwd(const float *zx, float *zh, int64_t z_stride, float *gates, const LnLstmParams &p, const float *cm,
                    int64_t s_stride, int nenv, int nh, float *c, float *cn, float *h, float *rstd_h, float *rstd_c,
                    int64_t r_stride, float *state_out, hipStream_t s) {
    if (nh > kLnThreads * kLnUnits) {
        set_error("lnlstm: nlstm=%d above the %d the cell kernels cover", nh, kLnThreads * kLnUnits);
        return PPO2HIP_EUNSUPPORTED;
    }
    LnParams P{p.gx, p.bx, p.gh, p.bh, p.b, p.gc, p.bc};
    hipLaunchKernelGGL(lnlstm_step_fwd_kernel, dim3(nenv), dim3(kLnThreads), 0, s, zx, zh, z_stride, gates, P, cm, s_stride,
                       nh, c, cn, h, rstd_h, rstd_c, r_stride, state_out);
    return check_launch("lnlstm_step_fwd_kernel");
}

int lnlstm_step_bwd(float *gates, const float *zx, const float *zh, float *dzx, float *dzh, int64_t z_stride,
                    const LnLstmParams &p, const float *cm, const float *cn, const float *dh_head, float *dl,
                    int64_t s_stride, const float *rstd_x, const float *rstd_h, const float *rstd_c, int64_t r_stride,
                    const float *dhm_next, float *dcm_carry, const uint8_t *mask_next, int64_t mask_stride, int nenv,
                    int nh, hipStream_t s) {
    LnParams P{p.gx, p.bx, p.gh, p.bh, p.b, p.gc, p.bc};
    hipLaunchKernelGGL(lnlstm_step_bwd_kernel, dim3(nenv), dim3(kLnThreads), 0, s, gates, zx, zh, dzx, dzh, z_stride, P, cm,
                       cn, dh_head, dl, s_stride, rstd_x, rstd_h, rstd_c, r_stride, dhm_next, dcm_carry, mask_next,
                       mask_stride, nh);
    return check_launch("lnlstm_step_bwd_kernel");
}

int ln_colsum(const float *a, const float *b, int64_t rows, int n, float *out, hipStream_t s) {
    hipLaunchKernelGGL(ln_colsum_kernel, dim3((unsigned)ceil_div(n, 64)), dim3(64), 0, s, a, b, rows, n, out);
    return check_launch("ln_colsum_kernel");
}

int lstm_mask(const float *c_prev, const float *h_prev, int64_t prev_stride, const uint8_t *masks, int64_t mask_stride,
              int nenv, int nh, float *cm, float *hm, int64_t out_stride, hipStream_t s) {
    const int64_t n = (int64_t)nenv * nh;
    hipLaunchKernelGGL(lstm_mask_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, c_prev, h_prev, prev_stride,
                       masks, mask_stride, nenv, nh, cm, hm, out_stride);
    return check_launch("lstm_mask_kernel");
}

int lstm_gates(float *z, int64_t z_stride, const float *zparts, int nparts, const float *cm, int64_t s_stride, int nenv,
               int nh, float *c, float *h, float *state_out, const uint8_t *mask_next, int64_t mask_stride, float *cm_next,
               float *hm_next, hipStream_t s) {
    const int64_t n = (int64_t)nenv * nh;
    hipLaunchKernelGGL(lstm_gate_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, z, z_stride, zparts, nparts, cm,
                       s_stride, nenv, nh, c, h, state_out, mask_next, mask_stride, cm_next, hm_next);
    return check_launch("lstm_gate_kernel");
}

int lstm_gates_bwd(float *gates, int64_t z_stride, const float *c, const float *cm, const float *dh_head,
                   int64_t s_stride, const float *dhm_parts, int nparts, float *dcm_carry, const uint8_t *mask_next,
                   int64_t mask_stride, int nenv, int nh, hipStream_t s) {
    const int64_t n = (int64_t)nenv * nh;
    hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, gates, z_stride, c, cm,
                       dh_head, s_stride, dhm_parts, nparts, dcm_carry, mask_next, mask_stride, nenv, nh);
    return check_launch("lstm_gate_bwd_kernel");
}

}  // namespace ppo2hip
