// Fused minibatch gradient of the two-trunk tanh MLP policy, 64-64 hidden units, on
// 16-sample tiles: gather -> advantage normalisation -> forward -> distribution + PPO loss ->
// backward -> weight-gradient accumulation in ONE launch.  Same mathematics as
// mlp_fused.hip (baselines/a2c/policies.py:116-146 under the loss of
// baselines/ppo2/common_functions.py:26-63), re-tiled so that TWO waves share every SIMD.
//
// Why 16-sample tiles: a wave's per-tile state halves (activations 16 registers per layer,
// LDS tiles 11 KB), which brings a wave under 256 registers, so a 512-thread workgroup
// (8 waves) fits one CU: while one wave of a SIMD waits on LDS or runs tanh / loss VALU work,
// the other keeps the matrix pipe busy.
//
// Mapping (v_mfma_f32_16x16x4_f32, exact fp32 fma chain; lane l: li = l & 15, lk = l >> 4):
//   operands  A[i = li][k = lk], B[k = lk][j = li];  result  D[row = 4 lk + r][col = li], r = 0..3.
//   Samples sit on the column (li) axis, units on rows: a layer's accumulator is directly the B
//   operand of the next product; its register r supplies unit 4 lk + r, so the A operand for
//   that k-step is the weight row of unit 16 mb + 4 lk + r.
//
// LDS images are laid out for 16-byte reads: every weight matrix is stored "packed", the four
// values one lane feeds to four consecutive products adjacent (one ds_read_b128 per four MFMAs),
// and the per-wave transposed tiles [unit][16 samples] keep sample s at column
// 4 ((s & 3) ^ swizzle(row)) + (s >> 2), so a lane's four samples {lk, lk + 4, lk + 8, lk + 12}
// of a weight-gradient product are one 16-byte chunk, and the XOR (2 on rows 8..15 of every
// 16) spreads the four 16-lane groups of a ds_read_b128 over distinct 16-byte bank slots.
// Head rows are permuted (accumulator row 4 lk + r holds output lk + 4 r) so that outputs
// 0..O-1 occupy the first ceil(O / 4) registers and the input gradient skips the rest.
#include <cstdlib>

#include "gemm_core.hpp"
#include "kernels.hpp"
#include "rowgather.hpp"

namespace ppo2hip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// 16 x 16 x 32 product of bf16 operands (eight per lane and operand, k = 8 lk .. 8 lk + 7), float32 accumulate
#define MFMA32B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
// 32 x 32 x 16 product of bf16 operands (lane l: row / column l & 31, k = the eight values of half l >> 5), float32 accumulate:
// result register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
#define MFMA16Q(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

namespace f16 {
constexpr int kThreads = 512;
constexpr int kWaves = 8;
constexpr int H = 64;             // both hidden widths
constexpr int HB = H / 16;        // 16-unit blocks
// weights (floats)
constexpr int W1P = 0;                         // [32 inputs][16 li][4 mb]    unit = 16 mb + li; row D = bias
// layer 2's weights as bf16 pieces (three per weight, see cut8) in the A-operand order of the 16 x 16 x 32 product:
// [k-step q (2)][16-row block (4)][piece (3)][lane (64)] x 8 bf16; slot s of lane quarter lk stands for contraction
// unit 32 q + 4 lk + s (s < 4) or 32 q + 16 + 4 lk + (s - 4): what the lane holds of two adjacent accumulator blocks
constexpr int W2Q = W1P + 32 * H;              // forward:  rows = output units, contraction over input units (scaled for tanh)
constexpr int W2TQ = W2Q + 2 * 4 * 3 * 64 * 4; // backward: rows = input units, contraction over output units
constexpr int B2 = W2TQ + 2 * 4 * 3 * 64 * 4;  // [64]
constexpr int WHP = B2 + H;                    // [16 k4][16 rows][4 r]       accumulator row -> output (row >> 2) + 4 (row & 3)
constexpr int WHTP = WHP + H * 16;             // [4 lk][64 units][4 r]       output = lk + 4 r
constexpr int BHP = WHTP + H * 16;             // [16 rows]
constexpr int WHV = BHP + 16;                  // [64 units]  the single head column of a one-output value trunk
constexpr int WEIGHTS = WHV + H;
// per-wave tiles (floats), all [rows][16 samples] in the swizzled column order
constexpr int XA = 0;                          // [32 features]   row D = ones, rows above D = zeros
constexpr int TA = XA + 32 * 16;               // [64 units]      h1, then d(loss)/d(pre-activation 1)
constexpr int TB = TA + H * 16;                // [64 units]      h2, then d(loss)/d(pre-activation 2)
constexpr int TC = TB + H * 16;                // [16 outputs]    d(loss)/d(head output)
constexpr int SPARE = TC + 16 * 16;              // [64] landing zone for masked-off observation stores
constexpr int WAVE = SPARE + 64;
constexpr int TOTAL = WEIGHTS + kWaves * WAVE;
// gradient image of one trunk (floats): [w1 32x64 | w2 64x64 | b2 64 | wh 64x16 | bh 16]
constexpr int S_W1 = 0;
constexpr int S_W2 = 32 * H;
constexpr int S_B2 = S_W2 + H * H;
constexpr int S_WH = S_B2 + H;
constexpr int S_BH = S_WH + H * 16;
constexpr int SLAB = S_BH + 16;
static_assert(4 * SLAB <= TOTAL, "LDS must hold four gradient images for the cross-wave reduction");
static_assert(TOTAL * 4 + 1024 <= 160 * 1024, "LDS budget");
constexpr int kLossStride = 8 + 64;
constexpr int kLossVals = 5;
}  // namespace f16

struct Fused16Args {
    const float *obs;
    const void *actions;
    const float *returns, *values, *neglogpacs;
    const void *advs;
    int adv_f64, T, E;
    int64_t n;               // rows in every rollout array
    unsigned nsamples;       // sample numbers the index list may name (T E, or n for flat arrays)
    unsigned *fault;         // the device's fault word
    const int32_t *idx;
    int64_t B;
    const double *stats;
    const float *params;
    int64_t a_w1, a_w2, a_wh, v_w1, v_w2, v_wh, logstd_off;
    int D, A, R, dist, upstream;
    float cliprange, ent_coef, vf_coef, inv_count;
    unsigned div_magic;      // smp / T = (t + ((smp - t) >> div_sh1)) >> div_sh2 with t = mulhi(div_magic, smp)
    int div_sh1, div_sh2;
    int nwg_policy;          // workgroups 0 .. nwg_policy-1 run the policy trunk, the rest the value trunk
    float *slabs;            // [workgroup][SLAB]: the policy trunk's images, then the value trunk's
    double *loss_partial;    // [gridDim.x][kLossStride]
};

// tanh of a pre-activation that arrives already multiplied by kTanhScale = 2 log2(e) (the forward copies of the
// weights and biases carry the factor): tanh z = 1 - 2 / (2^(kTanhScale z) + 1).  v_exp_f32 and v_rcp_f32 are 1 ulp:
// |error| <= ~1.5e-7 absolute.  The add and the final fma run on register pairs (v_pk_add_f32 / v_pk_fma_f32: two
// values per lane and issue slot); with the fp32 matrix products sharing the vector ALUs, every slot saved here is a
// slot for them.
constexpr float kTanhScale = 2.8853900817779268f;
__device__ __forceinline__ f32x4 tanh_prescaled4(f32x4 z) {
    f32x4 t;
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = __builtin_amdgcn_exp2f(z[r]);
    t = t + 1.0f;
#pragma unroll
    for (int r = 0; r < 4; r++) t[r] = __builtin_amdgcn_rcpf(t[r]);
    return 1.0f - 2.0f * t;
}
// MINUS dz * tanh'(z) from the saved activation t = tanh z, on register pairs: dz (t t - 1) is a v_pk_fma_f32 with
// the inline constant -1 and a v_pk_mul_f32, where 1 - t t would first have to negate t.  The backward pass applies
// it once per layer, so the sign returns at layer 1; what is built from the layer-2 gradient alone (dW2, db2) is
// accumulated negated and has its sign put right when the images are written out.
__device__ __forceinline__ f32x4 neg_times_tanh_grad4(f32x4 dz, f32x4 t) { return dz * (t * t - 1.0f); }

// A float32 is the exact sum of three bf16 numbers: its 24 significand bits cut into runs of eight by truncation
// (x = hi + mid + lo, each piece what truncating the remainder to bf16 leaves).  Layer 2's two 64 x 64 products per
// tile run on the bf16 matrix pipe from such pieces: weights cut once per launch, activations per tile.  Of the nine
// piece products the six whose weight is 2^-16 of the result or more are kept (hi hi, hi mid, mid hi, hi lo, lo hi,
// mid mid): every kept product is exact in the float32 accumulator, what is dropped is below 2^-24 of the term it
// belongs to — the rounding class of the float32 fma chain itself — and the bf16 pipe retires the 64-deep contraction
// in 2 x 6 instructions of 16 cycles where the float32 form takes 16 of 32.
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {        // the bf16 truncations of a (low half) and b (high half)
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ void cut8(const f32x4 &v0, const f32x4 &v1, uint4 &hi, uint4 &mid, uint4 &lo) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const f32x4 &v = d < 2 ? v0 : v1;
        const f32x2 x = {v[2 * (d & 1)], v[2 * (d & 1) + 1]};
        const f32x2 xh = {__uint_as_float(__float_as_uint(x[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(x[1]) & 0xffff0000u)};
        const f32x2 r1 = x - xh;                                          // exact: the low sixteen significand bits
        const f32x2 rh = {__uint_as_float(__float_as_uint(r1[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(r1[1]) & 0xffff0000u)};
        const f32x2 r2 = r1 - rh;                                         // exact: at most eight bits, a bf16 as it stands
        h[d] = pack_hi16(x[0], x[1]);
        m[d] = pack_hi16(r1[0], r1[1]);
        l[d] = pack_hi16(r2[0], r2[1]);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    mid = make_uint4(m[0], m[1], m[2], m[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// acc[m] += A_q,m x B over the six kept piece products, small ones first; the blocks' accumulators take turns so that
// consecutive instructions do not depend on each other
#define PIECE_PRODUCTS(acc, m0, a, b, NB)                                               \
    {                                                                                   \
        constexpr int pa_[6] = {2, 0, 1, 1, 0, 0}, pb_[6] = {0, 2, 1, 0, 1, 0};         \
        _Pragma("unroll") for (int t_ = 0; t_ < 6; t_++)                                \
            _Pragma("unroll") for (int m_ = 0; m_ < NB; m_++)                           \
                acc[(m0) + m_] = MFMA32B(a[m_][pa_[t_]], b[pb_[t_]], acc[(m0) + m_]);   \
    }

// KQ = k-steps of layer 1 (ob_dim + 1 <= 4 KQ; also the observation registers per lane);
// WIDE = ob_dim + 1 > 16 with the second block of dW1 rows as matrix products;
// XV = 1 or 2: ob_dim = 15 + XV, the one or two feature rows past the first sixteen (x16 if any, and the ones row)
//      get their weight gradients from vector fmas on the operand chunks the first block reads anyway — a second
//      16-row block of products would multiply fourteen rows of zeros.
template <int KQ, bool WIDE, int XV>
__global__ __launch_bounds__(f16::kThreads, 2) void mlp_fused16_kernel(Fused16Args g) {
    using namespace f16;
    __shared__ __attribute__((aligned(16))) float lds[TOTAL];
    __shared__ double red[kWaves * kLossVals];
#define LDS4(off) (*(const f32x4 *)(lds + (off)))
#define LDSQ(base, q, m, p) (*(const uint4 *)(lds + (base) + 4 * ((((q) * 4 + (m)) * 3 + (p)) * 64 + lane)))

    constexpr int PB = WIDE ? 1 : 2;      // accumulator blocks a piece product serves at a time (register budget)
    // Layer 2's weight gradient h1^T dz2 contracts over the tile's 16 samples: as 32 x 32 x 16 bf16 products (2 x 2
    // blocks, six piece products each) it takes 24 matrix instructions that leave the vector ALUs free for all but 8 of
    // their 32 cycles, where the float32 form takes 64 of 32 cycles that block them; the operands are the rows the
    // float32 form reads from the tiles, cut in registers (8 values per lane and 32-row block).
    constexpr bool DW2Q = !WIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const bool is_value = (int)blockIdx.x >= g.nwg_policy;
    const int nwg_trunk = is_value ? (int)gridDim.x - g.nwg_policy : g.nwg_policy;
    const int wg_in_trunk = is_value ? (int)blockIdx.x - g.nwg_policy : (int)blockIdx.x;
    const int64_t o_w1 = is_value ? g.v_w1 : g.a_w1, o_w2 = is_value ? g.v_w2 : g.a_w2, o_wh = is_value ? g.v_wh : g.a_wh;
    const int D = g.D;
    const int O = is_value ? g.R : g.A;
    const int nr = (O + 3) >> 2;                      // head registers in use
    const bool gaussian = g.dist == PPO2HIP_DIST_GAUSSIAN;
    // A value trunk with ONE output runs its head on the vector ALUs: the 16-row matrix shape would spend 36
    // products per tile (forward, weight gradient, input gradient) on one useful row in sixteen.
    const bool vhead = is_value && g.R == 1;

    // ---- first tiles' gather goes out before anything else: two dependent round trips (index, then the row it names)
    //      that the weight staging below covers --------------------------------------------------------------------
    const int64_t ntiles = (g.B + 15) >> 4;
    // Tiles are dealt to SIMDs, not to waves: waves w and w + 4 share SIMD w (and its vector / matrix ALUs), so what
    // bounds a workgroup is the tile count of its busiest SIMD.  SIMD q of the trunk's 4 nwg_trunk takes tiles
    // q, q + 4 nwg_trunk, ...; its two waves alternate over that list, so an odd count costs half a tile per wave, not one.
    const int64_t nsimd = (int64_t)nwg_trunk * 4;
    const int64_t wave_global = (int64_t)wg_in_trunk * 4 + (wave & 3) + nsimd * (wave >> 2);
    const int64_t wave_stride = 2 * nsimd;

    const rsrc_t r_obs = make_rsrc(g.obs, g.n * D * 4);
    const rsrc_t r_act = make_rsrc(g.actions, gaussian ? g.n * g.A * 4 : g.n * 4);
    const rsrc_t r_ret = make_rsrc(g.returns, g.n * g.R * 4), r_val = make_rsrc(g.values, g.n * g.R * 4);
    const rsrc_t r_nlp = make_rsrc(g.neglogpacs, g.n * 4), r_adv = make_rsrc(g.advs, g.n * (g.adv_f64 ? 8 : 4));
    const rsrc_t r_idx = make_rsrc(g.idx, g.idx ? g.B * 4 : 0);

    // rollout row of this lane's sample in a tile (row 0 stands in for samples past the end: the
    // index load past the list's end returns 0); sample numbers are int32, so rows are too
    auto tile_row = [&](int64_t tile) -> int {
        const int64_t i = tile * 16 + li;
        int smp;
        if (g.idx) {
            smp = buf_i32(r_idx, (unsigned)(i < g.B ? i : g.B) * 4u);
            if ((unsigned)smp >= g.nsamples) {       // a stale list: reported, not followed
                atomicOr(g.fault, kFaultIndexRange);
                smp = 0;
            }
        } else {
            smp = i < g.B ? (int)i : 0;
        }
        if (g.T == 0) return smp;
        // env = smp / T by multiplication (Granlund-Montgomery round-up form; magic and shifts from the host)
        const unsigned n = (unsigned)smp, t = __umulhi(g.div_magic, n);
        const int env = (int)((t + ((n - t) >> g.div_sh1)) >> g.div_sh2);
        return (smp - env * g.T) * g.E + env;
    };
    // this lane's share of its sample's observation row: features lk + 4 u (reads past the row's
    // end bring the next row's values or 0, and are parked in the spare row below)
    float xr[KQ];
    auto load_obs = [&](int row) {
        const unsigned base = (unsigned)row * (unsigned)(D * 4) + (unsigned)(lk * 4);
#pragma unroll
        for (int u = 0; u < KQ; u++) xr[u] = buf_f32(r_obs, base + 16u * u);
    };

    const float sc = g.inv_count / (float)g.R, rR = 1.f / (float)g.R;      // value-loss scalings
    int row = tile_row(wave_global), row_next = tile_row(wave_global + wave_stride);
    load_obs(row);
    // ---- stage this trunk's weights in the packed layouts ------------------------------------
    {
        const float *P = g.params;
#pragma unroll
        for (int u = 0; u < 32 * H / kThreads; u++) {
            const int e = tid + u * kThreads;
            const int k = e >> 6, j = e & 63;                              // input row k, unit j (coalesced read)
            const float v = k <= D ? P[o_w1 + (int64_t)k * H + j] : 0.f;   // rows 0..D-1 weights, row D = bias
            lds[W1P + k * 64 + (j & 15) * 4 + (j >> 4)] = kTanhScale * v;  // forward copy: pre-activations arrive scaled for tanh
        }
        {
            unsigned short *const fq = (unsigned short *)(lds + W2Q), *const bq = (unsigned short *)(lds + W2TQ);
            // where contraction unit c (0..63) sits: k-step, lane quarter, slot
            auto slot = [](int c, int &q, int &lq, int &sl) { q = c >> 5; lq = (c & 15) >> 2; sl = (c & 3) + ((c >> 4) & 1) * 4; };
            auto cut = [](float x, unsigned short (&pc)[3]) {
                const unsigned hb = __float_as_uint(x) & 0xffff0000u;
                const float r1 = x - __uint_as_float(hb);
                const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(mb);
                pc[0] = (unsigned short)(hb >> 16); pc[1] = (unsigned short)(mb >> 16); pc[2] = (unsigned short)(__float_as_uint(r2) >> 16);
            };
#pragma unroll
            for (int u = 0; u < H * H / kThreads; u++) {
                const int e = tid + u * kThreads;
                const int k = e >> 6, j = e & 63;                              // W2[input unit k][output unit j] (coalesced read)
                const float v = P[o_w2 + e];
                unsigned short pc[3];
                int q, lq, sl;
                cut(kTanhScale * v, pc);                                       // forward copy: pre-activations arrive scaled for tanh
                slot(k, q, lq, sl);
#pragma unroll
                for (int p = 0; p < 3; p++) fq[((((q * 4 + (j >> 4)) * 3 + p) * 64) + lq * 16 + (j & 15)) * 8 + sl] = pc[p];
                cut(v, pc);                                                    // backward copy
                slot(j, q, lq, sl);
#pragma unroll
                for (int p = 0; p < 3; p++) bq[((((q * 4 + (k >> 4)) * 3 + p) * 64) + lq * 16 + (k & 15)) * 8 + sl] = pc[p];
            }
        }
        if (tid < H) lds[B2 + tid] = kTanhScale * P[o_w2 + (int64_t)H * H + tid];
#pragma unroll
        for (int u = 0; u < H * 16 / kThreads; u++) {
            const int e = tid + u * kThreads;
            const int k = e >> 4, o = e & 15;                              // hidden unit k, real output o
            const float v = o < O ? P[o_wh + (int64_t)k * O + o] : 0.f;
            const int row = 4 * (o & 3) + (o >> 2);                        // accumulator row of output o
            lds[WHP + ((k >> 2) * 16 + row) * 4 + (k & 3)] = v;
            lds[WHTP + ((o & 3) * 64 + k) * 4 + (o >> 2)] = v;
        }
        if (tid < H) lds[WHV + tid] = vhead ? P[o_wh + tid] : 0.f;
        if (tid < 16) {
            const int o = (tid >> 2) + 4 * (tid & 3);                      // output held by accumulator row tid
            lds[BHP + tid] = o < O ? P[o_wh + (int64_t)H * O + o] : 0.f;
        }
        // feature rows above the ones row multiply as zeros (layer 1's k padding, the second block of dW1 rows); every
        // other tile is written whole before it is read
        for (int e = tid; e < kWaves * 32 * 16; e += kThreads) lds[WEIGHTS + (e >> 9) * WAVE + XA + (e & 511)] = 0.f;
    }
    __syncthreads();

    float *const xa = lds + WEIGHTS + wave * WAVE + XA;
    float *const ta = lds + WEIGHTS + wave * WAVE + TA;
    float *const tb = lds + WEIGHTS + wave * WAVE + TB;
    float *const tc = lds + WEIGHTS + wave * WAVE + TC;
    if (lane < 16) xa[D * 16 + lane] = 1.f;            // the ones feature (its column order is irrelevant)
    __builtin_amdgcn_wave_barrier();

    // lane-constant tile offsets.  Element (row, sample s) of a tile sits at
    //   row * 16 + 4 ((s & 3) ^ swz(row)) + (s >> 2),   swz(row) = 2 on rows 8..15 of every 16.
    const int c0 = ((li & 3) << 2) + (li >> 2);                         // sample li in a row with swz 0
    const int c1 = (((li & 3) ^ 2) << 2) + (li >> 2);                   // ... with swz 2
    // accumulator element (block m, register r) = unit 16 m + 4 lk + r of sample li: swz = lk & 2
    const int dpos = 4 * lk * 16 + ((lk & 2) ? c1 : c0);                // + (16 m + r) * 16
    // 16-byte chunk {samples lk, lk + 4, lk + 8, lk + 12} of row 16 m + li: swz = (li >> 2) & 2
    const int rpos = li * 16 + ((lk ^ ((li >> 2) & 2)) << 2);           // + 256 m
    const int wpos = (lk * 64 + li) * 4;                                // lane part of the packed 64-wide weight reads
    // the two adjacent 16-byte chunks {2 hh, 2 hh + 1} of row 32 b + (lane & 31): swz = (row >> 2) & 2
    const int qpos = (lane & 31) * 16 + ((((lane >> 5) << 1) ^ (((lane & 31) >> 2) & 2)) << 2);      // + 512 b, + 4 for the second chunk

    // advantage normalisation constants (baselines/ppo2/common_functions.py:67)
    double adv_mean = 0.0, adv_rstd = 1.0;
    if (g.stats) {
        const double n = g.stats[2];
        adv_mean = g.stats[0] / n;
        double var = g.stats[1] / n - adv_mean * adv_mean;
        var = var > 0.0 ? var : 0.0;
        adv_rstd = 1.0 / (sqrt(var) + 1e-8);
    }
    // wave-uniform: keep them in scalar registers
    auto uniform = [](double v) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(__double_as_longlong(v) & 0xffffffffll));
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)__double_as_longlong(v) >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    adv_mean = uniform(adv_mean);
    adv_rstd = uniform(adv_rstd);
    // head register r of this lane holds output lk + 4 r
    float inv_std[2] = {0.f, 0.f};
    float sum_logstd = 0.f;
    if (!is_value && gaussian) {
        for (int k = 0; k < g.A; k++) sum_logstd += g.params[g.logstd_off + k];
        sum_logstd = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sum_logstd)));
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int k = lk + 4 * r;
            if (k < g.A) inv_std[r] = expf(-g.params[g.logstd_off + k]);
        }
    }

    // ---- gradient accumulators (registers, whole launch) ------------------------------------
    f32x4 dW1[2][HB], dW2[DW2Q ? 1 : HB][DW2Q ? 1 : HB], dWh[HB];
    f32x16 dW2q[DW2Q ? 2 : 1][DW2Q ? 2 : 1];      // block (bi, bj) = input units 32 bi .., output units 32 bj ..
    float db2[HB], dbh = 0.f;                     // DW2Q: db2[bj] = this lane's share of unit 32 bj + (lane & 31)
    float dx16[HB], db1[HB];          // XV: rows 16 (feature x16) and ob_dim (ones) of dW1, per lane over its four samples
#pragma unroll
    for (int a = 0; a < HB; a++) {
        db2[a] = 0.f;
        dx16[a] = 0.f;
        db1[a] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            dWh[a][r] = 0.f;
            dW1[0][a][r] = 0.f;
            dW1[1][a][r] = 0.f;
#pragma unroll
            for (int b = 0; b < (DW2Q ? 1 : HB); b++) dW2[DW2Q ? 0 : a][b][r] = 0.f;
        }
    }
#pragma unroll
    for (int a = 0; a < (DW2Q ? 2 : 1); a++)
#pragma unroll
        for (int b = 0; b < (DW2Q ? 2 : 1); b++)
#pragma unroll
            for (int r = 0; r < 16; r++) dW2q[a][b][r] = 0.f;
    // per-lane loss sums stay short (one term per tile): float here, double from the wave sum on
    float s_pg = 0.f, s_vf = 0.f, s_ent = 0.f, s_kl = 0.f, s_cf = 0.f;


    for (int64_t tile = wave_global; tile < ntiles; tile += wave_stride) {
        const bool live = tile * 16 + li < g.B;

        // ---- x tile: feature row lk + 4 u, sample li (swz of that row = u & 2) --------------------------
#pragma unroll
        for (int u = 0; u < KQ; u++) {
            const int pos = (lk + 4 * u) * 16 + ((u & 2) ? c1 : c0);
            xa[lk + 4 * u < D ? pos : SPARE + lane] = xr[u];
        }

        // ---- this tile's loss inputs, then the next tile's observations and the index after that --------
        float adv = 0.f, oldnlp = 0.f, act[2] = {0.f, 0.f}, ret[2] = {0.f, 0.f}, oldv[2] = {0.f, 0.f};
        int act_i = 0;
        if (!is_value) {
            double a;
            if (g.adv_f64) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(r_adv, (int)((unsigned)row * 8u), 0, 0);
                a = __longlong_as_double((long long)(((unsigned long long)v[1] << 32) | v[0]));
            } else {
                a = (double)buf_f32(r_adv, (unsigned)row * 4u);
            }
            adv = (float)((a - adv_mean) * adv_rstd);
            oldnlp = buf_f32(r_nlp, (unsigned)row * 4u);
            if (gaussian) {      // outputs lk and lk + 4 (ac_dim <= 8)
                const unsigned base = (unsigned)row * (unsigned)(g.A * 4) + (unsigned)(lk * 4);
                act[0] = buf_f32(r_act, base);
                act[1] = buf_f32(r_act, base + 16u);
            } else {
                act_i = buf_i32(r_act, (unsigned)row * 4u);
            }
        } else {                 // value channels lk and lk + 4 (nvalue <= 8); one-output head: every lane its sample's own
            const unsigned base = (unsigned)row * (unsigned)(g.R * 4) + (vhead ? 0u : (unsigned)(lk * 4));
            ret[0] = buf_f32(r_ret, base);
            ret[1] = buf_f32(r_ret, base + 16u);
            oldv[0] = buf_f32(r_val, base);
            oldv[1] = buf_f32(r_val, base + 16u);
        }
        load_obs(row_next);
        const int row_after = tile_row(tile + 2 * wave_stride);
        __builtin_amdgcn_wave_barrier();

        // ---- layer 1: h1 = tanh(W1^T [x; 1]) ----------------------------------------------------
        f32x4 h1[HB], h2[HB], out;
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) h1[mb][r] = 0.f;
        {
            f32x4 w[2];
            float b[2];
            w[0] = LDS4(W1P + lk * 64 + li * 4);
            b[0] = xa[lk * 16 + c0];
#pragma unroll
            for (int q = 0; q < KQ; q++) {
                if (q + 1 < KQ) {       // operands of the next k-step are in flight while this one multiplies
                    w[(q + 1) & 1] = LDS4(W1P + (4 * (q + 1) + lk) * 64 + li * 4);
                    b[(q + 1) & 1] = xa[(4 * (q + 1) + lk) * 16 + (((q + 1) & 2) ? c1 : c0)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < HB; mb++) h1[mb] = MFMA16(w[q & 1][mb], b[q & 1], h1[mb]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int mb = 0; mb < HB; mb++) h1[mb] = tanh_prescaled4(h1[mb]);
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) ta[dpos + (16 * mb + r) * 16] = h1[mb][r];

        // ---- layer 2: h2 = tanh(W2^T h1 + b2); the bias seeds the accumulator ------------------------
        {
#pragma unroll
            for (int m = 0; m < HB; m++) h2[m] = LDS4(B2 + 16 * m + 4 * lk);
#pragma unroll
            for (int q = 0; q < 2; q++) {
                uint4 b[3];
                cut8(h1[2 * q], h1[2 * q + 1], b[0], b[1], b[2]);
#pragma unroll
                for (int m0 = 0; m0 < HB; m0 += PB) {        // PB output blocks at a time: 12 PB operand registers, not 48
                    uint4 a[PB][3];
#pragma unroll
                    for (int m = 0; m < PB; m++)
#pragma unroll
                        for (int p = 0; p < 3; p++) a[m][p] = LDSQ(W2Q, q, m0 + m, p);
                    PIECE_PRODUCTS(h2, m0, a, b, PB)
                }
            }
        }
#pragma unroll
        for (int m = 0; m < HB; m++) h2[m] = tanh_prescaled4(h2[m]);
#pragma unroll
        for (int m = 0; m < HB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) tb[dpos + (16 * m + r) * 16] = h2[m][r];

        // ---- head: register r of lane quarter lk = output lk + 4 r ----------------------------------
        if (vhead) {                     // this lane's 16 units of the head column
            float p = 0.f;
#pragma unroll
            for (int m = 0; m < HB; m++) {
                const f32x4 wv = LDS4(WHV + 16 * m + 4 * lk);
#pragma unroll
                for (int r = 0; r < 4; r++) p = fmaf(wv[r], h2[m][r], p);
            }
            p += __shfl_xor(p, 16, 64);      // the four lane quarters hold disjoint units of sample li
            p += __shfl_xor(p, 32, 64);
#pragma unroll
            for (int r = 0; r < 4; r++) out[r] = 0.f;
            out[0] = p + lds[BHP];
        } else {
            f32x4 w[HB];
            out = LDS4(BHP + 4 * lk);
#pragma unroll
            for (int mb = 0; mb < HB; mb++) w[mb] = LDS4(WHP + (lk * 16 + li) * 4 + 256 * mb);
#pragma unroll
            for (int mb = 0; mb < HB; mb++)
#pragma unroll
                for (int r = 0; r < 4; r++) out = MFMA16(w[mb][r], h2[mb][r], out);
        }

        // ---- loss and d(loss)/d(out) ------------------------------------------------------------------
        f32x4 dout;
#pragma unroll
        for (int r = 0; r < 4; r++) dout[r] = 0.f;
        if (!is_value) {
            float nlp, entropy_row;
            float zq[2] = {0.f, 0.f}, pr[4] = {0.f, 0.f, 0.f, 0.f}, lp[4] = {0.f, 0.f, 0.f, 0.f};
            if (gaussian) {
                float ss = 0.f;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    if (lk + 4 * r < g.A) {
                        zq[r] = (act[r] - out[r]) * inv_std[r];
                        ss += zq[r] * zq[r];
                    }
                }
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                nlp = 0.5f * ss + 0.91893853320467274178f * (float)g.A + sum_logstd;
                entropy_row = sum_logstd + (0.91893853320467274178f + 0.5f) * (float)g.A;
            } else {
                float mx = -3.0e38f;
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (lk + 4 * r < g.A) mx = fmaxf(mx, out[r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float z0 = 0.f, la = 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = lk + 4 * r;
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
                    const int k = lk + 4 * r;
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
                s_pg += fmaxf(pg1, pg2);
                const float d = nlp - oldnlp;
                s_kl += 0.5f * (d * d);
                s_cf += (fabsf(ratio - 1.0f) > g.cliprange) ? 1.f : 0.f;
                s_ent += entropy_row;
            }
            if (gaussian) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    if (lk + 4 * r < g.A) dout[r] = dnlp * (-zq[r] * inv_std[r]);
                }
                // d(loss)/d(logstd_o) of this sample rides in the unused head rows 8 + o (ac_dim <= 8): the
                // bias-gradient row sums below then add it up over samples; the matching weight rows are zero
                const float dent = live ? g.ent_coef * g.inv_count : 0.f;
#pragma unroll
                for (int r = 0; r < 2; r++)
                    if (lk + 4 * r < g.A) dout[r + 2] = dnlp * (1.f - zq[r] * zq[r]) - dent;
            } else {
                const float ec = live ? g.ent_coef * g.inv_count : 0.f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = lk + 4 * r;
                    if (k < g.A) dout[r] = dnlp * (pr[r] - (k == act_i ? 1.f : 0.f)) + ec * pr[r] * (lp[r] + entropy_row);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                // one-output head: all four lane quarters hold the sample's value and need its gradient; quarter 0 books the loss
                const bool mine = vhead ? r == 0 : lk + 4 * r < g.R;
                const float book = (!vhead || lk == 0) ? 1.f : 0.f;
                if (mine && live) {
                    const float v = out[r];
                    const float e1 = v - ret[r];
                    if (g.upstream) {
                        const float dv = v - oldv[r];
                        const float vc = oldv[r] + fminf(fmaxf(dv, -g.cliprange), g.cliprange);
                        const float e2 = vc - ret[r];
                        const float l1 = e1 * e1, l2 = e2 * e2;
                        s_vf += book * (0.5f * fmaxf(l1, l2) * rR);
                        const float gg = (l1 >= l2) ? e1 : ((dv >= -g.cliprange && dv <= g.cliprange) ? e2 : 0.f);
                        dout[r] = g.vf_coef * sc * gg;
                    } else {
                        s_vf += book * ((e1 * e1) * rR);
                        dout[r] = 2.f * e1 * sc;
                    }
                }
            }
        }
        f32x4 dz2[HB];
        const float d0 = dout[0];
        if (vhead) {
            // ---- one-output head on the vector ALUs: input gradient and bias gradient here, the per-lane partial weight
            //      gradients (units 16 m + 4 lk + r, sample li; summed over the sample lanes when the images are written)
            //      below, from the activations the tanh' step reads back anyway ------------------------------------
#pragma unroll
            for (int m = 0; m < HB; m++) dz2[m] = LDS4(WHV + 16 * m + 4 * lk) * d0;
            dbh += lk == 0 ? d0 : 0.f;
        } else {
        // output lk + 4 r is row lk + 4 r of the head tile (swz of that row = r & 2)
#pragma unroll
        for (int r = 0; r < 4; r++) tc[(lk + 4 * r) * 16 + ((r & 2) ? c1 : c0)] = dout[r];
        __builtin_amdgcn_wave_barrier();

        // ---- head: weight gradient (h2^T dout; bias gradient = row sums of dout^T) -------------------
        {
            f32x4 a[HB], wt[HB];
            const f32x4 b = LDS4(WEIGHTS + wave * WAVE + TC + rpos);
#pragma unroll
            for (int mi = 0; mi < HB; mi++) a[mi] = LDS4(WEIGHTS + wave * WAVE + TB + rpos + 256 * mi);
#pragma unroll
            for (int m = 0; m < HB; m++) wt[m] = LDS4(WHTP + wpos + 64 * m);
            dbh += (b[0] + b[1]) + (b[2] + b[3]);
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int mi = 0; mi < HB; mi++) dWh[mi] = MFMA16(a[mi][q], b[q], dWh[mi]);
            // ---- head: input gradient over the registers that hold real outputs ----------------------
#pragma unroll
            for (int m = 0; m < HB; m++)
#pragma unroll
                for (int r = 0; r < 4; r++) dz2[m][r] = 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (r < nr) {
#pragma unroll
                    for (int m = 0; m < HB; m++) dz2[m] = MFMA16(wt[m][r], dout[r], dz2[m]);
                }
        }
        }
        // tanh' from the activations still sitting in the tile, then the tile takes the gradient
#pragma unroll
        for (int m = 0; m < HB; m++) {
            f32x4 t;
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = tb[dpos + (16 * m + r) * 16];
            if (vhead) dWh[m] += t * d0;
            dz2[m] = neg_times_tanh_grad4(dz2[m], t);      // = -dz2
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < HB; m++)
#pragma unroll
            for (int r = 0; r < 4; r++) tb[dpos + (16 * m + r) * 16] = dz2[m][r];
        __builtin_amdgcn_wave_barrier();

        // ---- layer 2: input gradient first (dz2 is still in registers) ------------------------------------
        f32x4 dz1[HB];
        {
#pragma unroll
            for (int mb = 0; mb < HB; mb++)
#pragma unroll
                for (int r = 0; r < 4; r++) dz1[mb][r] = 0.f;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                uint4 b[3];
                cut8(dz2[2 * q], dz2[2 * q + 1], b[0], b[1], b[2]);
#pragma unroll
                for (int m0 = 0; m0 < HB; m0 += PB) {
                    uint4 a[PB][3];
#pragma unroll
                    for (int mb = 0; mb < PB; mb++)
#pragma unroll
                        for (int p = 0; p < 3; p++) a[mb][p] = LDSQ(W2TQ, q, m0 + mb, p);
                    PIECE_PRODUCTS(dz1, m0, a, b, PB)
                }
            }
        }
#pragma unroll
        for (int mb = 0; mb < HB; mb++) {
            f32x4 t;
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = ta[dpos + (16 * mb + r) * 16];
            dz1[mb] = neg_times_tanh_grad4(dz1[mb], t);    // -(W2 (-dz2)) tanh' = dz1
        }
        // ---- layer 2: weight gradient (h1^T dz2 from the tiles), bias gradient from the same operand reads ---
        if constexpr (DW2Q) {
            // lane (r32, hh) takes row 32 b + r32 of a tile and the sample chunks 2 hh, 2 hh + 1 (samples c + 4 j of
            // chunk c: any split of the sixteen samples over the lane halves serves, h1 and dz2 use the same one)
            uint4 ap[2][3];
#pragma unroll
            for (int bi = 0; bi < 2; bi++)
                cut8(LDS4(WEIGHTS + wave * WAVE + TA + qpos + 512 * bi), LDS4(WEIGHTS + wave * WAVE + TA + qpos + 4 + 512 * bi), ap[bi][0], ap[bi][1], ap[bi][2]);
#pragma unroll
            for (int bj = 0; bj < 2; bj++) {
                const f32x4 v0 = LDS4(WEIGHTS + wave * WAVE + TB + qpos + 512 * bj), v1 = LDS4(WEIGHTS + wave * WAVE + TB + qpos + 4 + 512 * bj);
                db2[bj] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
                uint4 bp[3];
                cut8(v0, v1, bp[0], bp[1], bp[2]);
                constexpr int pa_[6] = {2, 0, 1, 1, 0, 0}, pb_[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; t++)
#pragma unroll
                    for (int bi = 0; bi < 2; bi++) dW2q[bi][bj] = MFMA16Q(ap[bi][pa_[t]], bp[pb_[t]], dW2q[bi][bj]);
            }
        } else {
            f32x4 a[HB], b[HB];
#pragma unroll
            for (int m = 0; m < HB; m++) {
                a[m] = LDS4(WEIGHTS + wave * WAVE + TA + rpos + 256 * m);
                b[m] = LDS4(WEIGHTS + wave * WAVE + TB + rpos + 256 * m);
            }
#pragma unroll
            for (int mj = 0; mj < HB; mj++) db2[mj] += (b[mj][0] + b[mj][1]) + (b[mj][2] + b[mj][3]);
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int mi = 0; mi < HB; mi++)
#pragma unroll
                    for (int mj = 0; mj < HB; mj++) dW2[mi][mj] = MFMA16(a[mi][q], b[mj][q], dW2[mi][mj]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int mb = 0; mb < HB; mb++)
#pragma unroll
            for (int r = 0; r < 4; r++) ta[dpos + (16 * mb + r) * 16] = dz1[mb][r];
        __builtin_amdgcn_wave_barrier();

        // ---- layer 1: weight (+ bias, through the ones feature) gradient --------------------------------
        {
            f32x4 b[HB];
#pragma unroll
            for (int mj = 0; mj < HB; mj++) b[mj] = LDS4(WEIGHTS + wave * WAVE + TA + rpos + 256 * mj);
            const f32x4 a0 = LDS4(WEIGHTS + wave * WAVE + XA + rpos);
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int mj = 0; mj < HB; mj++) dW1[0][mj] = MFMA16(a0[q], b[mj][q], dW1[0][mj]);
            if (WIDE) {
                const f32x4 a1 = LDS4(WEIGHTS + wave * WAVE + XA + rpos + 256);
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int mj = 0; mj < HB; mj++) dW1[1][mj] = MFMA16(a1[q], b[mj][q], dW1[1][mj]);
            }
            if (XV) {       // the ones row: sums of the same operand chunks (unit 16 mj + li, samples lk + 4 q)
#pragma unroll
                for (int mj = 0; mj < HB; mj++) db1[mj] += (b[mj][0] + b[mj][1]) + (b[mj][2] + b[mj][3]);
            }
            if (XV == 2) {  // feature row 16 (swizzle 0): its samples lk + 4 q are one 16-byte chunk
                const f32x4 x16 = LDS4(WEIGHTS + wave * WAVE + XA + 256 + (lk << 2));
#pragma unroll
                for (int mj = 0; mj < HB; mj++)
#pragma unroll
                    for (int q = 0; q < 4; q++) dx16[mj] = fmaf(x16[q], b[mj][q], dx16[mj]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        row = row_next;
        row_next = row_after;
    }

    // ---- loss partial row of this workgroup: wave sums, then eight values per output in fixed order ----
    {
        double vals[kLossVals];
        vals[0] = (double)s_pg; vals[1] = (double)s_vf; vals[2] = (double)s_ent; vals[3] = (double)s_kl; vals[4] = (double)s_cf;
#pragma unroll
        for (int k = 0; k < kLossVals; k++) {
            const double v = wave_sum(vals[k]);
            if (lane == 0) red[wave * kLossVals + k] = v;
        }
    }

    // ---- cross-wave reduction of the gradient images (two rounds of four), one slab per workgroup ------
    // sums over the four lane quarters (a lane's partial covers its samples lk + 4 q only)
    auto quarters = [](float v) {
        v += __shfl_xor(v, 16, 64);
        return v + __shfl_xor(v, 32, 64);
    };
    if (vhead) {
        // the vector-ALU head kept one partial per sample lane: sum over the 16 lanes of each quarter
#pragma unroll
        for (int mi = 0; mi < HB; mi++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = dWh[mi][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
                dWh[mi][r] = v;
            }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) dbh += __shfl_xor(dbh, o, 64);
    }
    auto put = [&](float *img, bool add) {
        auto st = [&](int off, float v) { img[off] = add ? img[off] + v : v; };
#pragma unroll
        for (int f = 0; f < (WIDE ? 2 : 1); f++)
#pragma unroll
            for (int mj = 0; mj < HB; mj++)
#pragma unroll
                for (int r = 0; r < 4; r++) st(S_W1 + (16 * f + 4 * lk + r) * H + 16 * mj + li, dW1[f][mj][r]);
        if (XV) {
#pragma unroll
            for (int mj = 0; mj < HB; mj++) {
                const float vb = quarters(db1[mj]);
                if (lk == 0) st(S_W1 + (15 + XV) * H + 16 * mj + li, vb);          // the ones row = row ob_dim
                if (XV == 2) {
                    const float vx = quarters(dx16[mj]);
                    if (lk == 0) st(S_W1 + 16 * H + 16 * mj + li, vx);
                }
            }
        }
        // dW2 and db2 were accumulated from -dz2 (see neg_times_tanh_grad4)
        if constexpr (DW2Q) {
#pragma unroll
            for (int bi = 0; bi < 2; bi++)
#pragma unroll
                for (int bj = 0; bj < 2; bj++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        st(S_W2 + (32 * bi + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * H + 32 * bj + (lane & 31), -dW2q[bi][bj][r]);
        } else {
#pragma unroll
        for (int mi = 0; mi < HB; mi++)
#pragma unroll
            for (int mj = 0; mj < HB; mj++)
#pragma unroll
                for (int r = 0; r < 4; r++) st(S_W2 + (16 * mi + 4 * lk + r) * H + 16 * mj + li, -dW2[mi][mj][r]);
        }
        if (vhead) {
#pragma unroll
            for (int mi = 0; mi < HB; mi++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (li == 0) st(S_WH + (16 * mi + 4 * lk + r) * 16, dWh[mi][r]);
        } else {
#pragma unroll
            for (int mi = 0; mi < HB; mi++)
#pragma unroll
                for (int r = 0; r < 4; r++) st(S_WH + (16 * mi + 4 * lk + r) * 16 + li, dWh[mi][r]);
        }
        if constexpr (DW2Q) {
#pragma unroll
            for (int bj = 0; bj < 2; bj++) {
                const float v = db2[bj] + __shfl_xor(db2[bj], 32, 64);          // the two sample halves
                if (lane < 32) st(S_B2 + 32 * bj + lane, -v);
            }
        } else {
#pragma unroll
        for (int mj = 0; mj < HB; mj++) {
            const float v = quarters(db2[mj]);
            if (lk == 0) st(S_B2 + 16 * mj + li, -v);
        }
        }
        const float v = quarters(dbh);
        if (vhead) {
            if (lane == 0) st(S_BH, v);          // quarter 0 held the sum over the samples, the others zero
        } else if (lk == 0) {
            st(S_BH + li, v);
        }
    };
    __syncthreads();                                   // weights and tiles are dead: LDS holds images now
    if (wave >= 4) put(lds + (wave - 4) * SLAB, false);
    __syncthreads();
    if (wave < 4) put(lds + wave * SLAB, true);
    __syncthreads();
    float *slab = g.slabs + (int64_t)blockIdx.x * SLAB;
    for (int e = tid; e < SLAB; e += kThreads) {
        const float *p = lds + e;
        slab[e] = (p[0] + p[SLAB]) + (p[2 * SLAB] + p[3 * SLAB]);
    }
    if (tid < kLossVals) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) v += red[w * kLossVals + tid];
        g.loss_partial[(int64_t)blockIdx.x * kLossStride + tid] = v;
    }
#undef LDS4
#undef LDSQ
}

bool mlp_fused16_supported(const ppo2hip_policy_desc &d) {
    if (d.kind != PPO2HIP_POLICY_MLP) return false;
    if (d.hidden1 != 64 || d.hidden2 != 64) return false;
    if (d.ob_dim < 1 || d.ob_dim > 31) return false;
    if (d.nvalue > 8) return false;
    return d.ac_dim <= (d.dist == PPO2HIP_DIST_GAUSSIAN ? 8 : 16);   // Gaussian: rows 8.. carry the logstd gradient
}

bool mlp_fused16_fits(const ppo2hip_policy_desc &d, const ppo2hip_rollout &ro, int64_t B) {
    return rollout_fits_buffer_loads(d, ro, B);
}

int mlp_fused16_grid() { return 256; }

// How many of the grid's workgroups run the policy trunk.  A tile costs the value trunk less (a one-output head runs
// on the vector ALUs, no distribution arithmetic), so an even split leaves its workgroups idle at the end; the split
// that minimises max(tiles per SIMD x relative cost) over the two trunks is taken instead.
static int fused16_policy_workgroups(const ppo2hip_policy_desc &d, int64_t B, int grid) {
    if (const char *e = getenv("PPO2HIP_F16_POLICY_WGS")) {
        const int v = atoi(e);
        if (v >= 1 && v < grid) return v;
    }
    const int64_t ntiles = (B + 15) / 16;
    const double cost_value = d.nvalue == 1 ? 0.84 : 0.97;      // relative to a policy tile (measured on gfx950)
    int best = grid / 2;
    double best_t = 1e300;
    for (int wp = grid / 4; wp <= 3 * grid / 4; wp++) {
        const double tp = (double)ceil_div(ntiles, 4 * (int64_t)wp), tv = (double)ceil_div(ntiles, 4 * (int64_t)(grid - wp)) * cost_value;
        const double t = tp > tv ? tp : tv;
        if (t < best_t - 1e-9 || (t < best_t + 1e-9 && abs(wp - grid / 2) < abs(best - grid / 2))) { best_t = t; best = wp; }
    }
    return best;
}   // one 512-thread workgroup per CU (LDS-limited), 128 per trunk

size_t mlp_fused16_workspace_bytes() {
    return align_up((size_t)mlp_fused16_grid() * f16::SLAB * sizeof(float), 256) +
           align_up((size_t)mlp_fused16_grid() * f16::kLossStride * sizeof(double), 256) + fused_adam_extra_bytes();
}

int mlp_fused16_grad(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const float *params,
                     const FusedOffsets &fo, const ppo2hip_rollout &ro, const int32_t *idx, int64_t B,
                     const double *stats, float cliprange, double inv_count, float *grads, float *losses, void *ws,
                     size_t ws_bytes, hipStream_t s, AdamFuse *adam) {
    using namespace f16;
    if (ws_bytes < mlp_fused16_workspace_bytes()) {
        set_error("mlp_fused16_grad: workspace too small");
        return PPO2HIP_ENOSPACE;
    }
    const int grid = mlp_fused16_grid();
    float *slabs = (float *)ws;
    double *lpart = (double *)((char *)ws + align_up((size_t)grid * SLAB * sizeof(float), 256));
    char *extra = (char *)lpart + align_up((size_t)grid * kLossStride * sizeof(double), 256);
    if (adam) adam->words = (unsigned long long *)extra;
    Fused16Args g;
    g.obs = (const float *)ro.obs; g.actions = ro.actions; g.returns = ro.returns; g.values = ro.values;
    g.neglogpacs = ro.neglogpacs; g.advs = ro.advs; g.adv_f64 = ro.adv_f64; g.T = ro.T; g.E = ro.E; g.n = ro.n;
    { const int64_t ns = rollout_samples(ro); g.nsamples = (unsigned)(ns < 0x7fffffffll ? ns : 0x7fffffffll); }
    g.fault = fault_word();
    if (!g.fault) return PPO2HIP_EINVAL;
    g.idx = idx; g.B = B; g.stats = stats; g.params = params;
    g.a_w1 = fo.a_w1; g.a_w2 = fo.a_w2; g.a_wh = fo.a_wh; g.v_w1 = fo.v_w1; g.v_w2 = fo.v_w2; g.v_wh = fo.v_wh;
    g.logstd_off = fo.logstd; g.D = d.ob_dim; g.A = d.ac_dim; g.R = d.nvalue; g.dist = d.dist;
    g.upstream = lc.upstream; g.cliprange = cliprange; g.ent_coef = lc.ent_coef;
    g.vf_coef = lc.upstream ? lc.vf_coef : 1.0f; g.inv_count = (float)inv_count;
    g.slabs = slabs; g.loss_partial = lpart;
    g.nwg_policy = fused16_policy_workgroups(d, B, grid);
    {   // env = smp / T as a multiplication (smp < 2^31): s = ceil(log2 T), magic = floor(2^32 (2^s - T) / T) + 1
        const unsigned T = ro.T > 0 ? (unsigned)ro.T : 1u;
        int sh = 0;
        while ((1ull << sh) < T) sh++;
        g.div_magic = (unsigned)((((1ull << sh) - T) << 32) / T + 1);
        g.div_sh1 = sh < 1 ? sh : 1;
        g.div_sh2 = sh > 1 ? sh - 1 : 0;
    }
    if (d.ob_dim <= 7) hipLaunchKernelGGL((mlp_fused16_kernel<2, false, 0>), dim3(grid), dim3(kThreads), 0, s, g);
    else if (d.ob_dim <= 15) hipLaunchKernelGGL((mlp_fused16_kernel<4, false, 0>), dim3(grid), dim3(kThreads), 0, s, g);
    else if (d.ob_dim == 16) hipLaunchKernelGGL((mlp_fused16_kernel<5, false, 1>), dim3(grid), dim3(kThreads), 0, s, g);
    else if (d.ob_dim == 17) hipLaunchKernelGGL((mlp_fused16_kernel<5, false, 2>), dim3(grid), dim3(kThreads), 0, s, g);
    else if (d.ob_dim <= 19) hipLaunchKernelGGL((mlp_fused16_kernel<5, true, 0>), dim3(grid), dim3(kThreads), 0, s, g);
    else hipLaunchKernelGGL((mlp_fused16_kernel<8, true, 0>), dim3(grid), dim3(kThreads), 0, s, g);
    int rc = check_launch("mlp_fused16_kernel");
    if (rc) return rc;
    FusedImage im;
    im.slab_elems = SLAB; im.s_w1 = S_W1; im.s_w2 = S_W2; im.s_wh = S_WH; im.s_bh = S_BH; im.head_cols = 16; im.s_logstd = S_BH + 8;
    return fused_finalize(im, slabs, g.nwg_policy, grid - g.nwg_policy, lpart, grid, d, fo, inv_count, grads, losses, s, adam);
}

}  // namespace ppo2hip
