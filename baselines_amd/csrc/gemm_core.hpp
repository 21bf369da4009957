// fp32 MFMA GEMM core for gfx950: C = epilogue(A x B) with functor loaders, so that dense
// layers (plain / transposed operands), implicit-GEMM convolutions and weight-gradient
// products (split over the sample axis into slabs) all share one LDS-tiled main loop.
//
// Tile: BM x 64 x 16, 256 threads = 4 wave64.  v_mfma_f32_32x32x2_f32 is an exact fp32
// fma chain, so results are reproducible and within fp32 rounding of a float64 reference.
//   operand lanes : A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
//   result lanes  : C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
// LDS holds both tiles k-major ([k][m], [k][n]) so a wave's operand read is two runs of 32
// consecutive floats (conflict-free).  The next k-slab is prefetched into registers while
// the current one feeds the matrix cores.
#pragma once
#include "common.hpp"

namespace ppo2hip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BN = 64;
constexpr int GEMM_BK = 16;
constexpr int GEMM_THREADS = 256;

// Loader concept:   float operator()(int64_t row, int64_t col) const  — A: (m, k), B: (k, n);
//                   out-of-range coordinates are never requested.
//                   static constexpr bool kInnerFast — true when the SECOND coordinate is the
//                   contiguous one in memory (drives the thread->element mapping for coalescing).
// Epilogue concept: void operator()(int64_t m, int64_t n, float acc, int z) const

// WM = number of wave rows (4: BM = 128, each wave 32 x 64; 2: BM = 64, each wave 32 x 32)
template <class AL, class BL, class EP, int WM>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(AL a, BL b, EP ep, int64_t M, int N, int64_t K,
                                                            int64_t klen) {
    constexpr int BM = 32 * WM;
    constexpr int WN = 4 / WM;              // wave columns
    constexpr int NB = GEMM_BN / 32 / WN;   // 32-col blocks per wave (2 or 1)
    constexpr int LDA = BM + 1, LDB = GEMM_BN + 1;
    constexpr int A_PER = BM * GEMM_BK / GEMM_THREADS;
    constexpr int B_PER = GEMM_BN * GEMM_BK / GEMM_THREADS;
    __shared__ float As[GEMM_BK * LDA];
    __shared__ float Bs[GEMM_BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * GEMM_BN;
    const int z = blockIdx.z;
    const int64_t kbeg = (int64_t)z * klen;
    const int64_t kend = kbeg + klen < K ? kbeg + klen : K;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nb][r] = 0.f;

    float areg[A_PER], breg[B_PER];

    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int e = 0; e < A_PER; e++) {
            const int lin = tid + e * GEMM_THREADS;
            int ml, kl;
            if (AL::kInnerFast) { kl = lin % GEMM_BK; ml = lin / GEMM_BK; }
            else { ml = lin % BM; kl = lin / BM; }
            const int64_t m = m0 + ml, k = k0 + kl;
            areg[e] = (m < M && k < kend) ? a(m, k) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < B_PER; e++) {
            const int lin = tid + e * GEMM_THREADS;
            int kl, nl;
            if (BL::kInnerFast) { nl = lin % GEMM_BN; kl = lin / GEMM_BN; }
            else { kl = lin % GEMM_BK; nl = lin / GEMM_BK; }
            const int64_t k = k0 + kl;
            const int n = n0 + nl;
            breg[e] = (k < kend && n < N) ? b(k, n) : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int e = 0; e < A_PER; e++) {
            const int lin = tid + e * GEMM_THREADS;
            int ml, kl;
            if (AL::kInnerFast) { kl = lin % GEMM_BK; ml = lin / GEMM_BK; }
            else { ml = lin % BM; kl = lin / BM; }
            As[kl * LDA + ml] = areg[e];
        }
#pragma unroll
        for (int e = 0; e < B_PER; e++) {
            const int lin = tid + e * GEMM_THREADS;
            int kl, nl;
            if (BL::kInnerFast) { nl = lin % GEMM_BN; kl = lin / GEMM_BN; }
            else { kl = lin % GEMM_BK; nl = lin / GEMM_BK; }
            Bs[kl * LDB + nl] = breg[e];
        }
    };

    if (kbeg < kend) {
        fetch(kbeg);
        for (int64_t k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
            __syncthreads();           // previous slab fully consumed
            stash();
            __syncthreads();
            if (k0 + GEMM_BK < kend) fetch(k0 + GEMM_BK);
#pragma unroll
            for (int kk = 0; kk < GEMM_BK / 2; kk++) {
                const float av = As[(2 * kk + lh) * LDA + wm * 32 + li];
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    const float bv = Bs[(2 * kk + lh) * LDB + (wn * NB + nb) * 32 + li];
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        const int n = n0 + (wn * NB + nb) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < M && n < N) ep(m, n, acc[nb][r], z);
        }
    }
}

template <class AL, class BL, class EP>
inline int launch_gemm(const AL &a, const BL &b, const EP &ep, int64_t M, int N, int64_t K, int nsplit,
                       hipStream_t s, const char *what) {
    if (M <= 0 || N <= 0) return 0;
    if (nsplit < 1) nsplit = 1;
    int64_t klen = ceil_div(K > 0 ? K : 1, nsplit);
    klen = ceil_div(klen, GEMM_BK) * GEMM_BK;
    const bool small_m = M <= 64;
    const int BM = small_m ? 64 : 128;
    const int64_t gx = ceil_div(M, BM);
    if (gx > 0x7fffffffll || nsplit > 65535) {
        set_error("%s: grid too large", what);
        return PPO2HIP_EINVAL;
    }
    dim3 grid((unsigned)gx, (unsigned)ceil_div(N, GEMM_BN), (unsigned)nsplit);
    if (small_m)
        hipLaunchKernelGGL((gemm_kernel<AL, BL, EP, 2>), grid, dim3(GEMM_THREADS), 0, s, a, b, ep, M, N, K, klen);
    else
        hipLaunchKernelGGL((gemm_kernel<AL, BL, EP, 4>), grid, dim3(GEMM_THREADS), 0, s, a, b, ep, M, N, K, klen);
    return check_launch(what);
}

// ---- loaders ---------------------------------------------------------------------------
struct RowMajor {   // element (r, c) at p[r*ld + c]
    static constexpr bool kInnerFast = true;
    const float *p;
    int64_t ld;
    __device__ float operator()(int64_t r, int64_t c) const { return p[r * ld + c]; }
};
struct Transposed {  // element (r, c) at p[c*ld + r]
    static constexpr bool kInnerFast = false;
    const float *p;
    int64_t ld;
    __device__ float operator()(int64_t r, int64_t c) const { return p[c * ld + r]; }
};
struct TransposedOnes {  // as Transposed, plus a row `ones_row` of 1.0 (folds the bias gradient in)
    static constexpr bool kInnerFast = false;
    const float *p;
    int64_t ld;
    int64_t ones_row;
    __device__ float operator()(int64_t r, int64_t c) const { return r == ones_row ? 1.f : p[c * ld + r]; }
};

// ---- epilogues -------------------------------------------------------------------------
__device__ inline float apply_act(float x, int act) {
    if (act == 1) return tanhf(x);
    if (act == 2) return x > 0.f ? x : 0.f;
    return x;
}
// derivative expressed through the saved OUTPUT h of the activation
__device__ inline float act_grad(float h, int act) {
    if (act == 1) return 1.f - h * h;
    if (act == 2) return h > 0.f ? 1.f : 0.f;
    return 1.f;
}
struct EpiBiasAct {
    float *C;
    int64_t ldc;
    const float *bias;
    int act;
    int accumulate;
    __device__ void operator()(int64_t m, int64_t n, float v, int) const {
        if (bias) v += bias[n];
        float *d = C + m * ldc + n;
        if (accumulate) v += *d;
        *d = apply_act(v, act);
    }
};
struct EpiActGrad {
    float *C;
    int64_t ldc;
    const float *H;
    int64_t ldh;
    int act;
    int accumulate;
    __device__ void operator()(int64_t m, int64_t n, float v, int) const {
        if (H) v *= act_grad(H[m * ldh + n], act);
        float *d = C + m * ldc + n;
        if (accumulate) v += *d;
        *d = v;
    }
};
struct EpiSlab {
    float *C;
    int64_t ldc;
    int64_t zstride;
    __device__ void operator()(int64_t m, int64_t n, float v, int z) const { C[z * zstride + m * ldc + n] = v; }
};

}  // namespace ppo2hip
