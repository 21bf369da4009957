// Dense-layer products on the fp32 MFMA GEMM core: forward (bias + activation fused),
// input gradient (activation derivative fused) and weight gradient (bias gradient folded in
// as an extra row of ones; split over the sample axis into deterministic slabs).
// Mirrors what tf.matmul + tf.gradients build for `fc`, baselines/a2c/utils.py:63-68.
#include "gemm_core.hpp"
#include "kernels.hpp"

namespace ppo2hip {

int fc_forward(const float *X, int ldx, const float *W, const float *b, float *Y, int ldy, int M, int N,
               int K, int act, bool accumulate, hipStream_t s) {
    RowMajor A{X, ldx};
    RowMajor B{W, N};
    EpiBiasAct ep{Y, ldy, b, act, accumulate ? 1 : 0};
    return launch_gemm(A, B, ep, M, N, K, 1, s, "fc_forward");
}

int fc_dgrad(const float *dY, int lddy, const float *W, int Kin, int N, const float *H, int ldh, int act,
             float *dX, int lddx, int M, bool accumulate, hipStream_t s) {
    RowMajor A{dY, lddy};
    Transposed B{W, N};   // B(k = n, j = kin) = W[kin][n]
    EpiActGrad ep{dX, lddx, H, ldh, act, accumulate ? 1 : 0};
    return launch_gemm(A, B, ep, M, Kin, N, 1, s, "fc_dgrad");
}

static int pick_slabs(int64_t S, int64_t tiles) {
    // enough workgroups to fill 256 CUs a few times over, at least 256 sample rows per slab
    int64_t want = ceil_div(1024, tiles > 0 ? tiles : 1);
    int64_t cap = ceil_div(S, 256);
    int64_t n = want < cap ? want : cap;
    if (n < 1) n = 1;
    if (n > 128) n = 128;
    return (int)n;
}

int wgrad_max_slabs(int64_t S, int Kin, int N) {
    const int64_t tiles = ceil_div(Kin + 1, (Kin + 1) <= 64 ? 64 : 128) * ceil_div(N, GEMM_BN);
    return pick_slabs(S, tiles);
}

int fc_wgrad_slabs(const float *X, int ldx, const float *dY, int lddy, int64_t S, int Kin, int N,
                   float *slabs, int max_slabs, int *nslabs, hipStream_t s) {
    int n = wgrad_max_slabs(S, Kin, N);
    if (n > max_slabs) n = max_slabs;
    *nslabs = n;
    TransposedOnes A{X, ldx, Kin};   // A(i = kin, k = sample) = X[sample][kin]; row Kin = 1
    RowMajor B{dY, lddy};
    EpiSlab ep{slabs, N, (int64_t)(Kin + 1) * N};
    return launch_gemm(A, B, ep, Kin + 1, N, S, n, s, "fc_wgrad");
}

// 64 consecutive elements x 4 slab groups per workgroup: coalesced along the element axis,
// the slab axis split over the four waves and finished through LDS in a fixed order.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float *__restrict__ slabs, int nslabs, int64_t size,
                                                           float *__restrict__ dst, int accumulate) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < size)
        for (int z = grp; z < nslabs; z += 4) s += slabs[(int64_t)z * size + i];
    part[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && i < size) {
        const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        dst[i] = accumulate ? dst[i] + t : t;
    }
}

int reduce_slabs(const float *slabs, int nslabs, int64_t size, float *dst, bool accumulate, hipStream_t s) {
    if (size <= 0) return 0;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div(size, 64)), dim3(256), 0, s, slabs, nslabs,
                       size, dst, accumulate ? 1 : 0);
    return check_launch("reduce_slabs");
}

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" int ppo2hip_gemm_f32(const float *A, const float *B, float *C, int M, int N, int K, void *stream) {
    PPO2HIP_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: bad shape %d %d %d", M, N, K);
    PPO2HIP_REQUIRE(M == 0 || N == 0 || (A && B && C), "gemm: null pointer");
    return fc_forward(A, K, B, nullptr, C, N, M, N, K, ACT_NONE, false, (hipStream_t)stream);
}
