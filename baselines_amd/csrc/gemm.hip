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
    return launch_gemm<true>(A, B, ep, M, N, K, 1, s, "fc_forward");
}

int fc_dgrad(const float *dY, int lddy, const float *W, int Kin, int N, const float *H, int ldh, int act,
             float *dX, int lddx, int M, bool accumulate, hipStream_t s, float *colsum_tiles, int *ntiles) {
    RowMajor A{dY, lddy};
    Transposed B{W, N};   // B(k = n, j = kin) = W[kin][n]
    EpiActGrad ep{dX, lddx, H, ldh, act, accumulate ? 1 : 0, colsum_tiles};
    if (ntiles) *ntiles = (int)gemm_row_tiles(M, Kin, 1);
    return launch_gemm<true>(A, B, ep, M, Kin, N, 1, s, "fc_dgrad");
}

int fc_forward_parts(const float *X, int ldx, const float *W, int M, int N, int K, int nparts, float *parts, hipStream_t s) {
    RowMajor A{X, ldx};
    RowMajor B{W, N};
    EpiSlab ep{parts, N, (int64_t)M * N};
    return launch_gemm(A, B, ep, M, N, K, nparts, s, "fc_forward_parts");
}

int fc_dgrad_parts(const float *dY, int lddy, const float *W, int Kin, int N, int M, int nparts, float *parts, hipStream_t s) {
    RowMajor A{dY, lddy};
    Transposed B{W, N};
    EpiSlab ep{parts, Kin, (int64_t)M * Kin};
    return launch_gemm(A, B, ep, M, Kin, N, nparts, s, "fc_dgrad_parts");
}

bool wgrad_bias_separately(int Kin);

static int pick_slabs(int64_t S, int64_t tiles) {
    // A product with few output tiles has nothing but resident workgroups to hide its load latency with
    // (one k-slab of loads per 16 MFMA steps): aim at eight workgroups per CU, at least 256 sample rows per slab.
    int64_t want = ceil_div(2048, tiles > 0 ? tiles : 1);
    int64_t cap = ceil_div(S, 256);
    int64_t n = want < cap ? want : cap;
    if (n < 1) n = 1;
    if (n > 1024) n = 1024;
    return (int)n;
}

int wgrad_max_slabs(int64_t S, int Kin, int N) {
    const int rows = wgrad_bias_separately(Kin) ? Kin : Kin + 1;
    const int64_t tiles = ceil_div(rows, rows <= 64 ? 64 : 128) * ceil_div(N, (rows > 64 && N <= 32) ? 32 : GEMM_BN);
    return pick_slabs(S, tiles);
}

// Column sums of dY over each slab's rows, written to row Kin of that slab: the bias gradient where a row of
// ones inside the product would stand in the way of the operand's 16-byte loads (Kin a multiple of 16).
__global__ __launch_bounds__(256) void slab_colsum_kernel(const float *__restrict__ dY, int64_t lddy, int64_t S, int N,
                                                          int64_t klen, float *__restrict__ slabs, int64_t zstride,
                                                          int64_t row_off) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int64_t z = blockIdx.y;
    const int64_t r0 = z * klen, r1 = r0 + klen < S ? r0 + klen : S;
    float acc = 0.f;
    if (n < N) {
        constexpr int U = 8;
        for (int64_t r = r0 + grp; r < r1; r += 4 * U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; u++) x[u] = r + 4 * u < r1 ? dY[(r + 4 * u) * lddy + n] : 0.f;
#pragma unroll
            for (int u = 0; u < U; u++) acc += x[u];
        }
    }
    part[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && n < N) slabs[z * zstride + row_off + n] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

bool wgrad_bias_separately(int Kin) { return Kin % 16 == 0; }

int slab_bias_rows(const float *dY, int lddy, int64_t S, int Kin, int N, float *slabs, int nslabs, hipStream_t s) {
    int64_t klen = ceil_div(S > 0 ? S : 1, nslabs);
    klen = ceil_div(klen, GEMM_BK) * GEMM_BK;       // the slab boundaries launch_gemm uses
    hipLaunchKernelGGL(slab_colsum_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)nslabs), dim3(256), 0, s, dY,
                       (int64_t)lddy, S, N, klen, slabs, (int64_t)(Kin + 1) * N, (int64_t)Kin * N);
    return check_launch("slab_colsum_kernel");
}

// dst[n] = sum over the S rows of rows[S, N], in a fixed order: up to 256 chunks of rows summed side by side into
// scratch[chunks][N], then added
int colsum_rows(const float *rows, int64_t S, int N, float *scratch, float *dst, hipStream_t s) {
    if (S <= 0 || N <= 0) return 0;
    int nchunks = (int)(ceil_div(S, (int64_t)64) < 256 ? ceil_div(S, (int64_t)64) : 256);
    const int64_t klen = ceil_div(S, (int64_t)nchunks);
    nchunks = (int)ceil_div(S, klen);
    hipLaunchKernelGGL(slab_colsum_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)nchunks), dim3(256), 0, s, rows, (int64_t)N, S, N,
                       klen, scratch, (int64_t)N, (int64_t)0);
    int rc = check_launch("slab_colsum_kernel");
    if (rc) return rc;
    return reduce_slabs(scratch, nchunks, N, dst, false, s);
}

int fc_wgrad_slabs(const float *X, int ldx, const float *dY, int lddy, int64_t S, int Kin, int N,
                   float *slabs, int max_slabs, int *nslabs, hipStream_t s, bool bias_rows) {
    int n = wgrad_max_slabs(S, Kin, N);
    if (n > max_slabs) n = max_slabs;
    *nslabs = n;
    RowMajor B{dY, lddy};
    EpiSlab ep{slabs, N, (int64_t)(Kin + 1) * N};
    if (wgrad_bias_separately(Kin)) {
        TransposedOnes A{X, ldx, -1};    // no ones-row: rows 0 .. Kin-1 only
        int rc = launch_gemm<true>(A, B, ep, Kin, N, S, n, s, "fc_wgrad");
        if (rc || !bias_rows) return rc;
        return slab_bias_rows(dY, lddy, S, Kin, N, slabs, n, s);
    }
    TransposedOnes A{X, ldx, Kin};   // A(i = kin, k = sample) = X[sample][kin]; row Kin = 1
    return launch_gemm(A, B, ep, Kin + 1, N, S, n, s, "fc_wgrad");
}

// 64 consecutive elements x 16 slab groups per workgroup: coalesced along the element axis, the slab axis
// split over sixteen waves (a handful of independent loads each) and finished through LDS in a fixed order.
constexpr int kReduceGroups = 16;
__global__ __launch_bounds__(64 * kReduceGroups) void reduce_slabs_kernel(const float *__restrict__ slabs, int nslabs,
                                                                          int64_t size, float *__restrict__ dst,
                                                                          int accumulate, int64_t stride) {
    __shared__ float part[kReduceGroups][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < size) {
        constexpr int U = 8;
        for (int z0 = grp; z0 < nslabs; z0 += kReduceGroups * U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int z = z0 + u * kReduceGroups;
                x[u] = z < nslabs ? slabs[(int64_t)z * stride + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; u++) s += x[u];
        }
    }
    part[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && i < size) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < kReduceGroups; q++) t += part[q][lane];
        dst[i] = accumulate ? dst[i] + t : t;
    }
}

int reduce_slabs(const float *slabs, int nslabs, int64_t size, float *dst, bool accumulate, hipStream_t s, int64_t stride) {
    if (size <= 0) return 0;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div(size, 64)), dim3(64 * kReduceGroups), 0, s, slabs, nslabs,
                       size, dst, accumulate ? 1 : 0, stride > 0 ? stride : size);
    return check_launch("reduce_slabs");
}

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" int ppo2hip_gemm_f32(const float *A, const float *B, float *C, int M, int N, int K, void *stream) {
    PPO2HIP_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: bad shape %d %d %d", M, N, K);
    PPO2HIP_REQUIRE(M == 0 || N == 0 || (A && B && C), "gemm: null pointer");
    return fc_forward(A, K, B, nullptr, C, N, M, N, K, ACT_NONE, false, (hipStream_t)stream);
}
