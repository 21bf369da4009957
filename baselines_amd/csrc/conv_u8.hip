// First convolution of nature_cnn (baselines/a2c/utils.py:46 conv under baselines/common/models.py:22-31) on uint8
// observations, on the bf16 matrix pipe without giving up float32 results.
//
// A uint8 tap is an integer of at most eight significant bits, so it is a bf16 number exactly.  A float32 weight is the
// exact sum of three bf16 numbers (its 24 significand bits cut into three runs of eight: w = hi + mid + lo, each piece
// what truncation to bf16 leaves of the remainder).  Every product tap x piece has at most sixteen significant bits and is
// exact in the float32 the matrix instruction accumulates in, so
//     sum_k x_k w_k = sum_k x_k hi_k + sum_k x_k mid_k + sum_k x_k lo_k
// differs from the float32 fma chain of the plain product only in the rounding of the running sum — the error class of
// any float32 dot product in another order — while v_mfma_f32_32x32x16_bf16 retires 16 times the multiply-adds per
// cycle of the float32 form: three of them per 32 x 32 x 16 block cost 3/16 of the float32 block.
//
// The taps need no LDS: lane (r = lane & 31, h = lane >> 5) of the A operand holds row r, k = 8 h + j — eight
// consecutive bytes of the pixel's patch row.  One 16-byte load per lane and 32-byte run of a kernel row feeds two
// k-steps (the k order inside a run is permuted: lane half h takes bytes 16 h .. 16 h + 15; the weight image is laid
// out to match).  A wave owns 32 consecutive output pixels and all 32 output channels; the three-piece weight image of
// the whole layer (48 KB for K = 256) sits in LDS in fragment order, one conflict-free ds_read_b128 per product.
#include <cstdlib>

#include "kernels.hpp"

namespace ppo2hip {

namespace u8c {
constexpr int kThreads = 256;
constexpr int kCout = 32;
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// the two bf16 of float bit patterns a (low half) and b (high half): their upper 16 bits
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

struct Geom {
    int IH, IW, Cin, S, OH, OW;
};
}  // namespace u8c

// KH kernel rows of CPR 32-byte runs each (KW Cin = 32 CPR bytes)
template <int KH, int CPR>
__global__ __launch_bounds__(u8c::kThreads, 3) void conv_u8_forward_kernel(const uint8_t *__restrict__ X,
                                                                        const float *__restrict__ W,
                                                                        const float *__restrict__ bias,
                                                                        float *__restrict__ Y, u8c::Geom g, int64_t M,
                                                                        int64_t ntiles, float scale) {
    using namespace u8c;
    constexpr int DS = KH * CPR;                 // 32-tap runs = pairs of k-steps
    constexpr int K = DS * 32;
    __shared__ uint4 wimg[DS * 2 * 3 * 64];      // [run d][k-step u][piece p][lane] fragments of eight bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // first tile's taps go out before the weights are cut up
    uint4 raw[DS];
    auto load_tile = [&](int64_t tile) {
        int64_t m = tile * 128 + wave * 32 + r;
        m = m < M ? m : M - 1;
        const int ohw = g.OH * g.OW;
        const int64_t b = m / ohw;
        const int rem = (int)(m - b * ohw);
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        const uint8_t *p = X + ((b * g.IH + oh * g.S) * g.IW + ow * g.S) * g.Cin + 16 * h;
#pragma unroll
        for (int kh = 0; kh < KH; kh++)
#pragma unroll
            for (int c = 0; c < CPR; c++) raw[kh * CPR + c] = *(const uint4 *)(p + (int64_t)kh * g.IW * g.Cin + 32 * c);
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);

    // ---- weight image: task = eight consecutive taps of one output channel ------------------------------------
    for (int t = tid; t < (K / 8) * kCout; t += kThreads) {
        const int co = t & 31, kg = t >> 5;                      // taps 8 kg .. 8 kg + 7 in (kh, kw, ci) order
        const int kh = kg / (4 * CPR), q = kg - kh * (4 * CPR);  // q = 4 c + 2 h + u inside the kernel row
        const int d = kh * CPR + (q >> 2), hh = (q >> 1) & 1, u = q & 1;
        unsigned hi[8], mid[8], lo[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float w = W[(int64_t)(8 * kg + j) * kCout + co];
            hi[j] = __float_as_uint(w) & 0xffff0000u;
            const float r1 = w - __uint_as_float(hi[j]);         // exact: the low sixteen significand bits
            mid[j] = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mid[j]);       // exact: at most eight bits, a bf16 as it stands
            lo[j] = __float_as_uint(r2);
        }
        uint4 *dst = wimg + ((d * 2 + u) * 3) * 64 + hh * 32 + co;
        dst[0] = make_uint4(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]), pack_hi16(hi[4], hi[5]), pack_hi16(hi[6], hi[7]));
        dst[64] = make_uint4(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]), pack_hi16(mid[4], mid[5]), pack_hi16(mid[6], mid[7]));
        dst[128] = make_uint4(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]), pack_hi16(lo[4], lo[5]), pack_hi16(lo[6], lo[7]));
    }
    const float bco = bias ? bias[r] : 0.f;
    __syncthreads();

    for (; tile < ntiles; tile += gridDim.x) {
        // ---- bytes -> bf16 fragments (cvt_f32_ubyte, then the upper halves of two floats side by side) -------
        uint4 frag[DS][2];
#pragma unroll
        for (int d = 0; d < DS; d++) {
            const unsigned w[4] = {raw[d].x, raw[d].y, raw[d].z, raw[d].w};
#pragma unroll
            for (int u = 0; u < 2; u++) {
                unsigned o[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned src = w[2 * u + (q >> 1)];
                    const float f0 = (float)((src >> (16 * (q & 1))) & 0xffu);
                    const float f1 = (float)((src >> (16 * (q & 1) + 8)) & 0xffu);
                    o[q] = pack_hi16(__float_as_uint(f0), __float_as_uint(f1));
                }
                frag[d][u] = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        // ---- the next tile's taps fly while this one multiplies ------------------------------------------------
        const int64_t next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);

        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        uint4 wf[2][3];         // weight fragments one k-step ahead of the products that take them
#pragma unroll
        for (int p = 0; p < 3; p++) wf[0][p] = wimg[p * 64 + lane];
#pragma unroll
        for (int du = 0; du < 2 * DS; du++) {
            if (du + 1 < 2 * DS) {
#pragma unroll
                for (int p = 0; p < 3; p++) wf[(du + 1) & 1][p] = wimg[((du + 1) * 3 + p) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 a = __builtin_bit_cast(bf16x8, frag[du >> 1][du & 1]);
#pragma unroll
            for (int p = 2; p >= 0; p--)        // small pieces first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, wf[du & 1][p]), acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: rows (e & 3) + 8 (e >> 2) + 4 h of the wave's 32 pixels, column r = output channel --------
        const int64_t m0 = tile * 128 + wave * 32;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int64_t m = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float v = acc[e] * scale + bco;
            if (m < M) Y[m * kCout + r] = v > 0.f ? v : 0.f;
        }
    }
}

bool conv_u8_forward_ok(const void *X, const ConvShape &c) {
    static const bool plain = getenv("PPO2HIP_CONV_U8_PLAIN") != nullptr;
    if (plain) return false;
    return c.Cout == u8c::kCout && c.KH == 8 && c.KW * c.Cin == 32 && ((uintptr_t)X & 15) == 0 &&
           (c.stride * c.Cin) % 16 == 0 && (c.IW * c.Cin) % 16 == 0 && ((int64_t)c.IH * c.IW * c.Cin) % 16 == 0 && c.B > 0;
}

int conv_u8_forward(const void *X, const float *W, const float *b, float *Y, const ConvShape &c, hipStream_t s) {
    using namespace u8c;
    const int64_t M = (int64_t)c.B * c.OH * c.OW;
    const int64_t ntiles = ceil_div(M, 128);
    const int grid = (int)(ntiles < 768 ? ntiles : 768);       // three workgroups a CU (48 KB of LDS each), each walks its tiles
    Geom g{c.IH, c.IW, c.Cin, c.stride, c.OH, c.OW};
    hipLaunchKernelGGL((conv_u8_forward_kernel<8, 1>), dim3(grid), dim3(kThreads), 0, s, (const uint8_t *)X, W, b, Y, g, M,
                       ntiles, 1.0f / 255.0f);
    return check_launch("conv_u8_forward");
}

}  // namespace ppo2hip
