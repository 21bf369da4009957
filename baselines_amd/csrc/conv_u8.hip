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
//
// Both kernels of this file can take a list of frame numbers (`rows`) and read a minibatch's frames where they lie in the
// rollout instead of from a gathered copy; the forward can also leave one word of "output > 0" bits per output pixel
// (`relu_bits`), which the next layer's input gradient reads for the rectifier's derivative instead of the activations.
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
                                                                        int64_t ntiles, float scale,
                                                                        const int32_t *__restrict__ rows,
                                                                        uint32_t *__restrict__ relu_bits) {
    using namespace u8c;
    constexpr int DS = KH * CPR;                 // 32-tap runs = pairs of k-steps
    constexpr int K = DS * 32;
    __shared__ uint4 wimg[DS * 2 * 3 * 64];      // [run d][k-step u][piece p][lane] fragments of eight bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // first tile's taps go out before the weights are cut up; two register sets, so that a tile's loads have the whole
    // of the tile before it to arrive in
    uint4 raw0[DS], raw1[DS];
    auto load_tile = [&](int64_t tile, uint4 (&raw)[DS]) {
        int64_t mm = tile * 128 + wave * 32 + r;
        mm = mm < M ? mm : M - 1;
        const unsigned m = (unsigned)mm;                     // M < 2^31 (host)
        const unsigned ohw = (unsigned)(g.OH * g.OW);
        const unsigned b = m / ohw, rem = m - b * ohw;
        const unsigned oh = rem / (unsigned)g.OW, ow = rem - oh * (unsigned)g.OW;
        const int64_t frame = rows ? rows[b] : b;            // frames read where they lie in the rollout
        const uint8_t *p = X + ((frame * g.IH + oh * g.S) * g.IW + ow * g.S) * g.Cin + 16 * h;
#pragma unroll
        for (int kh = 0; kh < KH; kh++)
#pragma unroll
            for (int c = 0; c < CPR; c++) raw[kh * CPR + c] = *(const uint4 *)(p + (int64_t)kh * g.IW * g.Cin + 32 * c);
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile, raw0);

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

    // bytes -> bf16 fragments of one 32-tap run (cvt_f32_ubyte, then the upper halves of two floats side by side)
    auto convert = [&](const uint4 &v, uint4 (&f)[2]) {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
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
            f[u] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    };
    // one tile from the register set `cur` while the next tile's loads land in `nxt`; the conversion of run d + 1 is
    // independent vector work between the dependent products of run d
    auto do_tile = [&](uint4 (&cur)[DS], uint4 (&nxt)[DS]) {
        const int64_t next = tile + gridDim.x;
        if (next < ntiles) load_tile(next, nxt);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; e++) acc[e] = 0.f;
        uint4 frag[2][2];
        convert(cur[0], frag[0]);
#pragma unroll
        for (int d = 0; d < DS; d++) {
            uint4 wf[2][3];
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int p = 0; p < 3; p++) wf[u][p] = wimg[((d * 2 + u) * 3 + p) * 64 + lane];
            if (d + 1 < DS) convert(cur[d + 1], frag[(d + 1) & 1]);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, frag[d & 1][u]);
#pragma unroll
                for (int p = 2; p >= 0; p--)        // small pieces first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, wf[u][p]), acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // one run's reads, conversion and products at a time in flight
        }
        // ---- epilogue: rows (e & 3) + 8 (e >> 2) + 4 h of the wave's 32 pixels, column r = output channel --------
        const int64_t m0 = tile * 128 + wave * 32;
        float *y = Y + (m0 + 4 * h) * kCout + r;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = fmaxf(fmaf(acc[e], scale, bco), 0.f);
        if (m0 + 32 <= M) {
#pragma unroll
            for (int e = 0; e < 16; e++) y[((e & 3) + 8 * (e >> 2)) * kCout] = v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 16; e++)
                if (m0 + (e & 3) + 8 * (e >> 2) + 4 * h < M) y[((e & 3) + 8 * (e >> 2)) * kCout] = v[e];
        }
        if (relu_bits) {
            // which outputs the rectifier let through, one word per pixel (bit = channel = lane r): lane L < 32 collects
            // the word of the wave's pixel L — register (L & 3) + 4 (L >> 3), lane half (L >> 2) & 1
            unsigned mine = 0u;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const unsigned long long bal = __ballot(v[e] > 0.f);
                const unsigned word = ((r >> 2) & 1) ? (unsigned)(bal >> 32) : (unsigned)bal;
                mine = ((r & 3) + 4 * (r >> 3) == e) ? word : mine;
            }
            if (h == 0 && m0 + r < M) relu_bits[m0 + r] = mine;
        }
    };
    while (tile < ntiles) {
        do_tile(raw0, raw1);
        tile += gridDim.x;
        if (tile >= ntiles) break;
        do_tile(raw1, raw0);
        tile += gridDim.x;
    }
}

bool conv_u8_forward_ok(const void *X, const ConvShape &c) {
    static const bool plain = getenv("PPO2HIP_CONV_U8_PLAIN") != nullptr;
    if (plain) return false;
    return c.Cout == u8c::kCout && c.KH == 8 && c.KW * c.Cin == 32 && ((uintptr_t)X & 15) == 0 &&
           (int64_t)c.B * c.OH * c.OW < 0x7fffffffll && (c.stride * c.Cin) % 16 == 0 && (c.IW * c.Cin) % 16 == 0 && ((int64_t)c.IH * c.IW * c.Cin) % 16 == 0 && c.B > 0;
}

int conv_u8_forward(const void *X, const float *W, const float *b, float *Y, const ConvShape &c, hipStream_t s,
                    const int32_t *rows, uint32_t *relu_bits) {
    using namespace u8c;
    const int64_t M = (int64_t)c.B * c.OH * c.OW;
    const int64_t ntiles = ceil_div(M, 128);
    const int grid = (int)(ntiles < 768 ? ntiles : 768);       // three workgroups a CU (48 KB of LDS each), each walks its tiles
    Geom g{c.IH, c.IW, c.Cin, c.stride, c.OH, c.OW};
    hipLaunchKernelGGL((conv_u8_forward_kernel<8, 1>), dim3(grid), dim3(kThreads), 0, s, (const uint8_t *)X, W, b, Y, g, M,
                       ntiles, 1.0f / 255.0f, rows, relu_bits);
    return check_launch("conv_u8_forward");
}


// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layer: dW[tap][co] = (1/255) sum over pixels of x[pixel][tap] dY[pixel][co], again with exact
// bf16 products — here the float32 operand that is cut into three pieces is dY.
//
// The sum runs over pixels, the slow index of both operands in memory, so both pass through LDS once per image:
//   * the image itself, converted to bf16 once (56 KB for 84 x 84 x 4; no im2col copy): lane (tap r, half h) of the A
//     operand gathers its eight pixels' values with 16-bit reads — a group of eight pixels is a 4 (ow) x 2 (oh) block, so
//     the eight addresses are two bases plus immediates, and the 32 lanes of a half read 64 consecutive bytes (no bank
//     conflict);
//   * dY, cut up on the way in: thread (co, group) takes its eight pixels' values, splits them and writes three 16-byte
//     fragments [piece][group][co] — exactly what lane (co, h) of the B operand reads back with one ds_read_b128.
//     The same values give the bias gradient (column sums of dY) on the side.
// Eight waves, one per kernel row: wave kh holds dW[kh, :, :, :] (32 taps x 32 channels, 16 accumulator registers) for
// the whole launch; a workgroup walks images z, z + gridDim.x, ... and leaves one slab [K + 1][32] for the fixed-order
// reduction the other weight gradients use.  Loads of the next image (taps and dY) fly during the products of this one.
namespace u8c {
constexpr int kWgThreads = 512;
constexpr int kXR = 4;            // 16-byte image loads per thread: images up to 32 KB
constexpr int kNR = 4;            // dY tasks per thread: up to 64 pixel groups (512 pixels) per image
struct WGeom {
    int IH, IW, Cin, OH, OW, S;
    int img_bytes, row_bytes, G, Gpad, Q;     // Q = OW / 4 groups per pair of output rows
};
}  // namespace u8c

__global__ __launch_bounds__(u8c::kWgThreads) void conv_u8_wgrad_kernel(const uint8_t *__restrict__ X,
                                                                        const float *__restrict__ dY,
                                                                        float *__restrict__ slabs, u8c::WGeom g, int B,
                                                                        int K, float scale, int bias_rows,
                                                                        const int32_t *__restrict__ rows) {
    using namespace u8c;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *ximg = smem;                                                    // [img_bytes] bf16 (+ slack to a multiple of 32 bytes)
    uint4 *pimg = (uint4 *)(smem + 2 * ((g.img_bytes + 15) & ~15));          // [3][Gpad][32] fragments
    const int tid = threadIdx.x, lane = tid & 63, kh = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int co = tid & 31;

    // the padding group of an odd group count multiplies as zero
    for (int t = tid; t < 3 * (g.Gpad - g.G) * 32; t += kWgThreads) {
        const int p = t / ((g.Gpad - g.G) * 32), q = t - p * (g.Gpad - g.G) * 32;
        pimg[(p * g.Gpad + g.G) * 32 + q] = make_uint4(0u, 0u, 0u, 0u);
    }

    uint4 xr[kXR];
    float dyr[kNR][8];
    // task i of this thread: group tid / 32 + 16 i, channel co; its pixels (2 ohp + dy, 4 owq + dx), jj = 4 dy + dx
    int dyoff[kNR];
#pragma unroll
    for (int i = 0; i < kNR; i++) {
        const int grp = (tid >> 5) + 16 * i;
        const int ohp = grp / g.Q, owq = grp - ohp * g.Q;
        dyoff[i] = grp < g.G ? ((2 * ohp) * g.OW + 4 * owq) * kCout + co : -1;
    }
    auto prefetch = [&](int b) {
        const uint4 *src = (const uint4 *)(X + (int64_t)(rows ? rows[b] : b) * g.img_bytes);
#pragma unroll
        for (int i = 0; i < kXR; i++)
            if ((tid + i * kWgThreads) * 16 < g.img_bytes) xr[i] = src[tid + i * kWgThreads];
        const float *dsrc = dY + (int64_t)b * g.OH * g.OW * kCout;
#pragma unroll
        for (int i = 0; i < kNR; i++)
            if (dyoff[i] >= 0) {
#pragma unroll
                for (int jj = 0; jj < 8; jj++) dyr[i][jj] = dsrc[dyoff[i] + ((jj >> 2) * g.OW + (jj & 3)) * kCout];
            }
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    float bsum = 0.f;

    int b = blockIdx.x;
    if (b < B) prefetch(b);
    for (; b < B; b += gridDim.x) {
        // ---- phase A: this image as bf16 (every byte converted once, not once per kernel row and tap that meets it) and
        //      cut-up dY into LDS ------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < kXR; i++)
            if ((tid + i * kWgThreads) * 16 < g.img_bytes) {
                const unsigned w[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
                unsigned o[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const unsigned src = w[q >> 1];
                    const float f0 = (float)((src >> (16 * (q & 1))) & 0xffu);
                    const float f1 = (float)((src >> (16 * (q & 1) + 8)) & 0xffu);
                    o[q] = pack_hi16(__float_as_uint(f0), __float_as_uint(f1));
                }
                uint4 *dst = (uint4 *)ximg + 2 * (tid + i * kWgThreads);
                dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
                dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
            }
#pragma unroll
        for (int i = 0; i < kNR; i++)
            if (dyoff[i] >= 0) {
                unsigned hi[8], mid[8], lo[8];
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    const float w = dyr[i][jj];
                    bsum += w;
                    hi[jj] = __float_as_uint(w) & 0xffff0000u;
                    const float r1 = w - __uint_as_float(hi[jj]);
                    mid[jj] = __float_as_uint(r1) & 0xffff0000u;
                    lo[jj] = __float_as_uint(r1 - __uint_as_float(mid[jj]));
                }
                uint4 *dst = pimg + ((tid >> 5) + 16 * i) * 32 + co;
                dst[0] = make_uint4(pack_hi16(hi[0], hi[1]), pack_hi16(hi[2], hi[3]), pack_hi16(hi[4], hi[5]), pack_hi16(hi[6], hi[7]));
                dst[g.Gpad * 32] = make_uint4(pack_hi16(mid[0], mid[1]), pack_hi16(mid[2], mid[3]), pack_hi16(mid[4], mid[5]), pack_hi16(mid[6], mid[7]));
                dst[2 * g.Gpad * 32] = make_uint4(pack_hi16(lo[0], lo[1]), pack_hi16(lo[2], lo[3]), pack_hi16(lo[4], lo[5]), pack_hi16(lo[6], lo[7]));
            }
        __syncthreads();
        if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);

        // ---- phase B: groups 2 kb + h, two per product step ---------------------------------------------------------
        // lane (tap r, half h) takes the eight pixels of its group at this tap: element 2 q + e of the fragment is pixel
        // (dy, dx) = (q >> 1, 2 (q & 1) + e) of the 2 x 4 block; 16-bit reads land in register halves as they are
        const unsigned short *xp = (const unsigned short *)ximg + kh * g.row_bytes + 64 * h + r;      // group h of step 0 (Q >= 2)
        int owq = h;
        const int row4 = 4 * g.row_bytes, wrap = 8 * g.row_bytes - 64 * g.Q;
        auto fetch_frag = [&](uint4 &f) {
            const unsigned short *p1 = xp + row4;
            f.x = (unsigned)xp[0] | ((unsigned)xp[16] << 16);
            f.y = (unsigned)xp[32] | ((unsigned)xp[48] << 16);
            f.z = (unsigned)p1[0] | ((unsigned)p1[16] << 16);
            f.w = (unsigned)p1[32] | ((unsigned)p1[48] << 16);
        };
        auto advance = [&]() {          // two groups on: 128 taps-of-a-row further, or into the next pair of output rows
            owq += 2;
            xp += 128;
            if (owq >= g.Q) { owq -= g.Q; xp += wrap; }
        };
        uint4 xf[2];
        uint4 wf[2][3];
        fetch_frag(xf[0]);
#pragma unroll
        for (int p = 0; p < 3; p++) wf[0][p] = pimg[(p * g.Gpad + h) * 32 + r];
        const int nkb = g.Gpad >> 1;
        for (int kb = 0; kb < nkb; kb += 2) {
#pragma unroll
            for (int par = 0; par < 2; par++) {
                if (kb + par < nkb) {
                    advance();
                    if (kb + par + 1 < nkb) {
                        fetch_frag(xf[par ^ 1]);
#pragma unroll
                        for (int p = 0; p < 3; p++) wf[par ^ 1][p] = pimg[(p * g.Gpad + 2 * (kb + par + 1) + h) * 32 + r];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 a = __builtin_bit_cast(bf16x8, xf[par]);
#pragma unroll
                    for (int p = 2; p >= 0; p--)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, wf[par][p]), acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();
    }

    // ---- this workgroup's slab: rows kh 32 + tap, then the bias row --------------------------------------------------
    float *slab = slabs + (int64_t)blockIdx.x * (K + 1) * kCout;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int tap = (e & 3) + 8 * (e >> 2) + 4 * h;
        slab[(kh * 32 + tap) * kCout + r] = acc[e] * scale;
    }
    if (bias_rows) {
        float *red = (float *)smem;             // the images are spent
        red[tid] = bsum;
        __syncthreads();
        if (tid < kCout) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < kWgThreads / kCout; q++) t += red[q * kCout + tid];
            slab[K * kCout + tid] = t;
        }
    }
}

static size_t conv_u8_wgrad_lds(const ConvShape &c) {
    const int G = c.OH * c.OW / 8, Gpad = (G + 1) & ~1;
    const size_t img = ((size_t)c.IH * c.IW * c.Cin + 15) & ~(size_t)15;
    // the padding group's byte reads may run a few image rows past the image: they land in the fragment area
    return 2 * img + (size_t)3 * Gpad * 32 * 16;
}

bool conv_u8_wgrad_ok(const void *X, const float *dY, const ConvShape &c) {
    static const bool plain = getenv("PPO2HIP_CONV_U8_PLAIN") != nullptr;
    if (plain) return false;
    const int64_t img = (int64_t)c.IH * c.IW * c.Cin;
    const int G = c.OH * c.OW / 8;
    return c.Cout == u8c::kCout && c.KH == 8 && c.KW * c.Cin == 32 && c.stride * c.Cin == 16 && ((uintptr_t)X & 15) == 0 &&
           img % 16 == 0 && img <= (int64_t)u8c::kXR * u8c::kWgThreads * 16 && c.OW % 4 == 0 && c.OH % 2 == 0 && c.OW >= 8 &&
           G <= 16 * u8c::kNR && conv_u8_wgrad_lds(c) <= 150 * 1024 && c.B > 0 && dY != nullptr;
}

bool conv_u8_direct_ok(const void *X, const ConvShape &c) {
    static const bool gathered = getenv("PPO2HIP_CONV_U8_GATHER") != nullptr;
    return !gathered && conv_u8_forward_ok(X, c) && conv_u8_wgrad_ok(X, (const float *)X, c);
}

int conv_u8_wgrad_slabs(const void *X, const float *dY, const ConvShape &c, float *slabs, int max_slabs, int *nslabs,
                        hipStream_t s, bool bias_rows, const int32_t *rows) {
    using namespace u8c;
    int n = c.B < 256 ? c.B : 256;             // one workgroup (133 KB of LDS for 84 x 84 x 4) per CU
    if (n > max_slabs) n = max_slabs;
    if (n < 1) n = 1;
    *nslabs = n;
    WGeom g;
    g.IH = c.IH; g.IW = c.IW; g.Cin = c.Cin; g.OH = c.OH; g.OW = c.OW; g.S = c.stride;
    g.img_bytes = c.IH * c.IW * c.Cin; g.row_bytes = c.IW * c.Cin;
    g.G = c.OH * c.OW / 8; g.Gpad = (g.G + 1) & ~1; g.Q = c.OW / 4;
    const int K = c.KH * c.KW * c.Cin;
    const size_t lds = conv_u8_wgrad_lds(c);
    // per device, and cheap: asked for on every call rather than remembered per device
    if (hipFuncSetAttribute((const void *)conv_u8_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        set_error("conv_u8_wgrad: cannot raise the dynamic LDS limit");
        return PPO2HIP_EINVAL;
    }
    hipLaunchKernelGGL(conv_u8_wgrad_kernel, dim3(n), dim3(kWgThreads), lds, s, (const uint8_t *)X, dY, slabs, g, c.B, K,
                       1.0f / 255.0f, bias_rows ? 1 : 0, rows);
    return check_launch("conv_u8_wgrad");
}

}  // namespace ppo2hip
