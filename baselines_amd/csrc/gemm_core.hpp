// fp32 MFMA GEMM core for gfx950: C = epilogue(A x B) with functor loaders, so that dense
// layers (plain / transposed operands), implicit-GEMM convolutions and weight-gradient
// products (split over the sample axis into slabs) all share one LDS-tiled main loop.
//
// Tile: BM x BN x 16 (128 x 64, 64 x 64 or, for outputs of at most 32 columns, 128 x 32), 256 threads =
// 4 wave64; operands reach LDS element by element or, where pointers and strides allow, by 16-byte loads.  v_mfma_f32_32x32x2_f32 is an exact fp32
// fma chain, so results are reproducible and within fp32 rounding of a float64 reference.
//   operand lanes : A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]
//   result lanes  : C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
// LDS holds both tiles k-major ([k][m], [k][n]) so a wave's operand read is two runs of 32
// consecutive floats (conflict-free).  The next k-slab is prefetched into registers while
// the current one feeds the matrix cores.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.hpp"

namespace ppo2hip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short gemm_bf16x8 __attribute__((ext_vector_type(8)));

// A float32 is the exact sum of three bf16 numbers: h = x rounded to bf16 (half an ulp added, then truncated),
// m = x - h truncated to bf16 (x - h is exact and has at most 16 significant bits), l = x - h - m (at most 8 bits: a
// bf16 as it stands).  PIECES products cut both operands into such pieces and run on the bf16 matrix pipe: of the nine
// piece products the six whose weight is 2^-17 of the result or more are kept — every kept product is exact, what is
// dropped (m l, l m, l l) is below 2^-24 of its term — and v_mfma_f32_32x32x16_bf16 retires a 16-deep slab of a
// 32 x 32 block in 6 instructions of 32 cycles where v_mfma_f32_32x32x2_f32 takes 8 of 64 on the vector ALUs.
// Rounding h (not truncating it) is what keeps sums of same-signed terms unbiased: m and l then take either sign, and
// the matrix pipe's truncation of small addends against a large accumulator cancels instead of adding up.  Measured
// against float64 on K = 3136 with all-positive operands, mean error over sum |a b|: 4.3e-7 (float32 pipe 6.7e-7,
// all three pieces truncated 8.0e-6); with operands of either sign all forms give 1.7e-8 (float32 pipe 2.0e-8).
// The tiles stay float32 in LDS, k-major as for the float32 products, and each lane cuts the eight k of its operand
// row as it reads them.  Cutting on the way INTO LDS instead (piece tiles, 16-byte operand reads) measured the same
// for the k-contiguous operands and cannot serve the m / n-contiguous ones of the weight gradients at all; two LDS
// buffers with two slabs of loads in flight lost a resident workgroup per CU and with it more than the barrier saved;
// 128-column tiles (the A tile cut once for twice the columns) lost to the same: 64 accumulator registers more;
// v_cvt_pk_bf16_f32 for all three pieces (round to nearest even throughout) cost 12 % of the product's time, rounding
// h alone as here 2.5 % of an Atari update over truncating everything.
// eight k-consecutive elements -> their three pieces as MFMA operands
__device__ __forceinline__ void cut_run8(const float *v, uint4 *out) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        h[j] = (__float_as_uint(v[j]) + 0x8000u) & 0xffff0000u;
        const float r = v[j] - __uint_as_float(h[j]);
        m[j] = __float_as_uint(r) & 0xffff0000u;
        l[j] = __float_as_uint(r - __uint_as_float(m[j]));
    }
    // upper halves of two floats side by side: bytes 2, 3 of the first, then of the second
    auto pk = [](unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); };
    out[0] = make_uint4(pk(h[0], h[1]), pk(h[2], h[3]), pk(h[4], h[5]), pk(h[6], h[7]));
    out[1] = make_uint4(pk(m[0], m[1]), pk(m[2], m[3]), pk(m[4], m[5]), pk(m[6], m[7]));
    out[2] = make_uint4(pk(l[0], l[1]), pk(l[2], l[3]), pk(l[4], l[5]), pk(l[6], l[7]));
}
constexpr int GEMM_BN = 64;        // widest tile
constexpr int GEMM_BK = 16;
constexpr int GEMM_THREADS = 256;

// Loader concept (A: coordinates (m, k), B: (k, n)); out-of-range coordinates are never requested:
//     Row row(int64_t r) const;  Col col(int64_t c) const;      what the element address owes to each coordinate
//     void advance_col(Col &, int d) const;  void advance_row(Row &, int d) const;   move a coordinate by d >= 0
//     float at(const Row &, const Col &) const;
//     static constexpr bool kInnerFast — true when the SECOND coordinate is the contiguous one in memory
//                                        (drives the thread->element mapping for coalescing).
//     static constexpr int kVec;  int vec_width() const (host);  void at_vec(const Row &, const Col &, float (&)[kVec]) const
//         — kVec > 1: one wide load returns kVec consecutive elements along the contiguous coordinate, starting at
//           a multiple of kVec; vec_width() says on the host whether this operand's pointers and strides allow it
//           (kVec) or not (1).
//   A thread's elements share one coordinate and differ in the other, and only the k coordinate moves from
//   slab to slab, so the index arithmetic of an implicit-GEMM loader (divisions that split a row number into
//   image / pixel, a column into tap / channel) runs once per tile or once per slab instead of once per element;
//   on the vector path it also runs once per kVec elements, which is what keeps the main loop's VALU work below
//   its MFMA work for the narrow (few output columns) and byte-input products.
// Epilogue concept: Row row(int64_t m, int z) const — what the output address owes to the row (and the slab);
//                   void operator()(const Row &, int n, float acc) const

// a loader with `static constexpr bool kTileK = true` has begin_tile(m0, kbeg, kend) (A) / begin_tile(m0) (B)
template <class L, class = void>
struct tile_k : std::false_type {};
template <class L>
struct tile_k<L, std::enable_if_t<L::kTileK>> : std::true_type {};

// an epilogue with `static constexpr bool kColSum = true` returns what it stored and has a `float *colsum` (null: off)
template <class E, class = void>
struct col_sum : std::false_type {};
template <class E>
struct col_sum<E, std::enable_if_t<E::kColSum>> : std::true_type {};

// an epilogue with `static constexpr bool kColCtx = true` has `int64_t col(int n) const` — what an output element's
// address owes to its column, worked out once per thread and 32-column block instead of once per element — and takes
// that value where the others take n
template <class E, class = void>
struct col_ctx : std::false_type {};
template <class E>
struct col_ctx<E, std::enable_if_t<E::kColCtx>> : std::true_type {};

// LDS row pitch: rows lh = 0 / 1 of an operand read must land on different bank halves (pitch = 32 mod 64)
// when rows are written as vectors; the scalar path keeps the odd pitch that spreads its k-fast writes.
// Vectors along k written with the k group as the fastest lane index (four lanes cover 64 contiguous bytes of one
// operand row, sixteen rows per load) need pitch = 4 mod 16 for conflict-free writes: 36 mod 64 serves both but
// for four lanes of each operand read.
__host__ __device__ constexpr int lds_pitch(int width, int mode) {
    return mode == 0 ? width + 1 : mode == 1 ? (width % 64 == 0 ? width + 32 : width) : (width % 64 == 0 ? width + 36 : width + 4);
}

// WM = number of wave rows, NB = 32-column blocks per wave:
//   <4, 2>: 128 x 64, each wave 32 x 64;  <2, 1>: 64 x 64, each wave 32 x 32;  <4, 1>: 128 x 32, each wave 32 x 32
// AV / BV = vector width of the A / B loads (1 = element-wise)
// FULLK: every slab of every tile is a whole 16 steps of k (K, the split length and a tile-dependent k range are all
// multiples of 16), so the loads need neither tests nor zero-fill — with float32 matrix products every vector instruction
// saved in the main loop is matrix time won.
// PIECES: the products run as bf16 piece products on the bf16 pipe (see cut3)
template <class AL, class BL, class EP, int WM, int NB, int AV, int BV, bool FULLK = false, bool PIECES = false>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(AL a, BL b, EP ep, int64_t M, int N, int64_t K,
                                                            int64_t klen) {
    constexpr int BM = 32 * WM;
    constexpr int WN = 4 / WM;              // wave columns
    constexpr int BN = 32 * WN * NB;
    constexpr bool A_KG_FAST = AL::kInnerFast && AV > 1 && AV < GEMM_BK;   // several k groups per row and slab
    // m-contiguous A read sixteen at a time (uint8 taps): lanes run along k, so that a wave's sixteen-byte loads are
    // consecutive pixels (contiguous in memory for a stride of four 4-channel pixels) and its four 16-byte LDS writes per
    // lane land on sixteen different rows, which the 36-mod-64 pitch spreads over all banks
    constexpr bool A_K_LANES = !AL::kInnerFast && AV >= 8;
    constexpr int LDA = lds_pitch(BM, AV == 1 ? 0 : (A_KG_FAST || A_K_LANES) ? 2 : 1), LDB = lds_pitch(BN, BV == 1 ? 0 : BL::kInnerFast ? 1 : 2);
    // element-wise: elements per thread; vector: tasks (one wide load each) per thread
    constexpr int A_TASKS_ALL = BM * GEMM_BK / AV, B_TASKS_ALL = BN * GEMM_BK / BV;
    constexpr int A_PER = (A_TASKS_ALL + GEMM_THREADS - 1) / GEMM_THREADS;
    constexpr int B_PER = (B_TASKS_ALL + GEMM_THREADS - 1) / GEMM_THREADS;
    __shared__ __attribute__((aligned(16))) float As[GEMM_BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[GEMM_BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int z = blockIdx.z;
    int64_t kbeg = (int64_t)z * klen;
    int64_t kend = kbeg + klen < K ? kbeg + klen : K;
    // loaders whose k range depends on the tile (a product that skips what is zero for all of a tile's rows) set
    // themselves up for this tile and say which k it runs over
    if constexpr (tile_k<AL>::value) {
        a.begin_tile(m0, kbeg, kend);
        b.begin_tile(m0);
    }

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nb][r] = 0.f;

    float areg[A_PER][AV], breg[B_PER][BV];

    // thread -> element mapping, element-wise path (element e of the thread is tile entry lin = tid + e * 256)
    //   A, kInnerFast : k = lin % 16 is the thread's own, rows differ with e          (row contexts: once)
    //   A, otherwise  : m = lin % BM is the thread's own, k differs with e and slab    (col contexts: advanced)
    //   B, kInnerFast : n = lin % BN is the thread's own, k differs with e and slab    (row contexts: advanced)
    //   B, otherwise  : k = lin % 16 is the thread's own, n differs with e             (col contexts: once)
    // vector path (task t = tid + e * 256 is one wide load)
    //   A, kInnerFast : one k group per row and slab (AV = 16): row m = t % BM, lanes along m;
    //                   several (AV = 4): k group t % (16/AV) is the thread's own and fastest over the lanes, so a
    //                   load instruction touches sixteen rows' 64-byte pieces, not 64 rows' 16-byte pieces; rows t / (16/AV)
    //   A, otherwise  : m group t % (BM/AV), k = t / (BM/AV), advanced from slab to slab (LDS rows written as float4)
    //   B, kInnerFast : n group t % (BN/BV), k = t / (BN/BV), advanced from slab to slab
    //   B, otherwise  : k group t % (16/BV) fastest over the lanes (as for A), column n = t / (16/BV) the thread's own
    constexpr int A_KSTEP = GEMM_THREADS / BM;          // k distance between a thread's elements (A, m-fast mapping)
    constexpr int B_KSTEP = GEMM_THREADS / BN;
    constexpr int A_MG = BM / AV, B_NG = BN / BV;       // groups along the contiguous coordinate (vector path)
    constexpr bool A_ROWS_ONCE = AL::kInnerFast;        // which context array is per-element
    constexpr int A_KGS = GEMM_BK / AV;
    typename AL::Row arow[((AV == 1 && A_ROWS_ONCE) || A_KG_FAST) ? A_PER : 1];
    typename AL::Col acol[((AV == 1 && A_ROWS_ONCE) || A_KG_FAST) ? 1 : A_PER];
    constexpr int B_KGS = GEMM_BK / BV;
    typename BL::Row brow[BL::kInnerFast ? B_PER : 1];
    typename BL::Col bcol[BL::kInnerFast ? 1 : B_PER];
    bool a_ok[((AV == 1 && A_ROWS_ONCE) || A_KG_FAST) ? A_PER : 1], b_ok[BL::kInnerFast ? 1 : B_PER];
    if constexpr (AV > 1) {
        if constexpr (A_KG_FAST) {
#pragma unroll
            for (int e = 0; e < A_PER; e++) {
                const int64_t m = m0 + (tid + e * GEMM_THREADS) / A_KGS;
                a_ok[e] = m < M;
                arow[e] = a.row(a_ok[e] ? m : 0);
            }
            acol[0] = a.col(kbeg + (tid % A_KGS) * AV);
        } else if constexpr (AL::kInnerFast) {
            const int64_t m = m0 + tid % BM;
            a_ok[0] = m < M;
            arow[0] = a.row(a_ok[0] ? m : 0);
#pragma unroll
            for (int e = 0; e < A_PER; e++) acol[e] = a.col(kbeg + ((tid + e * GEMM_THREADS) / BM) * AV);
        } else {
            const int64_t m = m0 + (A_K_LANES ? tid / GEMM_BK : tid % A_MG) * AV;
            a_ok[0] = m < M;                             // M is a multiple of AV: a group is all in or all out
            arow[0] = a.row(a_ok[0] ? m : 0);
#pragma unroll
            for (int e = 0; e < A_PER; e++) acol[e] = a.col(kbeg + (A_K_LANES ? tid % GEMM_BK : (tid + e * GEMM_THREADS) / A_MG));
        }
    } else if constexpr (AL::kInnerFast) {
#pragma unroll
        for (int e = 0; e < A_PER; e++) {
            const int64_t m = m0 + (tid + e * GEMM_THREADS) / GEMM_BK;
            a_ok[e] = m < M;
            arow[e] = a.row(a_ok[e] ? m : 0);
        }
        acol[0] = a.col(kbeg + tid % GEMM_BK);
    } else {
        const int64_t m = m0 + tid % BM;
        a_ok[0] = m < M;
        arow[0] = a.row(a_ok[0] ? m : 0);
#pragma unroll
        for (int e = 0; e < A_PER; e++) acol[e] = a.col(kbeg + tid / BM + e * A_KSTEP);
    }
    if constexpr (BV > 1 && !BL::kInnerFast) {
#pragma unroll
        for (int e = 0; e < B_PER; e++) {
            const int n = n0 + (tid + e * GEMM_THREADS) / B_KGS;
            b_ok[e] = n < N && tid + e * GEMM_THREADS < B_TASKS_ALL;
            bcol[e] = b.col(n < N ? n : 0);
        }
        brow[0] = b.row(kbeg + (tid % B_KGS) * BV);
    } else if constexpr (BV > 1) {
        const int n = n0 + (tid % B_NG) * BV;
        b_ok[0] = n < N;                                 // N is a multiple of BV
        bcol[0] = b.col(b_ok[0] ? n : 0);
#pragma unroll
        for (int e = 0; e < B_PER; e++) brow[e] = b.row(kbeg + (tid + e * GEMM_THREADS) / B_NG);
    } else if constexpr (BL::kInnerFast) {
        const int n = n0 + tid % BN;
        b_ok[0] = n < N;
        bcol[0] = b.col(b_ok[0] ? n : 0);
#pragma unroll
        for (int e = 0; e < B_PER; e++) brow[e] = b.row(kbeg + tid / BN + e * B_KSTEP);
    } else {
#pragma unroll
        for (int e = 0; e < B_PER; e++) {
            const int n = n0 + (tid + e * GEMM_THREADS) / GEMM_BK;
            b_ok[e] = n < N;
            bcol[e] = b.col(b_ok[e] ? n : 0);
        }
        brow[0] = b.row(kbeg + tid % GEMM_BK);
    }

    // loads slab [k0, k0 + 16); the advancing contexts stand at k0 on entry and at k0 + 16 on exit
    auto fetch = [&](int64_t k0) {
        if constexpr (A_KG_FAST) {
            const bool kok = FULLK || k0 + (tid % A_KGS) * AV < kend;        // K is a multiple of AV
#pragma unroll
            for (int e = 0; e < A_PER; e++) {
                if (FULLK || (kok && a_ok[e])) {        // rows past M are clamped to row 0 and never stored
                    a.at_vec(arow[e], acol[0], areg[e]);
                } else {
#pragma unroll
                    for (int j = 0; j < AV; j++) areg[e][j] = 0.f;
                }
            }
            a.advance_col(acol[0], GEMM_BK);
        } else if constexpr (AV > 1 && AL::kInnerFast) {
#pragma unroll
            for (int e = 0; e < A_PER; e++) {
                const int t = tid + e * GEMM_THREADS;
                const int64_t k = k0 + (t / BM) * AV;                       // K is a multiple of AV
                if ((A_TASKS_ALL % GEMM_THREADS == 0 || t < A_TASKS_ALL) && (FULLK || (a_ok[0] && k < kend))) {
                    a.at_vec(arow[0], acol[e], areg[e]);
                } else {
#pragma unroll
                    for (int j = 0; j < AV; j++) areg[e][j] = 0.f;
                }
                a.advance_col(acol[e], GEMM_BK);
            }
        } else if constexpr (AV > 1) {
#pragma unroll
            for (int e = 0; e < A_PER; e++) {
                const int t = tid + e * GEMM_THREADS;
                const int64_t k = k0 + (A_K_LANES ? t % GEMM_BK : t / A_MG);
                if ((A_TASKS_ALL % GEMM_THREADS == 0 || t < A_TASKS_ALL) && (FULLK || (a_ok[0] && k < kend))) {
                    a.at_vec(arow[0], acol[e], areg[e]);
                } else {
#pragma unroll
                    for (int j = 0; j < AV; j++) areg[e][j] = 0.f;
                }
                a.advance_col(acol[e], GEMM_BK);
            }
        } else if constexpr (AL::kInnerFast) {
            const int64_t k = k0 + tid % GEMM_BK;
            const bool kok = FULLK || k < kend;
#pragma unroll
            for (int e = 0; e < A_PER; e++) areg[e][0] = (FULLK || (kok && a_ok[e])) ? a.at(arow[e], acol[0]) : 0.f;
            a.advance_col(acol[0], GEMM_BK);
        } else {
#pragma unroll
            for (int e = 0; e < A_PER; e++) {
                const int64_t k = k0 + tid / BM + e * A_KSTEP;
                areg[e][0] = (FULLK || (a_ok[0] && k < kend)) ? a.at(arow[0], acol[e]) : 0.f;
                a.advance_col(acol[e], GEMM_BK);
            }
        }
        if constexpr (BV > 1 && !BL::kInnerFast) {
            const bool kok = FULLK || k0 + (tid % B_KGS) * BV < kend;        // K is a multiple of BV
#pragma unroll
            for (int e = 0; e < B_PER; e++) {
                if ((FULLK && (B_TASKS_ALL % GEMM_THREADS == 0 || tid + e * GEMM_THREADS < B_TASKS_ALL)) || (kok && b_ok[e])) {
                    b.at_vec(brow[0], bcol[e], breg[e]);
                } else {
#pragma unroll
                    for (int j = 0; j < BV; j++) breg[e][j] = 0.f;
                }
            }
            b.advance_row(brow[0], GEMM_BK);
        } else if constexpr (BV > 1) {
#pragma unroll
            for (int e = 0; e < B_PER; e++) {
                const int t = tid + e * GEMM_THREADS;
                const int64_t k = k0 + t / B_NG;
                if ((B_TASKS_ALL % GEMM_THREADS == 0 || t < B_TASKS_ALL) && (FULLK || (b_ok[0] && k < kend))) {
                    b.at_vec(brow[e], bcol[0], breg[e]);
                } else {
#pragma unroll
                    for (int j = 0; j < BV; j++) breg[e][j] = 0.f;
                }
                b.advance_row(brow[e], GEMM_BK);
            }
        } else if constexpr (BL::kInnerFast) {
#pragma unroll
            for (int e = 0; e < B_PER; e++) {
                const int64_t k = k0 + tid / BN + e * B_KSTEP;
                breg[e][0] = (FULLK || (b_ok[0] && k < kend)) ? b.at(brow[e], bcol[0]) : 0.f;
                b.advance_row(brow[e], GEMM_BK);
            }
        } else {
            const int64_t k = k0 + tid % GEMM_BK;
            const bool kok = FULLK || k < kend;
#pragma unroll
            for (int e = 0; e < B_PER; e++) breg[e][0] = (FULLK || (kok && b_ok[e])) ? b.at(brow[0], bcol[e]) : 0.f;
            b.advance_row(brow[0], GEMM_BK);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int e = 0; e < A_PER; e++) {
            const int lin = tid + e * GEMM_THREADS;
            if constexpr (A_KG_FAST) {
                const int kg = lin % A_KGS, ml = lin / A_KGS;
#pragma unroll
                for (int j = 0; j < AV; j++) As[(kg * AV + j) * LDA + ml] = areg[e][j];
            } else if constexpr (AV > 1 && AL::kInnerFast) {
                if (lin < A_TASKS_ALL) {
                    const int ml = lin % BM, kg = lin / BM;
#pragma unroll
                    for (int j = 0; j < AV; j++) As[(kg * AV + j) * LDA + ml] = areg[e][j];
                }
            } else if constexpr (AV > 1) {
                if (lin < A_TASKS_ALL) {
                    float *dst = A_K_LANES ? &As[(lin % GEMM_BK) * LDA + (lin / GEMM_BK) * AV] : &As[(lin / A_MG) * LDA + (lin % A_MG) * AV];
#pragma unroll
                    for (int q = 0; q < AV / 4; q++)
                        *(float4 *)(dst + 4 * q) = make_float4(areg[e][4 * q], areg[e][4 * q + 1], areg[e][4 * q + 2], areg[e][4 * q + 3]);
                }
            } else {
                int ml, kl;
                if (AL::kInnerFast) { kl = lin % GEMM_BK; ml = lin / GEMM_BK; }
                else { ml = lin % BM; kl = lin / BM; }
                As[kl * LDA + ml] = areg[e][0];
            }
        }
#pragma unroll
        for (int e = 0; e < B_PER; e++) {
            const int lin = tid + e * GEMM_THREADS;
            if constexpr (BV > 1 && !BL::kInnerFast) {
                if (lin < B_TASKS_ALL) {
                    const int kg = lin % B_KGS, nl = lin / B_KGS;
#pragma unroll
                    for (int j = 0; j < BV; j++) Bs[(kg * BV + j) * LDB + nl] = breg[e][j];
                }
            } else if constexpr (BV > 1) {
                if (lin < B_TASKS_ALL)
                    *(float4 *)&Bs[(lin / B_NG) * LDB + (lin % B_NG) * BV] = make_float4(breg[e][0], breg[e][1], breg[e][2], breg[e][3]);
            } else {
                int kl, nl;
                if (BL::kInnerFast) { nl = lin % BN; kl = lin / BN; }
                else { kl = lin % GEMM_BK; nl = lin / GEMM_BK; }
                Bs[kl * LDB + nl] = breg[e][0];
            }
        }
    };

    if (kbeg < kend) {
        fetch(kbeg);
        for (int64_t k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
            __syncthreads();           // previous slab fully consumed
            stash();
            __syncthreads();
            if (k0 + GEMM_BK < kend) fetch(k0 + GEMM_BK);
            if constexpr (PIECES) {
                // lane (li, lh) takes k = lh, 2 + lh, ..., 14 + lh of its operand row (any split of the slab's sixteen k
                // over the two lane halves serves, as long as A and B agree): the rows the float32 products read, whose
                // pitch keeps the two halves on different banks
                float ar[8];
#pragma unroll
                for (int j = 0; j < 8; j++) ar[j] = As[(2 * j + lh) * LDA + wm * 32 + li];
                uint4 af[3];
                cut_run8(ar, af);
#pragma unroll
                for (int nb = 0; nb < NB; nb++) {
                    float br[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) br[j] = Bs[(2 * j + lh) * LDB + (wn * NB + nb) * 32 + li];
                    uint4 bf[3];
                    cut_run8(br, bf);
                    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; t++)
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gemm_bf16x8, af[pa[t]]),
                                                                          __builtin_bit_cast(gemm_bf16x8, bf[pb[t]]), acc[nb], 0, 0, 0);
                }
            } else
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
    // what an output element's address owes to its row (a division or two for the implicit-GEMM epilogues) is
    // worked out once per row of the tile and handed round through LDS
    typedef typename EP::Row EpRow;
    static_assert(sizeof(EpRow) * BM <= sizeof(float) * GEMM_BK * LDA, "row contexts must fit the A tile");
    __syncthreads();
    EpRow *rows = reinterpret_cast<EpRow *>(As);
    if (tid < BM) rows[tid] = ep.row(m0 + tid < M ? m0 + tid : M - 1, z);
    __syncthreads();
    float csum[NB];          // column sums of what this thread stored (epilogues with kColSum)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
        const int n = n0 + (wn * NB + nb) * 32 + li;
        csum[nb] = 0.f;
        if constexpr (col_ctx<EP>::value) {
            static_assert(col_sum<EP>::value, "column contexts come with column sums");
            const int64_t cc = ep.col(n < N ? n : 0);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m0 + ml < M && n < N) csum[nb] += ep(rows[ml], cc, acc[nb][r]);
            }
        } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m0 + ml < M && n < N) {
                if constexpr (col_sum<EP>::value) csum[nb] += ep(rows[ml], n, acc[nb][r]);
                else ep(rows[ml], n, acc[nb][r]);
            }
        }
        }
    }
    // The consumer of a gradient tensor wants its column sums too (the bias gradient of the layer it came out of):
    // a tile's sums leave here, added in a fixed order (lane halves, then wave rows), one row of N per (row tile, slab),
    // instead of a second pass over the tensor.
    if constexpr (col_sum<EP>::value) {
        if (ep.colsum) {
            static_assert(WM * BN <= GEMM_BK * LDB, "tile sums must fit the B tile");
            __syncthreads();
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const float v = csum[nb] + __shfl_xor(csum[nb], 32, 64);
                if (lh == 0) Bs[wm * BN + (wn * NB + nb) * 32 + li] = v;
            }
            __syncthreads();
            if (tid < BN && n0 + tid < N) {
                float t = Bs[tid];
#pragma unroll
                for (int q = 1; q < WM; q++) t += Bs[q * BN + tid];
                ep.colsum[((int64_t)blockIdx.x * gridDim.z + z) * N + n0 + tid] = t;
            }
        }
    }
}

// a loader whose k range depends on the tile says on the host whether every tile's range is whole slabs
template <class L, class = void>
struct has_full_slabs : std::false_type {};
template <class L>
struct has_full_slabs<L, std::void_t<decltype(std::declval<const L &>().full_slabs())>> : std::true_type {};

// PIECES: a call site's opt-in to the bf16-piece form for its wide, whole-slab shape (one more instantiation per
// product, so only the products that carry a workload ask for it); PPO2HIP_GEMM_FP32 in the environment turns it off
template <bool PIECES, class AL, class BL, class EP, int AV, int BV>
inline int launch_gemm_v(const AL &a, const BL &b, const EP &ep, int64_t M, int N, int64_t K, int64_t klen, dim3 grid,
                         bool small_m, bool narrow, hipStream_t s, const char *what) {
    static const bool no_fullk = getenv("PPO2HIP_GEMM_CHECKED") != nullptr;
    bool fullk = K > 0 && K % GEMM_BK == 0 && !no_fullk;       // klen is a multiple of 16 by construction
    if constexpr (tile_k<AL>::value) {
        if constexpr (has_full_slabs<AL>::value) fullk = fullk && a.full_slabs();
        else fullk = false;
    }
    // the wide, well-shaped products are the ones that matter; the 64-row tiles keep the checked loads only
    if (small_m)
        hipLaunchKernelGGL((gemm_kernel<AL, BL, EP, 2, 1, AV, BV>), grid, dim3(GEMM_THREADS), 0, s, a, b, ep, M, N, K, klen);
    else if (narrow && fullk)
        hipLaunchKernelGGL((gemm_kernel<AL, BL, EP, 4, 1, AV, BV, true>), grid, dim3(GEMM_THREADS), 0, s, a, b, ep, M, N, K, klen);
    else if (narrow)
        hipLaunchKernelGGL((gemm_kernel<AL, BL, EP, 4, 1, AV, BV>), grid, dim3(GEMM_THREADS), 0, s, a, b, ep, M, N, K, klen);
    else if (fullk) {
        if constexpr (PIECES && AV == AL::kVec && BV == BL::kVec) {
            static const bool no_pieces = getenv("PPO2HIP_GEMM_FP32") != nullptr;
            if (!no_pieces) {
                hipLaunchKernelGGL((gemm_kernel<AL, BL, EP, 4, 2, AV, BV, true, true>), grid, dim3(GEMM_THREADS), 0, s, a, b, ep, M, N, K, klen);
                return check_launch(what);
            }
        }
        hipLaunchKernelGGL((gemm_kernel<AL, BL, EP, 4, 2, AV, BV, true>), grid, dim3(GEMM_THREADS), 0, s, a, b, ep, M, N, K, klen);
    } else
        hipLaunchKernelGGL((gemm_kernel<AL, BL, EP, 4, 2, AV, BV>), grid, dim3(GEMM_THREADS), 0, s, a, b, ep, M, N, K, klen);
    return check_launch(what);
}

// rows per tile launch_gemm takes for a product of M x N with nsplit slabs, and the number of row tiles that makes
inline bool gemm_small_m(int64_t M, int N, int nsplit) {
    // 64-row tiles for short operands, and wherever 128-row tiles would leave most of the 256 CUs without a workgroup
    return M <= 64 || (M <= 512 && ceil_div(M, 128) * ceil_div(N, 64) * nsplit < 128);
}
inline int64_t gemm_row_tiles(int64_t M, int N, int nsplit) { return ceil_div(M, gemm_small_m(M, N, nsplit < 1 ? 1 : nsplit) ? 64 : 128); }

template <bool PIECES = false, class AL, class BL, class EP>
inline int launch_gemm(const AL &a, const BL &b, const EP &ep, int64_t M, int N, int64_t K, int nsplit,
                       hipStream_t s, const char *what) {
    if (M <= 0 || N <= 0) return 0;
    if (nsplit < 1) nsplit = 1;
    int64_t klen = ceil_div(K > 0 ? K : 1, nsplit);
    klen = ceil_div(klen, GEMM_BK) * GEMM_BK;
    const bool small_m = gemm_small_m(M, N, nsplit);
    const bool narrow = !small_m && N <= 32;        // half of a 64-column tile would multiply zeros
    const int BM = small_m ? 64 : 128, BN = narrow ? 32 : 64;
    const int64_t gx = ceil_div(M, BM);
    if (gx > 0x7fffffffll || nsplit > 65535) {
        set_error("%s: grid too large", what);
        return PPO2HIP_EINVAL;
    }
    dim3 grid((unsigned)gx, (unsigned)ceil_div(N, BN), (unsigned)nsplit);
    // wide loads where the operand allows them and the coordinate they run along comes in whole groups
    constexpr int AVmax = AL::kVec, BVmax = BL::kVec;
    static_assert(BVmax == 1 || BVmax == 4, "B operands load four floats at a time");
    const bool av = AVmax > 1 && a.vec_width() == AVmax && (AL::kInnerFast ? K % AVmax == 0 : M % AVmax == 0);
    bool bv = false;
    if constexpr (BVmax > 1) bv = b.vec_width() == BVmax && (BL::kInnerFast ? N % BVmax == 0 : K % BVmax == 0);
    if constexpr (AVmax > 1 && BVmax > 1) {
        if (av && bv) return launch_gemm_v<PIECES, AL, BL, EP, AVmax, BVmax>(a, b, ep, M, N, K, klen, grid, small_m, narrow, s, what);
    }
    if constexpr (AVmax > 1) {
        if (av) return launch_gemm_v<PIECES, AL, BL, EP, AVmax, 1>(a, b, ep, M, N, K, klen, grid, small_m, narrow, s, what);
    }
    if constexpr (BVmax > 1) {
        if (bv) return launch_gemm_v<PIECES, AL, BL, EP, 1, BVmax>(a, b, ep, M, N, K, klen, grid, small_m, narrow, s, what);
    }
    return launch_gemm_v<PIECES, AL, BL, EP, 1, 1>(a, b, ep, M, N, K, klen, grid, small_m, narrow, s, what);
}

// ---- loaders ---------------------------------------------------------------------------
inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

struct RowMajor {   // element (r, c) at p[r*ld + c]
    static constexpr bool kInnerFast = true;
    static constexpr int kVec = 4;
    int vec_width() const { return aligned16(p) && ld % 4 == 0 ? 4 : 1; }
    __device__ void at_vec(const int64_t &r, const int64_t &c, float (&o)[4]) const {
        const float4 v = *(const float4 *)(p + r + c);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    typedef int64_t Row;
    typedef int64_t Col;
    const float *p;
    int64_t ld;
    __device__ Row row(int64_t r) const { return r * ld; }
    __device__ Col col(int64_t c) const { return c; }
    __device__ void advance_row(Row &r, int d) const { r += d * ld; }
    __device__ void advance_col(Col &c, int d) const { c += d; }
    __device__ float at(const Row &r, const Col &c) const { return p[r + c]; }
};
struct Transposed {  // element (r, c) at p[c*ld + r]
    static constexpr bool kInnerFast = false;
    static constexpr int kVec = 4;
    int vec_width() const { return aligned16(p) && ld % 4 == 0 ? 4 : 1; }
    __device__ void at_vec(const int64_t &r, const int64_t &c, float (&o)[4]) const {
        const float4 v = *(const float4 *)(p + r + c);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    typedef int64_t Row;
    typedef int64_t Col;
    const float *p;
    int64_t ld;
    __device__ Row row(int64_t r) const { return r; }
    __device__ Col col(int64_t c) const { return c * ld; }
    __device__ void advance_row(Row &r, int d) const { r += d; }
    __device__ void advance_col(Col &c, int d) const { c += d * ld; }
    __device__ float at(const Row &r, const Col &c) const { return p[r + c]; }
};
struct TransposedOnes {  // as Transposed, plus a row `ones_row` of 1.0 (folds the bias gradient in)
    static constexpr bool kInnerFast = false;
    static constexpr int kVec = 4;
    int vec_width() const { return ones_row < 0 && aligned16(p) && ld % 4 == 0 ? 4 : 1; }   // no ones row among wide loads
    __device__ void at_vec(const int64_t &r, const int64_t &c, float (&o)[4]) const {
        const float4 v = *(const float4 *)(p + r + c);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    typedef int64_t Row;     // -1 = the row of ones
    typedef int64_t Col;
    const float *p;
    int64_t ld;
    int64_t ones_row;
    __device__ Row row(int64_t r) const { return r == ones_row ? -1 : r; }
    __device__ Col col(int64_t c) const { return c * ld; }
    __device__ void advance_row(Row &, int) const {}
    __device__ void advance_col(Col &c, int d) const { c += d * ld; }
    __device__ float at(const Row &r, const Col &c) const { return r < 0 ? 1.f : p[r + c]; }
};

// ---- epilogues -------------------------------------------------------------------------
// tanh x = 1 - 2 / (2^(2 log2(e) x) + 1) on the hardware exp2 / rcp (1 ulp each): |error| <= ~1.5e-7 absolute, five
// vector instructions where libm's tanhf takes about forty — with the fp32 matrix products running on the same vector
// ALUs, a 64-wide epilogue of tanhf calls cost a K = 256 product 40 % on top of its multiply time.
__device__ inline float fast_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
// GEMM epilogues: fast_tanh's absolute error (~1.5e-7) is all of a small result's value, so below 2^-8 the odd series
// x - x^3 / 3 (error 2 x^5 / 15 < 2e-13) stands in: one compare and select
__device__ inline float tanh_act(float x) {
    const float t = fast_tanh(x);
    return fabsf(x) < 0.00390625f ? x * (1.0f - x * x * 0.33333334f) : t;
}
__device__ inline float apply_act(float x, int act) {
    if (act == 1) return tanh_act(x);
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
    typedef int64_t Row;
    float *C;
    int64_t ldc;
    const float *bias;
    int act;
    int accumulate;
    float scale = 1.f;       // of the product, before the bias (operands that load unscaled integers)
    __device__ Row row(int64_t m, int) const { return m * ldc; }
    __device__ void operator()(const Row &r, int n, float v) const {
        v *= scale;
        if (bias) v += bias[n];
        float *d = C + r + n;
        if (accumulate) v += *d;
        *d = apply_act(v, act);
    }
};
struct EpiActGrad {
    static constexpr bool kColSum = true;
    struct Row { int64_t c, h; };
    float *C;
    int64_t ldc;
    const float *H;
    int64_t ldh;
    int act;
    int accumulate;
    float *colsum;           // [row tiles][N] column sums of the stored values, or null
    __device__ Row row(int64_t m, int) const { return Row{m * ldc, m * ldh}; }
    __device__ float operator()(const Row &r, int n, float v) const {
        if (H) v *= act_grad(H[r.h + n], act);
        float *d = C + r.c + n;
        if (accumulate) v += *d;
        *d = v;
        return v;
    }
};
struct EpiSlab {
    typedef int64_t Row;
    float *C;
    int64_t ldc;
    int64_t zstride;
    float scale = 1.f;
    __device__ Row row(int64_t m, int z) const { return z * zstride + m * ldc; }
    __device__ void operator()(const Row &r, int n, float v) const { C[r + n] = v * scale; }
};

}  // namespace ppo2hip
