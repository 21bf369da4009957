// VALID NHWC convolutions of nature_cnn (baselines/ppo2/policies.py:6-16, conv builder
// baselines/a2c/utils.py:42-61) as implicit GEMMs on the fp32 MFMA core: the im2col matrix
// is never materialised, the loader functors compute its elements' addresses on the fly.
//   forward : Y[(b,oh,ow), co]  = relu( sum_k col[(b,oh,ow), k] W[k, co] + bias ),  k = (kh, kw, ci)
//   wgrad   : dW[k, co]         = sum_m col[m, k] dY[m, co]   (+ a ones row -> bias gradient), slabs over m
//   dgrad   : dX[(b,ih,iw), ci] = relu'(X) * sum_{kh,kw,co} dY[b, (ih-kh)/s, (iw-kw)/s, co] W[kh,kw,ci,co]
// uint8 observations are scaled by 1/255 inside the loader (tf.cast(...)/255., policies.py:10).
#include <cstdlib>

#include "gemm_core.hpp"
#include "kernels.hpp"

namespace ppo2hip {

struct ConvGeom {
    int IH, IW, Cin, KH, KW, S, OH, OW, Cout;
};

// pixel (b, oh, ow) of output row m, and the input offset of its patch's first element
struct PatchRow {
    int b, oh, ow;
    int64_t base;
};
__device__ inline PatchRow patch_row(const ConvGeom &g, int64_t m) {
    const int ohw = g.OH * g.OW;
    PatchRow p;
    p.b = (int)(m / ohw);
    const int r = (int)(m - (int64_t)p.b * ohw);
    p.oh = r / g.OW;
    p.ow = r - p.oh * g.OW;
    p.base = (((int64_t)p.b * g.IH + p.oh * g.S) * g.IW + p.ow * g.S) * g.Cin;
    return p;
}
// tap k = (kh, kw, ci) inside a patch: its offset from the patch's first element, and where it stands in its
// kernel row (kw, ci run contiguously), so that moving on by d taps needs no division
struct TapOff {
    int off, rest;
};
__device__ inline TapOff tap_offset(const ConvGeom &g, int64_t k) {
    const int kc = g.KW * g.Cin;
    const int kh = (int)k / kc;
    const int rest = (int)k - kh * kc;
    return TapOff{kh * g.IW * g.Cin + rest, rest};
}
__device__ inline void tap_advance(const ConvGeom &g, TapOff &t, int d) {
    const int kc = g.KW * g.Cin;
    t.rest += d;
    t.off += d;
    if (d <= kc) {          // at most one kernel row further on: no per-lane loop (the test is uniform)
        const bool wrap = t.rest >= kc;
        t.rest -= wrap ? kc : 0;
        t.off += wrap ? g.IW * g.Cin - kc : 0;
        return;
    }
    while (t.rest >= kc) {
        t.rest -= kc;
        t.off += g.IW * g.Cin - kc;
    }
}

// uint8 observations load as the integers they are; the 1/255 of the network's input scaling
// (baselines/common/models.py:22) multiplies the finished product in the epilogue, once per output element instead of
// once per loaded tap.
constexpr float kU8Scale = 1.0f / 255.0f;

template <bool U8>
struct Im2col {   // A(m, k)
    static constexpr bool kInnerFast = true;
    typedef int64_t Row;
    typedef TapOff Col;
    const void *x;
    ConvGeom g;
    __device__ Row row(int64_t m) const { return patch_row(g, m).base; }
    __device__ Col col(int64_t k) const { return tap_offset(g, k); }
    __device__ void advance_col(Col &c, int d) const { tap_advance(g, c, d); }
    __device__ float at(const Row &r, const Col &c) const { return at_off(r, c.off); }
    __device__ float at_off(int64_t r, int off) const {
        if (U8) return (float)((const uint8_t *)x)[r + off];
        return ((const float *)x)[r + off];
    }
    // one load: eight uint8 taps (8 bytes), or four float taps (16 bytes), of one kernel row (kw, ci run contiguously).
    // Eight, not sixteen, bytes: a 128-row tile is then 256 loads per slab, one for every thread, where sixteen-byte
    // loads would leave the conversion and the LDS writes to half of the workgroup's waves.
    static constexpr int kVec = U8 ? 8 : 4;
    int vec_width() const {
        const int esz = U8 ? 1 : 4, bytes = U8 ? 8 : 16;
        const bool ok = aligned16(x) && (g.KW * g.Cin) % kVec == 0 && (g.Cin * esz) % 4 == 0 &&
                        (g.IW * g.Cin * esz) % bytes == 0 && (g.S * g.Cin * esz) % bytes == 0;
        return ok ? kVec : 1;
    }
    __device__ void at_vec(const Row &r, const Col &c, float (&o)[kVec]) const { at_vec_off(r, c.off, o); }
    __device__ void at_vec_off(int64_t r, int c, float (&o)[kVec]) const {
        if constexpr (U8) {
            const uint2 v = *(const uint2 *)((const uint8_t *)x + r + c);
            const uint32_t w[2] = {v.x, v.y};
#pragma unroll
            for (int q = 0; q < 2; q++)
#pragma unroll
                for (int j = 0; j < 4; j++) o[4 * q + j] = (float)((w[q] >> (8 * j)) & 0xffu);
        } else {
            const float4 v = *(const float4 *)((const float *)x + r + c);
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
    }
};

template <bool U8>
struct Im2colT {   // A(i = k, kk = m), row `ones_row` = 1 after the epilogue's scaling
    static constexpr bool kInnerFast = false;
    typedef int Row;          // tap offset, or -1 for the row of ones
    typedef PatchRow Col;
    Im2col<U8> c;
    int64_t ones_row;
    __device__ Row row(int64_t i) const { return i == ones_row ? -1 : tap_offset(c.g, i).off; }
    __device__ Col col(int64_t m) const { return patch_row(c.g, m); }
    __device__ void advance_col(Col &p, int d) const {      // d output pixels further on
        const ConvGeom &g = c.g;
        // d = dq OW + dr pixels on: dq rows and dr columns, then at most one carry each way (uniform tests, no per-lane
        // loops); the patch offset moves by sums of wave-uniform constants — no multiplication per step
        const int dq = d / g.OW, dr = d - dq * g.OW;
        if (dq + 1 <= g.OH) {
            const int step = (dq * g.S * g.IW + dr * g.S) * g.Cin;             // dq rows down, dr columns right
            const int carry_w = (g.S * g.IW - g.OW * g.S) * g.Cin;             // back OW columns, one row down
            const int carry_h = (g.IH * g.IW - g.OH * g.S * g.IW) * g.Cin;     // back OH rows, one image on
            p.ow += dr;
            p.oh += dq;
            const bool wc = p.ow >= g.OW;
            p.ow -= wc ? g.OW : 0;
            p.oh += wc ? 1 : 0;
            const bool wr = p.oh >= g.OH;
            p.oh -= wr ? g.OH : 0;
            p.b += wr ? 1 : 0;
            p.base += step + (wc ? carry_w : 0) + (wr ? carry_h : 0);
            return;
        }
        p.ow += d;
        while (p.ow >= g.OW) { p.ow -= g.OW; p.oh++; }
        while (p.oh >= g.OH) { p.oh -= g.OH; p.b++; }
        p.base = (((int64_t)p.b * g.IH + p.oh * g.S) * g.IW + p.ow * g.S) * g.Cin;
    }
    __device__ float at(const Row &r, const Col &p) const { return r < 0 ? (U8 ? 255.f : 1.f) : c.at_off(p.base, r); }
    static constexpr int kVec = Im2col<U8>::kVec;
    int vec_width() const { return ones_row < 0 ? c.vec_width() : 1; }     // no row of ones among wide loads
    __device__ void at_vec(const Row &r, const Col &p, float (&o)[kVec]) const { c.at_vec_off(p.base, r, o); }
};

struct PixelRow {   // pixel (ih, iw) of image b in the dgrad row space; bbase = offset of dY[b, 0, 0, 0]
    int64_t bbase;
    int ih, iw;
};
struct TapCol {
    int kh, kw, co;
};
__device__ inline TapCol tap_col(int KW, int Cout, int64_t k) {
    const int kc = KW * Cout;
    TapCol t;
    t.kh = (int)k / kc;
    const int r2 = (int)k - t.kh * kc;
    t.kw = r2 / Cout;
    t.co = r2 - t.kw * Cout;
    return t;
}

// d taps further on in (kh, kw, co) order
__device__ inline void tap_col_advance(int KW, int Cout, TapCol &t, int d) {
    t.co += d;
    if (d <= Cout) {        // at most one tap further on: no per-lane loops (the test is uniform)
        const bool wc = t.co >= Cout;
        t.co -= wc ? Cout : 0;
        t.kw += wc ? 1 : 0;
        const bool wk = t.kw >= KW;
        t.kw -= wk ? KW : 0;
        t.kh += wk ? 1 : 0;
        return;
    }
    while (t.co >= Cout) { t.co -= Cout; t.kw++; }
    while (t.kw >= KW) { t.kw -= KW; t.kh++; }
}

struct DgradCol {   // A(m = (b,ih,iw), k = (kh,kw,co)) = dY[b, (ih-kh)/s, (iw-kw)/s, co] or 0
    static constexpr bool kInnerFast = true;
    typedef PixelRow Row;
    typedef TapCol Col;
    const float *dy;
    ConvGeom g;
    __device__ Row row(int64_t m) const {
        const int ihw = g.IH * g.IW;
        const int b = (int)(m / ihw);
        const int r = (int)(m - (int64_t)b * ihw);
        Row p;
        p.ih = r / g.IW;
        p.iw = r - p.ih * g.IW;
        p.bbase = (int64_t)b * g.OH * g.OW * g.Cout;
        return p;
    }
    __device__ Col col(int64_t k) const { return tap_col(g.KW, g.Cout, k); }
    __device__ void advance_col(Col &t, int d) const { tap_col_advance(g.KW, g.Cout, t, d); }
    __device__ float at(const Row &p, const Col &t) const {
        const int th = p.ih - t.kh, tw = p.iw - t.kw;
        if (th < 0 || tw < 0) return 0.f;
        const int oh = th / g.S, ow = tw / g.S;
        if (oh * g.S != th || ow * g.S != tw || oh >= g.OH || ow >= g.OW) return 0.f;
        return dy[p.bbase + (int64_t)(oh * g.OW + ow) * g.Cout + t.co];
    }
    static constexpr int kVec = 4;      // four output channels of one tap
    int vec_width() const { return aligned16(dy) && g.Cout % 4 == 0 ? 4 : 1; }
    __device__ void at_vec(const Row &p, const Col &t, float (&o)[4]) const {
        const int th = p.ih - t.kh, tw = p.iw - t.kw;
        const int oh = th / g.S, ow = tw / g.S;
        if (th < 0 || tw < 0 || oh * g.S != th || ow * g.S != tw || oh >= g.OH || ow >= g.OW) {
            o[0] = o[1] = o[2] = o[3] = 0.f;
            return;
        }
        const float4 v = *(const float4 *)(dy + p.bbase + (int64_t)(oh * g.OW + ow) * g.Cout + t.co);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};

struct DgradW {   // B(k = (kh,kw,co), n = ci) = W[kh, kw, ci, co]
    static constexpr bool kInnerFast = false;
    static constexpr int kVec = 4;      // four output channels of one tap and input channel
    int vec_width() const { return aligned16(w) && g.Cout % 4 == 0 ? 4 : 1; }
    __device__ void at_vec(const TapCol &t, const int &c, float (&o)[4]) const {
        const float4 v = *(const float4 *)(w + (t.kh * g.KW + t.kw) * g.Cin * g.Cout + t.co + c);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    typedef TapCol Row;
    typedef int Col;
    const float *w;
    ConvGeom g;
    __device__ Row row(int64_t k) const { return tap_col(g.KW, g.Cout, k); }
    __device__ void advance_row(Row &t, int d) const { tap_col_advance(g.KW, g.Cout, t, d); }
    __device__ Col col(int64_t n) const { return (int)n * g.Cout; }
    __device__ float at(const Row &t, const Col &c) const { return w[(t.kh * g.KW + t.kw) * g.Cin * g.Cout + t.co + c]; }
};

// Input gradient without the border's zeros.  A border pixel of the input meets only some of the taps (for the 3x3
// stride-1 layer on 9x9 inputs, 441 of the 729 pixel-tap pairs exist; for a parity class of the 4x4 stride-2 layer 81 %),
// and in image order every tile of rows holds pixels of all kinds, so the plain product multiplies the missing pairs as
// zeros.  Here rows are ordered
//   m = (bt * npos + p) * 128 + bl,   image b = 128 bt + bl, pixel p = (ih', iw') of the class, npos = IHp IWp,
// so a 128-row tile is one pixel of 128 images: its taps kh' in [kh0, kh1], kw' in [kw0, kw1] are the same for every row,
// the k loop runs over exactly those (k' = ((kh' - kh0) nkw + (kw' - kw0)) Cout + co), no load needs a bounds test, and
// consecutive tiles walk the pixels of the same 128 images (their dY stays in L2).  Needs B a multiple of 128.
// A stride-1 layer is the single class ph = pw = 0, S = 1, IHp = IH, KHp = KH; for a strided layer (see conv_dgrad) input
// pixel (S ih' + ph, S iw' + pw) meets taps (ph + S kh', pw + S kw') at output pixel (ih' - kh', iw' - kw').
// The dY operand of a class does not depend on (ph, pw) at all, only the weights do: when the classes have one size
// (IH, IW multiples of S) they run as ONE product with the classes side by side along n = (ph, pw, ci), `merged`, which
// reads and stages every dY element once for S x S Cin output columns instead of once per class for Cin.
struct PixelMajor {
    ConvGeom g;
    int ph, pw, IHp, IWp, KHp, KWp;
    int merged;              // columns n = ((ph, pw), ci) for all S x S classes (ph, pw above are 0 then)
    int kh0, kw0, nkw;       // set per tile
    __device__ int taps_of_tile(int64_t m0) {
        const int p = (int)((m0 / 128) % (IHp * IWp));
        const int ih = p / IWp, iw = p - ih * IWp;
        kh0 = ih - (g.OH - 1) > 0 ? ih - (g.OH - 1) : 0;
        kw0 = iw - (g.OW - 1) > 0 ? iw - (g.OW - 1) : 0;
        const int kh1 = ih < KHp - 1 ? ih : KHp - 1, kw1 = iw < KWp - 1 ? iw : KWp - 1;
        nkw = kw1 - kw0 + 1;
        return (kh1 - kh0 + 1) * nkw;
    }
    // row m -> (image, pixel of the class)
    __device__ void decode(int64_t m, int64_t &b, int &ih, int &iw) const {
        const int npos = IHp * IWp;
        const int64_t bt = m / ((int64_t)npos * 128);
        const int rem = (int)(m - bt * npos * 128);
        const int p = rem >> 7;
        b = bt * 128 + (rem & 127);
        ih = p / IWp;
        iw = p - ih * IWp;
    }
};
struct DgradColPM {   // A(m, k')
    bool full_slabs() const { return q.g.Cout % GEMM_BK == 0; }      // a tile's k range is whole taps of Cout terms
    static constexpr bool kInnerFast = true;
    static constexpr bool kTileK = true;
    typedef PixelRow Row;
    typedef TapCol Col;       // (kh' - kh0, kw' - kw0, co)
    const float *dy;
    PixelMajor q;
    __device__ void begin_tile(int64_t m0, int64_t &kbeg, int64_t &kend) {
        kbeg = 0;
        kend = (int64_t)q.taps_of_tile(m0) * q.g.Cout;
    }
    __device__ Row row(int64_t m) const {
        Row r;
        int64_t b;
        q.decode(m, b, r.ih, r.iw);
        r.bbase = b * q.g.OH * q.g.OW * q.g.Cout;
        return r;
    }
    __device__ Col col(int64_t k) const { return tap_col(q.nkw, q.g.Cout, k); }
    __device__ void advance_col(Col &t, int d) const { tap_col_advance(q.nkw, q.g.Cout, t, d); }
    __device__ int64_t offset(const Row &p, const Col &t) const {
        return p.bbase + (int64_t)((p.ih - q.kh0 - t.kh) * q.g.OW + (p.iw - q.kw0 - t.kw)) * q.g.Cout + t.co;
    }
    __device__ float at(const Row &p, const Col &t) const { return dy[offset(p, t)]; }
    static constexpr int kVec = 4;
    int vec_width() const { return aligned16(dy) && q.g.Cout % 4 == 0 ? 4 : 1; }
    __device__ void at_vec(const Row &p, const Col &t, float (&o)[4]) const {
        const float4 v = *(const float4 *)(dy + offset(p, t));
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
struct DgradWPM {   // B(k', n = ci) = W[ph + S (kh0 + kh'), pw + S (kw0 + kw'), ci, co]
    static constexpr bool kInnerFast = false;
    static constexpr bool kTileK = true;
    static constexpr int kVec = 4;
    int vec_width() const { return aligned16(w) && q.g.Cout % 4 == 0 ? 4 : 1; }
    typedef TapCol Row;
    typedef int Col;
    const float *w;
    PixelMajor q;
    __device__ void begin_tile(int64_t m0) { q.taps_of_tile(m0); }
    __device__ Row row(int64_t k) const { return tap_col(q.nkw, q.g.Cout, k); }
    __device__ void advance_row(Row &t, int d) const { tap_col_advance(q.nkw, q.g.Cout, t, d); }
    // what a column owes to the offset: its class's first tap and its input channel
    __device__ Col col(int64_t n) const {
        const int cls = q.merged ? (int)n / q.g.Cin : 0, ci = (int)n - cls * q.g.Cin;
        const int ph = q.merged ? cls / q.g.S : q.ph, pw = q.merged ? cls - (cls / q.g.S) * q.g.S : q.pw;
        return ((ph * q.g.KW + pw) * q.g.Cin + ci) * q.g.Cout;
    }
    __device__ int offset(const Row &t, const Col &c) const {
        return (q.g.S * (q.kh0 + t.kh) * q.g.KW + q.g.S * (q.kw0 + t.kw)) * q.g.Cin * q.g.Cout + t.co + c;
    }
    __device__ float at(const Row &t, const Col &c) const { return w[offset(t, c)]; }
    __device__ void at_vec(const Row &t, const Col &c, float (&o)[4]) const {
        const float4 v = *(const float4 *)(w + offset(t, c));
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
struct EpiActGradPM {   // row m of the pixel-major order -> pixel (b, S ih' + ph, S iw' + pw) of dX, times relu'(X)
    static constexpr bool kColSum = true;
    static constexpr bool kColCtx = true;
    typedef int64_t Row;
    float *dX;
    const float *Xact;
    PixelMajor q;
    float *colsum;           // [row tiles][N] column sums of the stored values, or null
    const uint32_t *bits;    // packed (Xact > 0), read in place of Xact when set
    __device__ Row row(int64_t m, int) const {
        int64_t b;
        int ih, iw;
        q.decode(m, b, ih, iw);
        return ((b * q.g.IH + q.g.S * ih + q.ph) * q.g.IW + q.g.S * iw + q.pw) * q.g.Cin;
    }
    // column n of the merged product = ((ph, pw), ci): the pixel of its class
    __device__ int64_t col(int n) const {
        if (!q.merged) return n;
        const int cls = n / q.g.Cin, ci = n - cls * q.g.Cin, ph = cls / q.g.S, pw = cls - ph * q.g.S;
        return (int64_t)(ph * q.g.IW + pw) * q.g.Cin + ci;
    }
    __device__ float operator()(const Row &r, int64_t cc, float v) const {
        const int64_t o = r + cc;
        const bool on = bits ? ((bits[o >> 5] >> (o & 31)) & 1u) != 0u : Xact[o] > 0.f;
        v = on ? v : 0.f;
        dX[o] = v;
        return v;
    }
};

static ConvGeom geom(const ConvShape &c) {
    return ConvGeom{c.IH, c.IW, c.Cin, c.KH, c.KW, c.stride, c.OH, c.OW, c.Cout};
}

int conv_forward(const void *X, bool x_u8, const float *W, const float *b, float *Y, const ConvShape &c,
                 hipStream_t s, const int32_t *rows, uint32_t *relu_bits) {
    const int64_t M = (int64_t)c.B * c.OH * c.OW;
    const int K = c.KH * c.KW * c.Cin;
    if (x_u8 && conv_u8_forward_ok(X, c)) return conv_u8_forward(X, W, b, Y, c, s, rows, relu_bits);
    if (rows || relu_bits) { set_error("conv_forward: a frame list or rectifier bits need the uint8 kernels"); return PPO2HIP_EINVAL; }
    RowMajor Bm{W, c.Cout};
    EpiBiasAct ep{Y, c.Cout, b, ACT_RELU, 0, x_u8 ? kU8Scale : 1.f};
    if (x_u8) return launch_gemm(Im2col<true>{X, geom(c)}, Bm, ep, M, c.Cout, K, 1, s, "conv_forward");
    return launch_gemm<true>(Im2col<false>{X, geom(c)}, Bm, ep, M, c.Cout, K, 1, s, "conv_forward");
}

int conv_wgrad_max_slabs(const ConvShape &c) {
    const int K = c.KH * c.KW * c.Cin;
    return wgrad_max_slabs((int64_t)c.B * c.OH * c.OW, K, c.Cout);
}

int conv_wgrad_slabs(const void *X, bool x_u8, const float *dY, const ConvShape &c, float *slabs, int max_slabs,
                     int *nslabs, hipStream_t s, bool bias_rows, const int32_t *frame_rows) {
    const int64_t M = (int64_t)c.B * c.OH * c.OW;
    const int K = c.KH * c.KW * c.Cin;
    if (x_u8 && conv_u8_wgrad_ok(X, dY, c)) return conv_u8_wgrad_slabs(X, dY, c, slabs, max_slabs, nslabs, s, bias_rows, frame_rows);
    if (frame_rows) { set_error("conv_wgrad: a frame list needs the uint8 kernels"); return PPO2HIP_EINVAL; }
    int n = conv_wgrad_max_slabs(c);
    if (n > max_slabs) n = max_slabs;
    *nslabs = n;
    RowMajor Bm{dY, c.Cout};
    EpiSlab ep{slabs, c.Cout, (int64_t)(K + 1) * c.Cout, x_u8 ? kU8Scale : 1.f};
    const bool sep = wgrad_bias_separately(K);         // ones-row only where it does not stand in the way of wide loads
    const int64_t ones = sep ? -1 : K, rows = sep ? K : K + 1;
    int rc;
    if (x_u8) rc = launch_gemm(Im2colT<true>{Im2col<true>{X, geom(c)}, ones}, Bm, ep, rows, c.Cout, M, n, s, "conv_wgrad");
    else rc = launch_gemm<true>(Im2colT<false>{Im2col<false>{X, geom(c)}, ones}, Bm, ep, rows, c.Cout, M, n, s, "conv_wgrad");
    if (rc || !sep || !bias_rows) return rc;
    return slab_bias_rows(dY, c.Cout, M, K, c.Cout, slabs, n, s);
}

// Strided convolutions: an input pixel (ih, iw) only meets the taps with kh = ih, kw = iw (mod S), so the input
// gradient splits into S x S products, one per parity class of (ih, iw), each over (KH/S)(KW/S) Cout terms
// instead of KH KW Cout of which all but 1/S^2 are zero.
struct DgradColP {   // A(m' = (b, ih', iw'), k' = (kh', kw', co)),  ih = S ih' + ph, kh = ph + S kh'  =>  oh = ih' - kh'
    static constexpr bool kInnerFast = true;
    typedef PixelRow Row;     // (ih', iw')
    typedef TapCol Col;       // (kh', kw', co)
    const float *dy;
    ConvGeom g;
    int ph, pw, IHp, IWp, KWp;
    __device__ Row row(int64_t m) const {
        const int ihw = IHp * IWp;
        const int b = (int)(m / ihw);
        const int r = (int)(m - (int64_t)b * ihw);
        Row p;
        p.ih = r / IWp;
        p.iw = r - p.ih * IWp;
        p.bbase = (int64_t)b * g.OH * g.OW * g.Cout;
        return p;
    }
    __device__ Col col(int64_t k) const { return tap_col(KWp, g.Cout, k); }
    __device__ void advance_col(Col &t, int d) const { tap_col_advance(KWp, g.Cout, t, d); }
    __device__ float at(const Row &p, const Col &t) const {
        const int oh = p.ih - t.kh, ow = p.iw - t.kw;
        if (oh < 0 || ow < 0 || oh >= g.OH || ow >= g.OW) return 0.f;
        return dy[p.bbase + (int64_t)(oh * g.OW + ow) * g.Cout + t.co];
    }
    static constexpr int kVec = 4;
    int vec_width() const { return aligned16(dy) && g.Cout % 4 == 0 ? 4 : 1; }
    __device__ void at_vec(const Row &p, const Col &t, float (&o)[4]) const {
        const int oh = p.ih - t.kh, ow = p.iw - t.kw;
        if (oh < 0 || ow < 0 || oh >= g.OH || ow >= g.OW) {
            o[0] = o[1] = o[2] = o[3] = 0.f;
            return;
        }
        const float4 v = *(const float4 *)(dy + p.bbase + (int64_t)(oh * g.OW + ow) * g.Cout + t.co);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
};
struct DgradWP {   // B(k' = (kh', kw', co), n = ci) = W[ph + S kh', pw + S kw', ci, co]
    static constexpr bool kInnerFast = false;
    static constexpr int kVec = 4;
    int vec_width() const { return aligned16(w) && g.Cout % 4 == 0 ? 4 : 1; }
    __device__ void at_vec(const TapCol &t, const int &c, float (&o)[4]) const {
        const float4 v = *(const float4 *)(w + ((ph + g.S * t.kh) * g.KW + pw + g.S * t.kw) * g.Cin * g.Cout + t.co + c);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    typedef TapCol Row;
    typedef int Col;
    const float *w;
    ConvGeom g;
    int ph, pw, KWp;
    __device__ Row row(int64_t k) const { return tap_col(KWp, g.Cout, k); }
    __device__ void advance_row(Row &t, int d) const { tap_col_advance(KWp, g.Cout, t, d); }
    __device__ Col col(int64_t n) const { return (int)n * g.Cout; }
    __device__ float at(const Row &t, const Col &c) const {
        return w[((ph + g.S * t.kh) * g.KW + pw + g.S * t.kw) * g.Cin * g.Cout + t.co + c];
    }
};
struct EpiActGradP {   // row m' of a parity class -> pixel (b, S ih' + ph, S iw' + pw) of dX, times relu'(X)
    typedef int64_t Row;   // offset of that pixel's first channel
    float *dX;
    const float *Xact;
    ConvGeom g;
    int ph, pw, IHp, IWp;
    __device__ Row row(int64_t m, int) const {
        const int ihw = IHp * IWp;
        const int64_t b = m / ihw;
        const int r = (int)(m - b * ihw);
        const int ihp = r / IWp, iwp = r - ihp * IWp;
        return ((b * g.IH + g.S * ihp + ph) * g.IW + g.S * iwp + pw) * g.Cin;
    }
    __device__ void operator()(const Row &r, int n, float v) const { dX[r + n] = Xact[r + n] > 0.f ? v : 0.f; }
};

// pixel-major tiles: whole groups of 128 images, and enough rows that launch_gemm takes its 128-row tiles;
// PPO2HIP_DGRAD_PLAIN=1 keeps the image-major products
static bool pixel_major_ok(const ConvShape &c, int64_t M) {
    return c.B % 128 == 0 && M >= 1024 && !getenv("PPO2HIP_DGRAD_PLAIN");
}

// rows of Cin column sums conv_dgrad may hand back (row tiles x parity classes of the largest route)
int64_t conv_dgrad_colsum_rows(const ConvShape &c) {
    const int S = c.stride > 0 ? c.stride : 1;
    return ceil_div((int64_t)c.B * ceil_div(c.IH, S) * ceil_div(c.IW, S), (int64_t)64) * S * S;
}

int conv_dgrad(const float *dY, const float *W, const float *Xact, float *dX, const ConvShape &c, hipStream_t s,
               float *colsum_rows, int *nrows, const uint32_t *relu_bits) {
    if (nrows) *nrows = 0;
    if (c.Cin != 32) relu_bits = nullptr;       // one word per pixel: the packed bits line up with the tensor for 32 channels only
    if (c.stride > 1 && c.KH % c.stride == 0 && c.KW % c.stride == 0) {
        const int S = c.stride, KHp = c.KH / S, KWp = c.KW / S;
        if (c.IH % S == 0 && c.IW % S == 0 && pixel_major_ok(c, (int64_t)c.B * (c.IH / S) * (c.IW / S)) && KHp <= c.IH / S &&
            KWp <= c.IW / S && !getenv("PPO2HIP_DGRAD_PER_CLASS")) {
            // all S x S classes in one product: n = ((ph, pw), ci)
            const int IHp = c.IH / S, IWp = c.IW / S;
            const PixelMajor q{geom(c), 0, 0, IHp, IWp, KHp, KWp, 1, 0, 0, 0};
            const int64_t Mq = (int64_t)c.B * IHp * IWp;
            if (nrows && colsum_rows) *nrows = (int)(gemm_row_tiles(Mq, S * S * c.Cin, 1) * S * S);      // [tile][class][ci]
            return launch_gemm<true>(DgradColPM{dY, q}, DgradWPM{W, q}, EpiActGradPM{dX, Xact, q, colsum_rows, relu_bits}, Mq, S * S * c.Cin,
                               KHp * KWp * c.Cout, 1, s, "conv_dgrad");
        }
        for (int ph = 0; ph < S; ph++)
            for (int pw = 0; pw < S; pw++) {
                const int IHp = (c.IH - ph + S - 1) / S, IWp = (c.IW - pw + S - 1) / S;
                if (IHp <= 0 || IWp <= 0) continue;
                const int64_t M = (int64_t)c.B * IHp * IWp;
                const int K = KHp * KWp * c.Cout;
                if (pixel_major_ok(c, M) && KHp <= IHp && KWp <= IWp) {
                    const PixelMajor q{geom(c), ph, pw, IHp, IWp, KHp, KWp, 0, 0, 0, 0};
                    int rc = launch_gemm(DgradColPM{dY, q}, DgradWPM{W, q}, EpiActGradPM{dX, Xact, q, nullptr, relu_bits}, M, c.Cin, K, 1, s, "conv_dgrad");
                    if (rc) return rc;
                    continue;
                }
                int rc = launch_gemm(DgradColP{dY, geom(c), ph, pw, IHp, IWp, KWp}, DgradWP{W, geom(c), ph, pw, KWp},
                                     EpiActGradP{dX, Xact, geom(c), ph, pw, IHp, IWp}, M, c.Cin, K, 1, s, "conv_dgrad");
                if (rc) return rc;
            }
        return 0;
    }
    const int64_t M = (int64_t)c.B * c.IH * c.IW;
    const int K = c.KH * c.KW * c.Cout;
    if (c.stride == 1 && pixel_major_ok(c, M) && c.KH <= c.IH && c.KW <= c.IW) {
        const PixelMajor q{geom(c), 0, 0, c.IH, c.IW, c.KH, c.KW, 0, 0, 0, 0};
        if (nrows && colsum_rows) *nrows = (int)gemm_row_tiles(M, c.Cin, 1);
        return launch_gemm<true>(DgradColPM{dY, q}, DgradWPM{W, q}, EpiActGradPM{dX, Xact, q, colsum_rows, relu_bits}, M, c.Cin, K, 1, s, "conv_dgrad");
    }
    EpiActGrad ep{dX, c.Cin, Xact, c.Cin, ACT_RELU, 0, nullptr};
    return launch_gemm(DgradCol{dY, geom(c)}, DgradW{W, geom(c)}, ep, M, c.Cin, K, 1, s, "conv_dgrad");
}

}  // namespace ppo2hip
