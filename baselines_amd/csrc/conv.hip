// VALID NHWC convolutions of nature_cnn (baselines/ppo2/policies.py:6-16, conv builder
// baselines/a2c/utils.py:42-61) as implicit GEMMs on the fp32 MFMA core: the im2col matrix
// is never materialised, the loader functors compute its elements' addresses on the fly.
//   forward : Y[(b,oh,ow), co]  = relu( sum_k col[(b,oh,ow), k] W[k, co] + bias ),  k = (kh, kw, ci)
//   wgrad   : dW[k, co]         = sum_m col[m, k] dY[m, co]   (+ a ones row -> bias gradient), slabs over m
//   dgrad   : dX[(b,ih,iw), ci] = relu'(X) * sum_{kh,kw,co} dY[b, (ih-kh)/s, (iw-kw)/s, co] W[kh,kw,ci,co]
// uint8 observations are scaled by 1/255 inside the loader (tf.cast(...)/255., policies.py:10).
#include "gemm_core.hpp"
#include "kernels.hpp"

namespace ppo2hip {

struct ConvGeom {
    int IH, IW, Cin, KH, KW, S, OH, OW, Cout;
};

template <bool U8>
struct Im2col {   // A(m, k)
    static constexpr bool kInnerFast = true;
    const void *x;
    ConvGeom g;
    __device__ float operator()(int64_t m, int64_t k) const {
        const int ohw = g.OH * g.OW;
        const int64_t b = m / ohw;
        const int r = (int)(m - b * ohw);
        const int oh = r / g.OW, ow = r - oh * g.OW;
        const int kc = g.KW * g.Cin;
        const int kh = (int)k / kc;
        const int r2 = (int)k - kh * kc;
        const int kw = r2 / g.Cin, ci = r2 - kw * g.Cin;
        const int64_t off = ((b * g.IH + oh * g.S + kh) * g.IW + ow * g.S + kw) * g.Cin + ci;
        if (U8) return (float)((const uint8_t *)x)[off] / 255.0f;
        return ((const float *)x)[off];
    }
};

template <bool U8>
struct Im2colT {   // A(i = k, kk = m), row `ones_row` = 1
    static constexpr bool kInnerFast = false;
    Im2col<U8> c;
    int64_t ones_row;
    __device__ float operator()(int64_t i, int64_t m) const { return i == ones_row ? 1.f : c(m, i); }
};

struct DgradCol {   // A(m = (b,ih,iw), k = (kh,kw,co)) = dY[b, (ih-kh)/s, (iw-kw)/s, co] or 0
    static constexpr bool kInnerFast = true;
    const float *dy;
    ConvGeom g;
    __device__ float operator()(int64_t m, int64_t k) const {
        const int ihw = g.IH * g.IW;
        const int64_t b = m / ihw;
        const int r = (int)(m - b * ihw);
        const int ih = r / g.IW, iw = r - ih * g.IW;
        const int kc = g.KW * g.Cout;
        const int kh = (int)k / kc;
        const int r2 = (int)k - kh * kc;
        const int kw = r2 / g.Cout, co = r2 - kw * g.Cout;
        const int th = ih - kh, tw = iw - kw;
        if (th < 0 || tw < 0) return 0.f;
        const int oh = th / g.S, ow = tw / g.S;
        if (oh * g.S != th || ow * g.S != tw || oh >= g.OH || ow >= g.OW) return 0.f;
        return dy[((b * g.OH + oh) * g.OW + ow) * g.Cout + co];
    }
};

struct DgradW {   // B(k = (kh,kw,co), n = ci) = W[kh, kw, ci, co]
    static constexpr bool kInnerFast = false;
    const float *w;
    ConvGeom g;
    __device__ float operator()(int64_t k, int64_t n) const {
        const int kc = g.KW * g.Cout;
        const int kh = (int)k / kc;
        const int r2 = (int)k - kh * kc;
        const int kw = r2 / g.Cout, co = r2 - kw * g.Cout;
        return w[((int64_t)(kh * g.KW + kw) * g.Cin + n) * g.Cout + co];
    }
};

static ConvGeom geom(const ConvShape &c) {
    return ConvGeom{c.IH, c.IW, c.Cin, c.KH, c.KW, c.stride, c.OH, c.OW, c.Cout};
}

int conv_forward(const void *X, bool x_u8, const float *W, const float *b, float *Y, const ConvShape &c,
                 hipStream_t s) {
    const int64_t M = (int64_t)c.B * c.OH * c.OW;
    const int K = c.KH * c.KW * c.Cin;
    RowMajor Bm{W, c.Cout};
    EpiBiasAct ep{Y, c.Cout, b, ACT_RELU, 0};
    if (x_u8) return launch_gemm(Im2col<true>{X, geom(c)}, Bm, ep, M, c.Cout, K, 1, s, "conv_forward");
    return launch_gemm(Im2col<false>{X, geom(c)}, Bm, ep, M, c.Cout, K, 1, s, "conv_forward");
}

int conv_wgrad_max_slabs(const ConvShape &c) {
    const int K = c.KH * c.KW * c.Cin;
    return wgrad_max_slabs((int64_t)c.B * c.OH * c.OW, K, c.Cout);
}

int conv_wgrad_slabs(const void *X, bool x_u8, const float *dY, const ConvShape &c, float *slabs, int max_slabs,
                     int *nslabs, hipStream_t s) {
    const int64_t M = (int64_t)c.B * c.OH * c.OW;
    const int K = c.KH * c.KW * c.Cin;
    int n = conv_wgrad_max_slabs(c);
    if (n > max_slabs) n = max_slabs;
    *nslabs = n;
    RowMajor Bm{dY, c.Cout};
    EpiSlab ep{slabs, c.Cout, (int64_t)(K + 1) * c.Cout};
    if (x_u8) return launch_gemm(Im2colT<true>{Im2col<true>{X, geom(c)}, K}, Bm, ep, K + 1, c.Cout, M, n, s, "conv_wgrad");
    return launch_gemm(Im2colT<false>{Im2col<false>{X, geom(c)}, K}, Bm, ep, K + 1, c.Cout, M, n, s, "conv_wgrad");
}

int conv_dgrad(const float *dY, const float *W, const float *Xact, float *dX, const ConvShape &c, hipStream_t s) {
    const int64_t M = (int64_t)c.B * c.IH * c.IW;
    const int K = c.KH * c.KW * c.Cout;
    EpiActGrad ep{dX, c.Cin, Xact, c.Cin, ACT_RELU, 0};
    return launch_gemm(DgradCol{dY, geom(c)}, DgradW{W, geom(c)}, ep, M, c.Cin, K, 1, s, "conv_dgrad");
}

}  // namespace ppo2hip
