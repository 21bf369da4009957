// Host-side orchestration: parameter arena layout, policy forward, and the minibatch
// gradient (gather -> forward -> loss -> backward) as a fixed sequence of kernel launches
// on the caller's stream.  No device memory is allocated here; scratch is carved from the
// caller's workspace.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "rowgather.hpp"

namespace ppo2hip {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct Var {
    std::string name;
    int64_t off;
    int64_t shape[4];
    int64_t size;
};
struct Layout {
    std::vector<Var> vars;
    int64_t total = 0;
    void add(const char *name, int64_t d0, int64_t d1 = 0, int64_t d2 = 0, int64_t d3 = 0) {
        Var v;
        v.name = name;
        v.off = total;
        v.shape[0] = d0; v.shape[1] = d1; v.shape[2] = d2; v.shape[3] = d3;
        v.size = d0 * (d1 ? d1 : 1) * (d2 ? d2 : 1) * (d3 ? d3 : 1);
        total += v.size;
        vars.push_back(v);
    }
    int64_t off(const char *name) const {
        for (auto &v : vars)
            if (v.name == name) return v.off;
        return -1;
    }
};

static bool is_recurrent(const ppo2hip_policy_desc &d) {
    return d.kind == PPO2HIP_POLICY_CNN_LSTM || d.kind == PPO2HIP_POLICY_CNN_LNLSTM;
}

// conv output size, VALID padding
static int conv_out(int in, int k, int s) { return (in - k) / s + 1; }

static int build_layout(const ppo2hip_policy_desc &d, Layout &L) {
    if (d.ac_dim < 1 || d.nvalue < 1 || d.nvalue > 8) {
        set_error("policy_desc: ac_dim=%d nvalue=%d out of range", d.ac_dim, d.nvalue);
        return PPO2HIP_EINVAL;
    }
    const bool gauss = d.dist == PPO2HIP_DIST_GAUSSIAN;
    if (d.kind == PPO2HIP_POLICY_MLP) {
        if (d.ob_dim < 1 || d.hidden1 < 1 || d.hidden2 < 1) {
            set_error("policy_desc: MLP needs ob_dim, hidden1, hidden2 >= 1");
            return PPO2HIP_EINVAL;
        }
        auto actor = [&]() {
            L.add("pi_fc1/w", d.ob_dim, d.hidden1); L.add("pi_fc1/b", d.hidden1);
            L.add("pi_fc2/w", d.hidden1, d.hidden2); L.add("pi_fc2/b", d.hidden2);
        };
        auto head = [&]() {
            L.add("pi/w", d.hidden2, d.ac_dim); L.add("pi/b", d.ac_dim);
            if (gauss) L.add("logstd", 1, d.ac_dim);
        };
        auto value = [&]() {
            L.add("vf_fc1/w", d.ob_dim, d.hidden1); L.add("vf_fc1/b", d.hidden1);
            L.add("vf_fc2/w", d.hidden1, d.hidden2); L.add("vf_fc2/b", d.hidden2);
            L.add("vf/w", d.hidden2, d.nvalue); L.add("vf/b", d.nvalue);
        };
        if (d.order == PPO2HIP_ORDER_FORK) { actor(); head(); value(); }       // ppo2/policies.py:139-149
        else { actor(); value(); head(); }                                    // a2c/policies.py:123-129
        return 0;
    }
    if (d.kind == PPO2HIP_POLICY_CNN || is_recurrent(d)) {
        if (d.ob_h < 36 || d.ob_w < 36 || d.ob_c < 1 || d.nvalue != 1) {
            set_error("policy_desc: CNN needs ob_h, ob_w >= 36, ob_c >= 1, nvalue == 1");
            return PPO2HIP_EINVAL;
        }
        const int h1 = conv_out(d.ob_h, 8, 4), w1 = conv_out(d.ob_w, 8, 4);
        const int h2 = conv_out(h1, 4, 2), w2 = conv_out(w1, 4, 2);
        const int h3 = conv_out(h2, 3, 1), w3 = conv_out(w2, 3, 1);
        L.add("c1/w", 8, 8, d.ob_c, 32); L.add("c1/b", 1, 32, 1, 1);   // bias kept [1,nf,1,1], a2c/utils.py:53
        L.add("c2/w", 4, 4, 32, 64); L.add("c2/b", 1, 64, 1, 1);
        L.add("c3/w", 3, 3, 64, 64); L.add("c3/b", 1, 64, 1, 1);
        L.add("fc1/w", (int64_t)h3 * w3 * 64, 512); L.add("fc1/b", 512);
        int feat = 512;
        if (is_recurrent(d)) {
            if (d.nlstm < 1) { set_error("policy_desc: nlstm must be >= 1"); return PPO2HIP_EINVAL; }
            const int nh = d.nlstm;
            if (d.kind == PPO2HIP_POLICY_CNN_LSTM) {                      // a2c/utils.py:90-92
                L.add("lstm1/wx", 512, 4 * nh); L.add("lstm1/wh", nh, 4 * nh); L.add("lstm1/b", 4 * nh);
            } else {                                                      // a2c/utils.py:120-131
                L.add("lstm1/wx", 512, 4 * nh); L.add("lstm1/gx", 4 * nh); L.add("lstm1/bx", 4 * nh);
                L.add("lstm1/wh", nh, 4 * nh); L.add("lstm1/gh", 4 * nh); L.add("lstm1/bh", 4 * nh);
                L.add("lstm1/b", 4 * nh); L.add("lstm1/gc", nh); L.add("lstm1/bc", nh);
            }
            feat = nh;
        }
        L.add("pi/w", feat, d.ac_dim); L.add("pi/b", d.ac_dim);
        L.add("v/w", feat, 1); L.add("v/b", 1);
        return 0;
    }
    set_error("policy_desc: unknown kind %d", d.kind);
    return PPO2HIP_EINVAL;
}

// ---------------------------------------------------------------------------------------
// MLP: two separate tanh trunks
// ---------------------------------------------------------------------------------------
struct MlpBuf {
    float *X, *RET, *OLDV, *NLP, *ADV;
    void *ACT;
    float *H1a, *H2a, *PI, *H1v, *H2v, *VF, *dPI, *dVF, *dZ2, *dZ1, *slabs;
    float *dZ2v, *dZ1v;          // the value trunk's own gradient rows where both trunks run in one launch
    double *loss256;
    float *wslabs256;
    char *fused;
    size_t fused_bytes;
    double *loss_partial;
    int max_slabs;
    int64_t slab_elems;
};

static void carve_mlp(const ppo2hip_policy_desc &d, int64_t B, Carver &c, MlpBuf &w, bool grad) {
    const int D = d.ob_dim, h1 = d.hidden1, h2 = d.hidden2, A = d.ac_dim, R = d.nvalue;
    w.X = c.take<float>(B * D);
    w.H1a = c.take<float>(B * h1); w.H2a = c.take<float>(B * h2); w.PI = c.take<float>(B * A);
    w.H1v = c.take<float>(B * h1); w.H2v = c.take<float>(B * h2); w.VF = c.take<float>(B * R);
    if (!grad) return;
    w.ACT = c.take<float>(B * (d.dist == PPO2HIP_DIST_GAUSSIAN ? A : 1));
    w.RET = c.take<float>(B * R); w.OLDV = c.take<float>(B * R); w.NLP = c.take<float>(B); w.ADV = c.take<float>(B);
    w.dPI = c.take<float>(B * A); w.dVF = c.take<float>(B * R);
    w.dZ2 = c.take<float>(B * h2); w.dZ1 = c.take<float>(B * h1);
    int ms = 1;
    int64_t se = 0;
    const int kin[3] = {D, h1, h2};
    const int nout_a[3] = {h1, h2, A}, nout_v[3] = {h1, h2, R};
    for (int l = 0; l < 3; l++) {
        for (int n : {nout_a[l], nout_v[l]}) {
            const int s = wgrad_max_slabs(B, kin[l], n);
            const int64_t e = (int64_t)s * (kin[l] + 1) * n;
            if (s > ms) ms = s;
            if (e > se) se = e;
        }
    }
    w.max_slabs = ms;
    w.slab_elems = se;
    w.slabs = c.take<float>(se);
    w.loss_partial = c.take<double>((int64_t)loss_partial_blocks(B) * 80);
    w.fused_bytes = 0;
    if (mlp_fused_supported(d)) w.fused_bytes = mlp_fused_workspace_bytes();
    if (mlp_fused16_supported(d) && mlp_fused16_workspace_bytes() > w.fused_bytes) w.fused_bytes = mlp_fused16_workspace_bytes();
    w.fused = c.take<char>(w.fused_bytes);
    w.dZ2v = w.dZ1v = nullptr;
    w.loss256 = nullptr;
    w.wslabs256 = nullptr;
    if (mlp_fused256_supported(d)) {
        w.wslabs256 = (float *)c.take<char>(mlp_fused256_wgrad_bytes());
        w.dZ2v = c.take<float>(B * h2); w.dZ1v = c.take<float>(B * h1);
        w.loss256 = (double *)c.take<char>(mlp_fused256_loss_bytes());
    }
}

static int mlp_forward(const ppo2hip_policy_desc &d, const Layout &L, const float *p, const MlpBuf &w, int64_t B,
                       float *pi, float *vf, hipStream_t s) {
    const int D = d.ob_dim, h1 = d.hidden1, h2 = d.hidden2, A = d.ac_dim, R = d.nvalue;
    int rc;
    if ((rc = fc_forward(w.X, D, p + L.off("pi_fc1/w"), p + L.off("pi_fc1/b"), w.H1a, h1, B, h1, D, ACT_TANH, false, s))) return rc;
    if ((rc = fc_forward(w.H1a, h1, p + L.off("pi_fc2/w"), p + L.off("pi_fc2/b"), w.H2a, h2, B, h2, h1, ACT_TANH, false, s))) return rc;
    if ((rc = fc_forward(w.H2a, h2, p + L.off("pi/w"), p + L.off("pi/b"), pi, A, B, A, h2, ACT_NONE, false, s))) return rc;
    if ((rc = fc_forward(w.X, D, p + L.off("vf_fc1/w"), p + L.off("vf_fc1/b"), w.H1v, h1, B, h1, D, ACT_TANH, false, s))) return rc;
    if ((rc = fc_forward(w.H1v, h1, p + L.off("vf_fc2/w"), p + L.off("vf_fc2/b"), w.H2v, h2, B, h2, h1, ACT_TANH, false, s))) return rc;
    if ((rc = fc_forward(w.H2v, h2, p + L.off("vf/w"), p + L.off("vf/b"), vf, R, B, R, h2, ACT_NONE, false, s))) return rc;
    return 0;
}

// weight + bias gradient of one fc layer into grads[off .. off + (Kin+1)*N)
static int fc_wgrad_into(const float *X, int ldx, const float *dY, int lddy, int64_t S, int Kin, int N,
                         const MlpBuf &w, float *dst, hipStream_t s) {
    int ns = 0;
    int rc = fc_wgrad_slabs(X, ldx, dY, lddy, S, Kin, N, w.slabs, w.max_slabs, &ns, s);
    if (rc) return rc;
    return reduce_slabs(w.slabs, ns, (int64_t)(Kin + 1) * N, dst, false, s);
}

static int mlp_trunk_backward(const float *X, int D, const float *H1, int h1, const float *H2, int h2,
                              const float *dOut, int N, const float *p, float *g, int64_t o1, int64_t o2,
                              int64_t oh, const MlpBuf &w, int64_t B, hipStream_t s) {
    int rc;
    if ((rc = fc_wgrad_into(H2, h2, dOut, N, B, h2, N, w, g + oh, s))) return rc;
    if ((rc = fc_dgrad(dOut, N, p + oh, h2, N, H2, h2, ACT_TANH, w.dZ2, h2, B, false, s))) return rc;
    if ((rc = fc_wgrad_into(H1, h1, w.dZ2, h2, B, h1, h2, w, g + o2, s))) return rc;
    if ((rc = fc_dgrad(w.dZ2, h2, p + o2, h1, h2, H1, h1, ACT_TANH, w.dZ1, h1, B, false, s))) return rc;
    if ((rc = fc_wgrad_into(X, D, w.dZ1, h1, B, D, h1, w, g + o1, s))) return rc;
    return 0;
}

// PPO2HIP_MLP_PATH = layered | fused32 | fused16 pins one of the routes for shapes that several
// cover (the tests check them against each other); default: the fastest that supports the shape.
static int mlp_path_choice(const ppo2hip_policy_desc &d, const ppo2hip_rollout &ro, int64_t B) {
    const char *e = getenv("PPO2HIP_MLP_PATH");
    const bool f16 = mlp_fused16_supported(d) && mlp_fused16_fits(d, ro, B), f32 = mlp_fused_supported(d);
    if (e && std::strcmp(e, "layered") == 0) return 0;
    if (mlp_fused256_supported(d)) return 256;
    if (e && std::strcmp(e, "fused32") == 0 && f32) return 32;
    if (e && std::strcmp(e, "fused16") == 0 && f16) return 16;
    return f16 ? 16 : (f32 ? 32 : 0);
}

static int mlp_grad(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const Layout &L, const float *p,
                    const ppo2hip_rollout &ro, const int32_t *idx, int64_t B, const double *stats, float cliprange,
                    double inv_count, float *grads, float *losses, void *ws, size_t ws_bytes, hipStream_t s,
                    AdamFuse *adam = nullptr, bool *adam_done = nullptr) {
    Carver c(ws, ws_bytes);
    MlpBuf w;
    carve_mlp(d, B, c, w, true);
    if (!c.ok) {
        set_error("ppo_grad: workspace too small (%zu < %zu bytes)", ws_bytes, c.used());
        return PPO2HIP_ENOSPACE;
    }
    const int D = d.ob_dim, h1 = d.hidden1, h2 = d.hidden2, A = d.ac_dim, R = d.nvalue;
    const bool gauss = d.dist == PPO2HIP_DIST_GAUSSIAN;
    const int path = mlp_path_choice(d, ro, B);
    if (path == 16 || path == 32) {
        FusedOffsets fo{L.off("pi_fc1/w"), L.off("pi_fc2/w"), L.off("pi/w"), L.off("vf_fc1/w"), L.off("vf_fc2/w"),
                        L.off("vf/w"), gauss ? L.off("logstd") : 0};
        if (path == 16) {
            // the optimiser step rides in the finalize launch when the caller offers it and the arena is small enough
            // (its blocks, one per 64 parameters plus the head and loss blocks, wait for each other: one per CU of this device)
            const bool fold = adam && L.total <= 16384 && device_cu_count() >= (L.total + 63) / 64 + 8 && !(getenv("PPO2HIP_NO_FUSED_ADAM"));
            if (fold && adam_done) *adam_done = true;
            return mlp_fused16_grad(d, lc, p, fo, ro, idx, B, stats, cliprange, inv_count, grads, losses, w.fused,
                                    w.fused_bytes, s, fold ? adam : nullptr);
        }
        return mlp_fused_grad(d, lc, p, fo, ro, idx, B, stats, cliprange, inv_count, grads, losses, w.fused,
                              w.fused_bytes, s);
    }
    int rc;
    if (path == 256) {
        // forward, loss and both input-gradient products of both trunks in one launch, then every weight gradient in
        // another; both read the minibatch rows straight out of the rollout where 32-bit offsets reach them
        FusedOffsets fo{L.off("pi_fc1/w"), L.off("pi_fc2/w"), L.off("pi/w"), L.off("vf_fc1/w"), L.off("vf_fc2/w"),
                        L.off("vf/w"), gauss ? L.off("logstd") : 0};
        Fused256Buffers fb;
        if (rollout_fits_buffer_loads(d, ro, B) && !getenv("PPO2HIP_F256_GATHER")) {
            fb.obs = (const float *)ro.obs; fb.actions = ro.actions; fb.returns = ro.returns; fb.values = ro.values;
            fb.neglogpacs = ro.neglogpacs; fb.advs = ro.advs; fb.adv_f64 = ro.adv_f64; fb.T = ro.T; fb.E = ro.E; fb.n = ro.n;
            fb.idx = idx; fb.stats = stats; fb.xg = w.X;
        } else {
            GatherArgs g;
            g.ro = &ro; g.idx = idx; g.B = B; g.ob_bytes = (int64_t)D * 4; g.ac_bytes = gauss ? (int64_t)A * 4 : 4;
            g.nvalue = R; g.stats = stats; g.obs = w.X; g.actions = w.ACT; g.returns = w.RET; g.values = w.OLDV;
            g.neglogpacs = w.NLP; g.advs = w.ADV; g.masks = nullptr;
            if ((rc = gather_minibatch(g, s))) return rc;
            fb.obs = w.X; fb.actions = w.ACT; fb.returns = w.RET; fb.values = w.OLDV; fb.neglogpacs = w.NLP; fb.advs = w.ADV;
            fb.adv_f64 = 0; fb.T = 0; fb.E = 0; fb.n = B; fb.idx = nullptr; fb.stats = nullptr; fb.xg = nullptr;
        }
        fb.H1[0] = w.H1a; fb.H2[0] = w.H2a; fb.DZ1[0] = w.dZ1; fb.DZ2[0] = w.dZ2; fb.DOUT[0] = w.dPI;
        fb.H1[1] = w.H1v; fb.H2[1] = w.H2v; fb.DZ1[1] = w.dZ1v; fb.DZ2[1] = w.dZ2v; fb.DOUT[1] = w.dVF;
        fb.loss_partial = w.loss256; fb.wslabs = w.wslabs256;
        if ((rc = mlp_fused256_chain(d, lc, p, fo, fb, B, cliprange, inv_count, losses, gauss ? grads + L.off("logstd") : nullptr, s))) return rc;
        return mlp_fused256_wgrad(d, fo, fb, B, grads, s);
    }
    GatherArgs g;
    g.ro = &ro; g.idx = idx; g.B = B; g.ob_bytes = (int64_t)D * 4; g.ac_bytes = gauss ? (int64_t)A * 4 : 4;
    g.nvalue = R; g.stats = stats; g.obs = w.X; g.actions = w.ACT; g.returns = w.RET; g.values = w.OLDV;
    g.neglogpacs = w.NLP; g.advs = w.ADV; g.masks = nullptr;
    if ((rc = gather_minibatch(g, s))) return rc;
    if ((rc = mlp_forward(d, L, p, w, B, w.PI, w.VF, s))) return rc;
    LossArgs la;
    la.dist = d.dist; la.ac_dim = A; la.nvalue = R; la.B = B; la.pi = w.PI;
    la.logstd = gauss ? p + L.off("logstd") : nullptr; la.vf = w.VF; la.actions = w.ACT; la.advs = w.ADV;
    la.returns = w.RET; la.oldv = w.OLDV; la.oldnlp = w.NLP; la.cliprange = cliprange; la.cfg = lc;
    la.inv_count = inv_count; la.dpi = w.dPI; la.dvf = w.dVF; la.partial = w.loss_partial; la.losses = losses;
    la.dlogstd = gauss ? grads + L.off("logstd") : nullptr;
    if ((rc = ppo_loss_backward(la, s))) return rc;
    if ((rc = mlp_trunk_backward(w.X, D, w.H1a, h1, w.H2a, h2, w.dPI, A, p, grads, L.off("pi_fc1/w"),
                                 L.off("pi_fc2/w"), L.off("pi/w"), w, B, s))) return rc;
    if ((rc = mlp_trunk_backward(w.X, D, w.H1v, h1, w.H2v, h2, w.dVF, R, p, grads, L.off("vf_fc1/w"),
                                 L.off("vf_fc2/w"), L.off("vf/w"), w, B, s))) return rc;
    return 0;
}

// ---------------------------------------------------------------------------------------
// nature_cnn (+ LSTM / layer-normalised LSTM) policies: baselines/ppo2/policies.py:6-130, baselines/a2c/utils.py:86-147
// ---------------------------------------------------------------------------------------
struct CnnDims {
    int H, W, C, h1, w1, h2, w2, h3, w3, flat, feat, nh, nact;
    bool lstm;   // recurrent (either cell)
    bool ln;     // layer-normalised cell
};
static CnnDims cnn_dims(const ppo2hip_policy_desc &d) {
    CnnDims c;
    c.H = d.ob_h; c.W = d.ob_w; c.C = d.ob_c;
    c.h1 = conv_out(c.H, 8, 4); c.w1 = conv_out(c.W, 8, 4);
    c.h2 = conv_out(c.h1, 4, 2); c.w2 = conv_out(c.w1, 4, 2);
    c.h3 = conv_out(c.h2, 3, 1); c.w3 = conv_out(c.w2, 3, 1);
    c.flat = c.h3 * c.w3 * 64;
    c.lstm = d.kind == PPO2HIP_POLICY_CNN_LSTM || d.kind == PPO2HIP_POLICY_CNN_LNLSTM;
    c.ln = d.kind == PPO2HIP_POLICY_CNN_LNLSTM;
    c.nh = d.nlstm;
    c.feat = c.lstm ? d.nlstm : 512;
    c.nact = d.ac_dim;
    return c;
}
static ConvShape conv_shape(int B, int IH, int IW, int Cin, int K, int S, int Cout) {
    return ConvShape{B, IH, IW, Cin, K, K, S, conv_out(IH, K, S), conv_out(IW, K, S), Cout};
}

struct CnnBuf {
    uint8_t *X, *MASK;
    int32_t *ACT;
    float *RET, *OLDV, *NLP, *ADV;
    float *Y1, *Y2, *Y3, *F, *PI, *VF;
    float *Z, *HM, *CM, *C, *Hs, *S0;          // LSTM: gates [B,4nh], masked prev states, states, gathered initial state
    float *ZP;                                 // partial products of one step's hm wh, [kLstmFwdParts][nenv, 4nh]
    void *XCHG;                                // hand-over words of the whole-sequence LSTM launches
    float *ZX, *ZH, *CN, *RX, *RH, *RC;        // layer-normalised cell: normalised x wx / hm wh / c rows and their 1/std
    float *dPI, *dVF, *dF, *dY3, *dY2, *dY1, *DH, *dhm, *dcm;
    float *DZX, *DZH, *DL;                     // ... gradients w.r.t. the raw products and w.r.t. LN(c)
    float *slabs;
    int max_slabs;
    double *loss_partial;
    float *CS;                                 // per-tile column sums handed from an input-gradient product to the bias gradient
    uint32_t *RB1;                             // rectifier bits of conv1's output, one word per pixel
};

static void carve_cnn(const ppo2hip_policy_desc &d, int64_t B, int nenv, Carver &c, CnnBuf &w, bool grad) {
    const CnnDims k = cnn_dims(d);
    w.Y1 = c.take<float>(B * k.h1 * k.w1 * 32);
    w.Y2 = c.take<float>(B * k.h2 * k.w2 * 64);
    w.Y3 = c.take<float>(B * k.flat);
    w.F = c.take<float>(B * 512);
    if (k.lstm) {
        w.Z = c.take<float>(B * 4 * k.nh);
        w.HM = c.take<float>(B * k.nh); w.CM = c.take<float>(B * k.nh);
        w.C = c.take<float>(B * k.nh); w.Hs = c.take<float>(B * k.nh);
        w.S0 = c.take<float>((int64_t)nenv * 2 * k.nh);
        w.ZP = c.take<float>((int64_t)kLstmFwdParts * nenv * 4 * k.nh);
        {   // hand-over words of the whole-sequence launches: only env counts those launches take (all workgroups resident)
            const int seq_envs = nenv < 65536 / k.nh ? nenv : 65536 / k.nh;
            w.XCHG = c.take<uint8_t>(k.ln ? 0 : (int64_t)lstm_seq_xchg_bytes(seq_envs, k.nh));
        }
        if (k.ln) {
            w.ZX = c.take<float>(B * 4 * k.nh); w.ZH = c.take<float>(B * 4 * k.nh); w.CN = c.take<float>(B * k.nh);
            w.RX = c.take<float>(B); w.RH = c.take<float>(B); w.RC = c.take<float>(B);
        }
    }
    if (!grad) return;
    w.X = c.take<uint8_t>(B * k.H * k.W * k.C);
    w.MASK = c.take<uint8_t>(B);
    w.ACT = c.take<int32_t>(B);
    w.RET = c.take<float>(B); w.OLDV = c.take<float>(B); w.NLP = c.take<float>(B); w.ADV = c.take<float>(B);
    w.PI = c.take<float>(B * k.nact); w.VF = c.take<float>(B);
    w.dPI = c.take<float>(B * k.nact); w.dVF = c.take<float>(B);
    w.dF = c.take<float>(B * 512);
    w.dY3 = c.take<float>(B * k.flat);
    w.dY2 = c.take<float>(B * k.h2 * k.w2 * 64);
    w.dY1 = c.take<float>(B * k.h1 * k.w1 * 32);
    w.RB1 = c.take<uint32_t>(B * k.h1 * k.w1);
    if (k.lstm) {
        w.DH = c.take<float>(B * k.nh);
        w.dhm = c.take<float>((int64_t)kLstmBwdParts * nenv * k.nh); w.dcm = c.take<float>((int64_t)nenv * k.nh);
        if (k.ln) {
            w.DZX = c.take<float>(B * 4 * k.nh); w.DZH = c.take<float>(B * 4 * k.nh); w.DL = c.take<float>(B * k.nh);
        }
    }
    // weight-gradient slabs: the largest layer decides
    int64_t se = 0;
    int ms = 1;
    auto consider = [&](int64_t S, int Kin, int N) {
        const int s = wgrad_max_slabs(S, Kin, N);
        const int64_t e = (int64_t)s * (Kin + 1) * N;
        if (e > se) se = e;
        if (s > ms) ms = s;
    };
    consider(B * k.h1 * k.w1, 8 * 8 * k.C, 32);
    consider(B * k.h2 * k.w2, 4 * 4 * 32, 64);
    consider(B * k.h3 * k.w3, 3 * 3 * 64, 64);
    consider(B, k.flat, 512);
    consider(B, k.feat, k.nact);
    consider(B, k.feat, 1);
    if (k.lstm) { consider(B, 512, 4 * k.nh); consider(B, k.nh, 4 * k.nh); }
    w.max_slabs = ms;
    w.slabs = c.take<float>(se);
    w.loss_partial = c.take<double>((int64_t)loss_partial_blocks(B) * 80);
    {
        const ConvShape c3 = conv_shape((int)B, k.h2, k.w2, 64, 3, 1, 64), c2 = conv_shape((int)B, k.h1, k.w1, 32, 4, 2, 64);
        int64_t n = ceil_div(B, (int64_t)64) * k.flat;
        if (conv_dgrad_colsum_rows(c3) * 64 > n) n = conv_dgrad_colsum_rows(c3) * 64;
        if (conv_dgrad_colsum_rows(c2) * 32 > n) n = conv_dgrad_colsum_rows(c2) * 32;
        w.CS = c.take<float>(n);
    }
}

// conv trunk + fc1: X (uint8 NHWC) -> F [B, 512]
static int cnn_trunk_forward(const ppo2hip_policy_desc &d, const Layout &L, const float *p, const void *X,
                             const CnnBuf &w, int64_t B, hipStream_t s, const int32_t *rows = nullptr,
                             uint32_t *relu_bits1 = nullptr) {
    const CnnDims k = cnn_dims(d);
    int rc;
    if ((rc = conv_forward(X, true, p + L.off("c1/w"), p + L.off("c1/b"), w.Y1, conv_shape((int)B, k.H, k.W, k.C, 8, 4, 32), s, rows, relu_bits1))) return rc;
    if ((rc = conv_forward(w.Y1, false, p + L.off("c2/w"), p + L.off("c2/b"), w.Y2, conv_shape((int)B, k.h1, k.w1, 32, 4, 2, 64), s))) return rc;
    if ((rc = conv_forward(w.Y2, false, p + L.off("c3/w"), p + L.off("c3/b"), w.Y3, conv_shape((int)B, k.h2, k.w2, 64, 3, 1, 64), s))) return rc;
    return fc_forward(w.Y3, k.flat, p + L.off("fc1/w"), p + L.off("fc1/b"), w.F, 512, (int)B, 512, k.flat, ACT_RELU, false, s);
}

// LSTM over rows e*T + t (T steps, nenv envs): F [B,512] -> Hs [B,nh]; masks [B]; S0 [nenv, 2nh]
static int lstm_forward_all(const ppo2hip_policy_desc &d, const Layout &L, const float *p, const CnnBuf &w,
                            const uint8_t *masks, int nenv, int T, float *state_out, hipStream_t s) {
    const int nh = d.nlstm;
    const int64_t B = (int64_t)nenv * T;
    int rc;
    if (d.kind == PPO2HIP_POLICY_CNN_LNLSTM) {
        const LnLstmParams P{p + L.off("lstm1/gx"), p + L.off("lstm1/bx"), p + L.off("lstm1/gh"), p + L.off("lstm1/bh"),
                             p + L.off("lstm1/b"), p + L.off("lstm1/gc"), p + L.off("lstm1/bc")};
        // x wx for every row at once, normalised in place
        if ((rc = fc_forward(w.F, 512, p + L.off("lstm1/wx"), nullptr, w.ZX, 4 * nh, (int)B, 4 * nh, 512, ACT_NONE, false, s))) return rc;
        if ((rc = ln_rows(w.ZX, 4 * nh, B, 4 * nh, w.RX, s))) return rc;
        const float *wh = p + L.off("lstm1/wh");
        for (int t = 0; t < T; t++) {
            const float *cprev = t == 0 ? w.S0 : w.C + (int64_t)(t - 1) * nh;
            const float *hprev = t == 0 ? w.S0 + nh : w.Hs + (int64_t)(t - 1) * nh;
            const int64_t pstride = t == 0 ? 2 * nh : (int64_t)T * nh;
            if ((rc = lstm_mask(cprev, hprev, pstride, masks + t, T, nenv, nh, w.CM + (int64_t)t * nh, w.HM + (int64_t)t * nh, (int64_t)T * nh, s))) return rc;
            if ((rc = fc_forward(w.HM + (int64_t)t * nh, T * nh, wh, nullptr, w.ZH + (int64_t)t * 4 * nh, T * 4 * nh, nenv, 4 * nh, nh, ACT_NONE, false, s))) return rc;
            if ((rc = lnlstm_step_fwd(w.ZX + (int64_t)t * 4 * nh, w.ZH + (int64_t)t * 4 * nh, (int64_t)T * 4 * nh, w.Z + (int64_t)t * 4 * nh, P,
                                      w.CM + (int64_t)t * nh, (int64_t)T * nh, nenv, nh, w.C + (int64_t)t * nh, w.CN + (int64_t)t * nh,
                                      w.Hs + (int64_t)t * nh, w.RH + t, w.RC + t, T, t == T - 1 ? state_out : nullptr, s))) return rc;
        }
        return 0;
    }
    // x wx + b for every row at once
    if ((rc = fc_forward(w.F, 512, p + L.off("lstm1/wx"), p + L.off("lstm1/b"), w.Z, 4 * nh, (int)B, 4 * nh, 512, ACT_NONE, false, s))) return rc;
    const float *wh = p + L.off("lstm1/wh");
    // step 0 masks the gathered initial state; every later step's masked states come out of the step before
    if ((rc = lstm_mask(w.S0, w.S0 + nh, 2 * nh, masks, T, nenv, nh, w.CM, w.HM, (int64_t)T * nh, s))) return rc;
    // one launch per step (product and gates together) for the widths that have an instance; PPO2HIP_LSTM_STEPS=split
    // keeps the product on the GEMM and the gates in a launch of their own
    const char *steps_env = getenv("PPO2HIP_LSTM_STEPS");
    const bool fused_steps = lstm_step_fused_width(nh) && (uintptr_t)wh % 16 == 0 && !(steps_env && std::strcmp(steps_env, "split") == 0);
    // the whole sequence in one launch where its workgroups fit the chip together (PPO2HIP_LSTM_SEQ_OFF=1: a launch per step)
    if (fused_steps && T > 1 && lstm_seq_ok(nenv, nh))
        return lstm_seq_fwd(w.Z, wh, w.HM, w.CM, nenv, T, nh, w.C, w.Hs, state_out, masks, w.XCHG, s);
    for (int t = 0; t < T; t++) {
        const int64_t o1 = (int64_t)t * nh, o4 = (int64_t)t * 4 * nh;
        const bool last = t == T - 1;
        if (fused_steps) {
            if ((rc = lstm_step_fwd(w.Z + o4, (int64_t)T * 4 * nh, wh, w.HM + o1, w.CM + o1, (int64_t)T * nh, nenv, nh, w.C + o1,
                                    w.Hs + o1, last ? state_out : nullptr, last ? nullptr : masks + t + 1, T,
                                    last ? nullptr : w.CM + o1 + nh, last ? nullptr : w.HM + o1 + nh, s))) return rc;
            continue;
        }
        // partial products of hm_t wh (rows of step t are T rows apart), summed into z_t by the gate kernel
        if ((rc = fc_forward_parts(w.HM + o1, T * nh, wh, nenv, 4 * nh, nh, kLstmFwdParts, w.ZP, s))) return rc;
        if ((rc = lstm_gates(w.Z + o4, (int64_t)T * 4 * nh, w.ZP, kLstmFwdParts, w.CM + o1, (int64_t)T * nh, nenv, nh, w.C + o1,
                             w.Hs + o1, last ? state_out : nullptr, last ? nullptr : masks + t + 1, T,
                             last ? nullptr : w.CM + o1 + nh, last ? nullptr : w.HM + o1 + nh, s))) return rc;
    }
    return 0;
}

static int wgrad_into(const float *X, int ldx, const float *dY, int lddy, int64_t S, int Kin, int N, float *slabs,
                      int max_slabs, float *dst, hipStream_t s) {
    int ns = 0;
    int rc = fc_wgrad_slabs(X, ldx, dY, lddy, S, Kin, N, slabs, max_slabs, &ns, s);
    if (rc) return rc;
    return reduce_slabs(slabs, ns, (int64_t)(Kin + 1) * N, dst, false, s);
}

// bias_rows: `nbias` rows of Cout column sums of dY that the product which wrote dY left behind (its row tiles), or
// null — then the bias gradient costs a pass of its own over dY
static int conv_wgrad_into(const void *X, bool u8, const float *dY, const ConvShape &c, float *slabs, int max_slabs,
                           float *dst, hipStream_t s, const float *bias_rows = nullptr, int nbias = 0,
                           const int32_t *rows = nullptr) {
    int ns = 0;
    const int64_t K = (int64_t)c.KH * c.KW * c.Cin;
    const bool given = bias_rows && nbias > 0 && wgrad_bias_separately((int)K);
    int rc = conv_wgrad_slabs(X, u8, dY, c, slabs, max_slabs, &ns, s, !given, rows);
    if (rc) return rc;
    if (!given) return reduce_slabs(slabs, ns, (K + 1) * c.Cout, dst, false, s);
    if ((rc = reduce_slabs(slabs, ns, K * c.Cout, dst, false, s, (K + 1) * c.Cout))) return rc;
    return colsum_rows(bias_rows, nbias, c.Cout, slabs, dst + K * c.Cout, s);        // the slabs are spent: scratch
}

static int cnn_grad(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const Layout &L, const float *p,
                    const ppo2hip_rollout &ro, const int32_t *idx, int64_t B, const int32_t *mb_envs, int nsteps_mb,
                    const double *stats, float cliprange, double inv_count, float *grads, float *losses, void *ws,
                    size_t ws_bytes, hipStream_t s) {
    const CnnDims k = cnn_dims(d);
    if (d.dist != PPO2HIP_DIST_CATEGORICAL) { set_error("ppo_grad: nature_cnn policies are Categorical"); return PPO2HIP_EUNSUPPORTED; }
    int nenv = 1, T = 1;
    if (k.lstm) {
        if (nsteps_mb < 1 || B % nsteps_mb != 0 || !mb_envs || !ro.masks || !ro.states) {
            set_error("ppo_grad: recurrent policy needs masks, states, mb_envs and nsteps_mb dividing B");
            return PPO2HIP_EINVAL;
        }
        T = nsteps_mb;
        nenv = (int)(B / T);
    }
    Carver c(ws, ws_bytes);
    CnnBuf w{};
    carve_cnn(d, B, nenv, c, w, true);
    if (!c.ok) { set_error("ppo_grad: workspace too small (%zu < %zu bytes)", ws_bytes, c.used()); return PPO2HIP_ENOSPACE; }
    GatherArgs g;
    g.ro = &ro; g.idx = idx; g.B = B; g.ob_bytes = (int64_t)k.H * k.W * k.C; g.ac_bytes = 4; g.nvalue = 1; g.stats = stats;
    g.obs = w.X; g.actions = w.ACT; g.returns = w.RET; g.values = w.OLDV; g.neglogpacs = w.NLP; g.advs = w.ADV;
    g.masks = k.lstm ? w.MASK : nullptr;
    // Frames the uint8 kernels of conv1 can take are read where they lie in the rollout (a 231 MB copy per 8192-frame
    // minibatch saved): the gather then leaves the frames' row numbers at the head of the frame buffer instead.
    const bool direct = conv_u8_direct_ok(ro.obs, conv_shape((int)B, k.H, k.W, k.C, 8, 4, 32)) && ro.n < 0x7fffffffll;
    const int32_t *rows = nullptr;
    const void *frames = w.X;
    if (direct) {
        g.obs_rows = (int32_t *)w.X;
        rows = g.obs_rows;
        frames = ro.obs;
    }
    // ... and conv1's forward kernel leaves one word of "output > 0" bits per pixel, which the product that writes dY1 reads
    // for the rectifier's derivative in place of the 420 MB activation tensor (PPO2HIP_RELU_FROM_ACT=1 reads the tensor).
    const bool defer_relu1 = direct && !getenv("PPO2HIP_RELU_FROM_ACT");
    int rc;
    if ((rc = gather_minibatch(g, s))) return rc;
    if ((rc = cnn_trunk_forward(d, L, p, frames, w, B, s, rows, defer_relu1 ? w.RB1 : nullptr))) return rc;
    const float *feat = w.F;
    if (k.lstm) {
        if ((rc = gather_states(ro.states, mb_envs, nenv, 2 * k.nh, w.S0, s))) return rc;
        if ((rc = lstm_forward_all(d, L, p, w, w.MASK, nenv, T, nullptr, s))) return rc;
        feat = w.Hs;
    }
    if ((rc = fc_forward(feat, k.feat, p + L.off("pi/w"), p + L.off("pi/b"), w.PI, k.nact, (int)B, k.nact, k.feat, ACT_NONE, false, s))) return rc;
    if ((rc = fc_forward(feat, k.feat, p + L.off("v/w"), p + L.off("v/b"), w.VF, 1, (int)B, 1, k.feat, ACT_NONE, false, s))) return rc;
    LossArgs la;
    la.dist = d.dist; la.ac_dim = k.nact; la.nvalue = 1; la.B = B; la.pi = w.PI; la.logstd = nullptr; la.vf = w.VF;
    la.actions = w.ACT; la.advs = w.ADV; la.returns = w.RET; la.oldv = w.OLDV; la.oldnlp = w.NLP; la.cliprange = cliprange;
    la.cfg = lc; la.inv_count = inv_count; la.dpi = w.dPI; la.dvf = w.dVF; la.partial = w.loss_partial; la.losses = losses;
    la.dlogstd = nullptr;
    if ((rc = ppo_loss_backward(la, s))) return rc;
    // heads
    if ((rc = wgrad_into(feat, k.feat, w.dPI, k.nact, B, k.feat, k.nact, w.slabs, w.max_slabs, grads + L.off("pi/w"), s))) return rc;
    if ((rc = wgrad_into(feat, k.feat, w.dVF, 1, B, k.feat, 1, w.slabs, w.max_slabs, grads + L.off("v/w"), s))) return rc;
    if (!k.lstm) {
        // dF = (dPI Wpi^T + dVF Wv^T) * relu'(F)
        if ((rc = fc_dgrad(w.dPI, k.nact, p + L.off("pi/w"), 512, k.nact, w.F, 512, ACT_RELU, w.dF, 512, (int)B, false, s))) return rc;
        if ((rc = fc_dgrad(w.dVF, 1, p + L.off("v/w"), 512, 1, w.F, 512, ACT_RELU, w.dF, 512, (int)B, true, s))) return rc;
    } else {
        const int nh = k.nh;
        if ((rc = fc_dgrad(w.dPI, k.nact, p + L.off("pi/w"), nh, k.nact, nullptr, 0, ACT_NONE, w.DH, nh, (int)B, false, s))) return rc;
        if ((rc = fc_dgrad(w.dVF, 1, p + L.off("v/w"), nh, 1, nullptr, 0, ACT_NONE, w.DH, nh, (int)B, true, s))) return rc;
        const float *wh = p + L.off("lstm1/wh");
        if (k.ln) {
            const LnLstmParams P{p + L.off("lstm1/gx"), p + L.off("lstm1/bx"), p + L.off("lstm1/gh"), p + L.off("lstm1/bh"),
                                 p + L.off("lstm1/b"), p + L.off("lstm1/gc"), p + L.off("lstm1/bc")};
            for (int t = T - 1; t >= 0; t--) {
                const uint8_t *mnext = t == T - 1 ? nullptr : w.MASK + (t + 1);
                const int64_t o4 = (int64_t)t * 4 * nh, o1 = (int64_t)t * nh;
                if ((rc = lnlstm_step_bwd(w.Z + o4, w.ZX + o4, w.ZH + o4, w.DZX + o4, w.DZH + o4, (int64_t)T * 4 * nh, P, w.CM + o1,
                                          w.CN + o1, w.DH + o1, w.DL + o1, (int64_t)T * nh, w.RX + t, w.RH + t, w.RC + t, T, w.dhm,
                                          w.dcm, mnext, T, nenv, nh, s))) return rc;
                // dhm = (gradient w.r.t. hm wh) wh^T
                if (t > 0 && (rc = fc_dgrad(w.DZH + o4, T * 4 * nh, wh, nh, 4 * nh, nullptr, 0, ACT_NONE, w.dhm, nh, nenv, false, s))) return rc;
            }
            // Z holds dz, ZX / ZH the normalised products, DZX / DZH the gradients of the raw products.  The
            // products' appended ones-rows land on gx / gh, which the gain gradients then overwrite.
            if ((rc = wgrad_into(w.F, 512, w.DZX, 4 * nh, B, 512, 4 * nh, w.slabs, w.max_slabs, grads + L.off("lstm1/wx"), s))) return rc;
            if ((rc = wgrad_into(w.HM, nh, w.DZH, 4 * nh, B, nh, 4 * nh, w.slabs, w.max_slabs, grads + L.off("lstm1/wh"), s))) return rc;
            if ((rc = ln_colsum(w.Z, w.ZX, B, 4 * nh, grads + L.off("lstm1/gx"), s))) return rc;
            if ((rc = ln_colsum(w.Z, w.ZH, B, 4 * nh, grads + L.off("lstm1/gh"), s))) return rc;
            if ((rc = ln_colsum(w.Z, nullptr, B, 4 * nh, grads + L.off("lstm1/bx"), s))) return rc;   // bx, bh and b all add to z
            if ((rc = ln_colsum(w.Z, nullptr, B, 4 * nh, grads + L.off("lstm1/bh"), s))) return rc;
            if ((rc = ln_colsum(w.Z, nullptr, B, 4 * nh, grads + L.off("lstm1/b"), s))) return rc;
            if ((rc = ln_colsum(w.DL, w.CN, B, nh, grads + L.off("lstm1/gc"), s))) return rc;
            if ((rc = ln_colsum(w.DL, nullptr, B, nh, grads + L.off("lstm1/bc"), s))) return rc;
            if ((rc = fc_dgrad(w.DZX, 4 * nh, p + L.off("lstm1/wx"), 512, 4 * nh, w.F, 512, ACT_RELU, w.dF, 512, (int)B, false, s))) return rc;
        } else {
        const char *steps_env = getenv("PPO2HIP_LSTM_STEPS");
        const bool fused_steps = lstm_step_fused_width(nh) && (uintptr_t)wh % 16 == 0 && !(steps_env && std::strcmp(steps_env, "split") == 0);
        const bool seq = fused_steps && T > 1 && lstm_seq_ok(nenv, nh);
        if (seq && (rc = lstm_seq_bwd(w.Z, wh, w.C, w.CM, w.DH, w.MASK, nenv, T, nh, w.XCHG, s))) return rc;
        for (int t = T - 1; fused_steps && !seq && t >= 0; t--) {
            // dhm of step t + 1's gate gradients and step t's gate gradients in one launch
            const bool last = t == T - 1;
            if ((rc = lstm_step_bwd(w.Z + (int64_t)t * 4 * nh, last ? nullptr : w.Z + (int64_t)(t + 1) * 4 * nh, (int64_t)T * 4 * nh, wh,
                                    w.C + (int64_t)t * nh, w.CM + (int64_t)t * nh, w.DH + (int64_t)t * nh, (int64_t)T * nh, w.dcm,
                                    last ? nullptr : w.MASK + (t + 1), T, nenv, nh, s))) return rc;
        }
        for (int t = T - 1; !fused_steps && t >= 0; t--) {
            const uint8_t *mnext = t == T - 1 ? nullptr : w.MASK + (t + 1);
            if ((rc = lstm_gates_bwd(w.Z + (int64_t)t * 4 * nh, (int64_t)T * 4 * nh, w.C + (int64_t)t * nh, w.CM + (int64_t)t * nh,
                                     w.DH + (int64_t)t * nh, (int64_t)T * nh, w.dhm, kLstmBwdParts, w.dcm, mnext, T, nenv, nh, s))) return rc;
            // dhm = dz_t wh^T, as partial products that the step before sums
            if (t > 0 && (rc = fc_dgrad_parts(w.Z + (int64_t)t * 4 * nh, T * 4 * nh, wh, nh, 4 * nh, nenv, kLstmBwdParts, w.dhm, s))) return rc;
        }
        // Z now holds dz for every row.  wx first: its appended ones-row lands on the start of the (wh, b)
        // block that the next product overwrites.
        if ((rc = wgrad_into(w.F, 512, w.Z, 4 * nh, B, 512, 4 * nh, w.slabs, w.max_slabs, grads + L.off("lstm1/wx"), s))) return rc;
        if ((rc = wgrad_into(w.HM, nh, w.Z, 4 * nh, B, nh, 4 * nh, w.slabs, w.max_slabs, grads + L.off("lstm1/wh"), s))) return rc;
        if ((rc = fc_dgrad(w.Z, 4 * nh, p + L.off("lstm1/wx"), 512, 4 * nh, w.F, 512, ACT_RELU, w.dF, 512, (int)B, false, s))) return rc;
        }
    }
    // fc1
    if ((rc = wgrad_into(w.Y3, k.flat, w.dF, 512, B, k.flat, 512, w.slabs, w.max_slabs, grads + L.off("fc1/w"), s))) return rc;
    // Every product that writes a gradient tensor leaves that tensor's column sums per row tile in w.CS: they are the
    // bias gradient of the layer below, which would otherwise read the whole tensor once more (PPO2HIP_BIAS_PASS=1 does).
    const bool hand_on = !getenv("PPO2HIP_BIAS_PASS");
    int nb = 0;
    if ((rc = fc_dgrad(w.dF, 512, p + L.off("fc1/w"), k.flat, 512, w.Y3, k.flat, ACT_RELU, w.dY3, k.flat, (int)B, false, s,
                       hand_on ? w.CS : nullptr, &nb))) return rc;
    // conv3, conv2, conv1
    const ConvShape c3 = conv_shape((int)B, k.h2, k.w2, 64, 3, 1, 64), c2 = conv_shape((int)B, k.h1, k.w1, 32, 4, 2, 64),
                    c1 = conv_shape((int)B, k.H, k.W, k.C, 8, 4, 32);
    // dY3 rows are (h, w, c) with c fastest: a tile's row of `flat` sums is h3 w3 rows of 64
    if ((rc = conv_wgrad_into(w.Y2, false, w.dY3, c3, w.slabs, w.max_slabs, grads + L.off("c3/w"), s, hand_on ? w.CS : nullptr,
                              nb * k.h3 * k.w3))) return rc;
    nb = 0;
    if ((rc = conv_dgrad(w.dY3, p + L.off("c3/w"), w.Y2, w.dY2, c3, s, hand_on ? w.CS : nullptr, &nb))) return rc;
    if ((rc = conv_wgrad_into(w.Y1, false, w.dY2, c2, w.slabs, w.max_slabs, grads + L.off("c2/w"), s, hand_on ? w.CS : nullptr, nb))) return rc;
    nb = 0;
    if ((rc = conv_dgrad(w.dY2, p + L.off("c2/w"), w.Y1, w.dY1, c2, s, hand_on ? w.CS : nullptr, &nb, defer_relu1 ? w.RB1 : nullptr))) return rc;
    if ((rc = conv_wgrad_into(frames, true, w.dY1, c1, w.slabs, w.max_slabs, grads + L.off("c1/w"), s, hand_on ? w.CS : nullptr, nb, rows))) return rc;
    return 0;
}

static int cnn_policy_forward(const ppo2hip_policy_desc &d, const Layout &L, const float *p, const void *obs, int64_t B,
                              const uint8_t *masks, const float *states_in, float *pi, float *vf, float *states_out,
                              void *ws, size_t ws_bytes, hipStream_t s) {
    const CnnDims k = cnn_dims(d);
    Carver c(ws, ws_bytes);
    CnnBuf w{};
    carve_cnn(d, B, (int)B, c, w, false);
    if (!c.ok || !ws) { set_error("policy_forward: workspace too small (%zu < %zu)", ws_bytes, c.used()); return PPO2HIP_ENOSPACE; }
    int rc;
    if ((rc = cnn_trunk_forward(d, L, p, obs, w, B, s))) return rc;
    const float *feat = w.F;
    if (k.lstm) {
        if (!masks || !states_in || !states_out) { set_error("policy_forward: recurrent policy needs masks and states"); return PPO2HIP_EINVAL; }
        PPO2HIP_CHECK_HIP(hipMemcpyAsync(w.S0, states_in, (size_t)B * 2 * k.nh * sizeof(float), hipMemcpyDeviceToDevice, s));
        if ((rc = lstm_forward_all(d, L, p, w, masks, (int)B, 1, states_out, s))) return rc;
        feat = w.Hs;
    }
    if ((rc = fc_forward(feat, k.feat, p + L.off("pi/w"), p + L.off("pi/b"), pi, k.nact, (int)B, k.nact, k.feat, ACT_NONE, false, s))) return rc;
    return fc_forward(feat, k.feat, p + L.off("v/w"), p + L.off("v/b"), vf, 1, (int)B, 1, k.feat, ACT_NONE, false, s);
}

}  // namespace ppo2hip

using namespace ppo2hip;

extern "C" int ppo2hip_version(void) { return PPO2HIP_VERSION; }
extern "C" const char *ppo2hip_last_error(void) { return g_err; }

extern "C" int64_t ppo2hip_param_count(const ppo2hip_policy_desc *desc) {
    if (!desc) { set_error("param_count: null desc"); return PPO2HIP_EINVAL; }
    Layout L;
    int rc = build_layout(*desc, L);
    return rc ? rc : L.total;
}

extern "C" int ppo2hip_param_layout(const ppo2hip_policy_desc *desc, int max_vars, char (*names)[32],
                                    int64_t *offsets, int64_t *shapes) {
    if (!desc) { set_error("param_layout: null desc"); return PPO2HIP_EINVAL; }
    Layout L;
    int rc = build_layout(*desc, L);
    if (rc) return rc;
    const int n = (int)L.vars.size();
    for (int i = 0; i < n && i < max_vars; i++) {
        if (names) { std::strncpy(names[i], L.vars[i].name.c_str(), 31); names[i][31] = 0; }
        if (offsets) offsets[i] = L.vars[i].off;
        if (shapes) for (int k = 0; k < 4; k++) shapes[i * 4 + k] = L.vars[i].shape[k];
    }
    return n;
}

static bool cnn_kind(const ppo2hip_policy_desc &d) {
    return d.kind == PPO2HIP_POLICY_CNN || d.kind == PPO2HIP_POLICY_CNN_LSTM || d.kind == PPO2HIP_POLICY_CNN_LNLSTM;
}

extern "C" size_t ppo2hip_forward_workspace_bytes(const ppo2hip_policy_desc *desc, int64_t B) {
    if (!desc || B < 0) return 0;
    Carver c(nullptr, 0);
    if (desc->kind == PPO2HIP_POLICY_MLP) {
        MlpBuf w;
        carve_mlp(*desc, B, c, w, false);
        return c.used();
    }
    if (cnn_kind(*desc)) {
        CnnBuf w{};
        carve_cnn(*desc, B, (int)B, c, w, false);
        return c.used();
    }
    return 0;
}

extern "C" size_t ppo2hip_grad_workspace_bytes(const ppo2hip_policy_desc *desc, int64_t B) {
    if (!desc || B < 0) return 0;
    Carver c(nullptr, 0);
    if (desc->kind == PPO2HIP_POLICY_MLP) {
        MlpBuf w;
        carve_mlp(*desc, B, c, w, true);
        return c.used();
    }
    if (cnn_kind(*desc)) {
        CnnBuf w{};
        carve_cnn(*desc, B, (int)B, c, w, true);   // nenv <= B bounds the per-env carry buffers
        return c.used();
    }
    return 0;
}

extern "C" int ppo2hip_grad_workspace_region(const ppo2hip_policy_desc *desc, int64_t B, int32_t nsteps_mb, const char *name,
                                             size_t *offset_bytes, int64_t *nelems) {
    PPO2HIP_REQUIRE(desc && name && offset_bytes && nelems && B >= 1, "grad_workspace_region: bad argument");
    PPO2HIP_REQUIRE(cnn_kind(*desc), "grad_workspace_region: nature_cnn policies only");
    Layout L;
    int rc = build_layout(*desc, L);
    if (rc) return rc;
    const CnnDims k = cnn_dims(*desc);
    int nenv = 1;
    if (k.lstm) {
        PPO2HIP_REQUIRE(nsteps_mb >= 1 && B % nsteps_mb == 0, "grad_workspace_region: nsteps_mb must divide B");
        nenv = (int)(B / nsteps_mb);
    }
    char *const base = (char *)4096;           // never dereferenced: only the carve-outs' distances from it are used
    Carver c(base, (size_t)1 << 62);
    CnnBuf w{};
    carve_cnn(*desc, B, nenv, c, w, true);
    const float *ptr = nullptr;
    int64_t n = 0;
    if (!std::strcmp(name, "c1")) { ptr = w.Y1; n = B * k.h1 * k.w1 * 32; }
    else if (!std::strcmp(name, "c2")) { ptr = w.Y2; n = B * k.h2 * k.w2 * 64; }
    else if (!std::strcmp(name, "c3")) { ptr = w.Y3; n = B * k.flat; }
    else if (!std::strcmp(name, "fc1")) { ptr = w.F; n = B * 512; }
    else { set_error("grad_workspace_region: unknown region '%s'", name); return PPO2HIP_EINVAL; }
    *offset_bytes = (size_t)((const char *)ptr - base);
    *nelems = n;
    return PPO2HIP_OK;
}

extern "C" int ppo2hip_policy_forward(const ppo2hip_policy_desc *desc, const float *params, const void *obs,
                                      int64_t B, const uint8_t *masks, const float *states_in, float *pi,
                                      float *vf, float *states_out, void *workspace, size_t workspace_bytes,
                                      void *stream) {
    PPO2HIP_REQUIRE(desc && params && obs && pi && vf && B >= 0, "policy_forward: bad argument");
    if (B == 0) return PPO2HIP_OK;
    Layout L;
    int rc = build_layout(*desc, L);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (desc->kind == PPO2HIP_POLICY_MLP) {
        Carver c(workspace, workspace_bytes);
        MlpBuf w;
        carve_mlp(*desc, B, c, w, false);
        if (!c.ok || !workspace) { set_error("policy_forward: workspace too small (%zu < %zu)", workspace_bytes, c.used()); return PPO2HIP_ENOSPACE; }
        w.X = (float *)obs;   // observations are already a dense [B, ob_dim] block
        return mlp_forward(*desc, L, params, w, B, pi, vf, s);
    }
    if (cnn_kind(*desc))
        return cnn_policy_forward(*desc, L, params, obs, B, masks, states_in, pi, vf, states_out, workspace,
                                  workspace_bytes, s);
    set_error("policy_forward: unknown policy kind %d", desc->kind);
    return PPO2HIP_EUNSUPPORTED;
}

extern "C" int ppo2hip_neglogp(const ppo2hip_policy_desc *desc, const float *params, const float *pi,
                               const void *actions, int64_t B, float *neglogp, void *stream) {
    PPO2HIP_REQUIRE(desc && params && pi && actions && neglogp && B >= 0, "neglogp: bad argument");
    Layout L;
    int rc = build_layout(*desc, L);
    if (rc) return rc;
    const float *logstd = desc->dist == PPO2HIP_DIST_GAUSSIAN ? params + L.off("logstd") : nullptr;
    return neglogp_rows(desc->dist, desc->ac_dim, pi, logstd, actions, B, neglogp, (hipStream_t)stream);
}

extern "C" int ppo2hip_sample_actions(const ppo2hip_policy_desc *desc, const float *params, const float *pi, int64_t B,
                                      uint64_t seed, uint64_t step, int deterministic, void *actions, float *neglogp,
                                      float *noise_out, void *stream) {
    PPO2HIP_REQUIRE(desc && params && pi && actions && neglogp && B >= 0, "sample_actions: bad argument");
    PPO2HIP_REQUIRE(B <= 0xffffffffll, "sample_actions: more than 2^32 rows");
    Layout L;
    int rc = build_layout(*desc, L);
    if (rc) return rc;
    const float *logstd = desc->dist == PPO2HIP_DIST_GAUSSIAN ? params + L.off("logstd") : nullptr;
    return sample_rows(desc->dist, desc->ac_dim, pi, logstd, B, seed, step, deterministic, actions, neglogp, noise_out,
                       (hipStream_t)stream);
}

extern "C" size_t ppo2hip_act_workspace_bytes(const ppo2hip_policy_desc *desc, int64_t B) {
    if (!desc || B < 0) return 0;
    return align_up(ppo2hip_forward_workspace_bytes(desc, B), 256) + align_up((size_t)B * (size_t)desc->ac_dim * sizeof(float), 256);
}

extern "C" int ppo2hip_act_step(const ppo2hip_policy_desc *desc, const float *params, const void *obs, int64_t B,
                                const uint8_t *masks, const float *states_in, uint64_t seed, uint64_t step,
                                int deterministic, void *actions, float *values, float *neglogp, float *states_out,
                                void *workspace, size_t workspace_bytes, void *stream) {
    PPO2HIP_REQUIRE(desc && params && obs && actions && values && neglogp && B >= 0, "act_step: bad argument");
    if (B == 0) return PPO2HIP_OK;
    const size_t fwd = align_up(ppo2hip_forward_workspace_bytes(desc, B), 256);
    const size_t need = fwd + align_up((size_t)B * (size_t)desc->ac_dim * sizeof(float), 256);
    if (!workspace || workspace_bytes < need) {
        set_error("act_step: workspace too small (%zu < %zu)", workspace_bytes, need);
        return PPO2HIP_ENOSPACE;
    }
    float *pi = (float *)((char *)workspace + fwd);      // the distribution parameters never leave the workspace
    int rc = ppo2hip_policy_forward(desc, params, obs, B, masks, states_in, pi, values, states_out, workspace, fwd, stream);
    if (rc) return rc;
    return ppo2hip_sample_actions(desc, params, pi, B, seed, step, deterministic, actions, neglogp, nullptr, stream);
}

static int ppo_grad_impl(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss, const float *params,
                         const ppo2hip_rollout *rollout, const int32_t *idx, int64_t B,
                         const int32_t *mb_envs, int32_t nsteps_mb, const double *stats, float cliprange,
                         double inv_count, float *grads, float *losses, void *workspace,
                         size_t workspace_bytes, void *stream, AdamFuse *adam, bool *adam_done) {
    PPO2HIP_REQUIRE(desc && loss && params && rollout && grads && losses && workspace && B >= 1,
                    "ppo_grad: bad argument");
    PPO2HIP_REQUIRE(rollout->obs && rollout->actions && rollout->returns && rollout->values &&
                    rollout->neglogpacs && rollout->advs, "ppo_grad: rollout has null arrays");
    PPO2HIP_REQUIRE(rollout->T == 0 || (rollout->T > 0 && rollout->E > 0), "ppo_grad: bad rollout shape");
    PPO2HIP_REQUIRE(rollout->n > 0 && (rollout->T == 0 || rollout->n == (int64_t)rollout->T * rollout->E),
                    "ppo_grad: rollout.n=%lld does not match T=%d E=%d", (long long)rollout->n, rollout->T, rollout->E);
    PPO2HIP_REQUIRE(idx || B <= rollout->n, "ppo_grad: B=%lld rows asked of a %lld-row rollout", (long long)B,
                    (long long)rollout->n);
    Layout L;
    int rc = build_layout(*desc, L);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (desc->kind == PPO2HIP_POLICY_MLP)
        return mlp_grad(*desc, *loss, L, params, *rollout, idx, B, stats, cliprange, inv_count, grads, losses,
                        workspace, workspace_bytes, s, adam, adam_done);
    if (cnn_kind(*desc))
        return cnn_grad(*desc, *loss, L, params, *rollout, idx, B, mb_envs, nsteps_mb, stats, cliprange, inv_count, grads,
                        losses, workspace, workspace_bytes, s);
    set_error("ppo_grad: unknown policy kind %d", desc->kind);
    return PPO2HIP_EUNSUPPORTED;
}

extern "C" int ppo2hip_ppo_grad(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss, const float *params,
                                const ppo2hip_rollout *rollout, const int32_t *idx, int64_t B,
                                const int32_t *mb_envs, int32_t nsteps_mb, const double *stats, float cliprange,
                                double inv_count, float *grads, float *losses, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return ppo_grad_impl(desc, loss, params, rollout, idx, B, mb_envs, nsteps_mb, stats, cliprange, inv_count, grads, losses,
                         workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int ppo2hip_ppo_epoch(const ppo2hip_policy_desc *desc, const ppo2hip_loss_desc *loss, float *params,
                                 float *grads, float *adam_m, float *adam_v, const ppo2hip_rollout *rollout,
                                 const int32_t *idx, int64_t N, int64_t nbatch_train, const double *stats,
                                 float cliprange, double lr, int64_t t_first, double beta1, double beta2,
                                 double epsilon, double max_grad_norm, float *losses, void *workspace,
                                 size_t workspace_bytes, void *adam_scratch, void *stream) {
    PPO2HIP_REQUIRE(desc && idx && adam_m && adam_v && adam_scratch && losses, "ppo_epoch: null pointer");
    PPO2HIP_REQUIRE(nbatch_train >= 1 && N >= nbatch_train && N % nbatch_train == 0,
                    "ppo_epoch: N=%lld is not a multiple of nbatch_train=%lld", (long long)N, (long long)nbatch_train);
    PPO2HIP_REQUIRE(desc->kind == PPO2HIP_POLICY_MLP || desc->kind == PPO2HIP_POLICY_CNN,
                    "ppo_epoch: feed-forward policies only (recurrent minibatches are env sequences)");
    const int64_t nmb = N / nbatch_train;
    const int64_t P = ppo2hip_param_count(desc);
    if (P < 0) return PPO2HIP_EINVAL;
    for (int64_t j = 0; j < nmb; j++) {
        // where the gradient route can, the optimiser step of the minibatch rides in its last launch
        const int64_t t = t_first + j;
        AdamFuse af{};
        af.theta = params; af.m = adam_m; af.v = adam_v;
        af.a = (float)(lr * sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t)));
        af.beta1 = (float)beta1; af.beta2 = (float)beta2; af.omb1 = (float)(1.0 - beta1); af.omb2 = (float)(1.0 - beta2);
        af.eps = (float)epsilon; af.gscale = 1.f; af.max_norm = (float)max_grad_norm; af.header = (float *)adam_scratch;
        bool folded = false;
        int rc = ppo_grad_impl(desc, loss, params, rollout, idx + j * nbatch_train, nbatch_train, nullptr, 0,
                               stats ? stats + 3 * j : nullptr, cliprange, 1.0 / (double)nbatch_train, grads,
                               losses + 8 * j, workspace, workspace_bytes, stream, t >= 1 ? &af : nullptr, &folded);
        if (rc) return rc;
        if (folded) continue;
        rc = ppo2hip_adam_step(params, grads, adam_m, adam_v, P, lr, t, beta1, beta2, epsilon, 1.0,
                               max_grad_norm, adam_scratch, stream);
        if (rc) return rc;
    }
    return PPO2HIP_OK;
}
