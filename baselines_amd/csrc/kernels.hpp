// Internal launch interface between the translation units of libppo2hip.
#pragma once
#include "common.hpp"

namespace ppo2hip {

enum Act { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2 };

constexpr int kMaxR = 8;                       // value / reward channels a scan or bootstrap handles
struct GammaVec { double g[kMaxR]; };          // per-channel discount, passed by value to kernels
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// ---- dense layers on the fp32 MFMA GEMM (gemm.hip) -------------------------------------
// Y[M,N] = act(X[M,K] W[K,N] + b)      (accumulate: Y += X W (+ b), act must be ACT_NONE)
int fc_forward(const float *X, int ldx, const float *W, const float *b, float *Y, int ldy, int M, int N,
               int K, int act, bool accumulate, hipStream_t s);
// dX[M,Kin] = (dY[M,N] W^T) * act'(H)   H = saved activation of the layer that produced X (or null)
int fc_dgrad(const float *dY, int lddy, const float *W, int Kin, int N, const float *H, int ldh, int act,
             float *dX, int lddx, int M, bool accumulate, hipStream_t s, float *colsum_tiles = nullptr,
             int *ntiles = nullptr);
//   colsum_tiles (optional): [*ntiles][Kin] column sums of dX per row tile — the bias gradient of the layer dX came
//   out of, without a second pass over dX (add the rows with reduce_slabs)
// slabs[z][(Kin+1), N]: rows 0..Kin-1 = X^T dY, row Kin = column sums of dY (the bias gradient; left unwritten with
// bias_rows = false where it would take a pass of its own), each over the z-th slice of the S sample rows.
// Returns the number of slabs written in *nslabs.
int fc_wgrad_slabs(const float *X, int ldx, const float *dY, int lddy, int64_t S, int Kin, int N,
                   float *slabs, int max_slabs, int *nslabs, hipStream_t s, bool bias_rows = true);
int wgrad_max_slabs(int64_t S, int Kin, int N);
// split-k partial products for few-row operands: parts[z][M, N] = X[:, k-slice z] W[k-slice z, :]
int fc_forward_parts(const float *X, int ldx, const float *W, int M, int N, int K, int nparts, float *parts, hipStream_t s);
// parts[z][M, Kin] = dY[:, n-slice z] W[:, n-slice z]^T
int fc_dgrad_parts(const float *dY, int lddy, const float *W, int Kin, int N, int M, int nparts, float *parts, hipStream_t s);
// true when the bias row is better summed on its own than folded into the product as a row of ones
bool wgrad_bias_separately(int Kin);
// row Kin of every slab = column sums of dY over that slab's sample rows
int slab_bias_rows(const float *dY, int lddy, int64_t S, int Kin, int N, float *slabs, int nslabs, hipStream_t s);
// dst[n] = sum of the S rows of rows[S, N]; scratch: 256 N floats
int colsum_rows(const float *rows, int64_t S, int N, float *scratch, float *dst, hipStream_t s);
// dst[i] (+)= sum_z slabs[z * stride + i] for i < size (stride 0: slabs of `size` elements back to back)
int reduce_slabs(const float *slabs, int nslabs, int64_t size, float *dst, bool accumulate, hipStream_t s, int64_t stride = 0);

// ---- convolution as implicit GEMM (conv.hip) -------------------------------------------
struct ConvShape {
    int B, IH, IW, Cin, KH, KW, stride, OH, OW, Cout;
};
// Y[B,OH,OW,Cout] = relu(conv(X) + b); X float32 NHWC, or uint8 NHWC scaled by 1/255 (x_u8)
//   rows (uint8 only, optional): frame i of the batch is frame rows[i] of X — only where conv_u8_direct_ok(c) holds
//   relu_bits (optional, same condition): one word per output pixel, bit co = (Y[pixel][co] > 0)
int conv_forward(const void *X, bool x_u8, const float *W, const float *b, float *Y, const ConvShape &c,
                 hipStream_t s, const int32_t *rows = nullptr, uint32_t *relu_bits = nullptr);
// the uint8 layer on the bf16 matrix pipe with float32-exact products (8 x 8 kernels of 32-byte rows, 32 channels)
bool conv_u8_forward_ok(const void *X, const ConvShape &c);
int conv_u8_forward(const void *X, const float *W, const float *b, float *Y, const ConvShape &c, hipStream_t s,
                    const int32_t *rows, uint32_t *relu_bits);
bool conv_u8_wgrad_ok(const void *X, const float *dY, const ConvShape &c);
int conv_u8_wgrad_slabs(const void *X, const float *dY, const ConvShape &c, float *slabs, int max_slabs, int *nslabs,
                        hipStream_t s, bool bias_rows, const int32_t *rows);
// both uint8 kernels take this layer: its frames can then be read where they lie (a list of frame numbers instead of a
// gathered copy)
bool conv_u8_direct_ok(const void *X, const ConvShape &c);
// slabs[z][(KH*KW*Cin+1), Cout] as fc_wgrad_slabs
int conv_wgrad_slabs(const void *X, bool x_u8, const float *dY, const ConvShape &c, float *slabs,
                     int max_slabs, int *nslabs, hipStream_t s, bool bias_rows = true, const int32_t *rows = nullptr);
int conv_wgrad_max_slabs(const ConvShape &c);
// dX[B,IH,IW,Cin] = conv_transpose(dY, W) * relu'(Xact)   (Xact = the layer's float32 input activation)
//   relu_bits (optional, Cin = 32): bit i of the array = (Xact[i] > 0), read in place of Xact — 1/32 of the bytes
//   colsum_rows (optional): on return *nrows rows of Cin column sums of dX (one per row tile and parity class), or
//   *nrows = 0 when the route taken does not produce them; max_rows = conv_dgrad_colsum_rows(c)
int conv_dgrad(const float *dY, const float *W, const float *Xact, float *dX, const ConvShape &c, hipStream_t s,
               float *colsum_rows = nullptr, int *nrows = nullptr, const uint32_t *relu_bits = nullptr);
int64_t conv_dgrad_colsum_rows(const ConvShape &c);

// ---- gather (gather.hip) ---------------------------------------------------------------
// device address of this GPU's fault word (a static device array of the library; symbol lookup on the first call per
// device, no allocation); null with an error set when the lookup fails
unsigned *fault_word();
// compute units of the current device (0 if unknown): the bound for launches whose workgroups wait for each other
int device_cu_count();
// how many samples an index into a rollout may name
inline int64_t rollout_samples(const ppo2hip_rollout &ro) { return ro.T > 0 ? (int64_t)ro.T * ro.E : ro.n; }

struct GatherArgs {
    const ppo2hip_rollout *ro;
    const int32_t *idx;
    int64_t B;
    int64_t ob_bytes;   // bytes per observation row
    int64_t ac_bytes;   // bytes per action row
    int nvalue;
    const double *stats;  // {sum, sumsq, count} or null
    void *obs;
    void *actions;
    float *returns, *values, *neglogpacs, *advs;
    uint8_t *masks;      // may be null
    int32_t *obs_rows = nullptr;   // when set: the observation rows are NOT copied; row numbers (into ro->obs) are left here instead
};
int gather_minibatch(const GatherArgs &g, hipStream_t s);
int gather_states(const float *states, const int32_t *envs, int nenv_mb, int width, float *out, hipStream_t s);

// ---- distributions + PPO loss (loss.hip) -----------------------------------------------
struct LossArgs {
    int dist, ac_dim, nvalue;
    int64_t B;
    const float *pi;       // [B, ac_dim]
    const float *logstd;   // [ac_dim] (Gaussian)
    const float *vf;       // [B, nvalue]
    const void *actions;
    const float *advs, *returns, *oldv, *oldnlp;
    float cliprange;
    ppo2hip_loss_desc cfg;
    double inv_count;
    float *dpi;            // [B, ac_dim] out
    float *dvf;            // [B, nvalue] out
    double *partial;       // scratch: loss_partial_count(B) * 16 doubles
    float *losses;         // [8] out (device)
    float *dlogstd;        // [ac_dim] out (Gaussian)
};
int loss_partial_blocks(int64_t B);
int ppo_loss_backward(const LossArgs &a, hipStream_t s);
int neglogp_rows(int dist, int ac_dim, const float *pi, const float *logstd, const void *actions, int64_t B,
                 float *out, hipStream_t s);

// actions and their neglogp drawn from pi with counter-based noise named by (seed, step)  (sample.hip)
int sample_rows(int dist, int ac_dim, const float *pi, const float *logstd, int64_t B, uint64_t seed, uint64_t step,
                int deterministic, void *actions, float *neglogp, float *noise_out, hipStream_t s);

int loss_final(const double *partial, int nblocks, int A_gaussian, double inv_count, float *losses, float *dlogstd,
               hipStream_t s);

// ---- fused two-trunk MLP minibatch gradient (mlp_fused.hip) -----------------------------
struct FusedOffsets {
    int64_t a_w1, a_w2, a_wh, v_w1, v_w2, v_wh, logstd;   // offsets of the (w, b) pairs / logstd in the arena
};
// where a fused kernel's per-workgroup gradient image keeps each block, and the head's row width
struct FusedImage {
    int64_t slab_elems;
    int s_w1, s_w2, s_wh, s_bh, head_cols;
    int s_logstd;   // where the image holds the summed logstd gradient, or -1: it is in the loss partial rows
};
// Optimiser step folded into the finalize launch (single process, small arenas): every block that has summed its 64
// gradient elements meets the others at a grid barrier over the per-block sums of squares and then applies
// clip + Adam to those same elements, so a minibatch costs two launches instead of three.
struct AdamFuse {
    float *theta, *m, *v;
    float a, beta1, beta2, omb1, omb2, eps, gscale, max_norm;
    float *header;            // adam scratch header: [0] global norm, [1] applied scale
    unsigned long long *words; // [2 x blocks] per-block sums of squares as tagged halves (workspace of the fused gradient kernel)
    unsigned *fault;          // the device's fault word: a block that gives up waiting raises kFaultGridSync there
    unsigned tag;             // this launch's tag (never zero, never the tag of the launch before on the same workspace)
    int max_spins;            // polls before a lane gives up (negative: give up at once — a debugging aid, PPO2HIP_GRID_SYNC_SPINS)
};
size_t fused_adam_extra_bytes();
// slabs: nslabs_policy gradient images of the policy trunk, then nslabs_value of the value trunk
int fused_finalize(const FusedImage &im, const float *slabs, int nslabs_policy, int nslabs_value, const double *lpart, int nloss_rows,
                   const ppo2hip_policy_desc &d, const FusedOffsets &fo, double inv_count, float *grads, float *losses,
                   hipStream_t s, const AdamFuse *adam = nullptr);
bool mlp_fused16_supported(const ppo2hip_policy_desc &d);
// every rollout array and the index list stay under 4 GB (the kernel addresses them with 32-bit byte offsets)
bool mlp_fused16_fits(const ppo2hip_policy_desc &d, const ppo2hip_rollout &ro, int64_t B);
size_t mlp_fused16_workspace_bytes();
int mlp_fused16_grad(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const float *params,
                     const FusedOffsets &fo, const ppo2hip_rollout &ro, const int32_t *idx, int64_t B,
                     const double *stats, float cliprange, double inv_count, float *grads, float *losses, void *ws,
                     size_t ws_bytes, hipStream_t s, AdamFuse *adam = nullptr);
// ---- fused forward / loss / input-gradient chain of the 256-128 two-trunk MLP (mlp_fused256.hip) ---------------
struct Fused256Buffers {
    // rollout arrays read in place (time-major with T, E set and an index list) or already gathered (T = 0, idx null,
    // advantages normalised: stats null); n rows each
    const float *obs;
    const void *actions;
    const float *returns, *values, *neglogpacs;
    const void *advs;
    int adv_f64, T, E;
    int64_t n;
    const int32_t *idx;
    const double *stats;
    float *xg;                                // [B, ob_dim] scratch for the gathered observation rows, or null when obs is gathered already
    float *H1[2], *H2[2], *DZ1[2], *DZ2[2], *DOUT[2];      // per trunk (policy, value)
    double *loss_partial;                     // mlp_fused256_loss_bytes()
    float *wslabs;                            // mlp_fused256_wgrad_bytes(): per-workgroup weight-gradient sums
};
bool mlp_fused256_supported(const ppo2hip_policy_desc &d);
size_t mlp_fused256_loss_bytes();
size_t mlp_fused256_wgrad_bytes();
// every weight and bias gradient of both trunks from the rows mlp_fused256_chain wrote (two launches)
int mlp_fused256_wgrad(const ppo2hip_policy_desc &d, const FusedOffsets &fo, const Fused256Buffers &b, int64_t B, float *grads,
                       hipStream_t s);
int mlp_fused256_chain(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const float *params, const FusedOffsets &fo,
                       const Fused256Buffers &b, int64_t B, float cliprange, double inv_count, float *losses, float *dlogstd,
                       hipStream_t s);
// loss scalars (and the logstd gradient riding at 8 + o) from per-workgroup partial rows of 8 + 64 doubles
int fused_loss_finalize(const double *lpart, int nrows, int A_gaussian, double inv_count, float *losses, float *dlogstd,
                        hipStream_t s);
bool mlp_fused_supported(const ppo2hip_policy_desc &d);
size_t mlp_fused_workspace_bytes();
int mlp_fused_grad(const ppo2hip_policy_desc &d, const ppo2hip_loss_desc &lc, const float *params,
                   const FusedOffsets &fo, const ppo2hip_rollout &ro, const int32_t *idx, int64_t B,
                   const double *stats, float cliprange, double inv_count, float *grads, float *losses, void *ws,
                   size_t ws_bytes, hipStream_t s);

// ---- LSTM (lstm.hip): element-wise halves of the cell; products go through the GEMM -----
int lstm_mask(const float *c_prev, const float *h_prev, int64_t prev_stride, const uint8_t *masks, int64_t mask_stride,
              int nenv, int nh, float *cm, float *hm, int64_t out_stride, hipStream_t s);
// The per-step products (hm wh forward, dz wh^T backward) have few rows, so they are split over k into `parts`
// partial products [parts][nenv, width] that the element-wise kernel of the same step sums in a fixed order.
constexpr int kLstmFwdParts = 4, kLstmBwdParts = 8;
// z rows hold x wx + b; zparts the partial hm wh products.  Also writes the NEXT step's masked states
// (cm_next = c (1 - m_next), hm_next likewise) when mask_next is given.
int lstm_gates(float *z, int64_t z_stride, const float *zparts, int nparts, const float *cm, int64_t s_stride, int nenv,
               int nh, float *c, float *h, float *state_out, const uint8_t *mask_next, int64_t mask_stride, float *cm_next,
               float *hm_next, hipStream_t s);
// dhm_parts: partial products of dz_{t+1} wh^T, [nparts][nenv, nh]
int lstm_gates_bwd(float *gates, int64_t z_stride, const float *c, const float *cm, const float *dh_head,
                   int64_t s_stride, const float *dhm_parts, int nparts, float *dcm_carry, const uint8_t *mask_next,
                   int64_t mask_stride, int nenv, int nh, hipStream_t s);

// one launch per step and direction (product + element-wise half) for the widths lstm_step_fused_width() names
bool lstm_step_fused_width(int nh);
// all T steps in one launch (the workgroups of a group of 16 envs hand h round through tagged 64-bit words);
// xchg: lstm_seq_xchg_bytes(nenv, nh) of device scratch (sized for the backward launch, which hands the gate gradients round)
bool lstm_seq_ok(int nenv, int nh);
size_t lstm_seq_xchg_bytes(int nenv, int nh);
int lstm_seq_fwd(float *z, const float *wh, float *hm, float *cm, int nenv, int T, int nh, float *c, float *h, float *state_out,
                 const uint8_t *masks, void *xchg, hipStream_t s);
int lstm_seq_bwd(float *gates, const float *wh, const float *c, const float *cm, const float *dh_head, const uint8_t *masks,
                 int nenv, int T, int nh, void *xchg, hipStream_t s);
int lstm_step_fwd(float *z, int64_t z_stride, const float *wh, const float *hm, const float *cm, int64_t s_stride, int nenv,
                  int nh, float *c, float *h, float *state_out, const uint8_t *mask_next, int64_t mask_stride, float *cm_next,
                  float *hm_next, hipStream_t s);
int lstm_step_bwd(float *gates, const float *dz_next, int64_t z_stride, const float *wh, const float *c, const float *cm,
                  const float *dh_head, int64_t s_stride, float *dcm_carry, const uint8_t *mask_next, int64_t mask_stride,
                  int nenv, int nh, hipStream_t s);
// layer-normalised cell (a2c/utils.py:110-147); see lstm.hip for the buffer meanings
struct LnLstmParams {
    const float *gx, *bx, *gh, *bh, *b, *gc, *bc;
};
int ln_rows(float *x, int64_t stride, int64_t rows, int n, float *rstd, hipStream_t s);
int lnlstm_step_fwd(const float *zx, float *zh, int64_t z_stride, float *gates, const LnLstmParams &p, const float *cm,
                    int64_t s_stride, int nenv, int nh, float *c, float *cn, float *h, float *rstd_h, float *rstd_c,
                    int64_t r_stride, float *state_out, hipStream_t s);
int lnlstm_step_bwd(float *gates, const float *zx, const float *zh, float *dzx, float *dzh, int64_t z_stride,
                    const LnLstmParams &p, const float *cm, const float *cn, const float *dh_head, float *dl,
                    int64_t s_stride, const float *rstd_x, const float *rstd_h, const float *rstd_c, int64_t r_stride,
                    const float *dhm_next, float *dcm_carry, const uint8_t *mask_next, int64_t mask_stride, int nenv,
                    int nh, hipStream_t s);
int ln_colsum(const float *a, const float *b, int64_t rows, int n, float *out, hipStream_t s);

}  // namespace ppo2hip
