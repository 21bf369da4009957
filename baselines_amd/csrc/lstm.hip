// Element-wise halves of the LSTM cell of baselines/a2c/utils.py:86-108 (gates in the order
// i, f, o, u; state [c | h]; c and h zeroed where the done-before-step mask is set).  The
// matrix products (x wx for all steps at once, h wh per step, and their gradients) run on the
// MFMA GEMM; these kernels do the masking, the gate non-linearities and their backward.
// Rows are env-major: row(e, t) = e * T + t (batch_to_seq, a2c/utils.py:70-75).
#include "kernels.hpp"

namespace ppo2hip {

__device__ inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// cm = c_prev * (1 - m), hm = h_prev * (1 - m) for step t of every env
__global__ void lstm_mask_kernel(const float *__restrict__ c_prev, const float *__restrict__ h_prev,
                                 int64_t prev_stride, const uint8_t *__restrict__ masks, int64_t mask_stride,
                                 int nenv, int nh, float *__restrict__ cm, float *__restrict__ hm, int64_t out_stride) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nenv * nh) return;
    const int e = (int)(g / nh), j = (int)(g - (int64_t)e * nh);
    const float keep = masks[(int64_t)e * mask_stride] ? 0.f : 1.f;
    cm[(int64_t)e * out_stride + j] = c_prev[(int64_t)e * prev_stride + j] * keep;
    hm[(int64_t)e * out_stride + j] = h_prev[(int64_t)e * prev_stride + j] * keep;
}

// z rows hold x wx + hm wh + b on entry and the activated gates (i, f, o, u) on exit
__global__ void lstm_gate_kernel(float *__restrict__ z, int64_t z_stride, const float *__restrict__ cm,
                                 int64_t s_stride, int nenv, int nh, float *__restrict__ c, float *__restrict__ h,
                                 float *__restrict__ state_out /* [nenv, 2nh] or null */) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nenv * nh) return;
    const int e = (int)(g / nh), j = (int)(g - (int64_t)e * nh);
    float *zr = z + (int64_t)e * z_stride;
    const float i = sigmoidf(zr[j]), f = sigmoidf(zr[nh + j]), o = sigmoidf(zr[2 * nh + j]), u = tanhf(zr[3 * nh + j]);
    const float cn = f * cm[(int64_t)e * s_stride + j] + i * u;
    const float hn = o * tanhf(cn);
    zr[j] = i; zr[nh + j] = f; zr[2 * nh + j] = o; zr[3 * nh + j] = u;
    c[(int64_t)e * s_stride + j] = cn;
    h[(int64_t)e * s_stride + j] = hn;
    if (state_out) {
        state_out[(int64_t)e * 2 * nh + j] = cn;
        state_out[(int64_t)e * 2 * nh + nh + j] = hn;
    }
}

// Backward of one step.  dh = dH_head[t] + dhm_next * (1 - m_{t+1}), dc_in = dcm_next * (1 - m_{t+1}).
// Writes dz over the gate rows and the new (unmasked) dcm carry.
__global__ void lstm_gate_bwd_kernel(float *__restrict__ gates, int64_t z_stride, const float *__restrict__ c,
                                     const float *__restrict__ cm, const float *__restrict__ dh_head,
                                     int64_t s_stride, const float *__restrict__ dhm_next,
                                     float *__restrict__ dcm_carry, const uint8_t *__restrict__ mask_next,
                                     int64_t mask_stride, int nenv, int nh) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)nenv * nh) return;
    const int e = (int)(g / nh), j = (int)(g - (int64_t)e * nh);
    float keep = 0.f;   // no later step: no carry
    if (mask_next) keep = mask_next[(int64_t)e * mask_stride] ? 0.f : 1.f;
    const int64_t sr = (int64_t)e * s_stride + j, cr = (int64_t)e * nh + j;
    float *zr = gates + (int64_t)e * z_stride;
    const float i = zr[j], f = zr[nh + j], o = zr[2 * nh + j], u = zr[3 * nh + j];
    const float dh = dh_head[sr] + (mask_next ? dhm_next[cr] * keep : 0.f);
    const float tc = tanhf(c[sr]);
    const float dc = (mask_next ? dcm_carry[cr] * keep : 0.f) + dh * o * (1.f - tc * tc);
    zr[j] = dc * u * i * (1.f - i);
    zr[nh + j] = dc * cm[sr] * f * (1.f - f);
    zr[2 * nh + j] = dh * tc * o * (1.f - o);
    zr[3 * nh + j] = dc * i * (1.f - u * u);
    dcm_carry[cr] = dc * f;
}

int lstm_mask(const float *c_prev, const float *h_prev, int64_t prev_stride, const uint8_t *masks, int64_t mask_stride,
              int nenv, int nh, float *cm, float *hm, int64_t out_stride, hipStream_t s) {
    const int64_t n = (int64_t)nenv * nh;
    hipLaunchKernelGGL(lstm_mask_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, c_prev, h_prev, prev_stride,
                       masks, mask_stride, nenv, nh, cm, hm, out_stride);
    return check_launch("lstm_mask_kernel");
}

int lstm_gates(float *z, int64_t z_stride, const float *cm, int64_t s_stride, int nenv, int nh, float *c, float *h,
               float *state_out, hipStream_t s) {
    const int64_t n = (int64_t)nenv * nh;
    hipLaunchKernelGGL(lstm_gate_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, z, z_stride, cm, s_stride,
                       nenv, nh, c, h, state_out);
    return check_launch("lstm_gate_kernel");
}

int lstm_gates_bwd(float *gates, int64_t z_stride, const float *c, const float *cm, const float *dh_head,
                   int64_t s_stride, const float *dhm_next, float *dcm_carry, const uint8_t *mask_next,
                   int64_t mask_stride, int nenv, int nh, hipStream_t s) {
    const int64_t n = (int64_t)nenv * nh;
    hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, gates, z_stride, c, cm,
                       dh_head, s_stride, dhm_next, dcm_carry, mask_next, mask_stride, nenv, nh);
    return check_launch("lstm_gate_bwd_kernel");
}

}  // namespace ppo2hip
