// K7 (COMA counterfactual advantage) and K8 (TD(lambda) targets): small latency-bound kernels; one
// wavefront handles several rows / one thread handles one (chain, t).
#include <algorithm>

#include "ippm_internal.h"

// actor/learner.py:55-83: pi~ = pi*mask / max(sum, 1e-5), floor 1e-5; baseline = sum_a pi~(a) Q(a) mask(a);
// advantage = Q(chosen) - baseline.  A <= 27 <= 32: one 32-lane half-wave per row.
__global__ void __launch_bounds__(256)
k_coma_advantage(const float* __restrict__ probs, const float* __restrict__ q, const uint8_t* __restrict__ mask,
                 const int32_t* __restrict__ action, float* __restrict__ adv, float* __restrict__ pi_tilde, int A, int batch) {
  const int lane = threadIdx.x & 31;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;
  if (row >= batch) return;
  const bool on = lane < A;
  const float m = on ? (float)mask[(size_t)row * A + lane] : 0.f;
  const float qv = on ? q[(size_t)row * A + lane] : 0.f;
  float p = on ? probs[(size_t)row * A + lane] * m : 0.f;
  float s = p;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
  s = s < 1e-5f ? 1e-5f : s;
  float pn = p / s;
  pn = pn <= 1e-5f ? 1e-5f : pn;
  float b = on ? pn * qv * m : 0.f;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) b += __shfl_xor(b, o, 32);
  if (pi_tilde && on) pi_tilde[(size_t)row * A + lane] = pn;
  if (lane == 0) adv[row] = q[(size_t)row * A + action[row]] - b;
}

// batch_memory.py:120-162 restated per (chain, t) in O(len): the n-step return grows by one reward per n,
// accumulation stops ("leave") at the first transition whose predecessor is terminal.
__global__ void k_td_lambda(const float* __restrict__ reward, const uint8_t* __restrict__ done, const float* __restrict__ q_sel,
                            float* __restrict__ td, float* __restrict__ dr, double gamma, double lam, int chains, int len) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= chains * len) return;
  const int ch = idx / len, t = idx % len;
  const float* r = reward + (size_t)ch * len;
  const uint8_t* d = done + (size_t)ch * len;
  const float* qs = q_sel + (size_t)ch * len;
  double total = 0.0, g = 0.0, disc = 0.0;
  double gpow = 1.0;    // gamma^(n-1)
  double lpow = 1.0;    // lambda^(n-1)
  for (int n = 1; n <= len - t; ++n) {
    const int l = n - 1;  // the new term of this n
    const bool ok = (t + l == 0) || !d[t + l - 1];
    if (!ok) {            // leave: weight lambda^n on the partial return, then stop
      total += lpow * lam * g;
      disc = g;
      break;
    }
    g += gpow * (double)r[t + l];
    disc = g;
    double gn = g;
    if (t + n < len && !(d[t + n] || (t + n + 1 >= len))) gn += gpow * gamma * (double)qs[t + n];
    total += lpow * gn;
    gpow *= gamma;
    lpow *= lam;
  }
  td[idx] = (float)((1.0 - lam) * total);
  dr[idx] = (float)disc;
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int ippm_coma_advantage(ippm_ctx* ctx, const float* probs, const float* q, const uint8_t* mask,
                                   const int32_t* action, float* advantage, float* pi_tilde, int32_t batch, void* stream) {
  if (!ctx || !probs || !q || !mask || !action || !advantage) { ippm_set_error("ippm_coma_advantage: null argument"); return -1; }
  const int A = ctx->cfg.n_actions;
  const int rows_per_block = 256 / 32;
  hipLaunchKernelGGL(k_coma_advantage, dim3((batch + rows_per_block - 1) / rows_per_block), dim3(256), 0, S_(stream), probs, q,
                     mask, action, advantage, pi_tilde, A, batch);
  IPPM_LAUNCH_CHECK("coma_advantage");
  return 0;
}

extern "C" int ippm_td_lambda(ippm_ctx* ctx, const float* reward, const uint8_t* done, const float* q_sel, float* td_target,
                              float* disc_return, int32_t chains, int32_t len, void* stream) {
  if (!ctx || !reward || !done || !q_sel || !td_target || !disc_return) { ippm_set_error("ippm_td_lambda: null argument"); return -1; }
  const int total = chains * len;
  hipLaunchKernelGGL(k_td_lambda, dim3((total + 255) / 256), dim3(256), 0, S_(stream), reward, done, q_sel, td_target,
                     disc_return, ctx->cfg.gamma, ctx->cfg.lambda_, chains, len);
  IPPM_LAUNCH_CHECK("td_lambda");
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// col2im for the input gradient of a stride-1, unpadded K x K convolution computed as a GEMM
// (ippmarl/networks.py::_ConvDataGradAsGemm):  cols [B*Ho*Wo, K*K*C] = grad_out [B*Ho*Wo, O] x W [O, (ky, kx, c)]
//   grad_x[b, y, x, c] = sum over taps (ky, kx) with 0 <= y-ky < Ho, 0 <= x-kx < Wo of cols[(b, y-ky, x-kx), (ky, kx, c)]
// channels-last on both sides, 16 bytes per lane, every element of `cols` read exactly once.  (torch's col2im kernel takes
// 8.6 ms for the 3.2 GB of a 12 288-sample minibatch of conv2 -- 0.37 TB/s; this is a plain gather at streaming rate.)
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_col2im_nhwc(const float4* __restrict__ cols, float4* __restrict__ grad_x, int B, int Ho, int Wo, int K, int C4) {
  const int H = Ho + K - 1, W = Wo + K - 1;
  const size_t total = (size_t)B * H * W * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    size_t r = i / C4;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = max(0, y - Ho + 1); ky <= min(K - 1, y); ++ky)
      for (int kx = max(0, x - Wo + 1); kx <= min(K - 1, x); ++kx) {
        const size_t row = ((size_t)b * Ho + (y - ky)) * Wo + (x - kx);
        const float4 v = cols[(row * K * K + (size_t)ky * K + kx) * C4 + c];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    grad_x[i] = acc;
  }
}

extern "C" int ippm_col2im_nhwc(const float* cols, float* grad_x, int32_t batch, int32_t out_h, int32_t out_w, int32_t kernel,
                                int32_t channels, void* stream) {
  if (!cols || !grad_x) { ippm_set_error("ippm_col2im_nhwc: null argument"); return -1; }
  if (channels % 4 || kernel < 1 || out_h < 1 || out_w < 1 || ((reinterpret_cast<uintptr_t>(cols) | reinterpret_cast<uintptr_t>(grad_x)) & 15)) {
    ippm_set_error("ippm_col2im_nhwc: needs channels % 4 == 0 and 16-byte aligned buffers");
    return -1;
  }
  if (batch <= 0) return 0;
  const size_t total = (size_t)batch * (out_h + kernel - 1) * (out_w + kernel - 1) * (channels / 4);
  const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 1u << 20);
  hipLaunchKernelGGL(k_col2im_nhwc, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(cols), reinterpret_cast<float4*>(grad_x), batch, out_h, out_w, kernel, channels / 4);
  IPPM_LAUNCH_CHECK("col2im_nhwc");
  return 0;
}
