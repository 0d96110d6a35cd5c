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

// ------------------------------------------------------------------------------------------------------
// Bias + ReLU around the library convolutions (actor/network.py:72-80, critic/network.py:33-41: activation(conv(x))).
// PyTorch runs a convolution with bias as conv + a bias pass + a ReLU pass over the activation tensor (0.2 GB after conv1 of a
// 4096-observation rollout step, 0.6 GB for a 12 288-sample minibatch), and the backward as a threshold pass + a column
// reduction for the bias gradient.  Here each direction is ONE pass over channels-last rows [rows, C]:
//   forward : y = max(x + b, 0), in place on the convolution's (bias-free) output
//   backward: gx = gy * (y > 0) and gb[c] += sum_rows gx[r, c] -- a workgroup owns a block of rows, 256 / (C/4) row lanes of
//             C/4 four-channel groups each, sums its block in registers and LDS, then adds C floats atomically.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bias_relu(float4* __restrict__ x, const float4* __restrict__ bias, size_t n4, int c4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // one 16-byte access each way per lane, one trip per workgroup
  if (i >= n4) return;
  const float4 b = bias[i % (size_t)c4];
  float4 v = x[i];
  // relu that lets NaN through like torch.relu (fmaxf(NaN, 0) = 0 would hide a diverged activation): x > 0 ? x : (x != x ? x : 0)
  v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  v.x = v.x > 0.f ? v.x : (v.x != v.x ? v.x : 0.f); v.y = v.y > 0.f ? v.y : (v.y != v.y ? v.y : 0.f);
  v.z = v.z > 0.f ? v.z : (v.z != v.z ? v.z : 0.f); v.w = v.w > 0.f ? v.w : (v.w != v.w ? v.w : 0.f);
  x[i] = v;
}

#define IPPM_BR_ROWS 512   // rows per workgroup of the backward pass
__global__ void __launch_bounds__(256)
k_bias_relu_bwd(const float4* __restrict__ gy, const float4* __restrict__ y, float4* __restrict__ gx, float* __restrict__ gb,
                size_t rows, int c4) {
  __shared__ float4 part[256];
  const int lanes = 256 / c4;                      // row lanes of this workgroup
  const int cg = threadIdx.x % c4, rl = threadIdx.x / c4;
  const size_t r0 = (size_t)blockIdx.x * IPPM_BR_ROWS, r1 = min(rows, r0 + (size_t)IPPM_BR_ROWS);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < lanes) {
    for (size_t r = r0 + rl; r < r1; r += 4 * (size_t)lanes) {
      float4 g[4], a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {                // eight independent loads in flight
        const size_t rr = r + (size_t)u * lanes;
        if (rr < r1) { g[u] = gy[rr * c4 + cg]; a[u] = y[rr * c4 + cg]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t rr = r + (size_t)u * lanes;
        if (rr >= r1) continue;
        float4 o;
        o.x = a[u].x > 0.f ? g[u].x : 0.f; o.y = a[u].y > 0.f ? g[u].y : 0.f;
        o.z = a[u].z > 0.f ? g[u].z : 0.f; o.w = a[u].w > 0.f ? g[u].w : 0.f;
        gx[rr * c4 + cg] = o;
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
      }
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if ((int)threadIdx.x < c4) {
    float4 t = part[threadIdx.x];
    for (int l = 1; l < lanes; ++l) {
      const float4 v = part[l * c4 + threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float* o = gb + 4 * threadIdx.x;
    atomicAdd(o, t.x); atomicAdd(o + 1, t.y); atomicAdd(o + 2, t.z); atomicAdd(o + 3, t.w);
  }
}

static bool bias_relu_args_ok(const void* a, const void* b, const void* c, int64_t rows, int32_t channels, const char* who) {
  if (!a || !b || !c) { ippm_set_error(who); return false; }
  if (channels < 4 || channels % 4 || 256 % (channels / 4) || rows < 0 ||
      ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15)) {
    ippm_set_error("ippm_bias_relu: needs channels in {4, 8, ..., 1024} dividing 1024, and 16-byte aligned buffers");
    return false;
  }
  return true;
}

extern "C" int ippm_bias_relu_nhwc(float* x, const float* bias, int64_t rows, int32_t channels, void* stream) {
  if (!bias_relu_args_ok(x, bias, x, rows, channels, "ippm_bias_relu_nhwc: null argument")) return -1;
  const size_t n4 = (size_t)rows * (channels / 4);
  if (n4 == 0) return 0;
  hipLaunchKernelGGL(k_bias_relu, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<float4*>(x), reinterpret_cast<const float4*>(bias), n4, channels / 4);
  IPPM_LAUNCH_CHECK("bias_relu");
  return 0;
}

extern "C" int ippm_bias_relu_backward_nhwc(const float* grad_y, const float* y, float* grad_x, float* grad_bias, int64_t rows,
                                            int32_t channels, void* stream) {
  if (!bias_relu_args_ok(grad_y, y, grad_x, rows, channels, "ippm_bias_relu_backward_nhwc: null argument")) return -1;
  if (!grad_bias) { ippm_set_error("ippm_bias_relu_backward_nhwc: null argument"); return -1; }
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_bias_relu_bwd, dim3((unsigned)((rows + IPPM_BR_ROWS - 1) / IPPM_BR_ROWS)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(grad_y),
                     reinterpret_cast<const float4*>(y), reinterpret_cast<float4*>(grad_x), grad_bias, (size_t)rows, channels / 4);
  IPPM_LAUNCH_CHECK("bias_relu_bwd");
  return 0;
}

