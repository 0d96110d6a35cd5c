// Greedy information-gain planner (reference: IG_baseline.py:222-325) and evaluation metrics (utils/utils.py:43-76,
// IG_baseline.py:84-97): K9 expected information gain of every candidate footprint, K10 the per-env selection logic,
// and the target-class F1 counts.  The per-candidate reduction streams a footprint tile of the agent's local map
// exactly like K3 does (4 grid-aligned cells per lane), accumulates in float64 and is deterministic (no atomics), so
// candidates with identical cell multisets get bit-identical gains and argmax ties resolve like the reference's.
#include <cstdlib>

#include "ippm_internal.h"

__device__ __forceinline__ void ig_action_offset(int A, int a, int s, int& dx, int& dy, int& dz) {
  dx = dy = dz = 0;
  if (A == 4) {
    if (a == 0) dx = -s; else if (a == 1) dy = -s; else if (a == 2) dy = s; else dx = s;
  } else if (A == 6) {
    if (a == 0) dz = s; else if (a == 1) dx = -s; else if (a == 2) dy = -s; else if (a == 3) dy = s;
    else if (a == 4) dx = s; else dz = -s;
  } else if (A == 9) {
    dx = (a / 3 - 1) * s; dy = (a % 3 - 1) * s;
  } else {
    int layer = a / 9, c9 = a % 9;
    dz = (1 - layer) * s; dx = (c9 / 3 - 1) * s; dy = (c9 % 3 - 1) * s;
  }
}

// Expected weighted entropy reduction of one cell with (clipped) belief log-odds l under a measurement of log-odds +-ln
// (IG_baseline.py:236-268):  p (H(l) - H(l + ln)) w(l + ln) + (1 - p) (H(l) - H(l - ln)) w(l - ln),  w = the posterior itself
// inside the weight band, 1 / 0 beyond it.  All three entropies and posteriors hang off ONE exponential: with e = exp(-|l|),
// exp(-|l + s|) = exp(-l) exp(-s) or exp(l) exp(s), and exp(+-l) is e or 1/e; kp / km = exp(+-ln), ec = exp(-clip).
// 8 transcendentals per cell instead of 15.
__device__ __forceinline__ float ig_entropy_from_e(float a, float e, float& rd) {  // H of |L| = a, e = exp(-a); rd = 1/(1+e)
  const float d = 1.0f + e;
  rd = __builtin_amdgcn_rcpf(d);
  // log2(1 + e): for a saturated cell e = 1e-4, and 1 + e rounded to float32 keeps only three digits of e -- the series
  // log2(e_) (e - e^2/2 + e^3/3 - e^4/4) below 2^-6 (remainder < 2e-10) keeps them all.  It matters here and not in the reward
  // terms: a candidate over cells that are all saturated has a gain made of nothing but such entropies' differences.
  const float lg = e < 0.015625f ? e * (1.44269504f + e * (-0.72134752f + e * (0.48089835f - 0.36067376f * e))) : __log2f(d);
  return lg + (a * 1.44269504f) * (e * rd);
}
__device__ __forceinline__ float ig_cell(float l, float ln, float kp, float km, float ec, float lc, float wt) {
  const float a = fabsf(l), e = __expf(-a), re = __builtin_amdgcn_rcpf(e);
  const bool pos = l >= 0.f;
  const float en = pos ? e : re, ep = pos ? re : e;   // exp(-l), exp(l)
  float rd;
  const float hh = ig_entropy_from_e(a, e, rd);
  const float pb = pos ? rd : e * rd;                  // sigmoid(l)
  const float qb = pos ? e * rd : rd;                  // 1 - sigmoid(l), formed directly: `1.f - pb` of a saturated cell (pb = 0.9999)
                                                       // keeps 3 digits, and its branch then carries the whole gain of the cell
  const float l1 = l + ln, l0 = l - ln;
  // exp(-|l +- ln|), floored at exp(-clip) like the entropy's clipped argument
  const float e1 = fmaxf(l1 >= 0.f ? en * km : ep * kp, ec), e0 = fmaxf(l0 >= 0.f ? en * kp : ep * km, ec);
  float rd1, rd0;
  const float h1 = ig_entropy_from_e(fminf(fabsf(l1), lc), e1, rd1), h0 = ig_entropy_from_e(fminf(fabsf(l0), lc), e0, rd0);
  // inside the weight band |l +- ln| < clip, so e1 / e0 are the unfloored exponentials there
  const float s1 = l1 >= 0.f ? rd1 : e1 * rd1, s0 = l0 >= 0.f ? rd0 : e0 * rd0;
  const float cw1 = l1 > wt ? 1.f : (l1 < -wt ? 0.f : s1);
  const float cw0 = l0 > wt ? 1.f : (l0 < -wt ? 0.f : s0);
  return pb * (hh - h1) * cw1 + qb * (hh - h0) * cw0;
}

// K9: one workgroup per (env, agent, action)
__global__ void __launch_bounds__(256)
k_ig_candidates(const ippm_config* __restrict__ c, const float* __restrict__ local, const int32_t* __restrict__ pos,
                const uint8_t* __restrict__ mask, float* __restrict__ gains, int tl) {
  const int n = c->n_agents, A = c->n_actions;
  const int cand = blockIdx.x;
  const int a = cand % A, i = (cand / A) % n, e = cand / (A * n);
  if (!mask[cand]) { if (threadIdx.x == 0) gains[cand] = 0.f; return; }
  const int32_t* p = pos + (size_t)(e * n + i) * 3;
  int dx, dy, dz;
  ig_action_offset(A, a, c->spacing, dx, dy, dz);
  int r[4];
  ippm_footprint_rect(c, p[0] + dx, p[1] + dy, p[2] + dz, r, nullptr);
  const int yu = r[0], yd = r[1], xl = r[2], xr = r[3];
  const int gx = c->grid_x, gy = c->grid_y;
  const int k = ippm_alt_index(c, p[2] + dz);
  const float ln = c->logit_noise[k];  // ln((1-noise)/noise) from the float64 noise level: update_cells(section, 1-noise)
  const float kp = __expf(ln), km = __expf(-ln), ec = __expf(-c->logit_clip);
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  const float* map = local + (size_t)(e * n + i) * gx * gy;
  const bool vec = gy >= 4;   // 16-byte groups at any row alignment; a row's last group is read cell by cell
  const int y0 = vec ? (yu & ~3) : yu;
  const int step = vec ? 4 : 1;
  const int groups = (yd - y0 + step - 1) / step;
  const int h = xr - xl;
  double acc = 0.0;
  for (int idx = threadIdx.x; idx < h * groups; idx += blockDim.x) {
    const int row = idx / groups, gi = idx - row * groups;
    const int x = xl + row, y = y0 + gi * step;
    float v[4];
    // (tl: tile storage of the maps, ippm_internal.h -- a grid-aligned group is 16 contiguous bytes there too)
    if (vec && y + 4 <= gy) { const float4 t = *reinterpret_cast<const float4*>(map + ippm_cell_index(x, y, gy, tl)); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if (vec) { for (int q = 0; q < 4; ++q) v[q] = y + q < gy ? map[(size_t)x * gy + y + q] : 0.f; }
    else v[0] = map[(size_t)x * gy + y];
    for (int q = 0; q < step; ++q) {
      if (y + q < yu || y + q >= yd) continue;
      // IG_baseline.py:236-268 in log-odds: belief clipped once, hypothetical posteriors L +- ln.  Every cell goes into the
      // float64 sum on its own: a float32 partial over the group showed at 2e-5 on a candidate whose gain nearly cancels
      acc += (double)ig_cell(ippm_clampl(v[q], lc), ln, kp, km, ec, lc, wt);
    }
  }
  // deterministic block reduction in float64
  __shared__ double s[256];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) gains[cand] = (float)(s[0] / 1000.0);
}


// K9, union walk for the 3 x 3 action sets (9 actions: one layer; 27: three altitude layers): the nine candidates of a layer are
// the same footprint shifted by one lattice step, so they overlap heavily (at 15 m: 9 footprints of 90 x 90 cells inside a
// 150 x 150 hull, 3.6 cells of candidate per cell of hull) and a cell's expected gain depends on the layer's altitude only.
// One workgroup per (env, agent, layer) evaluates every cell of the hull ONCE and adds it to every candidate that holds it:
// membership is separable (the row offset decides the rows, the column offset the columns), i.e. 3 + 3 range tests per cell.
// (For the 6-action set -- 5 candidates at 3 altitudes -- the per-candidate kernel stays: measured 1.4x fewer evaluations
// there, not enough to pay for the bookkeeping.)  Sums in float64, lane-striped then a fixed tree: deterministic.
__global__ void __launch_bounds__(256)
k_ig_union(const ippm_config* __restrict__ c, const float* __restrict__ local, const int32_t* __restrict__ pos,
           const uint8_t* __restrict__ mask, float* __restrict__ gains, int tl) {
  const int n = c->n_agents, A = c->n_actions, layers = A / 9;
  const int layer = blockIdx.x % layers, i = (blockIdx.x / layers) % n, e = blockIdx.x / (layers * n);
  const int gx = c->grid_x, gy = c->grid_y, s = c->spacing;
  const int32_t* p = pos + (size_t)(e * n + i) * 3;
  const size_t cand0 = (size_t)(e * n + i) * A + layer * 9;
  const int dz = A == 27 ? (1 - layer) * s : 0;
  // the three row ranges (offset -1, 0, +1 lattice steps in x) and the three column ranges of the layer's footprints
  int xl[3], xr[3], yu[3], yd[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    int r[4];
    ippm_footprint_rect(c, p[0] + (o - 1) * s, p[1], p[2] + dz, r, nullptr);
    xl[o] = r[2]; xr[o] = r[3];
    ippm_footprint_rect(c, p[0], p[1] + (o - 1) * s, p[2] + dz, r, nullptr);
    yu[o] = r[0]; yd[o] = r[1];
  }
  // (a masked candidate may lie outside the lattice: its table lookups above read a neighbouring entry, and it is never used)
  bool ok[9];
  int X0 = gx, X1 = 0, Y0 = gy, Y1 = 0;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    ok[q] = mask[cand0 + q] != 0;
    if (ok[q]) { X0 = min(X0, xl[q / 3]); X1 = max(X1, xr[q / 3]); Y0 = min(Y0, yu[q % 3]); Y1 = max(Y1, yd[q % 3]); }
  }
  const int k = ippm_alt_index(c, p[2] + dz);
  const float ln = c->logit_noise[k];
  const float kp = __expf(ln), km = __expf(-ln), ec = __expf(-c->logit_clip);
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  const float* map = local + (size_t)(e * n + i) * gx * gy;
  double acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) acc[q] = 0.0;
  const int y0 = Y0 & ~3;
  const int groups = X1 > X0 && Y1 > Y0 ? (Y1 - y0 + 3) >> 2 : 0, h = X1 - X0;
  for (int idx = threadIdx.x; idx < h * groups; idx += blockDim.x) {
    const int row = idx / groups, gi = idx - row * groups;
    const int x = X0 + row, y = y0 + gi * 4;
    float v[4];
    if (y + 4 <= gy) { const float4 t = *reinterpret_cast<const float4*>(map + ippm_cell_index(x, y, gy, tl)); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else { for (int q = 0; q < 4; ++q) v[q] = y + q < gy ? map[(size_t)x * gy + y + q] : 0.f; }
    bool inx[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) inx[o] = x >= xl[o] && x < xr[o];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = y + j;
      bool iny[3], any = false;
#pragma unroll
      for (int o = 0; o < 3; ++o) iny[o] = yy >= yu[o] && yy < yd[o];
#pragma unroll
      for (int q = 0; q < 9; ++q) any |= ok[q] && inx[q / 3] && iny[q % 3];
      if (!any) continue;
      // IG_baseline.py:236-268 in log-odds: belief clipped once, hypothetical posteriors L +- ln
      const double g = (double)ig_cell(ippm_clampl(v[j], lc), ln, kp, km, ec, lc, wt);
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] += ok[q] && inx[q / 3] && iny[q % 3] ? g : 0.0;
    }
  }
  __shared__ double sh[9][256];
#pragma unroll
  for (int q = 0; q < 9; ++q) sh[q][threadIdx.x] = acc[q];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
#pragma unroll
      for (int q = 0; q < 9; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 9) gains[cand0 + threadIdx.x] = ok[threadIdx.x] ? (float)(sh[threadIdx.x][0] / 1000.0) : 0.f;
}

// K10: get_relative_ig + get_cell_utilities + select_action (IG_baseline.py:270-325), one wavefront per env.
// The reference walks the agents in order and discounts candidate (i, a1) IN PLACE by every other agent's candidate that lands
// on the same lattice point: rel[i,a1] = g1 * (1 - rel[j,a2]), last match wins, where rel[j,a2] is already discounted for
// j < i and still raw for j > i.  A row i never reads its own entries, so its candidates are independent: the agent loop
// stays serial, lane a1 does the (j, a2) scan of its candidate against the table in LDS.
__global__ void __launch_bounds__(64)
k_ig_select(const ippm_config* __restrict__ c, const int32_t* __restrict__ pos, const uint8_t* __restrict__ mask,
            const float* __restrict__ gains, int communication, int32_t* __restrict__ action,
            float* __restrict__ utilities, int n_envs) {
  const int e = blockIdx.x, lane = threadIdx.x;
  const int n = c->n_agents, A = c->n_actions;
  __shared__ float rel[IPPM_MAX_AGENTS * IPPM_MAX_ACTIONS];
  __shared__ int cpos[IPPM_MAX_AGENTS * IPPM_MAX_ACTIONS][3];   // lattice point of every candidate
  __shared__ uint8_t cmask[IPPM_MAX_AGENTS * IPPM_MAX_ACTIONS];
  for (int q = lane; q < n * A; q += 64) {
    const int i = q / A, a = q - i * A;
    // relative gain: the row sum in the reference's order (a = 0, 1, ...)
    float total = 0.f;
    for (int b = 0; b < A; ++b) total += gains[(size_t)(e * n + i) * A + b];
    rel[q] = gains[(size_t)(e * n + i) * A + a] / total;
    int dx, dy, dz;
    ig_action_offset(A, a, c->spacing, dx, dy, dz);
    const int32_t* p = pos + (size_t)(e * n + i) * 3;
    cpos[q][0] = p[0] + dx; cpos[q][1] = p[1] + dy; cpos[q][2] = p[2] + dz;
    cmask[q] = mask[(size_t)(e * n + i) * A + a];
  }
  __syncthreads();
  if (communication) {
    for (int i = 0; i < n; ++i) {
      for (int a1 = lane; a1 < A; a1 += 64) {   // (A <= 27: one trip)
        const int q1 = i * A + a1;
        if (!cmask[q1]) continue;   // masked candidates carry the placeholder position 0
        const float g1 = rel[q1];
        float v = g1;
        for (int q2 = 0; q2 < n * A; ++q2) {
          if (q2 / A == i || !cmask[q2]) continue;
          if (cpos[q1][0] == cpos[q2][0] && cpos[q1][1] == cpos[q2][1] && cpos[q1][2] == cpos[q2][2]) v = g1 * (1.f - rel[q2]);
        }
        rel[q1] = v;   // row i is read by nobody during this pass
      }
      __syncthreads();
    }
  }
  if (lane < n) {
    int best = 0;
    float bv = rel[lane * A];
    for (int a = 1; a < A; ++a) {
      const float v = rel[lane * A + a];
      // np.argmax: first maximum, and a nan counts as the maximum (an agent whose candidates all have zero gain has 0 / 0 in
      // every entry, and candidates of others that share a cell with one of those inherit the nan)
      if (v > bv || (v != v && bv == bv)) { bv = v; best = a; }
    }
    action[e * n + lane] = best;
  }
  if (utilities)
    for (int q = lane; q < n * A; q += 64) utilities[(size_t)e * n * A + q] = rel[q];
}

// target-class confusion counts of a map thresholded at L > thr (thr = 0 <=> p > 0.5): out int64 [n_maps,3] = tp, fp, fn
__global__ void __launch_bounds__(256)
k_f1_counts(const ippm_config* __restrict__ c, const float* __restrict__ maps, const uint8_t* __restrict__ truth, int maps_per_truth,
            float thr, unsigned long long* __restrict__ out, int tl) {
  const int m = blockIdx.y;
  const size_t total = (size_t)c->grid_x * c->grid_y;
  const float* p = maps + (size_t)m * total;
  const uint8_t* t = truth + (size_t)(m / maps_per_truth) * ippm_truth_bytes(c->grid_x, c->grid_y);
  unsigned tp = 0, fp = 0, fn = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const bool pred = p[i] > thr, tr = ippm_truth1(t, ippm_stored_cell(i, c->grid_y, tl)) != 0;
    tp += pred && tr; fp += pred && !tr; fn += !pred && tr;
  }
  const float a = ippm_wave_sum((float)tp), b = ippm_wave_sum((float)fp), d = ippm_wave_sum((float)fn);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&out[m * 3 + 0], (unsigned long long)a);
    atomicAdd(&out[m * 3 + 1], (unsigned long long)b);
    atomicAdd(&out[m * 3 + 2], (unsigned long long)d);
  }
}

static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

extern "C" int ippm_ig_candidates(ippm_ctx* ctx, const float* local, const int32_t* pos, const uint8_t* mask, float* gains,
                                  int32_t n_envs, void* stream) {
  if (!ctx || !local || !pos || !mask || !gains) { ippm_set_error("ippm_ig_candidates: null argument"); return -1; }
  const int total = n_envs * ctx->cfg.n_agents * ctx->cfg.n_actions;
  if (total <= 0) return 0;
  const int A = ctx->cfg.n_actions;
  if ((A == 9 || A == 27) && ctx->cfg.grid_y >= 4 && !getenv("IPPM_IG_PER_CANDIDATE"))   // 3 x 3 sets: one walk of the hull per layer
    hipLaunchKernelGGL(k_ig_union, dim3(n_envs * ctx->cfg.n_agents * (A / 9)), dim3(256), 0, S_(stream), ctx->dcfg, local, pos, mask, gains, ctx->tl);
  else
    hipLaunchKernelGGL(k_ig_candidates, dim3(total), dim3(256), 0, S_(stream), ctx->dcfg, local, pos, mask, gains, ctx->tl);
  IPPM_LAUNCH_CHECK("ig_candidates");
  return 0;
}

extern "C" int ippm_ig_select(ippm_ctx* ctx, const int32_t* pos, const uint8_t* mask, const float* gains, int32_t communication,
                              int32_t* action, float* utilities, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !mask || !gains || !action) { ippm_set_error("ippm_ig_select: null argument"); return -1; }
  if (n_envs <= 0) return 0;
  hipLaunchKernelGGL(k_ig_select, dim3(n_envs), dim3(64), 0, S_(stream), ctx->dcfg, pos, mask, gains, communication, action,
                     utilities, n_envs);
  IPPM_LAUNCH_CHECK("ig_select");
  return 0;
}

extern "C" int ippm_f1_counts(ippm_ctx* ctx, const float* maps, const uint8_t* truth, int32_t maps_per_truth, float logodds_threshold,
                              int64_t* out, int32_t n_maps, void* stream) {
  if (!ctx || !maps || !truth || !out) { ippm_set_error("ippm_f1_counts: null argument"); return -1; }
  IPPM_HIP(hipMemsetAsync(out, 0, sizeof(int64_t) * 3 * n_maps, S_(stream)));
  const size_t cells = (size_t)ctx->cfg.grid_x * ctx->cfg.grid_y;
  const int gxb = (int)std::min<size_t>(32, (cells + 255) / 256);
  hipLaunchKernelGGL(k_f1_counts, dim3(gxb, n_maps), dim3(256), 0, S_(stream), ctx->dcfg, maps, truth, maps_per_truth > 0 ? maps_per_truth : 1,
                     logodds_threshold, reinterpret_cast<unsigned long long*>(out), ctx->tl);
  IPPM_LAUNCH_CHECK("f1_counts");
  return 0;
}
