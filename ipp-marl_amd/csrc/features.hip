// K6: network-input builders (actor 7 planes, critic 12 planes) and the full-grid reward of two given maps.
//
// The expensive part is the exact area average G x G -> 11 x 11 (the reference's cv2.resize INTER_AREA,
// utils/state.py:22-41).  It is separable: one pass streams the map row by row (coalesced: consecutive
// lanes own consecutive columns) into per-column partial sums of the 11 row bins held in LDS, a second tiny
// pass folds the columns.  The footprint planes are analytic functions of the rectangles and are
// accumulated in the same pass without touching memory.
#include "ippm_internal.h"

struct TabView {
  const int32_t* bin0;    // [n] first output bin a source index overlaps
  const float* w0;        // [n] weight into bin0 (already divided by the scale)
  const float* w1;        // [n] weight into bin0+1 (0 if none)
  const int32_t* bstart;  // [12] first source index whose bin0 >= b
  int n;
};

// Column-owner area reduction: thread `tid` owns columns tid, tid+nthr, ...; walks all rows.
// SRC(row, col) -> value.  colsum: LDS [planes][11][n_cols]; out: LDS [planes][121].
template <int PLANES, class SRC>
__device__ void area_reduce(SRC src, const TabView rows, const TabView cols, float* colsum, float* out) {
  const int n_rows = rows.n, n_cols = cols.n;
  for (int col = threadIdx.x; col < n_cols; col += blockDim.x) {
    float a0[PLANES], a1[PLANES];
#pragma unroll
    for (int p = 0; p < PLANES; ++p) { a0[p] = 0.f; a1[p] = 0.f; }
    int cur = 0;
    for (int r = 0; r < n_rows; ++r) {
      const int b = rows.bin0[r];
      while (b > cur) {
#pragma unroll
        for (int p = 0; p < PLANES; ++p) { colsum[(p * IPPM_FEAT + cur) * n_cols + col] = a0[p]; a0[p] = a1[p]; a1[p] = 0.f; }
        ++cur;
      }
      const float w0 = rows.w0[r], w1 = rows.w1[r];
      float v[PLANES];
      src(r, col, v);
#pragma unroll
      for (int p = 0; p < PLANES; ++p) { a0[p] += w0 * v[p]; a1[p] += w1 * v[p]; }
    }
    while (cur < IPPM_FEAT) {
#pragma unroll
      for (int p = 0; p < PLANES; ++p) { colsum[(p * IPPM_FEAT + cur) * n_cols + col] = a0[p]; a0[p] = a1[p]; a1[p] = 0.f; }
      ++cur;
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < PLANES * IPPM_FEAT * IPPM_FEAT; o += blockDim.x) {
    const int p = o / (IPPM_FEAT * IPPM_FEAT), ox = (o / IPPM_FEAT) % IPPM_FEAT, oy = o % IPPM_FEAT;
    const int i0 = oy > 0 ? max(cols.bstart[oy] - 1, 0) : 0;
    const int i1 = cols.bstart[oy + 1];
    const float* cs = colsum + (p * IPPM_FEAT + ox) * n_cols;
    float acc = 0.f;
    for (int i = i0; i < i1; ++i) {
      const int b = cols.bin0[i];
      const float wgt = b == oy ? cols.w0[i] : (b == oy - 1 ? cols.w1[i] : 0.f);
      acc += wgt * cs[i];
    }
    out[o] = acc;
  }
  __syncthreads();
}

struct RectSet {
  int n;
  int r[IPPM_MAX_AGENTS][4];  // [yu,yd,xl,xr]
};

// ------------------------------------------------------------------------------------------------------
// actor observation [11,11,7] (actor/transformations.py:14-176)
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_actor_features(const ippm_config* __restrict__ c, const float* __restrict__ local, const uint8_t* __restrict__ code,
                 const int32_t* __restrict__ rect, const int32_t* __restrict__ pos, const uint8_t* __restrict__ comm,
                 TabView trows, TabView tcols, const int32_t* __restrict__ tab_i, const float* __restrict__ tab_f0,
                 const float* __restrict__ tab_f1, const int32_t* __restrict__ fp_off, const int32_t* __restrict__ fp_n,
                 int t, float* __restrict__ obs, unsigned long long* __restrict__ counters) {
  extern __shared__ float smem[];
  const int n = c->n_agents;
  const int e = blockIdx.x / n, i = blockIdx.x % n;
  const int gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  float* colsum = smem;                               // [2][11][gy] (reused as [1][11][2r] for the footprint image)
  float* red = smem + 2 * IPPM_FEAT * max(gy, S);     // [2][121]
  float* red_fp = red + 2 * IPPM_FEAT * IPPM_FEAT;    // [121]
  __shared__ int s_rect[IPPM_MAX_AGENTS][4];
  __shared__ int s_recv[IPPM_MAX_AGENTS];
  __shared__ int s_idx[IPPM_MAX_AGENTS][3];
  if (threadIdx.x < n) {
    const int j = threadIdx.x;
    for (int q = 0; q < 4; ++q) s_rect[j][q] = rect[(size_t)(e * n + j) * 4 + q];
    s_recv[j] = comm[(size_t)(e * n + i) * n + j];
    const int32_t* pj = pos + (size_t)(e * n + j) * 3;
    ippm_pos_to_index(c, pj[0], pj[1], pj[2], s_idx[j][0], s_idx[j][1], s_idx[j][2]);
  }
  __syncthreads();
  const float* map = local + (size_t)(e * n + i) * gx * gy;
  // pass 1: plane q = R(local map), plane F = R(footprint indicator)
  auto src_map = [&](int r, int col, float* v) {
    v[0] = ippm_sigmoid(map[(size_t)r * gy + col]);  // maps hold log-odds; the resize averages probabilities
    float f = 0.5f;
    for (int j = 0; j < n; ++j) {
      if (j == i || !s_recv[j]) continue;
      if (r >= s_rect[j][2] && r < s_rect[j][3] && col >= s_rect[j][0] && col < s_rect[j][1]) f = 0.f;
    }
    if (r >= s_rect[i][2] && r < s_rect[i][3] && col >= s_rect[i][0] && col < s_rect[i][1]) f = 1.f;
    v[1] = f;
  };
  area_reduce<2>(src_map, trows, tcols, colsum, red);
  // pass 2: R(footprint_img): unclipped-size image, 0.5 with the measurement pasted at its border-aware offset
  // (mappings.py:41-43,72-76; utils/utils.py:79-98)
  const int32_t* pi = pos + (size_t)(e * n + i) * 3;
  const int k = ippm_alt_index(c, pi[2]);
  int cl[4], fu[4];
  ippm_footprint_rect(c, pi[0], pi[1], pi[2], cl, fu);
  const int hx = cl[3] - cl[2], wy = cl[1] - cl[0];
  const int full_x = fu[3] - fu[2], full_y = fu[1] - fu[0];
  const int xoff = (cl[2] > fu[2]) ? full_x - hx : 0;
  const int yoff = (cl[0] > fu[0]) ? full_y - wy : 0;
  const int vec = (gy & 3) == 0 ? 4 : 1;
  const uint8_t* cd = code + (size_t)(e * n + i) * ippm_tile_bytes(S, vec);
  const int ycode0 = cl[0] - (cl[0] & ~3);
  const float mv0 = c->meas_value[k][0], mv1 = c->meas_value[k][1];
  TabView tfp;
  tfp.n = fp_n[k];
  tfp.bin0 = tab_i + fp_off[k];
  tfp.w0 = tab_f0 + fp_off[k];
  tfp.w1 = tab_f1 + fp_off[k];
  tfp.bstart = tab_i + fp_off[k] + tfp.n;
  auto src_fp = [&](int r, int col, float* v) {
    const int u = r - xoff, w = col - yoff;
    float val = 0.5f;
    if (u >= 0 && u < hx && w >= 0 && w < wy) {
      const int col = w + ycode0;
      const uint32_t bit = vec == 4 ? (cd[(size_t)u * (S >> 2) + (col >> 2)] >> (col & 3)) & 1u : cd[(size_t)u * S + col];
      val = bit ? mv1 : mv0;
    }
    v[0] = val;
  };
  area_reduce<1>(src_fp, tfp, tfp, colsum, red_fp);
  // assemble
  const float lo = c->clip_lo, hi = c->clip_hi;
  const int Z = c->space_z;
  const int ox = s_idx[i][0], oy = s_idx[i][1], oz = s_idx[i][2];
  float* out = obs + (size_t)(e * n + i) * IPPM_FEAT * IPPM_FEAT * IPPM_ACTOR_PLANES;
  for (int o = threadIdx.x; o < IPPM_FEAT * IPPM_FEAT; o += blockDim.x) {
    const int a = o / IPPM_FEAT, b = o % IPPM_FEAT;
    float pm = 1.f;
    if (ox < 5 && a < 5 - ox) pm = 0.f;
    if (oy < 5 && b < 5 - oy) pm = 0.f;
    if (ox > 5 && a >= c->space_x + 5 - ox) pm = 0.f;
    if (oy > 5 && b >= c->space_y + 5 - oy) pm = 0.f;
    if (a == 5 && b == 5) pm = (float)(oz + 1) / (float)(Z + 1);
    for (int j = 0; j < n; ++j) {
      if (j == i || !s_recv[j]) continue;
      if (s_idx[j][0] - ox + 5 == a && s_idx[j][1] - oy + 5 == b) pm = (float)(s_idx[j][2] + 1) / (float)(Z + 1);
    }
    const float q = red[o], f = red[IPPM_FEAT * IPPM_FEAT + o], fp = red_fp[o];
    float* dst = out + (size_t)o * IPPM_ACTOR_PLANES;
    dst[0] = (float)(c->budget - t) / (float)c->budget;
    dst[1] = (float)(i + 1) / (float)n;
    dst[2] = pm;
    dst[3] = ippm_weight(q) * ippm_entropy(q, lo, hi);
    dst[4] = ippm_weight(fp) * ippm_entropy(fp, lo, hi);
    dst[5] = ippm_clipf(q, lo, hi);
    dst[6] = f;
  }
  if (counters && threadIdx.x == 0) atomicAdd(&counters[(blockIdx.x & (IPPM_COUNTER_SLOTS - 1)) * 8 + 5], (unsigned long long)gx * gy);
}

// ------------------------------------------------------------------------------------------------------
// critic state [11,11,12] (critic/transformations.py:17-132): one workgroup per env builds the shared
// global planes once and writes them for every agent
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_critic_features(const ippm_config* __restrict__ c, const float* __restrict__ global, const int32_t* __restrict__ rect,
                  const int32_t* __restrict__ pos_pre, const int32_t* __restrict__ action, const float* __restrict__ obs,
                  TabView trows, TabView tcols, float* __restrict__ state, unsigned long long* __restrict__ counters) {
  extern __shared__ float smem[];
  const int n = c->n_agents;
  const int e = blockIdx.x;
  const int gx = c->grid_x, gy = c->grid_y;
  float* colsum = smem;
  float* red = smem + 2 * IPPM_FEAT * gy;
  __shared__ int s_rect[IPPM_MAX_AGENTS][4];
  __shared__ int s_idx[IPPM_MAX_AGENTS][3];
  __shared__ int s_act[IPPM_MAX_AGENTS];
  if (threadIdx.x < n) {
    const int j = threadIdx.x;
    for (int q = 0; q < 4; ++q) s_rect[j][q] = rect[(size_t)(e * n + j) * 4 + q];
    const int32_t* pj = pos_pre + (size_t)(e * n + j) * 3;
    ippm_pos_to_index(c, pj[0], pj[1], pj[2], s_idx[j][0], s_idx[j][1], s_idx[j][2]);
    s_act[j] = action[e * n + j];
  }
  __syncthreads();
  const float* map = global + (size_t)e * gx * gy;
  auto src_map = [&](int r, int col, float* v) {
    v[0] = ippm_sigmoid(map[(size_t)r * gy + col]);
    float f = 0.5f;
    for (int j = 0; j < n; ++j)
      if (r >= s_rect[j][2] && r < s_rect[j][3] && col >= s_rect[j][0] && col < s_rect[j][1]) f = 1.f;
    v[1] = f;
  };
  area_reduce<2>(src_map, trows, tcols, colsum, red);
  const float lo = c->clip_lo, hi = c->clip_hi;
  const int Z = c->space_z, A = c->n_actions;
  const int cellsf = IPPM_FEAT * IPPM_FEAT;
  for (int w = threadIdx.x; w < n * cellsf; w += blockDim.x) {
    const int i = w / cellsf, o = w % cellsf;
    const int a = o / IPPM_FEAT, b = o % IPPM_FEAT;
    float pm = 0.f, am = 0.f;
    for (int j = 0; j < n; ++j) {
      if (s_idx[j][0] == a && s_idx[j][1] == b) {
        pm = (float)(s_idx[j][2] + 1) / (float)Z;
        if (j != i) am = (float)(s_act[j] + 1) / (float)A;
      }
    }
    // "other actions": later agents overwrite earlier ones on a shared cell, the own agent never writes
    // (handled above: am only changes for j != i, in ascending j)
    const float q = red[o], f = red[cellsf + o];
    const float* src = obs + ((size_t)(e * n + i) * cellsf + o) * IPPM_ACTOR_PLANES;
    float* dst = state + ((size_t)(e * n + i) * cellsf + o) * IPPM_CRITIC_PLANES;
#pragma unroll
    for (int p = 0; p < IPPM_ACTOR_PLANES; ++p) dst[p] = src[p];
    dst[7] = pm;
    dst[8] = ippm_weight(q) * ippm_entropy(q, lo, hi);
    dst[9] = ippm_clipf(q, lo, hi);
    dst[10] = f;
    dst[11] = am;
  }
  if (counters && threadIdx.x == 0) atomicAdd(&counters[(blockIdx.x & (IPPM_COUNTER_SLOTS - 1)) * 8 + 5], (unsigned long long)gx * gy);
}

// ------------------------------------------------------------------------------------------------------
// reward of two explicit maps (drop-in get_global_reward, utils/reward.py:11-82): full-grid sums
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_reward_pair(const ippm_config* __restrict__ c, const float* __restrict__ before, const float* __restrict__ after,
              double* __restrict__ out /* [n,2] S1,S2 */) {
  const int m = blockIdx.y;
  const size_t total = (size_t)c->grid_x * c->grid_y;
  const float* b = before + (size_t)m * total;
  const float* a = after + (size_t)m * total;
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  float s1 = 0.f, s2 = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float wa = ippm_weight_l(a[i], wt);
    const float hb = ippm_entropy_l(b[i], lc), ha = ippm_entropy_l(a[i], lc);
    s1 += wa * (hb - ha);
    s2 += wa * hb;
  }
  s1 = ippm_wave_sum(s1);
  s2 = ippm_wave_sum(s2);
  __shared__ float s[4][2];
  if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6][0] = s1; s[threadIdx.x >> 6][1] = s2; }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(&out[m * 2 + threadIdx.x], (double)(s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x]));
}

// ======================================================================================================
// host API
// ======================================================================================================
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }

static TabView make_view(const ippm_ctx* ctx, int off, int n) {
  TabView t;
  t.n = n;
  t.bin0 = ctx->tab_bin0 + off;
  t.w0 = ctx->tab_w0 + off;
  t.w1 = ctx->tab_w1 + off;
  t.bstart = ctx->tab_bin0 + off + n;
  return t;
}

static int feature_checks(const ippm_ctx* ctx, const char* who) {
  const ippm_config& c = ctx->cfg;
  if (c.space_x != IPPM_FEAT || c.space_y != IPPM_FEAT) {
    ippm_set_error(std::string(who) + ": the network inputs are hard-wired to an 11x11 lattice (actor/network.py:19-21)");
    return -2;
  }
  if (c.grid_x < IPPM_FEAT || c.grid_y < IPPM_FEAT) { ippm_set_error(std::string(who) + ": grid smaller than 11x11"); return -2; }
  for (int k = 0; k < c.space_z; ++k) {
    if (c.radius_x[k] != c.radius_y[k]) { ippm_set_error(std::string(who) + ": needs a square field of view"); return -2; }
    if (2 * c.radius_x[k] < IPPM_FEAT) { ippm_set_error(std::string(who) + ": footprint image smaller than 11 cells"); return -2; }
  }
  return 0;
}

extern "C" int ippm_actor_features(ippm_ctx* ctx, const float* local, const uint8_t* code, const int32_t* rect,
                                   const int32_t* pos, const uint8_t* comm, int32_t t, float* obs, int32_t n_envs,
                                   void* stream) {
  if (!ctx || !local || !code || !rect || !pos || !comm || !obs) { ippm_set_error("ippm_actor_features: null argument"); return -1; }
  if (int rc = feature_checks(ctx, "ippm_actor_features")) return rc;
  const ippm_config& c = ctx->cfg;
  const size_t lds = sizeof(float) * (2 * IPPM_FEAT * (size_t)std::max(c.grid_y, c.tile_stride) + 3 * IPPM_FEAT * IPPM_FEAT);
  IPPM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_actor_features), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_actor_features, dim3(n_envs * c.n_agents), dim3(256), lds, S_(stream), ctx->dcfg, local, code, rect, pos,
                     comm, make_view(ctx, ctx->off_rows, c.grid_x), make_view(ctx, ctx->off_cols, c.grid_y), ctx->tab_bin0,
                     ctx->tab_w0, ctx->tab_w1, ctx->d_fp_off, ctx->d_fp_n, t, obs, ctx->dcounters);
  IPPM_LAUNCH_CHECK("actor_features");
  return 0;
}

extern "C" int ippm_critic_features(ippm_ctx* ctx, const float* global, const int32_t* rect, const int32_t* pos_pre,
                                    const int32_t* action, const float* obs, float* state, int32_t n_envs, void* stream) {
  if (!ctx || !global || !rect || !pos_pre || !action || !obs || !state) { ippm_set_error("ippm_critic_features: null argument"); return -1; }
  if (int rc = feature_checks(ctx, "ippm_critic_features")) return rc;
  const ippm_config& c = ctx->cfg;
  const size_t lds = sizeof(float) * (2 * IPPM_FEAT * (size_t)c.grid_y + 2 * IPPM_FEAT * IPPM_FEAT);
  IPPM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_critic_features), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_critic_features, dim3(n_envs), dim3(256), lds, S_(stream), ctx->dcfg, global, rect, pos_pre, action, obs,
                     make_view(ctx, ctx->off_rows, c.grid_x), make_view(ctx, ctx->off_cols, c.grid_y), state, ctx->dcounters);
  IPPM_LAUNCH_CHECK("critic_features");
  return 0;
}

extern "C" int ippm_reward_from_maps(ippm_ctx* ctx, const float* before, const float* after, double* sums, float* reward,
                                     int32_t n_maps, void* stream) {
  if (!ctx || !before || !after || !sums) { ippm_set_error("ippm_reward_from_maps: null argument"); return -1; }
  IPPM_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * n_maps, S_(stream)));
  const size_t cells = (size_t)ctx->cfg.grid_x * ctx->cfg.grid_y;
  const int gxb = (int)std::min<size_t>(32, (cells + 255) / 256);
  hipLaunchKernelGGL(k_reward_pair, dim3(gxb, n_maps), dim3(256), 0, S_(stream), ctx->dcfg, before, after, sums);
  IPPM_LAUNCH_CHECK("reward_pair");
  (void)reward;
  return 0;
}
