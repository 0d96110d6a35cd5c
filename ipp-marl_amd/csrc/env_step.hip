// Environment-step kernels for gfx950: reset (device MT19937), K1 mask/act/move, K2 footprint,
// K3 sense+update, comm matrix, K4 local fusion, K5 global fusion + information-gain reward.
//
// All of this is HBM-bound byte/float streaming over map tiles: no MFMA.  The layout rules are
//   - a map row (y contiguous) is covered by lanes holding 4 grid-aligned cells each (one 16-byte access),
//     truth is bit-packed, measurement codes / flips are one nibble per lane group: one byte load per lane each;
//   - narrow footprints pack several rows into one 64-lane wavefront (lanes-per-row = next pow2);
//   - every map cell is read and written at most once per kernel, whatever the number of fused measurements.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "ippm_internal.h"

// ======================================================================================================
// reset: legacy NumPy MT19937 streams regenerated on the device
// ======================================================================================================
__global__ void k_reset_scalars(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                                int32_t* __restrict__ pos, int32_t* __restrict__ split_pct,
                                float* __restrict__ comm_range, int32_t* __restrict__ ws, double* __restrict__ sums,
                                int n_envs) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int n = c->n_agents;
  int per = n + 1;
  if (tid >= n_envs * per) return;
  int e = tid / per, k = tid % per;
  int64_t ep = episode[e];
  // clear this map's workspace (deferred clamp state + plan)
  int32_t* w = ws + (size_t)(e * per + k) * IPPM_WS_WORDS;
  for (int i = 0; i < WS_OPS; ++i) w[i] = 0;
  if (k < n) {
    ippm_start_state(c->env_seed, ep, k, c->spacing, c->space_x, c->space_y, pos + (size_t)(e * n + k) * 3);
  } else {  // the truth split and the per-episode comm range share the stream np.random.seed(episode)
    int split, pct;
    ippm_truth_params(ep, &split, &pct);
    if (split_pct) { split_pct[e * 2] = split; split_pct[e * 2 + 1] = pct; }
    if (comm_range) {
      const float ranges[4] = {0.f, 15.f, 25.f, 100.f};
      comm_range[e] = c->fix_range ? (float)c->comm_range : ranges[split];
    }
    if (sums) {
      double* s = sums + (size_t)e * 8;
      for (int i = 0; i < 8; ++i) s[i] = 0.0;
      // weighted entropy of the all-prior map: w(0.5) * H(0.5) = 0.5 per cell
      s[SUM_T] = 0.5 * (double)c->grid_x * (double)c->grid_y;
    }
  }
}

__global__ void k_fill_truth(const ippm_config* __restrict__ c, const int32_t* __restrict__ split_pct,
                             uint8_t* __restrict__ truth, int n_envs) {
  int e = blockIdx.y;
  int gx = c->grid_x, gy = c->grid_y;
  int split = split_pct[e * 2], pct = split_pct[e * 2 + 1];
  // Python: int((dim * pct) / 100) and int((dim * (1 - pct)) / 100) (truncation toward zero), negative
  // slice starts count from the end (ground_truths.py:49-56)
  int dim = (split < 2) ? gx : gy;
  int lo = 0, hi = dim;
  if ((split & 1) == 0) {
    hi = min((dim * pct) / 100, dim);
  } else {
    int start = -((dim * (pct - 1)) / 100);
    lo = start == 0 ? 0 : max(dim + start, 0);
  }
  uint8_t* t = truth + (size_t)e * ippm_truth_bytes(gx, gy);
  const size_t total = (size_t)gx * gy, nbytes = ippm_truth_bytes(gx, gy);
  for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < nbytes; b += (size_t)gridDim.x * blockDim.x) {
    uint32_t bits = 0;
    for (int q = 0; q < 8; ++q) {
      const size_t i = b * 8 + q;
      if (i >= total) break;
      const int x = (int)(i / gy), y = (int)(i - (size_t)x * gy);
      const int v = (split < 2) ? x : y;
      bits |= (v >= lo && v < hi) ? (1u << q) : 0u;
    }
    t[b] = (uint8_t)bits;
  }
}

__global__ void k_fill_f32(float* __restrict__ p, float v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// posterior <-> log-odds conversion at the API boundary (drop-in classes exchange probabilities)
__global__ void k_logodds_to_prob(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = ippm_sigmoid(src[i]);
}
__global__ void k_prob_to_logodds(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float p = src[i];
    dst[i] = __logf(p) - __logf(1.0f - p);  // p in {0,1} gives -inf/+inf: clamped on first use like the reference's clip
  }
}

// full-grid input clip of one map (stateless drop-in fuse_map: the deferred-clamp bookkeeping has no history there)
__global__ void k_clamp_logodds(float* __restrict__ p, float lc, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = ippm_clampl(p[i], lc);
}

// ======================================================================================================
// K2: footprint projection
// ======================================================================================================
__global__ void k_footprint(const ippm_config* __restrict__ c, const int32_t* __restrict__ pos,
                            int32_t* __restrict__ rect, int32_t* __restrict__ rect_full, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cl[4], fu[4];
  ippm_footprint_rect(c, pos[i * 3], pos[i * 3 + 1], pos[i * 3 + 2], cl, fu);
  for (int k = 0; k < 4; ++k) {
    rect[i * 4 + k] = cl[k];
    if (rect_full) rect_full[i * 4 + k] = fu[k];
  }
}

// ======================================================================================================
// row/lane geometry shared by K3 and the fusion kernel
// ======================================================================================================
struct RowGeom {
  int y0;      // grid-aligned first column
  int groups;  // VEC-wide groups per row
  int lpr;     // lanes per row (power of two <= 64)
  int rpw;     // rows per wavefront
  int shift;   // log2(lpr)
};
template <int VEC>
__device__ __forceinline__ RowGeom make_geom(int ya, int yb) {
  RowGeom g;
  g.y0 = ya & ~(VEC - 1);
  g.groups = (yb - g.y0 + VEC - 1) / VEC;
  const int gm1 = max(g.groups - 1, 0);
  g.shift = gm1 == 0 ? 0 : min(32 - __clz(gm1), 6);
  g.lpr = 1 << g.shift;
  g.rpw = 64 >> g.shift;
  return g;
}

template <int VEC>
struct CellVec {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ CellVec<VEC> load_cells(const float* p) {
  CellVec<VEC> r;
  if (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1 % VEC] = t.y; r.v[2 % VEC] = t.z; r.v[3 % VEC] = t.w;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int VEC>
__device__ __forceinline__ void store_cells(float* p, const CellVec<VEC>& r) {
  if (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % VEC], r.v[2 % VEC], r.v[3 % VEC]);
  } else {
    p[0] = r.v[0];
  }
}
// observation bits of a lane's cell group in a code / flips tile: low nibble of one byte (VEC == 4) or one byte per cell
template <int VEC>
__device__ __forceinline__ size_t tile_index(int row, int col, int S) {  // col = y - (yu & ~3)
  return VEC == 4 ? (size_t)row * (S >> 2) + (col >> 2) : (size_t)row * S + col;
}
template <int VEC>
__device__ __forceinline__ uint32_t load_bits(const uint8_t* tile, int row, int col, int S) {
  return tile[tile_index<VEC>(row, col, S)] & (VEC == 4 ? 0xFu : 1u);
}
template <int VEC>
__device__ __forceinline__ void store_bits(uint8_t* tile, int row, int col, int S, uint32_t bits) {
  tile[tile_index<VEC>(row, col, S)] = (uint8_t)bits;
}

// ======================================================================================================
// K3: sense + Bayesian update of the agent's own footprint tile
// ======================================================================================================
template <int VEC, int UNR>
__global__ void __launch_bounds__(256)
k_sense_update(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
               const int32_t* __restrict__ pos, const uint8_t* __restrict__ truth, float* __restrict__ local,
               const uint8_t* __restrict__ flips, uint8_t* __restrict__ code, int32_t* __restrict__ rect_out,
               int32_t* __restrict__ ws, unsigned long long* __restrict__ counters, int stage, int agent_sel,
               int split) {
  const int n = c->n_agents;
  const int tile = blockIdx.x / split, part = blockIdx.x % split;  // (tile, row part) flattened: grid.x has no 65535 limit
  int e, i;
  if (agent_sel >= 0) { e = tile; i = agent_sel; }
  else { e = tile / n; i = tile % n; }
  const int gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  const int32_t* p = pos + (size_t)(e * n + i) * 3;
  int r[4];
  ippm_footprint_rect(c, p[0], p[1], p[2], r, nullptr);
  const int yu = r[0], yd = r[1], xl = r[2], xr = r[3];
  if (part == 0 && threadIdx.x < 4) rect_out[(size_t)(e * n + i) * 4 + threadIdx.x] = r[threadIdx.x];
  const int h = xr - xl, w = yd - yu;
  if (h <= 0 || w <= 0) return;
  const int k = ippm_alt_index(c, p[2]);
  const float lm0 = c->logit_meas[k][0], lm1 = c->logit_meas[k][1];
  const uint32_t thr = c->flip_threshold[k];
  const float lc = c->logit_clip;
  const RowGeom g = make_geom<VEC>(yu, yd);
  const int tile_y0 = yu & ~3;
  const int rows_per_wg = (h + split - 1) / split;
  const int r0 = part * rows_per_wg, r1 = min(h, r0 + rows_per_wg);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane >> g.shift, gl = lane & (g.lpr - 1);
  float* map = local + (size_t)(e * n + i) * gx * gy;
  const uint8_t* tr = truth + (size_t)e * ippm_truth_bytes(gx, gy);
  const size_t TB = ippm_tile_bytes(S, VEC);
  uint8_t* cd = code + (size_t)(e * n + i) * TB;
  const uint8_t* fl = flips ? flips + (size_t)(e * n + i) * TB : nullptr;
  const int64_t ep = episode ? episode[e] : 0;
  const uint32_t sw = ippm_stream_word((uint32_t)i, (uint32_t)stage, IPPM_DOMAIN_FLIP);
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
  const int stride = 4 * g.rpw;
  bool exceed = false;
  for (int gi = gl; gi < g.groups; gi += g.lpr) {
    const int y = g.y0 + gi * VEC;
    for (int row = r0 + wv * g.rpw + sub; row < r1; row += stride * UNR) {
      // UNR independent rows per lane: all their loads are in flight before the first use
      CellVec<VEC> m[UNR];
      uint32_t tw[UNR], fw[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int rr = row + u * stride;
        tw[u] = 0; fw[u] = 0;
        if (rr < r1) {
          const size_t cell = (size_t)(xl + rr) * gy + y;
          m[u] = load_cells<VEC>(map + cell);
          tw[u] = VEC == 4 ? ippm_truth4(tr, cell) : ippm_truth1(tr, cell);
          if (fl) fw[u] = load_bits<VEC>(fl, rr, y - tile_y0, S);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int rr = row + u * stride;
        if (rr >= r1) continue;
        const size_t cell = (size_t)(xl + rr) * gy + y;
        Philox4 ph;
        if (!fl && VEC == 4) ph = ippm_philox((uint32_t)(cell >> 2), (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1);
        uint32_t cw = 0;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          // branch-free: cells of an edge group that lie outside the footprint keep their value
          const bool in = (unsigned)(y + q - yu) < (unsigned)w;
          uint32_t flip;
          if (fl) flip = (fw[u] >> q) & 1u;
          else if (VEC == 4) flip = ph.v[q] < thr ? 1u : 0u;
          else {
            Philox4 p1 = ippm_philox((uint32_t)((cell + q) >> 2), (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1);
            flip = p1.v[(cell + q) & 3] < thr ? 1u : 0u;
          }
          const uint32_t obs = ((tw[u] >> q) & 1u) ^ flip;
          // mappings.py:109-124 in log-odds: clip the prior belief, add the measurement's log-odds
          const float l = ippm_clampl(m[u].v[q], lc) + (obs ? lm1 : lm0);
          exceed |= in & (fabsf(l) > lc);
          m[u].v[q] = in ? l : m[u].v[q];
          cw |= (in ? obs : 0u) << q;
        }
        store_cells<VEC>(map + cell, m[u]);
        store_bits<VEC>(cd, rr, y - tile_y0, S, cw);
      }
    }
  }
  if (ws && __any(exceed) && lane == 0) ws[(size_t)(e * (n + 1) + i) * IPPM_WS_WORDS + WS_FLAG_S] = 1;
  if (counters && part == 0 && threadIdx.x == 0)
    atomicAdd(&counters[(tile & (IPPM_COUNTER_SLOTS - 1)) * 8 + 0], (unsigned long long)h * w);
}

// ======================================================================================================
// comm matrix
// ======================================================================================================
// row i of the comm matrix of env e: bit j set <=> agent i hears agent j (communication_log.py:39-58); also stored as bytes
__device__ __forceinline__ uint32_t comm_row(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                                             const int32_t* __restrict__ pos, const float* __restrict__ comm_range,
                                             const double* __restrict__ draws, uint8_t* __restrict__ comm, int t, int e, int i) {
  const int n = c->n_agents;
  const int32_t* pi = pos + (size_t)(e * n + i) * 3;
  const double range = comm_range ? (double)comm_range[e] : c->comm_range;
  const int64_t ep = episode ? episode[e] : 0;
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
  uint32_t row = 0;
  for (int j = 0; j < n; ++j) {
    const int32_t* pj = pos + (size_t)(e * n + j) * 3;
    long long dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
    long long d2 = dx * dx + dy * dy + dz * dz;
    double u;
    if (draws) u = draws[(size_t)(e * n + i) * n + j];
    else {
      Philox4 ph = ippm_philox((uint32_t)j, (uint32_t)ep, ippm_stream_word((uint32_t)i, (uint32_t)t, IPPM_DOMAIN_COMM),
                               (uint32_t)(ep >> 32), k0, k1);
      u = (double)ph.v[0] * (1.0 / 4294967296.0);
    }
    bool ok = d2 == 0;
    if (d2 > 0) {
      double dist = sqrt((double)d2);
      if (dist <= range && u >= c->failure_rate) ok = true;
    }
    comm[(size_t)(e * n + i) * n + j] = ok ? 1 : 0;
    row |= ok ? (1u << j) : 0u;
  }
  return row;
}

__global__ void k_comm(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                       const int32_t* __restrict__ pos, const float* __restrict__ comm_range,
                       const double* __restrict__ draws, uint8_t* __restrict__ comm, int t, int n_envs) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = c->n_agents;
  if (tid >= n_envs * n) return;
  comm_row(c, episode, pos, comm_range, draws, comm, t, tid / n, tid % n);
}

// ======================================================================================================
// fusion planning (one thread per map): builds the ordered op list of K4 / K5 and maintains the
// deferred-clamp state (the reference's full-grid input clip, applied only where it can matter)
// ======================================================================================================
__device__ __forceinline__ void plan_push(int32_t* w, int& nops, int type, int src, int alt, const int32_t* r,
                                          int& x0, int& x1, int& y0, int& y1) {
  if (r[3] <= r[2] || r[1] <= r[0]) return;
  int32_t* op = w + WS_OPS + nops * OP_WORDS;
  op[OP_TYPE] = type; op[OP_SRC] = src; op[OP_ALT] = alt;
  op[OP_YU] = r[0]; op[OP_YD] = r[1]; op[OP_XL] = r[2]; op[OP_XR] = r[3];
  x0 = min(x0, r[2]); x1 = max(x1, r[3]); y0 = min(y0, r[0]); y1 = max(y1, r[1]);
  ++nops;
}

// plans map i of env e (i == n: the global map); recv = agents whose measurements map i receives this step
__device__ __forceinline__ void plan_map(const ippm_config* __restrict__ c, const int32_t* __restrict__ rect,
                                         const int32_t* __restrict__ pos, uint32_t recv, int32_t* __restrict__ ws,
                                         int global_maps, int e, int i) {
  const int n = c->n_agents;
  int32_t* w = ws + (size_t)(e * (n + 1) + i) * IPPM_WS_WORDS;
  int nops = 0, x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
  int last_src = -1;
  for (int j = 0; j < n; ++j) {
    bool take = global_maps ? true : (j != i && ((recv >> j) & 1u) != 0);
    if (take) last_src = j;
  }
  int32_t* hdr = w + WS_PLAN;
  if (last_src < 0) {  // nothing received: the map is untouched; carry possible out-of-range regions forward
    if (!global_maps && w[WS_FLAG_S]) {
      const int32_t* ri = rect + (size_t)(e * n + i) * 4;
      if (w[WS_FLAG_A]) {
        w[WS_RECT_A + 0] = min(w[WS_RECT_A + 0], ri[0]); w[WS_RECT_A + 1] = max(w[WS_RECT_A + 1], ri[1]);
        w[WS_RECT_A + 2] = min(w[WS_RECT_A + 2], ri[2]); w[WS_RECT_A + 3] = max(w[WS_RECT_A + 3], ri[3]);
      } else {
        for (int q = 0; q < 4; ++q) w[WS_RECT_A + q] = ri[q];
      }
      w[WS_FLAG_A] = 1;
      w[WS_FLAG_S] = 0;
    }
    hdr[PL_NOPS] = 0;
    return;
  }
  if (w[WS_FLAG_A]) plan_push(w, nops, 0, -1, 0, w + WS_RECT_A, x0, x1, y0, y1);
  if (!global_maps && w[WS_FLAG_S]) plan_push(w, nops, 0, -1, 0, rect + (size_t)(e * n + i) * 4, x0, x1, y0, y1);
  int last_op = -1;
  for (int j = 0; j < n; ++j) {
    bool take = global_maps ? true : (j != i && ((recv >> j) & 1u) != 0);
    if (!take) continue;
    const int32_t* rj = rect + (size_t)(e * n + j) * 4;
    int before = nops;
    plan_push(w, nops, 1, j, ippm_alt_index(c, pos[(size_t)(e * n + j) * 3 + 2]), rj, x0, x1, y0, y1);
    if (j == last_src) {
      last_op = nops > before ? nops - 1 : -1;  // an empty last footprint leaves no unclamped outputs
      for (int q = 0; q < 4; ++q) w[WS_RECT_A + q] = rj[q];
    }
  }
  w[WS_FLAG_A] = 0;  // set again by the fusion kernel if the last op leaves out-of-range values
  w[WS_FLAG_S] = 0;
  hdr[PL_NOPS] = nops;
  hdr[PL_X0] = x0; hdr[PL_X1] = x1; hdr[PL_Y0] = y0; hdr[PL_Y1] = y1;
  hdr[PL_LAST] = last_op;
}

__global__ void k_plan(const ippm_config* __restrict__ c, const int32_t* __restrict__ rect,
                       const int32_t* __restrict__ pos, const uint8_t* __restrict__ comm, int32_t* __restrict__ ws,
                       int global_maps, int n_envs, int agent_sel) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = c->n_agents;
  const int per = (global_maps || agent_sel >= 0) ? 1 : n;
  if (tid >= n_envs * per) return;
  const int e = tid / per;
  const int i = global_maps ? n : (agent_sel >= 0 ? agent_sel : tid % n);
  uint32_t recv = 0;
  if (!global_maps)
    for (int j = 0; j < n; ++j) recv |= comm[(size_t)(e * n + i) * n + j] ? (1u << j) : 0u;
  plan_map(c, rect, pos, recv, ws, global_maps, e, i);
}

// comm matrix + local-fusion plans in one launch (one thread per (env, agent)): the two small kernels sit on the critical
// path of every step and each costs a launch latency of its own (18 + 28 us while K5 keeps the GPU busy)
__global__ void k_comm_plan(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                            const int32_t* __restrict__ pos, const float* __restrict__ comm_range,
                            const double* __restrict__ draws, uint8_t* __restrict__ comm, const int32_t* __restrict__ rect,
                            int32_t* __restrict__ ws, int t, int n_envs) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = c->n_agents;
  if (tid >= n_envs * n) return;
  const int e = tid / n, i = tid % n;
  const uint32_t recv = comm_row(c, episode, pos, comm_range, draws, comm, t, e, i);
  plan_map(c, rect, pos, recv, ws, 0, e, i);
}

// ======================================================================================================
// K4 / K5: apply the planned ops to a map, each touched cell read once and written once.
// REWARD: also accumulate the information-gain reward terms of K5 (utils/reward.py:68-82).
// ======================================================================================================
// Work decomposition: one workgroup column per (map, op).  The workgroups of op k walk the rows of ITS
// rectangle (dense lanes, like K3) and own every 4-cell group that no later op touches; an owned group gets the
// complete ordered chain of all ops covering each of its cells.  Every group of the union is therefore read
// and written exactly once, by exactly one workgroup, whatever the overlap pattern.
//
// A lane keeps its column group while it walks down the rows, so everything that depends on columns only (which
// cells of the group each op covers) is folded into a few bit masks once per column chunk; per row only the
// row-range tests remain.  NK = ops held in registers (scalar loads, fully unrolled).
struct OpRec {
  int info;  // type | src << 8 | alt << 16
  int yu, yd, xl, xr;
};

template <int VEC, bool REWARD, int NK>
__global__ void __launch_bounds__(256)
k_apply_ops(const ippm_config* __restrict__ c, float* __restrict__ maps, const uint8_t* __restrict__ code,
            const int32_t* __restrict__ plan_ro, int32_t* __restrict__ ws, double* __restrict__ sums,
            unsigned long long* __restrict__ counters, int split, int min_ops, int agent_sel) {
  const int n = c->n_agents;
  const int part = blockIdx.x % split;
  // map index: (e,i) for local maps (one agent per env when agent_sel >= 0), e for global maps
  const int m = (!REWARD && agent_sel >= 0) ? (blockIdx.x / split) * n + agent_sel : blockIdx.x / split;
  const int k = blockIdx.y;  // op whose rectangle this workgroup walks
  const int e = REWARD ? m : m / n;
  const int slot = REWARD ? n : m % n;
  const size_t wbase = (size_t)(e * (n + 1) + slot) * IPPM_WS_WORDS;
  const int32_t* __restrict__ hdr = plan_ro + wbase + WS_PLAN;
  const int nops = hdr[PL_NOPS];
  if (k >= nops || nops > NK || nops < min_ops) return;  // (another instantiation handles other plan sizes)
  __shared__ float s_red[4][6];
  OpRec op[NK];
#pragma unroll
  for (int o = 0; o < NK; ++o) {
    const int32_t* p = plan_ro + wbase + WS_OPS + o * OP_WORDS;  // uniform address: scalar loads
    const bool on = o < nops;
    op[o].info = on ? (p[OP_TYPE] | (p[OP_SRC] << 8) | (p[OP_ALT] << 16)) : 0;
    op[o].yu = on ? p[OP_YU] : 0; op[o].yd = on ? p[OP_YD] : 0;
    op[o].xl = on ? p[OP_XL] : 0; op[o].xr = on ? p[OP_XR] : 0;  // empty rect: never covers
  }
  int kyu = 0, kyd = 0, kxl = 0, kxr = 0;
#pragma unroll
  for (int o = 0; o < NK; ++o)
    if (o == k) { kyu = op[o].yu; kyd = op[o].yd; kxl = op[o].xl; kxr = op[o].xr; }
  const int gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  const bool k_is_last = hdr[PL_LAST] == k;
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  const RowGeom g = make_geom<VEC>(kyu, kyd);
  const int rows = kxr - kxl;
  const int rows_per_wg = (rows + split - 1) / split;
  const int r0 = part * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane >> g.shift, gl = lane & (g.lpr - 1);
  float* map = maps + (size_t)m * gx * gy;
  const size_t TB = ippm_tile_bytes(S, VEC);
  const uint8_t* code_e = code + (size_t)e * n * TB;
  bool exceed = false;
  float a1 = 0.f, aD = 0.f, aT = 0.f;
  unsigned cells = 0, opcells = 0;
  // Does any other op's rectangle intersect the rows/columns this workgroup walks?  If not (the common case) every
  // group is covered by op k alone: a short branch-free loop does the job.
  bool alone = true;
  unsigned hitmask = 0;  // ops (including k) that can touch a group this workgroup walks: all others are skipped wholesale
#pragma unroll
  for (int o = 0; o < NK; ++o) {
    // column ranges widened to whole VEC-cell groups: ownership is decided per group, so two rectangles that merely
    // share an edge group already interact
    const bool hit = op[o].xl < kxl + r1 && op[o].xr > kxl + r0 && (op[o].yu & ~(VEC - 1)) < ((kyd + VEC - 1) & ~(VEC - 1)) &&
                     ((op[o].yd + VEC - 1) & ~(VEC - 1)) > (kyu & ~(VEC - 1));
    alone &= (o == k) || !hit;
    hitmask |= (hit || o == k) ? (1u << o) : 0u;
  }
  if (alone) {
    int kinfo = 0;
#pragma unroll
    for (int o = 0; o < NK; ++o)
      if (o == k) kinfo = op[o].info;
    const bool isf = (kinfo & 0xFF) != 0;
    const int alt = (kinfo >> 16) & 0xFF;
    const float lm0 = isf ? c->logit_meas[alt][0] : 0.f, lm1 = isf ? c->logit_meas[alt][1] : 0.f;
    const uint8_t* ctile = code_e + (size_t)((kinfo >> 8) & 0xFF) * TB;
    const int wdt = kyd - kyu;
    for (int gi = gl; gi < g.groups; gi += g.lpr) {
      const int y = g.y0 + gi * VEC;
      unsigned inm = 0;
#pragma unroll
      for (int q = 0; q < VEC; ++q) inm |= ((unsigned)(y + q - kyu) < (unsigned)wdt) ? (1u << q) : 0u;
      constexpr int FU = 2;  // independent rows in flight per lane (latency hiding at the low occupancy of these launches)
      const int rstride = 4 * g.rpw;
      for (int row0 = r0 + wv * g.rpw + sub; row0 < r1; row0 += rstride * FU) {
        CellVec<VEC> mvu[FU];
        uint32_t cwu[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int row = row0 + u * rstride;
          cwu[u] = 0;
          if (row < r1) {
            mvu[u] = load_cells<VEC>(map + (size_t)(kxl + row) * gy + y);
            if (isf) cwu[u] = load_bits<VEC>(ctile, row, y - (kyu & ~3), S);
          }
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int row = row0 + u * rstride;
          if (row >= r1) continue;
          CellVec<VEC>& mv = mvu[u];
          const uint32_t cw = cwu[u];
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            const float b = mv.v[q];
            float a = ippm_clampl(b, lc) + (((cw >> q) & 1u) ? lm1 : lm0);
            a = k_is_last ? a : ippm_clampl(a, lc);
            const bool in = (inm >> q) & 1u;
            a = in ? a : b;
            exceed |= fabsf(a) > lc && in;
            mv.v[q] = a;
            if (REWARD) {
              const float sel = (in && isf) ? 1.f : 0.f;
              const float wa = ippm_weight_l(a, wt), wb = ippm_weight_l(b, wt);
              const float hb = ippm_entropy_l(b, lc), ha = ippm_entropy_l(a, lc);
              a1 += sel * (wa * (hb - ha));
              aD += sel * ((wa - wb) * hb);
              aT += sel * (wa * ha - wb * hb);
            }
          }
          cells += __popc(inm);
          store_cells<VEC>(map + (size_t)(kxl + row) * gy + y, mv);
        }
      }
    }
    opcells = cells;
  } else {
  using Mask = typename std::conditional<(NK * VEC > 32), unsigned long long, unsigned>::type;
  static_assert(NK * VEC <= 64, "op masks are at most 64 bits");
  constexpr unsigned QM = (1u << VEC) - 1u;
  for (int gi = gl; gi < g.groups; gi += g.lpr) {
    const int y = g.y0 + gi * VEC;
    // column-only part: cmask holds, VEC bits per op, which cells of my group lie inside the op's column range
    Mask cmask = 0;
#pragma unroll
    for (int o = 0; o < NK; ++o) {
      if (!((hitmask >> o) & 1u)) continue;
      unsigned mq = 0;
#pragma unroll
      for (int q = 0; q < VEC; ++q) mq |= ((unsigned)(y + q - op[o].yu) < (unsigned)(op[o].yd - op[o].yu)) ? (1u << q) : 0u;
      cmask |= (Mask)mq << (o * VEC);
    }
    for (int row = r0 + wv * g.rpw + sub; row < r1; row += 4 * g.rpw) {
      const int x = kxl + row;
      // row part: act = cells covered by op o in this row, for all ops
      Mask act = 0;
#pragma unroll
      for (int o = 0; o < NK; ++o) {
        if (!((hitmask >> o) & 1u)) continue;
        const bool rowin = (unsigned)(x - op[o].xl) < (unsigned)(op[o].xr - op[o].xl);
        act |= rowin ? (cmask & ((Mask)QM << (o * VEC))) : (Mask)0;
      }
      // ownership: a later op touching any cell of this group takes it over
      if (k + 1 < NK && (act >> ((k + 1) * VEC)) != 0) continue;
      const size_t cell = (size_t)x * gy + y;
      // issue every load of this group (map cells + the measurement codes of all covering ops) before any use
      CellVec<VEC> mv = load_cells<VEC>(map + cell);
      uint32_t cw[NK];
#pragma unroll
      for (int o = 0; o < NK; ++o) {
        cw[o] = 0;
        if (!((hitmask >> o) & 1u)) continue;
        if (o <= k && (op[o].info & 0xFF) && ((unsigned)(act >> (o * VEC)) & QM))
          cw[o] = load_bits<VEC>(code_e + (size_t)((op[o].info >> 8) & 0xFF) * TB, x - op[o].xl, y - (op[o].yu & ~3), S);
      }
      const CellVec<VEC> old = mv;
      float L[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) L[q] = mv.v[q];
      unsigned touched = 0, fusedm = 0;
      // ordered clamp/add chain (mappings.py:80-124 in log-odds); ops that cover no lane of the wavefront are skipped
#pragma unroll
      for (int o = 0; o < NK; ++o) {
        if (o > k) break;
        if (!((hitmask >> o) & 1u)) continue;
        const unsigned inm = (unsigned)(act >> (o * VEC)) & QM;
        if (!__any(inm != 0u)) continue;
        const bool isf = (op[o].info & 0xFF) != 0;
        const int alt = (op[o].info >> 16) & 0xFF;
        const float lm0 = isf ? c->logit_meas[alt][0] : 0.f, lm1 = isf ? c->logit_meas[alt][1] : 0.f;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          // every op of the reference clips its input over the whole grid (mappings.py:110-111)
          const float l = ippm_clampl(L[q], lc) + (((cw[o] >> q) & 1u) ? lm1 : lm0);
          L[q] = ((inm >> q) & 1u) ? l : L[q];
        }
        touched |= inm;
        fusedm |= isf ? inm : 0u;
        opcells += __popc(inm);
      }
      cells += __popc(touched);
      // outputs of the plan's last op stay unclamped; every other cell was clipped again by a later full-grid op
      const unsigned keep = k_is_last ? ((unsigned)(act >> (k * VEC)) & QM) : 0u;
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        const float b = mv.v[q];
        float a = ((keep >> q) & 1u) ? L[q] : ippm_clampl(L[q], lc);
        a = ((touched >> q) & 1u) ? a : b;
        exceed |= fabsf(a) > lc && ((touched >> q) & 1u);
        mv.v[q] = a;
      }
      if (REWARD && __any(fusedm != 0)) {
        // information-gain terms of the cells that received a measurement (utils/reward.py:68-82)
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const float sel = ((fusedm >> q) & 1u) ? 1.f : 0.f;
          const float b = old.v[q], a = mv.v[q];
          const float wa = ippm_weight_l(a, wt), wb = ippm_weight_l(b, wt);
          const float hb = ippm_entropy_l(b, lc), ha = ippm_entropy_l(a, lc);
          a1 += sel * (wa * (hb - ha));
          aD += sel * ((wa - wb) * hb);
          aT += sel * (wa * ha - wb * hb);
        }
      }
      store_cells<VEC>(map + cell, mv);
    }
  }
  }  // !alone
  if (__any(exceed) && lane == 0) ws[wbase + WS_FLAG_A] = 1;
  // block reduction of the reward terms and work counters: one atomic per workgroup and quantity
  {
    const float fc = ippm_wave_sum((float)cells), fo = ippm_wave_sum((float)opcells);
    if (REWARD) { a1 = ippm_wave_sum(a1); aD = ippm_wave_sum(aD); aT = ippm_wave_sum(aT); }
    if (lane == 0) { s_red[wv][0] = a1; s_red[wv][1] = aD; s_red[wv][2] = aT; s_red[wv][3] = fc; s_red[wv][4] = fo; }
    __syncthreads();
    if (threadIdx.x < 5) {
      const float t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
      if (threadIdx.x < 3) {
        if (REWARD && t != 0.f) atomicAdd(&sums[(size_t)e * 8 + SUM_ACC1 + threadIdx.x], (double)t);
      } else if (counters && t > 0.f) {
        const int cslot = blockIdx.x & (IPPM_COUNTER_SLOTS - 1);
        atomicAdd(&counters[cslot * 8 + (REWARD ? 3 : 1) + (threadIdx.x - 3)], (unsigned long long)t);
      }
    }
  }
}

// Fallback for plans with more than 10 ops (more than 8 agents): walks the bounding hull of the plan with the op
// table in LDS.  Same per-cell semantics, no attempt at speed.
template <int VEC, bool REWARD>
__global__ void __launch_bounds__(256)
k_apply_ops_generic(const ippm_config* __restrict__ c, float* __restrict__ maps, const uint8_t* __restrict__ code,
                    int32_t* __restrict__ ws, double* __restrict__ sums, unsigned long long* __restrict__ counters, int split,
                    int min_ops, int agent_sel) {
  const int n = c->n_agents;
  const int part = blockIdx.x % split;
  const int m = (!REWARD && agent_sel >= 0) ? (blockIdx.x / split) * n + agent_sel : blockIdx.x / split;
  const int e = REWARD ? m : m / n;
  const int slot = REWARD ? n : m % n;
  int32_t* w = ws + (size_t)(e * (n + 1) + slot) * IPPM_WS_WORDS;
  const int32_t* hdr = w + WS_PLAN;
  const int nops = hdr[PL_NOPS];
  if (nops < min_ops) return;
  __shared__ int32_t s_ops[IPPM_MAX_OPS * OP_WORDS];
  __shared__ float s_red[4][6];
  for (int q = threadIdx.x; q < nops * OP_WORDS; q += blockDim.x) s_ops[q] = w[WS_OPS + q];
  __syncthreads();
  const int gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  const int X0 = hdr[PL_X0], X1 = hdr[PL_X1], last_op = hdr[PL_LAST];
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  const RowGeom g = make_geom<VEC>(hdr[PL_Y0], hdr[PL_Y1]);
  const int rows = X1 - X0;
  const int rows_per_wg = (rows + split - 1) / split;
  const int r0 = part * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane >> g.shift, gl = lane & (g.lpr - 1);
  float* map = maps + (size_t)m * gx * gy;
  const size_t TB = ippm_tile_bytes(S, VEC);
  const uint8_t* code_e = code + (size_t)e * n * TB;
  bool exceed = false;
  float a1 = 0.f, aD = 0.f, aT = 0.f;
  unsigned cells = 0, opcells = 0;
  for (int row = r0 + wv * g.rpw + sub; row < r1; row += 4 * g.rpw) {
    const int x = X0 + row;
    for (int gi = gl; gi < g.groups; gi += g.lpr) {
      const int y = g.y0 + gi * VEC;
      bool need = false;
      for (int o = 0; o < nops; ++o) {
        const int32_t* op = s_ops + o * OP_WORDS;
        need |= (x >= op[OP_XL] && x < op[OP_XR] && y + VEC > op[OP_YU] && y < op[OP_YD]);
      }
      if (!need) continue;
      const size_t cell = (size_t)x * gy + y;
      CellVec<VEC> mv = load_cells<VEC>(map + cell);
      float L[VEC];
      int lastt[VEC];
      bool fused[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) { L[q] = mv.v[q]; lastt[q] = -1; fused[q] = false; }
      for (int o = 0; o < nops; ++o) {
        const int32_t* op = s_ops + o * OP_WORDS;
        if (!(x >= op[OP_XL] && x < op[OP_XR] && y + VEC > op[OP_YU] && y < op[OP_YD])) continue;
        uint32_t cw = 0;
        float lm0 = 0.f, lm1 = 0.f;
        if (op[OP_TYPE]) {
          cw = load_bits<VEC>(code_e + (size_t)op[OP_SRC] * TB, x - op[OP_XL], y - (op[OP_YU] & ~3), S);
          lm0 = c->logit_meas[op[OP_ALT]][0];
          lm1 = c->logit_meas[op[OP_ALT]][1];
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const int yy = y + q;
          if (yy >= op[OP_YU] && yy < op[OP_YD]) {
            L[q] = ippm_clampl(L[q], lc);
            if (op[OP_TYPE]) { L[q] += ((cw >> q) & 1u) ? lm1 : lm0; fused[q] = true; }
            lastt[q] = o;
            ++opcells;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        if (lastt[q] < 0) continue;
        ++cells;
        const float b = mv.v[q];
        float a = L[q];
        if (lastt[q] != last_op) a = ippm_clampl(a, lc);
        exceed |= fabsf(a) > lc;
        mv.v[q] = a;
        if (REWARD && fused[q]) {
          const float wa = ippm_weight_l(a, wt), wb = ippm_weight_l(b, wt);
          const float hb = ippm_entropy_l(b, lc), ha = ippm_entropy_l(a, lc);
          a1 += wa * (hb - ha);
          aD += (wa - wb) * hb;
          aT += wa * ha - wb * hb;
        }
      }
      store_cells<VEC>(map + cell, mv);
    }
  }
  if (__any(exceed) && lane == 0) w[WS_FLAG_A] = 1;
  {
    const float fc = ippm_wave_sum((float)cells), fo = ippm_wave_sum((float)opcells);
    if (REWARD) { a1 = ippm_wave_sum(a1); aD = ippm_wave_sum(aD); aT = ippm_wave_sum(aT); }
    if (lane == 0) { s_red[wv][0] = a1; s_red[wv][1] = aD; s_red[wv][2] = aT; s_red[wv][3] = fc; s_red[wv][4] = fo; }
    __syncthreads();
    if (threadIdx.x < 5) {
      const float t = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
      if (threadIdx.x < 3) {
        if (REWARD && t != 0.f) atomicAdd(&sums[(size_t)e * 8 + SUM_ACC1 + threadIdx.x], (double)t);
      } else if (counters && t > 0.f) {
        const int cslot = blockIdx.x & (IPPM_COUNTER_SLOTS - 1);
        atomicAdd(&counters[cslot * 8 + (REWARD ? 3 : 1) + (threadIdx.x - 3)], (unsigned long long)t);
      }
    }
  }
}

__global__ void k_reward_finalize(const ippm_config* __restrict__ c, double* __restrict__ sums,
                                  float* __restrict__ reward, int n_envs) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  double* s = sums + (size_t)e * 8;
  const double s1 = s[SUM_ACC1];
  const double s2 = s[SUM_T] + s[SUM_ACCD];
  s[SUM_S1] = s1;
  s[SUM_S2] = s2;
  s[SUM_T] += s[SUM_ACCT];
  s[SUM_ACC1] = 0; s[SUM_ACCD] = 0; s[SUM_ACCT] = 0;
  const double cells = (double)c->grid_x * (double)c->grid_y;
  reward[e * 2] = (float)(22.0 * (s1 / s2) - 0.5);        // utils/reward.py:38-40
  reward[e * 2 + 1] = (float)(10.0 * (s1 / cells) - 0.17);  // utils/reward.py:37
}

// full-grid weighted entropy per map (initialisation of T, evaluation metrics)
__global__ void __launch_bounds__(256)
k_weighted_entropy(const ippm_config* __restrict__ c, const float* __restrict__ maps, const uint8_t* __restrict__ truth,
                   double* __restrict__ out, int maps_per_truth) {
  const int m = blockIdx.y;
  const size_t total = (size_t)c->grid_x * c->grid_y;
  const float* p = maps + (size_t)m * total;
  const uint8_t* t = truth ? truth + (size_t)(m / maps_per_truth) * ippm_truth_bytes(c->grid_x, c->grid_y) : nullptr;
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float v = p[i];
    const float wgt = t ? (float)ippm_truth1(t, i) : ippm_weight_l(v, wt);
    acc += wgt * ippm_entropy_l(v, lc);
  }
  acc = ippm_wave_sum(acc);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[m], (double)(s[0] + s[1] + s[2] + s[3]));
}

// ======================================================================================================
// K1: action mask + collision mask + action choice + move, sequential over the agents of one env
// ======================================================================================================
__device__ __forceinline__ void action_offset(int A, int a, int s, int& dx, int& dy, int& dz) {
  dx = dy = dz = 0;
  if (A == 4) {
    if (a == 0) dx = -s; else if (a == 1) dy = -s; else if (a == 2) dy = s; else dx = s;
  } else if (A == 6) {
    if (a == 0) dz = s; else if (a == 1) dx = -s; else if (a == 2) dy = -s; else if (a == 3) dy = s;
    else if (a == 4) dx = s; else dz = -s;
  } else if (A == 9) {
    dx = (a / 3 - 1) * s; dy = (a % 3 - 1) * s;
  } else {  // 27: layer 0 = +z (action_space.py:249-303)
    int layer = a / 9, c9 = a % 9;
    dz = (1 - layer) * s; dx = (c9 / 3 - 1) * s; dy = (c9 % 3 - 1) * s;
  }
}

__device__ __forceinline__ uint32_t boundary_mask(const ippm_config* c, int px, int py, int pz) {
  const int A = c->n_actions, s = c->spacing;
  const int max_alt = c->min_altitude + (c->space_z - 1) * s;
  uint32_t m = 0;
  for (int a = 0; a < A; ++a) {
    int dx, dy, dz;
    action_offset(A, a, s, dx, dy, dz);
    int nx = px + dx, ny = py + dy, nz = pz + dz;
    bool ok = nx >= 0 && nx <= c->x_dim_m && ny >= 0 && ny <= c->y_dim_m;
    if (A == 6 || A == 27) ok = ok && nz >= c->min_altitude && nz <= max_alt;
    if ((A == 9 || A == 27) && dx == 0 && dy == 0 && dz == 0) ok = false;
    if (ok) m |= 1u << a;
  }
  return m;
}

// actions zeroed when an already-moved agent sits at lattice offset (dx,dy,dz) (action_space.py:309-589)
__device__ __forceinline__ uint32_t collision_bits(int A, int dx, int dy, int dz) {
  if (A == 4) {
    if (dx == -1 && dy == 0) return 1u; if (dx == 0 && dy == -1) return 2u;
    if (dx == 0 && dy == 1) return 4u; if (dx == 1 && dy == 0) return 8u;
    return 0;
  }
  if (A == 6) {
    if (dx == 0 && dy == 0) return (1u << 0) | (1u << 5);
    if (dx == -1 && dy == 0) return 1u << 1; if (dx == 0 && dy == -1) return 1u << 2;
    if (dx == 0 && dy == 1) return 1u << 3; if (dx == 1 && dy == 0) return 1u << 4;
    return 0;
  }
  if (dx < -1 || dx > 1 || dy < -1 || dy > 1) return 0;
  int c9 = (dx + 1) * 3 + (dy + 1);
  if (A == 9) return (dx == 0 && dy == 0) ? 0u : (1u << c9);
  if (dz < -1 || dz > 1 || (dx == 0 && dy == 0 && dz == 0)) return 0;
  if (dx == 0 && dy == 0) return (1u << 4) | (1u << 22);
  return (1u << c9) | (1u << (c9 + 9)) | (1u << (c9 + 18));
}

// stand-alone mask query of the drop-in AgentActionSpace (get_action_mask / apply_collision_mask)
__global__ void k_action_mask(const ippm_config* __restrict__ c, const int32_t* __restrict__ pos,
                              const int32_t* __restrict__ others, const int32_t* __restrict__ n_others, int max_others,
                              const uint8_t* __restrict__ mask_in, uint8_t* __restrict__ mask_out,
                              int32_t* __restrict__ next_pos, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const int A = c->n_actions;
  const int px = pos[b * 3], py = pos[b * 3 + 1], pz = pos[b * 3 + 2];
  if (next_pos) {  // AgentActionSpace.action_to_position for every action (action_space.py:198-307)
    for (int a = 0; a < A; ++a) {
      int dx, dy, dz;
      action_offset(A, a, c->spacing, dx, dy, dz);
      int32_t* o = next_pos + ((size_t)b * A + a) * 3;
      o[0] = px + dx; o[1] = py + dy; o[2] = pz + dz;
    }
  }
  uint32_t m = 0;
  if (mask_in) { for (int q = 0; q < A; ++q) m |= (mask_in[(size_t)b * A + q] ? 1u : 0u) << q; }
  else m = boundary_mask(c, px, py, pz);
  int ix, iy, iz;
  ippm_pos_to_index(c, px, py, pz, ix, iy, iz);
  const int no = n_others ? n_others[b] : 0;
  for (int j = 0; j < no; ++j) {
    const int32_t* o = others + ((size_t)b * max_others + j) * 3;
    int jx, jy, jz;
    ippm_pos_to_index(c, o[0], o[1], o[2], jx, jy, jz);
    const uint32_t z = collision_bits(A, jx - ix, jy - iy, jz - iz);
    if (!z) continue;
    if (A == 6) { if (__popc(m) > 1) m &= ~z; }
    else if (A == 9) { m &= ~z; if (m == 0) m |= z; }
    else m &= ~z;
  }
  for (int q = 0; q < A; ++q) mask_out[(size_t)b * A + q] = (m >> q) & 1u;
}

__global__ void k_mask_act_move(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                                int32_t* __restrict__ pos, const float* __restrict__ probs,
                                const int32_t* __restrict__ action_in, int policy, int t, uint8_t* __restrict__ mask_out,
                                int32_t* __restrict__ action_out, int32_t* __restrict__ fault, int n_envs) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  const int n = c->n_agents, A = c->n_actions, s = c->spacing;
  const int64_t ep = episode ? episode[e] : 0;
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
  // The agents of an env move one after the other (agent i is masked against the already-moved j < i), so this is a
  // latency chain per thread.  All positions are fetched up front into LDS (3n independent loads in flight) and
  // written back once at the end: the loop itself never waits on global memory for a position.
  __shared__ int32_t s_pos[64][IPPM_MAX_AGENTS * 3 + 1];   // +1: odd row stride, no bank conflicts between threads
  int32_t* pe = s_pos[threadIdx.x];
  int32_t* pg = pos + (size_t)e * n * 3;
  for (int q = 0; q < n * 3; ++q) pe[q] = pg[q];
  int flt = 0;
  for (int i = 0; i < n; ++i) {
    const int px = pe[i * 3], py = pe[i * 3 + 1], pz = pe[i * 3 + 2];
    const uint32_t bmask = boundary_mask(c, px, py, pz);
    uint32_t m = bmask;
    int ix, iy, iz;
    ippm_pos_to_index(c, px, py, pz, ix, iy, iz);
    for (int j = 0; j < i; ++j) {  // pe[j] already holds agent j's post-move position
      int jx, jy, jz;
      ippm_pos_to_index(c, pe[j * 3], pe[j * 3 + 1], pe[j * 3 + 2], jx, jy, jz);
      const uint32_t z = collision_bits(A, jx - ix, jy - iy, jz - iz);
      if (!z) continue;
      if (A == 6) { if (__popc(m) > 1) m &= ~z; }
      else if (A == 9) { m &= ~z; if (m == 0) m |= z; }
      else m &= ~z;
    }
    int a = -1;
    if (m == 0) {
      flt |= 1 << i;  // the reference's torch.multinomial raises on an all-zero distribution
    } else if (policy == 0) {
      a = action_in[e * n + i];
    } else if (policy == 1) {
      Philox4 ph = ippm_philox(0u, (uint32_t)ep, ippm_stream_word((uint32_t)i, (uint32_t)t, IPPM_DOMAIN_ACTION),
                               (uint32_t)(ep >> 32), k0, k1);
      int kth = (int)__umulhi(ph.v[0], (uint32_t)__popc(m));
      for (int q = 0; q < A; ++q)
        if ((m >> q) & 1u) { if (kth == 0) { a = q; break; } --kth; }
    } else {
      const float* pr = probs + (size_t)(e * n + i) * A;
      if (policy == 3) {  // eval: argmax of probs*mask (first maximum)
        float best = -1.f;
        for (int q = 0; q < A; ++q) {
          float v = ((m >> q) & 1u) ? pr[q] : 0.f;
          if (v > best) { best = v; a = q; }
        }
      } else {  // train: inverse CDF over probs*mask, sequential float32 sums without FMA contraction
        float total = 0.f;
        for (int q = 0; q < A; ++q) total = __fadd_rn(total, ((m >> q) & 1u) ? pr[q] : 0.f);
        Philox4 ph = ippm_philox(0u, (uint32_t)ep, ippm_stream_word((uint32_t)i, (uint32_t)t, IPPM_DOMAIN_ACTION),
                                 (uint32_t)(ep >> 32), k0, k1);
        const float u = (float)(ph.v[0] >> 8) * (1.0f / 16777216.0f);
        const float target = __fmul_rn(u, total);
        float acc = 0.f;
        int lastv = -1;
        for (int q = 0; q < A && a < 0; ++q) {
          float v = ((m >> q) & 1u) ? pr[q] : 0.f;
          if (v > 0.f) { lastv = q; acc = __fadd_rn(acc, v); if (acc > target) a = q; }
        }
        if (a < 0) a = lastv;
        if (a < 0) flt |= 1 << i;
      }
    }
    if (a < 0 || a >= A) {  // keep the state sane: first boundary-valid action
      a = 0;
      for (int q = 0; q < A; ++q) if ((bmask >> q) & 1u) { a = q; break; }
    }
    int dx, dy, dz;
    action_offset(A, a, s, dx, dy, dz);
    pe[i * 3] = px + dx; pe[i * 3 + 1] = py + dy; pe[i * 3 + 2] = pz + dz;
    action_out[e * n + i] = a;
    for (int q = 0; q < A; ++q) mask_out[(size_t)(e * n + i) * A + q] = (m >> q) & 1u;
  }
  for (int q = 0; q < n * 3; ++q) pg[q] = pe[q];
  if (fault) fault[e] = flt;
}

// ======================================================================================================
// host API
// ======================================================================================================
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }
// tuning knob (row splits per tile/map); the defaults are the measured best on MI355X
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

extern "C" int ippm_reset_episode(ippm_ctx* ctx, const int64_t* episode, int32_t* pos, uint8_t* truth, float* local,
                                  float* global, int32_t* split_pct, float* comm_range_out, int32_t* ws, double* sums,
                                  int32_t n_envs, void* stream) {
  if (!ctx || !episode || !pos || !ws) { ippm_set_error("ippm_reset_episode: null argument"); return -1; }
  if (truth && !split_pct) { ippm_set_error("ippm_reset_episode: truth generation needs split_pct scratch"); return -1; }
  const ippm_config& c = ctx->cfg;
  const int per = c.n_agents + 1;
  hipLaunchKernelGGL(k_reset_scalars, dim3(grid1((size_t)n_envs * per, 64)), dim3(64), 0, S_(stream), ctx->dcfg, episode, pos,
                     split_pct, comm_range_out, ws, sums, n_envs);
  IPPM_LAUNCH_CHECK("reset_scalars");
  const size_t cells = (size_t)c.grid_x * c.grid_y;
  if (truth) {
    hipLaunchKernelGGL(k_fill_truth, dim3(min(64, grid1(cells)), n_envs), dim3(256), 0, S_(stream), ctx->dcfg, split_pct,
                       truth, n_envs);
    IPPM_LAUNCH_CHECK("fill_truth");
  }
  if (local) {
    hipLaunchKernelGGL(k_fill_f32, dim3(min(4096, grid1(cells * n_envs * c.n_agents))), dim3(256), 0, S_(stream), local,
                       c.logit_prior, cells * n_envs * c.n_agents);
    IPPM_LAUNCH_CHECK("fill_local");
  }
  if (global) {
    hipLaunchKernelGGL(k_fill_f32, dim3(min(4096, grid1(cells * n_envs))), dim3(256), 0, S_(stream), global, c.logit_prior,
                       cells * n_envs);
    IPPM_LAUNCH_CHECK("fill_global");
  }
  return 0;
}

extern "C" int ippm_logodds_to_prob(ippm_ctx* ctx, const float* src, float* dst, int64_t n, void* stream) {
  if (!ctx || !src || !dst) { ippm_set_error("ippm_logodds_to_prob: null argument"); return -1; }
  hipLaunchKernelGGL(k_logodds_to_prob, dim3(min(4096, grid1((size_t)n))), dim3(256), 0, S_(stream), src, dst, (size_t)n);
  IPPM_LAUNCH_CHECK("logodds_to_prob");
  return 0;
}

extern "C" int ippm_prob_to_logodds(ippm_ctx* ctx, const float* src, float* dst, int64_t n, void* stream) {
  if (!ctx || !src || !dst) { ippm_set_error("ippm_prob_to_logodds: null argument"); return -1; }
  hipLaunchKernelGGL(k_prob_to_logodds, dim3(min(4096, grid1((size_t)n))), dim3(256), 0, S_(stream), src, dst, (size_t)n);
  IPPM_LAUNCH_CHECK("prob_to_logodds");
  return 0;
}

extern "C" int ippm_clamp_logodds(ippm_ctx* ctx, float* maps, int64_t n, void* stream) {
  if (!ctx || !maps) { ippm_set_error("ippm_clamp_logodds: null argument"); return -1; }
  hipLaunchKernelGGL(k_clamp_logodds, dim3(min(4096, grid1((size_t)n))), dim3(256), 0, S_(stream), maps, ctx->cfg.logit_clip, (size_t)n);
  IPPM_LAUNCH_CHECK("clamp_logodds");
  return 0;
}

extern "C" int ippm_footprint(ippm_ctx* ctx, const int32_t* pos, int32_t* rect, int32_t* rect_unclipped, int32_t n_envs,
                              void* stream) {
  if (!ctx || !pos || !rect) { ippm_set_error("ippm_footprint: null argument"); return -1; }
  const int n = n_envs * ctx->cfg.n_agents;
  hipLaunchKernelGGL(k_footprint, dim3(grid1(n)), dim3(256), 0, S_(stream), ctx->dcfg, pos, rect, rect_unclipped, n);
  IPPM_LAUNCH_CHECK("footprint");
  return 0;
}

extern "C" int ippm_sense_update(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const uint8_t* truth,
                                 float* local, const uint8_t* flips, uint8_t* code, int32_t* rect, int32_t* ws,
                                 int32_t stage, int32_t agent_sel, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !truth || !local || !code || !rect) { ippm_set_error("ippm_sense_update: null argument"); return -1; }
  if (!flips && !episode) { ippm_set_error("ippm_sense_update: Philox flips need the episode ids"); return -1; }
  if (agent_sel >= ctx->cfg.n_agents) { ippm_set_error("ippm_sense_update: agent_sel out of range"); return -1; }
  const int maps = agent_sel >= 0 ? n_envs : n_envs * ctx->cfg.n_agents;
  const int split = std::max(1, env_int("IPPM_SPLIT_K3", 2));
  dim3 grid((unsigned)maps * split), block(256);
  const int unr = env_int("IPPM_UNROLL_K3", 2);
#define IPPM_K3_LAUNCH(V, U)                                                                                              \
  hipLaunchKernelGGL((k_sense_update<V, U>), grid, block, 0, S_(stream), ctx->dcfg, episode, pos, truth, local, flips, code, \
                     rect, ws, ctx->dcounters, stage, agent_sel, split)
  if (ctx->vec == 4) {
    if (unr >= 4) IPPM_K3_LAUNCH(4, 4);
    else if (unr >= 2) IPPM_K3_LAUNCH(4, 2);
    else IPPM_K3_LAUNCH(4, 1);
  } else {
    IPPM_K3_LAUNCH(1, 1);
  }
#undef IPPM_K3_LAUNCH
  IPPM_LAUNCH_CHECK("sense_update");
  return 0;
}

extern "C" int ippm_comm_matrix(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const float* comm_range,
                                const double* draws, uint8_t* comm, int32_t t, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !comm) { ippm_set_error("ippm_comm_matrix: null argument"); return -1; }
  if (!draws && !episode) { ippm_set_error("ippm_comm_matrix: Philox draws need the episode ids"); return -1; }
  hipLaunchKernelGGL(k_comm, dim3(grid1((size_t)n_envs * ctx->cfg.n_agents)), dim3(256), 0, S_(stream), ctx->dcfg, episode,
                     pos, comm_range, draws, comm, t, n_envs);
  IPPM_LAUNCH_CHECK("comm");
  return 0;
}

// The three instantiations share one plan: <= 6 ops and 7..10 ops take the register paths (workgroup column per
// op), larger plans the generic path.  Each launch returns immediately for plans it does not own.
template <bool REWARD>
static void launch_apply(ippm_ctx* ctx, float* maps, const uint8_t* code, int32_t* ws, double* sums, int n_maps, int split,
                         hipStream_t st, int agent_sel = -1) {
  const int max_ops = ctx->cfg.n_agents + 1;
  dim3 block(256);
#define IPPM_APPLY(V, NK, MINOPS)                                                                                      \
  hipLaunchKernelGGL((k_apply_ops<V, REWARD, NK>), dim3((unsigned)n_maps* split, std::min(max_ops, NK)), block, 0, st, \
                     ctx->dcfg, maps, code, ws, ws, sums, ctx->dcounters, split, MINOPS, agent_sel)
  if (ctx->vec == 4) {
    IPPM_APPLY(4, 6, 1);
    if (max_ops > 6) IPPM_APPLY(4, 10, 7);
    if (max_ops > 10)
      hipLaunchKernelGGL((k_apply_ops_generic<4, REWARD>), dim3((unsigned)n_maps * 8), block, 0, st, ctx->dcfg, maps, code, ws,
                         sums, ctx->dcounters, 8, 11, agent_sel);
  } else {
    IPPM_APPLY(1, 6, 1);
    if (max_ops > 6) IPPM_APPLY(1, 10, 7);
    if (max_ops > 10)
      hipLaunchKernelGGL((k_apply_ops_generic<1, REWARD>), dim3((unsigned)n_maps * 8), block, 0, st, ctx->dcfg, maps, code, ws,
                         sums, ctx->dcounters, 8, 11, agent_sel);
  }
#undef IPPM_APPLY
}

extern "C" int ippm_fuse_local(ippm_ctx* ctx, float* local, const uint8_t* code, const int32_t* rect, const int32_t* pos,
                               const uint8_t* comm, int32_t* ws, int32_t agent_sel, int32_t n_envs, void* stream) {
  if (!ctx || !local || !code || !rect || !pos || !comm || !ws) { ippm_set_error("ippm_fuse_local: null argument"); return -1; }
  if (agent_sel >= ctx->cfg.n_agents) { ippm_set_error("ippm_fuse_local: agent_sel out of range"); return -1; }
  const int maps = agent_sel >= 0 ? n_envs : n_envs * ctx->cfg.n_agents;
  hipLaunchKernelGGL(k_plan, dim3(grid1(maps, 64)), dim3(64), 0, S_(stream), ctx->dcfg, rect, pos, comm, ws, 0, n_envs, agent_sel);
  IPPM_LAUNCH_CHECK("plan_local");
  launch_apply<false>(ctx, local, code, ws, nullptr, maps, std::max(1, env_int("IPPM_SPLIT_K4", 1)), S_(stream), agent_sel);
  IPPM_LAUNCH_CHECK("fuse_local");
  return 0;
}

extern "C" int ippm_comm_fuse_local(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const float* comm_range,
                                    const double* draws, uint8_t* comm, float* local, const uint8_t* code, const int32_t* rect,
                                    int32_t* ws, int32_t t, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !comm || !local || !code || !rect || !ws) { ippm_set_error("ippm_comm_fuse_local: null argument"); return -1; }
  if (!draws && !episode) { ippm_set_error("ippm_comm_fuse_local: Philox draws need the episode ids"); return -1; }
  const int maps = n_envs * ctx->cfg.n_agents;
  hipLaunchKernelGGL(k_comm_plan, dim3(grid1(maps, 64)), dim3(64), 0, S_(stream), ctx->dcfg, episode, pos, comm_range, draws, comm,
                     rect, ws, t, n_envs);
  IPPM_LAUNCH_CHECK("comm_plan");
  launch_apply<false>(ctx, local, code, ws, nullptr, maps, std::max(1, env_int("IPPM_SPLIT_K4", 1)), S_(stream), -1);
  IPPM_LAUNCH_CHECK("fuse_local");
  return 0;
}

extern "C" int ippm_fuse_global_reward(ippm_ctx* ctx, float* global, const uint8_t* code, const int32_t* rect,
                                       const int32_t* pos, int32_t* ws, double* sums, float* reward, int32_t n_envs,
                                       void* stream) {
  if (!ctx || !global || !code || !rect || !pos || !ws || !sums || !reward) {
    ippm_set_error("ippm_fuse_global_reward: null argument");
    return -1;
  }
  hipLaunchKernelGGL(k_plan, dim3(grid1(n_envs, 64)), dim3(64), 0, S_(stream), ctx->dcfg, rect, pos, nullptr, ws, 1, n_envs, -1);
  IPPM_LAUNCH_CHECK("plan_global");
  launch_apply<true>(ctx, global, code, ws, sums, n_envs, std::max(1, env_int("IPPM_SPLIT_K5", 1)), S_(stream));
  IPPM_LAUNCH_CHECK("fuse_global");
  hipLaunchKernelGGL(k_reward_finalize, dim3(grid1(n_envs)), dim3(256), 0, S_(stream), ctx->dcfg, sums, reward, n_envs);
  IPPM_LAUNCH_CHECK("reward_finalize");
  return 0;
}

extern "C" int ippm_weighted_entropy(ippm_ctx* ctx, const float* maps, const uint8_t* truth, int32_t maps_per_truth,
                                     double* out, int32_t n_maps, void* stream) {
  if (!ctx || !maps || !out) { ippm_set_error("ippm_weighted_entropy: null argument"); return -1; }
  IPPM_HIP(hipMemsetAsync(out, 0, sizeof(double) * n_maps, S_(stream)));
  const size_t cells = (size_t)ctx->cfg.grid_x * ctx->cfg.grid_y;
  hipLaunchKernelGGL(k_weighted_entropy, dim3(min(32, grid1(cells)), n_maps), dim3(256), 0, S_(stream), ctx->dcfg, maps, truth,
                     out, maps_per_truth > 0 ? maps_per_truth : 1);
  IPPM_LAUNCH_CHECK("weighted_entropy");
  return 0;
}

extern "C" int ippm_action_mask(ippm_ctx* ctx, const int32_t* pos, const int32_t* others, const int32_t* n_others,
                                int32_t max_others, const uint8_t* mask_in, uint8_t* mask_out, int32_t* next_pos, int32_t batch,
                                void* stream) {
  if (!ctx || !pos || !mask_out) { ippm_set_error("ippm_action_mask: null argument"); return -1; }
  if (n_others && !others) { ippm_set_error("ippm_action_mask: n_others without others"); return -1; }
  hipLaunchKernelGGL(k_action_mask, dim3(grid1(batch, 64)), dim3(64), 0, S_(stream), ctx->dcfg, pos, others, n_others, max_others,
                     mask_in, mask_out, next_pos, batch);
  IPPM_LAUNCH_CHECK("action_mask");
  return 0;
}

extern "C" int ippm_mask_act_move(ippm_ctx* ctx, const int64_t* episode, int32_t* pos, const float* probs,
                                  const int32_t* action_in, int32_t policy, int32_t t, uint8_t* mask, int32_t* action,
                                  int32_t* fault, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !mask || !action) { ippm_set_error("ippm_mask_act_move: null argument"); return -1; }
  if (policy == 0 && !action_in) { ippm_set_error("ippm_mask_act_move: policy 0 needs action_in"); return -1; }
  if ((policy == 2 || policy == 3) && !probs) { ippm_set_error("ippm_mask_act_move: policy 2/3 needs probs"); return -1; }
  if ((policy == 1 || policy == 2) && !episode) { ippm_set_error("ippm_mask_act_move: sampling needs episode ids"); return -1; }
  if (policy < 0 || policy > 3) { ippm_set_error("ippm_mask_act_move: unknown policy"); return -1; }
  hipLaunchKernelGGL(k_mask_act_move, dim3(grid1(n_envs, 64)), dim3(64), 0, S_(stream), ctx->dcfg, episode, pos, probs,
                     action_in, policy, t, mask, action, fault, n_envs);
  IPPM_LAUNCH_CHECK("mask_act_move");
  return 0;
}
