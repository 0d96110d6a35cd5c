// K6: network-input builders (actor 7 planes, critic 12 planes), the exact area-average resize they rest on, and the
// full-grid reward / entropy planes of explicitly given maps.
//
//   actor/transformations.py:14-176, critic/transformations.py:17-132, utils/state.py:14-121
//
// The expensive input of every plane set is the exact area average G x G -> 11 x 11 of a belief map (the reference's
// cv2.resize INTER_AREA, utils/state.py:22-41).  Round 1 streamed all N+1 maps of every env at every step to get it
// (4 (N+1) G^2 bytes per env step: 757 + 181 us at 1024 envs x 4 UAVs x 256^2, a third of a learned-policy step).  Now the
// kernels that WRITE maps (K3, K4, K5) keep the 121 area sums of each map up to date (ippm_tiles.h), and K6 is a
// 121-cell assembly per (env, agent):
//   q      = area sums / (gx gy)                                   (map planes 3, 5 / 8, 9)
//   F      = analytic: the footprint-indicator planes are unions of rectangles.  Along x the set of rectangles covering
//            a row changes only at rectangle edges (slabs); in 1/11-cell units every overlap length is an integer, so
//            R(F) = 0.5 + sum_slabs Wx(slab, bx) (Uown - Uother)(slab, by) / (2 gx gy) exactly
//   fp     = R(footprint_img): the measurement bits of the agent's own tile (<= 2r x 2r one-bit cells), integer counts of
//            the two measurement values per bin
// ippm_area_sums is the streaming form (16-byte loads, 2 rows in flight) for maps that were written from outside.
#include <algorithm>

#include "ippm_tiles.h"

#define FEAT2 (IPPM_FEAT * IPPM_FEAT)
#define MAX_EDGES (2 * IPPM_MAX_AGENTS + 2)

// 11-scaled overlap of the cell interval [a, b) with bin `bin` of an axis of n cells: an integer
__device__ __forceinline__ int overlap11(int a, int b, int bin, int n) {
  return max(0, min(11 * b, (bin + 1) * n) - max(11 * a, bin * n));
}

// ------------------------------------------------------------------------------------------------------
// full recomputation of area sums: area[m] = sum nr nc f(map[m]) with f = sigmoid (log-odds maps) or identity
// ------------------------------------------------------------------------------------------------------
template <int VEC, bool SIGMOID>
__global__ void __launch_bounds__(256)
k_area_sums(const float* __restrict__ maps, double* __restrict__ area, int gx, int gy, int chunk_rows, int maps_per_env,
            int slots_per_env, int slot0, int tl) {
  const int m = blockIdx.y;
  const int r0 = blockIdx.x * chunk_rows, r1 = min(gx, r0 + chunk_rows);
  if (r0 >= r1) return;
  const float* map = maps + (size_t)m * gx * gy;
  double* out = area + (size_t)((m / maps_per_env) * slots_per_env + slot0 + m % maps_per_env) * FEAT2;
  __shared__ double s_area[(IPPM_FEAT + 1) * IPPM_AREA_LD];
  area_lds_clear(s_area);
  const float inv_gx = __builtin_amdgcn_rcpf((float)gx), inv_gy = __builtin_amdgcn_rcpf((float)gy);
  __syncthreads();
  const RowGeom g = make_geom<VEC>(0, gy);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane >> g.shift, gl = lane & (g.lpr - 1);
  const int stride = 4 * g.rpw;
  constexpr int UNR = 2;
  for (int gi = gl; gi < g.groups; gi += g.lpr) {
    const int y = gi * VEC;
    const AreaCols<VEC> ac = area_cols<VEC>(y, gy, inv_gy);
    AreaAcc acc;
    acc.init();
    for (int row = r0 + wv * g.rpw + sub; row < r1; row += stride * UNR) {
      CellVec<VEC> v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int rr = row + u * stride;
        if (rr < r1) v[u] = tl ? load_cells<VEC>(map + ippm_cell_index(rr, y, gy, 1))   // tile storage (whole tiles: no row tail)
                               : load_cells_row<VEC>(map + (size_t)rr * gy, y, gy);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int rr = row + u * stride;
        if (rr >= r1) continue;
        float d[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) d[q] = y + q < gy ? (SIGMOID ? ippm_sigmoid(v[u].v[q]) : v[u].v[q]) : 0.f;   // (a row's last group may hang over)
        area_row<VEC>(acc, s_area, ac, rr, gx, inv_gx, d);
      }
    }
    acc.flush(s_area, ac.cb);
  }
  __syncthreads();
  area_lds_commit(s_area, out);
}

// ------------------------------------------------------------------------------------------------------
// shared pieces of the two feature kernels (workgroups of K6_THREADS threads)
// ------------------------------------------------------------------------------------------------------
#define K6_THREADS 128
#define BIN_WORDS 4  // a column bin spans at most ceil(gy/11) + 1 cells: 128-cell masks cover grids up to 1397 cells wide

struct RectList {  // in LDS
  int r[IPPM_MAX_AGENTS][4];   // [yu,yd,xl,xr]
  int kind[IPPM_MAX_AGENTS];   // 0 = not in the plane, 1 = "own" (value 1), 2 = "other"
  int edges[MAX_EDGES];        // sorted x-edges of the participating rectangles
  int n_edges;
  int U[MAX_EDGES][IPPM_FEAT]; // per slab and column bin: 11-scaled signed column measure
};

// bits [lo, hi) of a 128-bit mask held in BIN_WORDS 32-bit words
__device__ __forceinline__ void mask_range(uint32_t* m, int lo, int hi) {
#pragma unroll
  for (int w = 0; w < BIN_WORDS; ++w) {
    const int a = max(lo - 32 * w, 0), b = min(hi - 32 * w, 32);
    m[w] = b > a ? ((b - a == 32 ? 0xFFFFFFFFu : ((1u << (b - a)) - 1u)) << a) : 0u;
  }
}

// R(F) of a footprint-indicator plane: F = 1 on "own", `other_val` (0 or 1) on other \ own, 0.5 elsewhere; into out[121].
// Along x the set of rectangles covering a row changes only at rectangle edges (slabs); per (slab, column bin) the covered
// cells of the bin are a bit mask (interval arithmetic, no per-cell loop), weighted 11 per cell except the bin's two
// fractional border cells.  Everything is integer until the final scale.
__device__ void indicator_plane(RectList& L, int n, int gx, int gy, float other_val, float* out) {
  const int tid = threadIdx.x;
  // x-edges of the participating rectangles, rank-sorted in parallel (ties by index)
  if (tid == 0) {
    int m = 0;
    for (int j = 0; j < n; ++j) m += L.kind[j] ? 2 : 0;
    L.n_edges = m;
  }
  __syncthreads();
  if (tid < 2 * n) {
    const int j = tid >> 1;
    if (L.kind[j]) {
      const int v = L.r[j][2 + (tid & 1)];
      int rank = 0;
      for (int k = 0; k < 2 * n; ++k) {
        if (!L.kind[k >> 1]) continue;
        const int w = L.r[k >> 1][2 + (k & 1)];
        rank += (w < v || (w == v && k < tid)) ? 1 : 0;
      }
      L.edges[rank] = v;
    }
  }
  __syncthreads();
  const int n_slabs = max(L.n_edges - 1, 0);
  const int sign_other = other_val > 0.5f ? 1 : -1;
  for (int item = tid; item < n_slabs * IPPM_FEAT; item += blockDim.x) {
    const int k = item / IPPM_FEAT, by = item % IPPM_FEAT;
    const int xa = L.edges[k], xb = L.edges[k + 1];
    int u = 0;
    if (xb > xa) {
      const int c0 = (by * gy) / 11, c1 = min(gy, ((by + 1) * gy + 10) / 11);  // cells of the bin: bit i = cell c0 + i
      uint32_t own[BIN_WORDS], oth[BIN_WORDS], t[BIN_WORDS];
#pragma unroll
      for (int w = 0; w < BIN_WORDS; ++w) { own[w] = 0; oth[w] = 0; }
      for (int j = 0; j < n; ++j) {
        if (!L.kind[j] || !(L.r[j][2] <= xa && xb <= L.r[j][3])) continue;
        mask_range(t, max(L.r[j][0], c0) - c0, min(L.r[j][1], c1) - c0);
#pragma unroll
        for (int w = 0; w < BIN_WORDS; ++w) { if (L.kind[j] == 1) own[w] |= t[w]; else oth[w] |= t[w]; }
      }
      const int last = c1 - 1 - c0;
      const int w_first = overlap11(c0, c0 + 1, by, gy), w_last = overlap11(c1 - 1, c1, by, gy);
      int cnt_own = 0, cnt_oth = 0;
#pragma unroll
      for (int w = 0; w < BIN_WORDS; ++w) { oth[w] &= ~own[w]; cnt_own += __popc(own[w]); cnt_oth += __popc(oth[w]); }
      auto bit = [&](const uint32_t* m, int i) { return (int)((m[i >> 5] >> (i & 31)) & 1u); };
      // every covered cell weighs 11, except the bin's first and last cell (fractional overlap with the bin)
      int so = 11 * cnt_own - (11 - w_first) * bit(own, 0), sn = 11 * cnt_oth - (11 - w_first) * bit(oth, 0);
      if (last > 0) { so -= (11 - w_last) * bit(own, last); sn -= (11 - w_last) * bit(oth, last); }
      u = so + sign_other * sn;
    }
    L.U[k][by] = u;
  }
  __syncthreads();
  const double norm = 0.5 / ((double)gx * (double)gy);
  for (int o = tid; o < FEAT2; o += blockDim.x) {
    const int bx = o / IPPM_FEAT, by = o % IPPM_FEAT;
    long long acc = 0;
    for (int k = 0; k < n_slabs; ++k) acc += (long long)overlap11(L.edges[k], L.edges[k + 1], bx, gx) * L.U[k][by];
    out[o] = (float)(0.5 + norm * (double)acc);
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------
// actor observation [11,11,7] (actor/transformations.py:14-176): one workgroup per (env, agent).  Every global load (the
// agent's code tile, its map's area sums, rectangles, positions, comm row) is issued up front: one memory round trip.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(K6_THREADS)
k_actor_features(const ippm_config* __restrict__ c, const double* __restrict__ area, const uint8_t* __restrict__ code,
                 const int32_t* __restrict__ rect, const int32_t* __restrict__ pos, const uint8_t* __restrict__ comm, int t,
                 float* __restrict__ obs, const int32_t* __restrict__ n_active) {
  const int n = c->n_agents;
  const int e = blockIdx.x / n, i = blockIdx.x % n;
  const int na = n_active ? min(max(n_active[e], 0), n) : n;   // agents flying in this env (ippm_set_team_sizes)
  if (i >= na) {   // not flying: an all-zero observation
    float* out0 = obs + (size_t)(e * n + i) * FEAT2 * IPPM_ACTOR_PLANES;
    for (int q = threadIdx.x; q < FEAT2 * IPPM_ACTOR_PLANES; q += blockDim.x) out0[q] = 0.f;
    return;
  }
  const int gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  extern __shared__ uint32_t s_tile[];  // the agent's code tile
  __shared__ RectList L;
  __shared__ int s_recv[IPPM_MAX_AGENTS];
  __shared__ int s_idx[IPPM_MAX_AGENTS][3];
  __shared__ float s_F[FEAT2];
  __shared__ int s_c1[FEAT2], s_call[FEAT2];  // footprint image: 11-scaled weight of the "occupied" cells / of all pasted cells
  const int tid = threadIdx.x;
  const int vec = gy >= 4 * IPPM_FEAT ? 4 : 1;   // as ippm_ctx::vec
  const int tile_bytes = (int)ippm_tile_bytes(S, vec);
  // up-front loads
  const uint32_t* cd32 = reinterpret_cast<const uint32_t*>(code + (size_t)(e * n + i) * tile_bytes);
  for (int q = tid; q < tile_bytes / 4; q += blockDim.x) s_tile[q] = cd32[q];
  const double* am = area + (size_t)(e * (n + 1) + i) * FEAT2;
  const double a_mine = tid < FEAT2 ? am[tid] : 0.0;
  if (tid < n) {
    const int j = tid;
    for (int q = 0; q < 4; ++q) L.r[j][q] = rect[(size_t)(e * n + j) * 4 + q];
    const int rcv = j < na ? comm[(size_t)(e * n + i) * n + j] : 0;
    s_recv[j] = rcv;
    L.kind[j] = j == i ? 1 : (rcv ? 2 : 0);
    const int32_t* pj = pos + (size_t)(e * n + j) * 3;
    ippm_pos_to_index(c, pj[0], pj[1], pj[2], s_idx[j][0], s_idx[j][1], s_idx[j][2]);
  }
  for (int o = tid; o < FEAT2; o += blockDim.x) { s_c1[o] = 0; s_call[o] = 0; }
  __syncthreads();
  // plane 6: 1 where the own measurement lies, 0 where a received other's does (own wins), 0.5 elsewhere (:62-83)
  indicator_plane(L, n, gx, gy, 0.f, s_F);
  // plane 4 input: R(footprint_img): the unclipped-size (2r x 2r) image, 0.5 with the measurement pasted at its
  // border-aware offset (mappings.py:41-43,72-76; utils/utils.py:79-98)
  const int32_t* pi = pos + (size_t)(e * n + i) * 3;
  const int k = ippm_alt_index(c, pi[2]);
  int cl[4], fu[4];
  ippm_footprint_rect(c, pi[0], pi[1], pi[2], cl, fu);
  const int hx = cl[3] - cl[2], wy = cl[1] - cl[0];
  const int full_x = fu[3] - fu[2], full_y = fu[1] - fu[0];
  const int xoff = (cl[2] > fu[2]) ? full_x - hx : 0;
  const int yoff = (cl[0] > fu[0]) ? full_y - wy : 0;
  const uint8_t* cd = reinterpret_cast<const uint8_t*>(s_tile);
  const int ycode0 = cl[0] - (cl[0] & ~3);  // tile column of the first footprint cell
  const int tile_cols = wy + ycode0;        // tile columns in use
  const int n_groups = vec == 4 ? (tile_cols + 3) >> 2 : tile_cols;
  const int per = vec == 4 ? 4 : 1;
  for (int item = tid; item < IPPM_FEAT * n_groups; item += blockDim.x) {
    const int a = item / n_groups, gcol = item % n_groups;
    // image rows overlapping row bin a, restricted to the pasted rows [xoff, xoff + hx)
    const int u0 = max((a * full_x) / 11, xoff), u1 = min(min(full_x, ((a + 1) * full_x + 10) / 11), xoff + hx);
    int n1[4] = {0, 0, 0, 0}, nall = 0;
    for (int ui = u0; ui < u1; ++ui) {
      const int nr = overlap11(ui, ui + 1, a, full_x);
      const uint32_t bits = vec == 4 ? cd[(ui - xoff) * (S >> 2) + gcol] : cd[(ui - xoff) * S + gcol];
      nall += nr;
#pragma unroll
      for (int q = 0; q < 4; ++q) n1[q] += ((bits >> q) & 1u) ? nr : 0;
    }
    if (nall == 0) continue;
    for (int q = 0; q < per; ++q) {
      const int w = gcol * per + q - ycode0;  // footprint column of this tile cell
      if (w < 0 || w >= wy) continue;
      const int wi = w + yoff;                // image column
      const int b0 = (11 * wi) / full_y;
      const int ncA = min(11, (b0 + 1) * full_y - 11 * wi), ncB = 11 - ncA;
      atomicAdd(&s_c1[a * IPPM_FEAT + b0], n1[q] * ncA);
      atomicAdd(&s_call[a * IPPM_FEAT + b0], nall * ncA);
      if (ncB) {  // (b0 + 1 <= 10 whenever the cell reaches into it)
        atomicAdd(&s_c1[a * IPPM_FEAT + b0 + 1], n1[q] * ncB);
        atomicAdd(&s_call[a * IPPM_FEAT + b0 + 1], nall * ncB);
      }
    }
  }
  __syncthreads();
  // assemble
  const float lo = c->clip_lo, hi = c->clip_hi;
  const double mv0 = (double)c->meas_value[k][0] - 0.5, mv1 = (double)c->meas_value[k][1] - 0.5;
  const double inv_img = 1.0 / ((double)full_x * (double)full_y), inv_map = 1.0 / ((double)gx * (double)gy);
  const int Z = c->space_z;
  const int ox = s_idx[i][0], oy = s_idx[i][1], oz = s_idx[i][2];
  float* out = obs + (size_t)(e * n + i) * FEAT2 * IPPM_ACTOR_PLANES;
  if (tid < FEAT2) {
    const int o = tid;
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
    const float q = (float)(a_mine * inv_map);
    const float fp = (float)(0.5 + (mv1 * (double)s_c1[o] + mv0 * (double)(s_call[o] - s_c1[o])) * inv_img);
    float* dst = out + (size_t)o * IPPM_ACTOR_PLANES;
    dst[0] = (float)(c->budget - t) / (float)c->budget;
    dst[1] = (float)(i + 1) / (float)na;
    dst[2] = pm;
    dst[3] = ippm_weight(q) * ippm_entropy(q, lo, hi);
    dst[4] = ippm_weight(fp) * ippm_entropy(fp, lo, hi);
    dst[5] = ippm_clipf(q, lo, hi);
    dst[6] = s_F[o];
  }
}

// ------------------------------------------------------------------------------------------------------
// critic state [11,11,12] (critic/transformations.py:17-132): one workgroup per env builds the shared
// global planes once and writes them for every agent
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(K6_THREADS)
k_critic_features(const ippm_config* __restrict__ c, const double* __restrict__ area, const int32_t* __restrict__ rect,
                  const int32_t* __restrict__ pos_pre, const int32_t* __restrict__ action, const float* __restrict__ obs,
                  float* __restrict__ state, const int32_t* __restrict__ n_active) {
  const int n = c->n_agents;
  const int e = blockIdx.x;
  const int na = n_active ? min(max(n_active[e], 0), n) : n;   // agents flying in this env: the planes below know no others
  const int gx = c->grid_x, gy = c->grid_y;
  __shared__ RectList L;
  __shared__ int s_idx[IPPM_MAX_AGENTS][3];
  __shared__ int s_act[IPPM_MAX_AGENTS];
  __shared__ float s_F[FEAT2];
  __shared__ float s_q[FEAT2];
  const int tid = threadIdx.x;
  const double* am = area + (size_t)(e * (n + 1) + n) * FEAT2;
  if (tid < FEAT2) s_q[tid] = (float)(am[tid] * (1.0 / ((double)gx * (double)gy)));
  if (tid < n) {
    const int j = tid;
    for (int q = 0; q < 4; ++q) L.r[j][q] = rect[(size_t)(e * n + j) * 4 + q];
    L.kind[j] = j < na ? 2 : 0;
    const int32_t* pj = pos_pre + (size_t)(e * n + j) * 3;
    ippm_pos_to_index(c, pj[0], pj[1], pj[2], s_idx[j][0], s_idx[j][1], s_idx[j][2]);
    s_act[j] = action[e * n + j];
  }
  __syncthreads();
  indicator_plane(L, n, gx, gy, 1.f, s_F);  // plane 10: 1 where any agent's measurement lies, else 0.5 (:91-108)
  const float lo = c->clip_lo, hi = c->clip_hi;
  const int Z = c->space_z, A = c->n_actions;
  for (int w = tid; w < na * FEAT2; w += blockDim.x) {
    const int i = w / FEAT2, o = w % FEAT2;
    const int a = o / IPPM_FEAT, b = o % IPPM_FEAT;
    float pm = 0.f, am_ = 0.f;
    for (int j = 0; j < na; ++j) {
      if (s_idx[j][0] == a && s_idx[j][1] == b) {
        pm = (float)(s_idx[j][2] + 1) / (float)Z;
        if (j != i) am_ = (float)(s_act[j] + 1) / (float)A;
      }
    }
    // "other actions": later agents overwrite earlier ones on a shared cell, the own agent never writes
    // (handled above: am_ only changes for j != i, in ascending j)
    const float q = s_q[o];
    const float* src = obs + ((size_t)(e * n + i) * FEAT2 + o) * IPPM_ACTOR_PLANES;
    float* dst = state + ((size_t)(e * n + i) * FEAT2 + o) * IPPM_CRITIC_PLANES;
#pragma unroll
    for (int p = 0; p < IPPM_ACTOR_PLANES; ++p) dst[p] = src[p];
    dst[7] = pm;
    dst[8] = ippm_weight(q) * ippm_entropy(q, lo, hi);
    dst[9] = ippm_clipf(q, lo, hi);
    dst[10] = s_F[o];
    dst[11] = am_;
  }
  // agents that do not fly: an all-zero state, like their observation (k_actor_features) -- the rows travel through the target
  // critic with everyone else's in COMATrainer.td_targets and must not hold whatever the allocation held
  for (int w = na * FEAT2 * IPPM_CRITIC_PLANES + tid; w < n * FEAT2 * IPPM_CRITIC_PLANES; w += blockDim.x)
    state[(size_t)e * n * FEAT2 * IPPM_CRITIC_PLANES + w] = 0.f;
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

// ------------------------------------------------------------------------------------------------------
// calculate_w_entropy on explicit probability arrays (utils/state.py:53-121): per element
//   grid = clip(p, 1e-4, 0.9999) (get_shannon_entropy clips its argument in place), se = H(grid),
//   weightings = w(target) with target = p itself ("reward"/"actor"/"global") or the given ground truth ("eval"),
//   w_entropy = weightings * se
// ------------------------------------------------------------------------------------------------------
__global__ void k_entropy_maps(const ippm_config* __restrict__ c, const float* __restrict__ p, const float* __restrict__ target,
                               float* __restrict__ w_entropy, float* __restrict__ weightings, float* __restrict__ se,
                               float* __restrict__ grid, size_t n) {
  const float lo = c->clip_lo, hi = c->clip_hi;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = p[i];
    const float w = ippm_weight(target ? target[i] : v);
    const float h = ippm_entropy(v, lo, hi);
    if (w_entropy) w_entropy[i] = w * h;
    if (weightings) weightings[i] = w;
    if (se) se[i] = h;
    if (grid) grid[i] = ippm_clipf(v, lo, hi);
  }
}

// ======================================================================================================
// host API
// ======================================================================================================
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }

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

static int launch_area_sums(const float* maps, double* area, int rows, int cols, int n_maps, int maps_per_env, int slots_per_env,
                            int slot0, bool sigmoid, hipStream_t st, int tl = 0) {
  const int chunk_rows = 32;
  dim3 grid((rows + chunk_rows - 1) / chunk_rows, n_maps), block(256);
  const bool v4 = cols >= 4 * IPPM_FEAT;   // 16-byte groups at any row alignment (a row's last group is read cell by cell)
#define IPPM_AS(V, SG) \
  hipLaunchKernelGGL((k_area_sums<V, SG>), grid, block, 0, st, maps, area, rows, cols, chunk_rows, maps_per_env, slots_per_env, slot0, v4 ? tl : 0)
  if (v4) { if (sigmoid) IPPM_AS(4, true); else IPPM_AS(4, false); }
  else { if (sigmoid) IPPM_AS(1, true); else IPPM_AS(1, false); }
#undef IPPM_AS
  IPPM_LAUNCH_CHECK("area_sums");
  return 0;
}

extern "C" int ippm_area_sums(ippm_ctx* ctx, const float* maps, double* area, int32_t n_maps, int32_t maps_per_env, int32_t slot0,
                              void* stream) {
  if (!ctx || !maps || !area) { ippm_set_error("ippm_area_sums: null argument"); return -1; }
  const ippm_config& c = ctx->cfg;
  if (c.grid_x < IPPM_FEAT || c.grid_y < IPPM_FEAT) { ippm_set_error("ippm_area_sums: grid smaller than 11x11"); return -2; }
  if (maps_per_env < 1 || slot0 < 0 || slot0 + maps_per_env > c.n_agents + 1 || n_maps % maps_per_env != 0) {
    ippm_set_error("ippm_area_sums: maps_per_env / slot0 do not fit the [E, N+1, 121] layout");
    return -1;
  }
  if (n_maps <= 0) return 0;
  // zero the addressed slots (contiguous only when a whole env's slots are rebuilt; do it per env otherwise)
  const int per = c.n_agents + 1, envs = n_maps / maps_per_env;
  if (maps_per_env == per) {
    IPPM_HIP(hipMemsetAsync(area, 0, sizeof(double) * FEAT2 * (size_t)n_maps, S_(stream)));
  } else {
    IPPM_HIP(hipMemset2DAsync(area + (size_t)slot0 * FEAT2, sizeof(double) * FEAT2 * per, 0, sizeof(double) * FEAT2 * maps_per_env, envs,
                              S_(stream)));
  }
  return launch_area_sums(maps, area, c.grid_x, c.grid_y, n_maps, maps_per_env, per, slot0, true, S_(stream), ctx->tl);
}

__global__ void k_area_scale(const double* __restrict__ src, float* __restrict__ dst, double scale, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)(src[i] * scale);
}

extern "C" int ippm_area_resize(ippm_ctx* ctx, const float* src, int32_t rows, int32_t cols, float* dst, double* scratch,
                                int32_t n_arrays, void* stream) {
  if (!ctx || !src || !dst || !scratch) { ippm_set_error("ippm_area_resize: null argument"); return -1; }
  if (rows < IPPM_FEAT || cols < IPPM_FEAT) { ippm_set_error("ippm_area_resize: source smaller than 11x11"); return -2; }
  if (n_arrays <= 0) return 0;
  IPPM_HIP(hipMemsetAsync(scratch, 0, sizeof(double) * FEAT2 * (size_t)n_arrays, S_(stream)));
  if (int rc = launch_area_sums(src, scratch, rows, cols, n_arrays, 1, 1, 0, false, S_(stream))) return rc;
  hipLaunchKernelGGL(k_area_scale, dim3(grid1((size_t)n_arrays * FEAT2)), dim3(256), 0, S_(stream), scratch, dst,
                     1.0 / ((double)rows * (double)cols), n_arrays * FEAT2);
  IPPM_LAUNCH_CHECK("area_scale");
  return 0;
}

extern "C" int ippm_actor_features(ippm_ctx* ctx, const double* area, const uint8_t* code, const int32_t* rect,
                                   const int32_t* pos, const uint8_t* comm, int32_t t, float* obs, int32_t n_envs,
                                   void* stream) {
  if (!ctx || !area || !code || !rect || !pos || !comm || !obs) { ippm_set_error("ippm_actor_features: null argument"); return -1; }
  if (int rc = feature_checks(ctx, "ippm_actor_features")) return rc;
  if (n_envs <= 0) return 0;
  const ippm_config& c = ctx->cfg;
  const size_t tile = ippm_tile_bytes(c.tile_stride, ctx->vec);
  IPPM_LAUNCH_SH(ctx, IPPM_T_ACTOR_FEAT, k_actor_features, dim3(n_envs * c.n_agents), dim3(K6_THREADS), (tile + 3) / 4 * 4, S_(stream), ctx->dcfg, area,
                     code, rect, pos, comm, t, obs, ctx->n_active);
  IPPM_LAUNCH_CHECK("actor_features");
  return 0;
}

extern "C" int ippm_critic_features(ippm_ctx* ctx, const double* area, const int32_t* rect, const int32_t* pos_pre,
                                    const int32_t* action, const float* obs, float* state, int32_t n_envs, void* stream) {
  if (!ctx || !area || !rect || !pos_pre || !action || !obs || !state) { ippm_set_error("ippm_critic_features: null argument"); return -1; }
  if (int rc = feature_checks(ctx, "ippm_critic_features")) return rc;
  if (n_envs <= 0) return 0;
  IPPM_LAUNCH(ctx, IPPM_T_CRITIC_FEAT, k_critic_features, dim3(n_envs), dim3(K6_THREADS), S_(stream), ctx->dcfg, area, rect, pos_pre, action, obs, state, ctx->n_active);
  IPPM_LAUNCH_CHECK("critic_features");
  return 0;
}

extern "C" int ippm_reward_from_maps(ippm_ctx* ctx, const float* before, const float* after, double* sums, float* reward,
                                     int32_t n_maps, void* stream) {
  if (!ctx || !before || !after || !sums) { ippm_set_error("ippm_reward_from_maps: null argument"); return -1; }
  if (n_maps <= 0) return 0;
  IPPM_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * n_maps, S_(stream)));
  const size_t cells = (size_t)ctx->cfg.grid_x * ctx->cfg.grid_y;
  const int gxb = (int)std::min<size_t>(32, (cells + 255) / 256);
  hipLaunchKernelGGL(k_reward_pair, dim3(gxb, n_maps), dim3(256), 0, S_(stream), ctx->dcfg, before, after, sums);
  IPPM_LAUNCH_CHECK("reward_pair");
  (void)reward;
  return 0;
}

extern "C" int ippm_entropy_maps(ippm_ctx* ctx, const float* prob, const float* target, float* w_entropy, float* weightings,
                                 float* se, float* grid, int64_t n, void* stream) {
  if (!ctx || !prob) { ippm_set_error("ippm_entropy_maps: null argument"); return -1; }
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_entropy_maps, dim3(std::min(4096, grid1((size_t)n))), dim3(256), 0, S_(stream), ctx->dcfg, prob, target,
                     w_entropy, weightings, se, grid, (size_t)n);
  IPPM_LAUNCH_CHECK("entropy_maps");
  return 0;
}
