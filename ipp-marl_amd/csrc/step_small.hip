// Small-state kernels of an env step (gfx950): comm matrix, fusion planning, K1 mask/act/move.
//
// None of this touches a map; it is a few hundred bytes per env and pure latency.  Round 1 ran it as five launches per
// step (plan(global), comm+plan(local), K1, reward finalize, each 5-18 us with 4-11 us gaps).  Here one wavefront per env
// does all of it in ONE launch (k_plan_step): lanes = agents for the comm rows and the plans, lanes = actions for the
// mask work of K1, the agent loop of K1 stays serial (agent i is masked against the already-moved j < i).
#include <algorithm>

#include "ippm_k1.h"

// ======================================================================================================
// comm matrix
// ======================================================================================================
// does agent i hear agent j in env e at step t? (communication_log.py:39-58)  pos_e = the env's [N,3] positions (global memory or
// LDS); u = the pair's uniform draw: explicit (parity mode), Philox, or moot when links never fail
__device__ __forceinline__ bool comm_pair(const ippm_config* __restrict__ c, int64_t ep, const int32_t* pos_e, double range,
                                          const double* __restrict__ draws, int t, int e, int i, int j) {
  const int n = c->n_agents;
  const int32_t* pi = pos_e + i * 3;
  const int32_t* pj = pos_e + j * 3;
  long long dx = pi[0] - pj[0], dy = pi[1] - pj[1], dz = pi[2] - pj[2];
  long long d2 = dx * dx + dy * dy + dz * dz;
  double u = 1.0;
  if (c->failure_rate > 0.0) {  // u >= 0 always passes otherwise: the draw (one per ordered pair in the reference) is moot
    if (draws) u = draws[(size_t)(e * n + i) * n + j];
    else {
      const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
      Philox4 ph = ippm_philox((uint32_t)j, (uint32_t)ep, ippm_stream_word((uint32_t)i, (uint32_t)t, IPPM_DOMAIN_COMM),
                               (uint32_t)(ep >> 32), k0, k1);
      u = (double)ph.v[0] * (1.0 / 4294967296.0);
    }
  }
  bool ok = d2 == 0;
  if (d2 > 0) {
    double dist = sqrt((double)d2);
    if (dist <= range && u >= c->failure_rate) ok = true;
  }
  return ok;
}

// row i of the comm matrix of env e: bit j set <=> agent i hears agent j; also stored as bytes
__device__ __forceinline__ uint32_t comm_row(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                                             const int32_t* pos_e, const float* __restrict__ comm_range,
                                             const double* __restrict__ draws, uint8_t* __restrict__ comm, int t, int e, int i) {
  const int n = c->n_agents;
  const double range = comm_range ? (double)comm_range[e] : c->comm_range;
  const int64_t ep = episode ? episode[e] : 0;
  uint32_t row = 0;
  for (int j = 0; j < n; ++j) {
    const bool ok = comm_pair(c, ep, pos_e, range, draws, t, e, i, j);
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
  const int e = tid / n;
  comm_row(c, episode, pos + (size_t)e * n * 3, comm_range, draws, comm, t, e, tid % n);
}

// ======================================================================================================
// fusion planning (one lane per map): builds the ordered op list of K4 / K5 and maintains the
// deferred-clamp state (the reference's full-grid input clip, applied only where it can matter)
// ======================================================================================================
__device__ __forceinline__ void plan_push(const ippm_config* __restrict__ c, int32_t* w, int& nops, int type, int src, int alt,
                                          const int32_t* r, int& x0, int& x1, int& y0, int& y1, int4* s_ops) {
  if (r[3] <= r[2] || r[1] <= r[0]) return;
  if (s_ops) s_ops[nops] = make_int4(r[0], r[1], r[2], r[3]);  // LDS mirror of the rectangle for the tile builder
  int32_t* op = w + WS_OPS + nops * OP_WORDS;
  op[OP_TYPE] = type; op[OP_SRC] = src;
  // the measurement's log-odds ride in the op record: the fusion kernel needs no dependent table lookup
  op[OP_LM0] = type ? __float_as_int(c->logit_meas[alt][0]) : 0;
  op[OP_LM1] = type ? __float_as_int(c->logit_meas[alt][1]) : 0;
  op[OP_YU] = r[0]; op[OP_YD] = r[1]; op[OP_XL] = r[2]; op[OP_XR] = r[3];
  x0 = min(x0, r[2]); x1 = max(x1, r[3]); y0 = min(y0, r[0]); y1 = max(y1, r[1]);
  ++nops;
}

// plans map i of env e (i == n: the global map); recv = agents whose measurements map i receives this step.
// Returns the number of rows of the plan's hull (0: nothing to fuse).
// rect_e = the env's [N,4] published footprints, st = the map's first 6 workspace words (deferred-clamp state), both possibly
// prefetched by the caller (global memory or LDS / registers).
__device__ __forceinline__ int plan_map(const ippm_config* __restrict__ c, const int32_t* rect_e, const int32_t* pos_e, uint32_t recv,
                                         int32_t* __restrict__ ws, int global_maps, int e, int i, const int32_t* st,
                                         int4* s_ops = nullptr, int32_t* s_nops = nullptr, int n_act = -1) {
  const int n_all = c->n_agents;
  const int n = n_act >= 0 ? n_act : n_all;   // agents flying in this env (messages come from them only)
  int32_t* w = ws + (size_t)(e * (n_all + 1) + i) * IPPM_WS_WORDS;
  int nops = 0, x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
  int last_src = -1;
  for (int j = 0; j < n; ++j) {
    bool take = global_maps ? true : (j != i && ((recv >> j) & 1u) != 0);
    if (take) last_src = j;
  }
  int32_t* hdr = w + WS_PLAN;
  const int whole = c->logit_prior != 0.f;
  if (last_src < 0) {  // nothing received: the map is untouched; carry possible out-of-range regions forward
    if (!global_maps && st[WS_FLAG_S]) {
      const int32_t* ri = rect_e + i * 4;
      if (st[WS_FLAG_A]) {
        w[WS_RECT_A + 0] = min(st[WS_RECT_A + 0], ri[0]); w[WS_RECT_A + 1] = max(st[WS_RECT_A + 1], ri[1]);
        w[WS_RECT_A + 2] = min(st[WS_RECT_A + 2], ri[2]); w[WS_RECT_A + 3] = max(st[WS_RECT_A + 3], ri[3]);
      } else {
        for (int q = 0; q < 4; ++q) w[WS_RECT_A + q] = ri[q];
      }
      w[WS_FLAG_A] = 1;
      w[WS_FLAG_S] = 0;
    }
    hdr[PL_NOPS] = 0;
    if (s_nops) *s_nops = 0;
    return 0;
  }
  if (st[WS_FLAG_A]) plan_push(c, w, nops, 0, -1, 0, st + WS_RECT_A, x0, x1, y0, y1, s_ops);
  if (!global_maps && st[WS_FLAG_S]) plan_push(c, w, nops, 0, -1, 0, rect_e + i * 4, x0, x1, y0, y1, s_ops);
  int last_op = -1;
  for (int j = 0; j < n; ++j) {
    bool take = global_maps ? true : (j != i && ((recv >> j) & 1u) != 0);
    if (!take) continue;
    const int32_t* rj = rect_e + j * 4;
    int before = nops;
    plan_push(c, w, nops, 1, j, ippm_alt_index(c, pos_e[j * 3 + 2]), rj, x0, x1, y0, y1, s_ops);
    if (j == last_src) {
      last_op = nops > before ? nops - 1 : -1;  // an empty last footprint leaves no unclamped outputs
      for (int q = 0; q < 4; ++q) w[WS_RECT_A + q] = rj[q];
    }
  }
  w[WS_FLAG_A] = 0;  // set again by the fusion kernel if the last op leaves out-of-range values
  w[WS_FLAG_S] = 0;
  hdr[PL_NOPS] = nops;
  if (s_nops) *s_nops = nops;
  if (whole && nops > 0) { x0 = 0; x1 = c->grid_x; y0 = 0; y1 = c->grid_y; }
  hdr[PL_X0] = x0; hdr[PL_X1] = x1; hdr[PL_Y0] = y0; hdr[PL_Y1] = y1;
  hdr[PL_LAST] = last_op;
  return nops > 0 ? x1 - x0 : 0;
}

// plan_map for k_plan_step: the op record of a measurement (type, source, log-odds of its altitude, footprint) is the same for
// every map that receives it, so lane j has prepared it once (s_rec: two int4 per agent, s_rect4: the footprints) and a map's lane
// copies 32 bytes per received agent instead of re-deriving altitude index and table values op by op (the planning wavefront's
// serial loop was 4 of the kernel's 12 us).  Same results as plan_map, field by field.
__device__ __forceinline__ int plan_map_fast(const ippm_config* __restrict__ c, const int4* s_rect4, const int4* s_rec, uint32_t recv,
                                              int32_t* __restrict__ ws, int global_maps, int e, int i, const int32_t* st,
                                              int4* s_ops, int32_t* s_nops, int n_act) {
  const int n = n_act;    // agents flying in this env (messages come from them only); strides use the configured team size
  int32_t* w = ws + (size_t)(e * (c->n_agents + 1) + i) * IPPM_WS_WORDS;
  const uint32_t all = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
  const uint32_t takes = global_maps ? all : (recv & ~(1u << i) & all);
  int32_t* hdr = w + WS_PLAN;
  if (takes == 0) {  // nothing received: the map is untouched; carry possible out-of-range regions forward
    if (!global_maps && st[WS_FLAG_S]) {
      const int4 ri = s_rect4[i];
      if (st[WS_FLAG_A]) {
        w[WS_RECT_A + 0] = min(st[WS_RECT_A + 0], ri.x); w[WS_RECT_A + 1] = max(st[WS_RECT_A + 1], ri.y);
        w[WS_RECT_A + 2] = min(st[WS_RECT_A + 2], ri.z); w[WS_RECT_A + 3] = max(st[WS_RECT_A + 3], ri.w);
      } else {
        w[WS_RECT_A + 0] = ri.x; w[WS_RECT_A + 1] = ri.y; w[WS_RECT_A + 2] = ri.z; w[WS_RECT_A + 3] = ri.w;
      }
      w[WS_FLAG_A] = 1;
      w[WS_FLAG_S] = 0;
    }
    hdr[PL_NOPS] = 0;
    *s_nops = 0;
    return 0;
  }
  const int last_src = 31 - __clz(takes);
  int nops = 0, x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
  int4* ops = reinterpret_cast<int4*>(w + WS_OPS);
  auto clamp_op = [&](int4 r) {   // an op that only clips (type 0)
    if (r.w <= r.z || r.y <= r.x) return;
    s_ops[nops] = r;
    ops[nops * 2] = make_int4(0, -1, 0, r.x);
    ops[nops * 2 + 1] = make_int4(r.y, r.z, r.w, 0);
    x0 = min(x0, r.z); x1 = max(x1, r.w); y0 = min(y0, r.x); y1 = max(y1, r.y);
    ++nops;
  };
  if (st[WS_FLAG_A]) clamp_op(make_int4(st[WS_RECT_A], st[WS_RECT_A + 1], st[WS_RECT_A + 2], st[WS_RECT_A + 3]));
  if (!global_maps && st[WS_FLAG_S]) clamp_op(s_rect4[i]);
  int last_op = -1;
  for (uint32_t rem = takes; rem != 0; rem &= rem - 1u) {
    const int j = __ffs(rem) - 1;
    const int4 r = s_rect4[j];
    const bool some = r.w > r.z && r.y > r.x;
    if (some) {
      s_ops[nops] = r;
      ops[nops * 2] = s_rec[j * 2];
      ops[nops * 2 + 1] = s_rec[j * 2 + 1];
      x0 = min(x0, r.z); x1 = max(x1, r.w); y0 = min(y0, r.x); y1 = max(y1, r.y);
      ++nops;
    }
    if (j == last_src) {
      last_op = some ? nops - 1 : -1;  // an empty last footprint leaves no unclamped outputs
      w[WS_RECT_A + 0] = r.x; w[WS_RECT_A + 1] = r.y; w[WS_RECT_A + 2] = r.z; w[WS_RECT_A + 3] = r.w;
    }
  }
  w[WS_FLAG_A] = 0;  // set again by the fusion kernel if the last op leaves out-of-range values
  w[WS_FLAG_S] = 0;
  hdr[PL_NOPS] = nops;
  *s_nops = nops;
  if (c->logit_prior != 0.f && nops > 0) { x0 = 0; x1 = c->grid_x; y0 = 0; y1 = c->grid_y; }
  hdr[PL_X0] = x0; hdr[PL_X1] = x1; hdr[PL_Y0] = y0; hdr[PL_Y1] = y1;
  hdr[PL_LAST] = last_op;
  return nops > 0 ? x1 - x0 : 0;
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
  int32_t st[6];
  for (int q = 0; q < 6; ++q) st[q] = ws[(size_t)(e * (n + 1) + i) * IPPM_WS_WORDS + q];
  plan_map(c, rect + (size_t)e * n * 4, pos + (size_t)e * n * 3, recv, ws, global_maps, e, i, st);
}

// ======================================================================================================
// One-trip tile items of the fusion (fuse_tiles.hip), built here so that the fusion's wavefronts start from a self-contained
// item instead of deriving rows and columns from the plan: trip 1 = the item, trip 2 = op records + every map cell of the item.
//
// The ops of a plan are rectangles.  Along x the set of ops covering a row changes only at rectangle edges (SLABS); inside a
// slab the covered columns are the union of the active ops' column ranges, i.e. a few disjoint INTERVALS of 4-cell groups.
// An item is a RUN of one (slab, interval)'s lane-loads in row-major order: at most 64 * slots lane-loads of 16 bytes (slots = 4 or 2
// loads in flight per lane, by the number of ops that meet the interval), with the mask of exactly those ops.  Every cell of the
// union belongs to exactly one item; gaps between side-by-side rectangles belong to none.
// Wave-cooperative per map: lane l ranks edge l (rank sort + one ds_permute, as the row walker did per item), lane s then owns
// slab s and merges the intervals of the ops sorted by their first column; finished intervals go out as the walk proceeds.
// ======================================================================================================
__device__ __forceinline__ int tb_lane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// The items of one REGION = rows [xa, xb) x groups [g0, g1) of a (slab, interval), met by the ops of `mask` (all arguments
// wave-uniform; every lane takes part).  The region's lane-loads in row-major order, t = (row - xa) * W + (group - g0), are cut
// into runs of 64 * slots: item i is the run [i * cap, i * cap + count) -- it starts wherever in a row the previous one ended, so
// every item but a region's last is full whatever the width (round 4 cut regions into blocks of WHOLE rows, floor(cap / W) of
// them: a 150-group interval filled 150 of 256 lane-loads, and at config 5's shape the list as a whole 77 %).  One LDS atomic
// reserves the region's slots in the env's list, then lane l writes items l, l + 64, ...
#ifndef IPPM_TILE_COOP_ITEMS   // regions of more items than this are written by the whole builder wavefront
#define IPPM_TILE_COOP_ITEMS 6
#endif
__device__ __forceinline__ int tile_region_items(int rows, int W, unsigned mask, int& sh) {
  sh = ippm_tile_slots(__popc(mask)) == 4 ? 8 : 7;   // cap = 64 * slots = 1 << sh
  return (rows * W + (1 << sh) - 1) >> sh;
}
// items first, first + stride, ... of the region's n, into list slots k0 + i
__device__ __forceinline__ void tile_write_items(int4* items, int env_cap, int k0, int first, int stride, int n, int sh, int slot, int xa, int xb,
                                                 int g0, int g1, unsigned mask) {
  const int W = g1 - g0, total = (xb - xa) * W;
  const float inv_w = __builtin_amdgcn_rcpf((float)W);
  for (int i = first; i < n; i += stride) {
    const int t0 = i << sh;
    int r = (int)((float)t0 * inv_w);                     // floor(t0 / W) to within one (t0 < 2^24), then exact
    int gs = t0 - r * W;
    if (gs < 0) { --r; gs += W; }
    else if (gs >= W) { ++r; gs -= W; }
    const int cnt = min(1 << sh, total - t0);
    if (k0 + i < env_cap) items[k0 + i] = make_int4(gs | (cnt << 16), xa + r, g0 | (W << 16), (int)(mask | ((unsigned)slot << 24)));
  }
}
// a LARGE region (all arguments wave-uniform; every lane takes part): lane l writes items l, l + 64, ...
__device__ __forceinline__ void tile_emit_region(int4* items, int env_cap, int32_t* s_items, int slot, int xa, int xb, int g0, int g1,
                                                 unsigned mask, int lane) {
  int sh;
  const int n = tile_region_items(xb - xa, g1 - g0, mask, sh);
  int k0 = 0;
  if (lane == 0) k0 = atomicAdd(s_items, n);
  k0 = __builtin_amdgcn_readfirstlane(k0);
  tile_write_items(items, env_cap, k0, lane, 64, n, sh, slot, xa, xb, g0, g1, mask);
}

// All items of one map's plan (nops > 0, uniform) into the env's list.  s_ops: the plan's rectangles in LDS.
// Lane l ranks edge l among the plan's 2 nops row edges; lane s then owns slab s and walks the plan's ops in ascending first
// column (a second rank sort), merging their group ranges into intervals.  Whenever lanes have finished an interval the wave
// stops walking and emits those regions one after the other, ALL lanes writing each region's items (tile_emit_region): with
// every lane writing its own slab's items one by one the wave ran as long as its longest slab at every step of the walk -- at
// config 5's shape (17 plans of up to 17 rectangles on 1024^2, thousands of items per map) that was most of the plan kernel's
// 144 us.  The order of an env's items in its list is whatever the atomics make it; no result depends on it.
// round_mask / G: column intervals are rounded OUTWARDS to multiples of round_mask + 1 groups (7: whole 128-byte lines; 0: not), capped at the
// G groups of a row, before they are merged -- so that a row segment of an item starts and ends on line boundaries and every line
// it touches is written whole.  The cells this adds are met by no op: the fusion clips them (as the reference clips EVERY cell of a map
// at every fusion, mappings.py:110-111) and writes them back; intervals whose rounded ranges touch are merged, so no cell is in two items.
// round_mask < 0: TILE STORAGE of the maps (ippm_internal.h).  The same walk in units of tiles: a rectangle's rows become the rows of tiles it meets
// ([xl >> 2, (xr + 3) >> 2)), its columns the lane-loads of those tiles in a row of tiles (8 per tile: [8 (yu >> 3), 8 ((yd + 7) >> 3)), G = 8 tiles per
// row = grid_y of them).  Slabs are then rows of tiles with one set of ops, intervals runs of whole tiles, an item a run of whole 128-byte lines; the
// fusion's lanes test their own row and cells against each op of the mask (an edge tile holds cells of the slab's ops and cells of none).
__device__ __forceinline__ void tile_build_map(const int4* s_ops, int nops, int env, int slot, int4* items, int env_cap, int32_t* s_items,
                                               int lane, int round_mask, int G) {
  const int n_edges = 2 * nops;
  const bool tl = round_mask < 0;
  int4 rc = make_int4(0, 0, 0, 0);
  if (lane < nops) rc = s_ops[lane];
  if (tl) { rc.z >>= 2; rc.w = (rc.w + 3) >> 2; round_mask = 0; }
  const int r_yu = rc.x, r_yd = rc.y, r_xl = rc.z, r_xr = rc.w;
  // lane l holds edge l = xl / xr of op l >> 1
  int edge = 0;
  {
    const int o = min(lane >> 1, IPPM_MAX_OPS - 1);
    const int exl = __builtin_amdgcn_ds_bpermute(o << 2, r_xl), exr = __builtin_amdgcn_ds_bpermute(o << 2, r_xr);
    edge = (lane & 1) ? exr : exl;
  }
  int rank = 0, orank = 0;
  for (int j = 0; j < n_edges; ++j) {
    const int ej = tb_lane_i(edge, j);
    rank += (ej < edge || (ej == edge && j < lane)) ? 1 : 0;
  }
  for (int j = 0; j < nops; ++j) {
    const int yj = tb_lane_i(r_yu, j);
    orank += (yj < r_yu || (yj == r_yu && j < lane)) ? 1 : 0;
  }
  const int sorted = __builtin_amdgcn_ds_permute((lane < n_edges ? rank : lane) << 2, edge);  // lane `rank` receives my edge
  const int order = __builtin_amdgcn_ds_permute((lane < nops ? orank : lane) << 2, lane);      // lane r receives the op of rank r
  const int xa = sorted;
  int xb = __builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, sorted);
  const bool slab_on = lane + 1 < n_edges && xb > xa;
  if (!slab_on) xb = xa;
  int g0 = 0, g1 = -1;
  unsigned mask = 0;
  for (int r = 0; r <= nops; ++r) {      // r == nops: the intervals still open
    bool in = false, done = mask != 0;
    int lo = 0, hi = 0, o = 0;
    if (r < nops) {
      o = tb_lane_i(order, r);  // op with the r-th smallest first column
      const int yu = tb_lane_i(r_yu, o), yd = tb_lane_i(r_yd, o), xl = tb_lane_i(r_xl, o), xr = tb_lane_i(r_xr, o);
      in = slab_on && xl <= xa && xa < xr;
      if (tl) { lo = (yu >> 3) << 3; hi = ((yd + 7) >> 3) << 3; }
      else { lo = (yu >> 2) & ~round_mask; hi = min(G, (((yd + 3) >> 2) + round_mask) & ~round_mask); }
      done = in && mask != 0 && lo > g1;  // a gap of at least one group: the interval so far is complete
    }
    // finished intervals go out now.  A small region (config 2: three items on average) is written by its own lane, all lanes
    // at once; a large one (config 5's shape: 75 items) by the whole wave, one region after the other
    int sh = 0;
    const int n_mine = done ? tile_region_items(xb - xa, g1 - g0, mask, sh) : 0;
    const bool big = n_mine > IPPM_TILE_COOP_ITEMS;
    if (done && !big) tile_write_items(items, env_cap, atomicAdd(s_items, n_mine), 0, 1, n_mine, sh, slot, xa, xb, g0, g1, mask);
    for (unsigned long long pend = __ballot(big); pend != 0; pend &= pend - 1) {
      const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)pend) - 1);
      tile_emit_region(items, env_cap, s_items, slot, tb_lane_i(xa, src), tb_lane_i(xb, src), tb_lane_i(g0, src), tb_lane_i(g1, src),
                       (unsigned)tb_lane_i((int)mask, src), lane);
    }
    if (done) mask = 0;
    if (in) {
      if (mask == 0) { g0 = lo; g1 = hi; }
      else g1 = max(g1, hi);
      mask |= 1u << o;
    }
  }
}

// ======================================================================================================
// K1: action mask + collision mask + action choice + move
// ======================================================================================================
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
    m = collide(A, m, collision_bits(A, jx - ix, jy - iy, jz - iz));
  }
  for (int q = 0; q < A; ++q) mask_out[(size_t)b * A + q] = (m >> q) & 1u;
}

// ======================================================================================================
// k_plan_step: everything small of an env step in one launch, one wavefront per env
//   IPPM_STEP_COMM   comm matrix + local-fusion plans (lanes = agents)
//   IPPM_STEP_GLOBAL global-fusion plan (lane n)
//   IPPM_STEP_MOVE   K1 on the same positions, then the footprints of the NEW positions into rect_next, so that K3
//                    starts with its rectangle in hand instead of a pos -> lattice index -> centre table chain
// comm and the plans read the pre-move positions (LDS copy taken before K1 writes anything).
// ======================================================================================================
#ifdef IPPM_PLAN_STAMPS   // variant builds: env 0 leaves wall-clock stamps of its phases in word 7 of the counter slots
#define PLAN_STAMP(k) do { if (blockIdx.x == 0 && lane == 0 && stamps) stamps[((wv * 8 + (k)) & 63) * 8 + 7] = wall_clock64(); \
    if (blockIdx.x == gridDim.x - 1 && wv == 0 && lane == 0 && stamps) stamps[(48 + (k)) * 8 + 7] = wall_clock64(); \
    if (blockIdx.x == gridDim.x / 2 && wv == 0 && lane == 0 && stamps) stamps[(56 + (k)) * 8 + 7] = wall_clock64(); } while (0)
#else
#define PLAN_STAMP(k) do { } while (0)
#endif
#define IPPM_PLAN_BUILDERS 14  // most wavefronts that build tile items next to wavefront 0 (maps are dealt out round-robin); + K1 = 16 wavefronts
#define IPPM_PLAN_BUILDERS_DEFAULT 3   // measured at 1024 envs x 4 UAVs: 1 / 2 / 3 / 5 builders -> 32.6 / 25.8 / 21.1 / 26.5 us (16 wavefronts
                                       // per CU is what one round of the launch holds; a sixth wavefront per env makes it two rounds)
__global__ void __launch_bounds__(64 * (1 + IPPM_PLAN_BUILDERS))
k_plan_step(int32_t* __restrict__ pos, const int32_t* __restrict__ rect, int32_t* __restrict__ ws, int n, int flags, int t, int policy,
            const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
            const float* __restrict__ comm_range, const double* __restrict__ draws, uint8_t* __restrict__ comm,
            const float* __restrict__ probs,
            const int32_t* __restrict__ action_in, uint8_t* __restrict__ mask, int32_t* __restrict__ action,
            int32_t* __restrict__ fault, int32_t* __restrict__ rect_next, int agent_sel, int32_t* __restrict__ work,
            int wave_rows, int env_cap, unsigned long long* __restrict__ stamps, const int32_t* __restrict__ n_active,
            int32_t* __restrict__ slabs, int n_slabs, int tile_round) {
  // (argument order = latency order: what the first loads need -- positions, footprints, the maps' clamp state, the team size --
  // arrives in SGPRs with the wavefront, so the loads go out before anything else has been read)
  // Wavefront 0 plans (comm matrix, fusion plans, written-cells boxes).  With a tile-form work list the workgroup carries more
  // wavefronts: builders, which cut the plans into one-trip items (a map each, round-robin) as soon as the plans are in LDS --
  // wavefront 0 joins them once it has planned -- and, when the launch also moves the agents, wavefront 1 for K1, which needs
  // nothing but the positions and runs beside the planning (in line K1 was 7 of the 13 us of wavefront 0's chain; one wavefront
  // per SIMD executes ~2.5 ns per instruction, so anything serial here is expensive and anything that can go to a neighbour
  // wavefront is almost free -- up to 4 wavefronts per env: a round of the launch holds 16 per CU).
  // Hand-overs: ONE workgroup barrier, after the loads; the builders then wait for `s_ready` (an LDS flag the planning wavefront
  // releases), K1 waits for nobody.  K1's LDS traffic stays inside its wavefront (wave_sync_lds).
  const int e = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int A = c->n_actions;
  // na: the agents that fly in this env (ippm_set_team_sizes; n = the configured team size = the arrays' stride, by default all fly).
  // The others are heard by nobody and hear nobody, get no plan, do not move and get no sense record: K3 finds an empty footprint.
  const int na = n_active ? min(max(n_active[e], 0), n) : n;
  __shared__ int32_t s_pos[IPPM_MAX_AGENTS * 3];    // pre-move positions: what comm and the plans see
  __shared__ int32_t s_pos1[IPPM_MAX_AGENTS * 3];   // K1's working copy (moved in place)
  __shared__ int4 s_rect4[IPPM_MAX_AGENTS];
  __shared__ int4 s_rec[IPPM_MAX_AGENTS * 2];        // per agent: the op record of its measurement (plan_map_fast)
  int32_t* s_rect = reinterpret_cast<int32_t*>(s_rect4);
  __shared__ int4 s_ops[(IPPM_MAX_AGENTS + 1) * IPPM_MAX_OPS];   // the op rectangles of every plan, for the tile builders
  __shared__ int32_t s_nops[IPPM_MAX_AGENTS + 1];
  __shared__ uint32_t s_recv[IPPM_MAX_AGENTS];     // row i of the comm matrix as a bit mask
  __shared__ int32_t s_items, s_done, s_ready;   // items handed out so far; builders that have finished; plans are in LDS
  const bool plans = (flags & (IPPM_STEP_COMM | IPPM_STEP_GLOBAL)) != 0;
  const bool tiled = (flags & IPPM_STEP_TILES) != 0 && work && plans;
  const bool move = (flags & IPPM_STEP_MOVE) != 0;
  const int waves = (int)(blockDim.x >> 6);
  const int k1_wave = move ? (tiled && waves >= 3 ? 1 : 0) : -1;   // K1 beside the planning only when builders exist besides it
  int32_t* pg = pos + (size_t)e * n * 3;
  int32_t st[6] = {0, 0, 0, 0, 0, 0};
  PLAN_STAMP(0);
  if (wv == 0) {
    // everything the workgroup will read is requested now, in one round trip: positions, published footprints, the
    // deferred-clamp state of my map (lane i: local map i, lane n: the global map)
    // (unconditional loads from clamped addresses: behind lane predicates each load was waited for before the next went out)
    const int32_t vpos = pg[min(lane, n * 3 - 1)];
    int32_t vrect = 0;
    if (plans) {
      vrect = rect[(size_t)e * n * 4 + min(lane, n * 4 - 1)];
      const int4* wsm = reinterpret_cast<const int4*>(ws + (size_t)(e * (n + 1) + min(lane, n)) * IPPM_WS_WORDS);
      const int4 a = wsm[0];
      const int2 b = *reinterpret_cast<const int2*>(wsm + 1);
      st[0] = a.x; st[1] = a.y; st[2] = a.z; st[3] = a.w; st[4] = b.x; st[5] = b.y;
    }
    if (lane <= n) s_nops[lane] = 0;
    if (lane < n) s_recv[lane] = 0;
    if (lane == 0) { s_items = 0; s_done = 0; s_ready = 0; }
    if (lane < n * 3) { s_pos[lane] = vpos; s_pos1[lane] = vpos; }
    if (plans && lane < n * 4) s_rect[lane] = vrect;
  }
  if (waves > 1) __syncthreads(); else wave_sync_lds();
  PLAN_STAMP(1);
  if (wv == 0 && plans) {
    int hull_rows = 0;
    if (tiled && lane < n) {   // lane j: the op record of agent j's measurement, once for all the maps that will take it
      const int k = ippm_alt_index(c, s_pos[lane * 3 + 2]);
      const int4 r = s_rect4[lane];
      s_rec[lane * 2] = make_int4(1, lane, __float_as_int(c->logit_meas[k][0]), r.x);
      s_rec[lane * 2 + 1] = make_int4(r.y, r.z, r.w, __float_as_int(c->logit_meas[k][1]));
    }
    uint32_t recv = 0;
    if (flags & IPPM_STEP_COMM) {
      // the comm matrix with lanes = ordered pairs (a lane per row walked its n pairs one after the other: a float64 square root
      // and a Philox call each); the rows' receive masks are gathered in LDS
      const double range = comm_range ? (double)comm_range[e] : c->comm_range;
      const int64_t ep = episode ? episode[e] : 0;
      const float inv_n = __builtin_amdgcn_rcpf((float)n);
      for (int p0 = 0; p0 < n * n; p0 += 64) {
        const int p = p0 + lane;
        if (p < n * n) {
          const int i = (int)(((float)p + 0.5f) * inv_n), j = p - i * n;
          const bool ok = i < na && j < na && comm_pair(c, ep, s_pos, range, draws, t, e, i, j);
          comm[(size_t)e * n * n + p] = ok ? 1 : 0;
          if (ok) atomicOr(&s_recv[i], 1u << j);
        }
      }
      wave_sync_lds();
      if (lane < n) recv = s_recv[lane];
    }
    PLAN_STAMP(7);
    if (tiled) {
      wave_sync_lds();
      if ((flags & IPPM_STEP_COMM) && lane < na && (agent_sel < 0 || agent_sel == lane))
        hull_rows = plan_map_fast(c, s_rect4, s_rec, recv, ws, 0, e, lane, st, s_ops + lane * IPPM_MAX_OPS, s_nops + lane, na);
      if ((flags & IPPM_STEP_GLOBAL) && lane == n)
        hull_rows = plan_map_fast(c, s_rect4, s_rec, 0u, ws, 1, e, n, st, s_ops + n * IPPM_MAX_OPS, s_nops + n, na);
    } else {
      if ((flags & IPPM_STEP_COMM) && lane < na && (agent_sel < 0 || agent_sel == lane))
        hull_rows = plan_map(c, s_rect, s_pos, recv, ws, 0, e, lane, st, nullptr, nullptr, na);
      if ((flags & IPPM_STEP_GLOBAL) && lane == n)
        hull_rows = plan_map(c, s_rect, s_pos, 0u, ws, 1, e, n, st, nullptr, nullptr, na);
    }
    // an agent that does not fly (any more: team sizes may change between steps) gets an EMPTY plan, so that a fusion which
    // enumerates every map's plan from ws (no work list) never re-applies the plan it was left with when it last flew
    if ((flags & IPPM_STEP_COMM) && lane >= na && lane < n && agent_sel < 0)
      ws[(size_t)(e * (n + 1) + lane) * IPPM_WS_WORDS + WS_PLAN + PL_NOPS] = 0;
    // the map's fused-cells box takes in this step's plan hull (ippm_reset_maps fills only the boxes at the next reset)
    if (lane <= n && hull_rows > 0) {
      int32_t* wm = ws + (size_t)(e * (n + 1) + lane) * IPPM_WS_WORDS;
      const int32_t* hdr = wm + WS_PLAN;
      box_union(wm, hdr[PL_X0], hdr[PL_X1], hdr[PL_Y0], hdr[PL_Y1]);
      if (slabs) {
        // ... and, finer, the dirty slabs: per 16-row slab the column interval of the ops that meet it (tile form: the op rectangles are
        // in LDS; row form: the plan's hull, which is what that fusion walks)
        int32_t* sl = slabs + (size_t)(e * (n + 1) + lane) * 2 * n_slabs;
        if (tiled) {
          const int nops = s_nops[lane];
          for (int k = 0; k < nops; ++k) {
            const int4 r = s_ops[lane * IPPM_MAX_OPS + k];      // {yu, yd, xl, xr}
            slab_mark(sl, n_slabs, r.z, r.w, r.x, r.y);
          }
        } else {
          slab_mark(sl, n_slabs, hdr[PL_X0], hdr[PL_X1], hdr[PL_Y0], hdr[PL_Y1]);
        }
      }
    }
    // row-run form of the work list: items of this env's plans into the env's own slice (exclusive scan of the lanes' counts)
    if (work && !tiled) {
      const int items = (hull_rows + wave_rows - 1) / wave_rows;
      // the global map's runs go first (they carry the reward arithmetic: longest items first balances the env's wavefronts)
      const int g_items = __builtin_amdgcn_readlane(items, n);
      int before = lane == n ? 0 : g_items, total = g_items;
      for (int j = 0; j < n; ++j) {
        const int v = __builtin_amdgcn_readlane(items, j);
        before += (j < lane && lane != n) ? v : 0;
        total += v;
      }
      if (lane == 0) work[e] = total;
      int32_t* dst = work + gridDim.x + (size_t)e * env_cap + before;
      if (lane <= n)
        for (int k = 0; k < items; ++k) dst[k] = ((e * (n + 1) + lane) << 8) | k;
    }
    if (tiled && lane == 0) __hip_atomic_store(&s_ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // the plans are in LDS
  }
  PLAN_STAMP(2);
  if (tiled && (wv != k1_wave || k1_wave == 0)) {   // (fewer than 3 wavefronts: the planning wavefront builds, then does K1)
    // Tile items: builder b takes maps b, b + nb, ...; an interval's items go wherever the env's running count says (an LDS
    // atomic), so no builder waits for another; the last one to finish writes the env's count.
    const int nb = waves - (k1_wave > 0 ? 1 : 0), b = (k1_wave > 0 && wv > k1_wave) ? wv - 1 : wv;
    if (wv != 0)
      while (__hip_atomic_load(&s_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    int4* items = reinterpret_cast<int4*>(work + ((gridDim.x + 3) & ~3)) + (size_t)e * env_cap;
    PLAN_STAMP(3);
    // Builder 0 is the planning wavefront: it starts last, and it takes the global map, whose plan holds every footprint (about
    // as many items as two or three local plans).  The local maps go to the other builders in turn, every third round of them
    // to builder 0 (4 UAVs, 3 builders: global | locals 0, 2 | locals 1, 3 instead of global + local 2 on the late one: -1 us).
    for (int mm = 0; mm <= n; ++mm) {
      const int k = mm - 1;
      const int owner = (mm == 0 || nb == 1) ? 0 : (((k / (nb - 1)) % 3 == 2) ? 0 : 1 + k % (nb - 1));
      if (owner != b) continue;
      const int m = mm == 0 ? n : mm - 1;   // the global map's items come early: they carry the reward arithmetic
      const int nops = __builtin_amdgcn_readfirstlane(s_nops[m]);
      if (nops > 0) tile_build_map(s_ops + m * IPPM_MAX_OPS, nops, e, m, items, env_cap, &s_items, lane, tile_round, (c->grid_y + 3) >> 2);
    }
    PLAN_STAMP(4);
    int last = 0;
    if (lane == 0) last = atomicAdd(&s_done, 1) == nb - 1 ? 1 : 0;
    last = __builtin_amdgcn_readfirstlane(last);
    if (last && lane == 0) {
      // (the capacity bound of ippm_tile_env_cap covers every plan; a list that would not fit is cut and reported)
      const int total = atomicAdd(&s_items, 0);
      work[e] = min(total, env_cap) | IPPM_WORK_TILED | (total > env_cap ? IPPM_WORK_OVERFLOW : 0);
      // the fusion skips an overflowed list (and counts it); the env says so too, stickily: its maps are no longer fused
      if (total > env_cap && fault) atomicOr(fault + e, IPPM_FAULT_WORK_OVERFLOW);
    }
  }
  if (wv != k1_wave) return;
  // ---- K1 on its own positions; the footprints of the NEW positions for K3, and into the map's sensed-cells box
  wave_sync_lds();
  PLAN_STAMP(5);
  k1_env(c, episode ? episode[e] : 0, s_pos1, probs ? probs + (size_t)e * n * A : nullptr,
         action_in ? action_in + (size_t)e * n : nullptr, policy, t, mask + (size_t)e * n * A, action + (size_t)e * n,
         fault ? fault + e : nullptr, na);
  if (lane < n * 3) pg[lane] = s_pos1[lane];
  if (rect_next && lane >= na && lane < n) {   // not flying: an empty sense record (K3's workgroups for it leave at once)
    int4* r = reinterpret_cast<int4*>(rect_next + (size_t)(e * n + lane) * IPPM_SENSE_REC_WORDS);
    r[0] = make_int4(0, 0, 0, 0);
    r[1] = make_int4(0, 0, 0, 0);
  }
  if (rect_next && lane < na) {
    // the agent's sense record: K3 starts from these 32 bytes alone (footprint + the measurement constants of the new altitude)
    int cl[4];
    ippm_footprint_rect(c, s_pos1[lane * 3], s_pos1[lane * 3 + 1], s_pos1[lane * 3 + 2], cl, nullptr);
    const int k = ippm_alt_index(c, s_pos1[lane * 3 + 2]);
    const float lp = c->logit_prior;
    int4* r = reinterpret_cast<int4*>(rect_next + (size_t)(e * n + lane) * IPPM_SENSE_REC_WORDS);
    r[0] = make_int4(cl[0], cl[1], cl[2], cl[3]);
    r[1] = make_int4(__float_as_int(c->logit_meas[k][0] - lp), __float_as_int(c->logit_meas[k][1] - lp), (int)c->flip_threshold[k], 0);
    if (ws) box_union(ws + (size_t)(e * (n + 1) + lane) * IPPM_WS_WORDS, cl[2], cl[3], cl[0], cl[1], WS_SBOX_X, WS_SBOX_Y);   // what K3 senses next
    if (slabs) slab_mark(slabs + (size_t)(e * (n + 1) + lane) * 2 * n_slabs, n_slabs, cl[2], cl[3], cl[0], cl[1]);
  }
  PLAN_STAMP(6);
}

// ======================================================================================================
// host API
// ======================================================================================================
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }

int ippm_launch_plan(ippm_ctx* ctx, const int32_t* rect, const int32_t* pos, const uint8_t* comm, int32_t* ws, int global_maps,
                     int n_envs, int agent_sel, hipStream_t st) {
  const int maps = (global_maps || agent_sel >= 0) ? n_envs : n_envs * ctx->cfg.n_agents;
  if (maps <= 0) return 0;
  hipLaunchKernelGGL(k_plan, dim3(grid1(maps, 64)), dim3(64), 0, st, ctx->dcfg, rect, pos, comm, ws, global_maps, n_envs, agent_sel);
  IPPM_LAUNCH_CHECK("plan");
  return 0;
}

extern "C" int ippm_comm_matrix(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const float* comm_range,
                                const double* draws, uint8_t* comm, int32_t t, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !comm) { ippm_set_error("ippm_comm_matrix: null argument"); return -1; }
  if (!draws && !episode) { ippm_set_error("ippm_comm_matrix: Philox draws need the episode ids"); return -1; }
  if (n_envs <= 0) return 0;
  hipLaunchKernelGGL(k_comm, dim3(grid1((size_t)n_envs * ctx->cfg.n_agents)), dim3(256), 0, S_(stream), ctx->dcfg, episode,
                     pos, comm_range, draws, comm, t, n_envs);
  IPPM_LAUNCH_CHECK("comm");
  return 0;
}

int ippm_work_env_cap(const ippm_ctx* ctx, int n_envs) {
  const int wave_rows = ippm_fuse_wave_rows(ctx, n_envs);
  const int chunks = (ctx->cfg.grid_x + wave_rows - 1) / wave_rows;
  return (ctx->cfg.n_agents + 1) * chunks;
}

// Tile items an env's slice can hold.  Per map: every full item covers more than half of its capacity of 64 * slots groups,
// capacity >= that of the config's largest plan, and the items are disjoint, so there are at most
// gx * G / (32 * slots(max_ops)) of them (G = groups per row); each (slab, interval) pair adds at most one partial row block
// per column chunk: <= max_ops * (2 max_ops - 1) * ceil(G / 64).
int ippm_tile_env_cap(const ippm_ctx* ctx) {
  const ippm_config& c = ctx->cfg;
  const int max_ops = c.n_agents + 1;
  const int G = (c.grid_y + 3) / 4;
  const int full = (c.grid_x * G + 32 * ippm_tile_slots(max_ops) - 1) / (32 * ippm_tile_slots(max_ops));
  const int partial = max_ops * (2 * max_ops - 1) * ((G + 63) / 64);
  return (c.n_agents + 1) * (full + partial);
}

extern "C" int ippm_work_words(ippm_ctx* ctx, int32_t n_envs, int64_t* words) {
  if (!ctx || !words || n_envs < 0) { ippm_set_error("ippm_work_words: bad argument"); return -1; }
  const int64_t rows_form = (int64_t)n_envs * (1 + ippm_work_env_cap(ctx, n_envs));
  const int64_t tile_form = ctx->tiles ? (int64_t)((n_envs + 3) & ~3) + (int64_t)n_envs * ippm_tile_env_cap(ctx) * 4 : 0;
  *words = std::max(rows_form, tile_form);   // either form of the list fits (ippm_plan_step's IPPM_STEP_TILES chooses)
  return 0;
}

extern "C" int ippm_plan_step(ippm_ctx* ctx, const int64_t* episode, int32_t* pos, const float* comm_range, const double* draws,
                              uint8_t* comm, const int32_t* rect, int32_t* ws, int32_t t, int32_t flags, const float* probs,
                              const int32_t* action_in, int32_t policy, uint8_t* mask, int32_t* action, int32_t* fault,
                              int32_t* rect_next, int32_t* work, int32_t n_envs, void* stream) {
  if (!ctx || !pos) { ippm_set_error("ippm_plan_step: null argument"); return -1; }
  if ((flags & (IPPM_STEP_COMM | IPPM_STEP_GLOBAL | IPPM_STEP_MOVE)) == 0 || (flags & ~15)) { ippm_set_error("ippm_plan_step: bad flags"); return -1; }
  // The form of the work list follows the CONTEXT, not the caller: ippm_fuse_step hands a list to the tile fusion exactly when
  // the context has the tile form (and no measurement knob routes it to the row walker), so that is the form built here --
  // IPPM_STEP_TILES is implied there and ignored elsewhere.  (Until round 5 the flag decided, and a caller who followed the
  // header -- plan without the flag, fuse with area sums -- got a list the tile fusion skipped: no fusion, rc 0.)
  const bool tile_ctx = ctx->tiles && !ctx->knob_nowork && !ctx->knob_split;
  if (tile_ctx && work && (flags & (IPPM_STEP_COMM | IPPM_STEP_GLOBAL))) flags |= IPPM_STEP_TILES;
  else flags &= ~IPPM_STEP_TILES;
  if ((flags & IPPM_STEP_COMM) && (!comm || !rect || !ws)) { ippm_set_error("ippm_plan_step: comm/plan needs comm, rect, ws"); return -1; }
  if ((flags & IPPM_STEP_COMM) && !draws && !episode) { ippm_set_error("ippm_plan_step: Philox draws need the episode ids"); return -1; }
  if ((flags & IPPM_STEP_GLOBAL) && (!rect || !ws)) { ippm_set_error("ippm_plan_step: global plan needs rect, ws"); return -1; }
  if (flags & IPPM_STEP_MOVE) {
    if (!mask || !action) { ippm_set_error("ippm_plan_step: move needs mask, action"); return -1; }
    if (policy == 0 && !action_in) { ippm_set_error("ippm_plan_step: policy 0 needs action_in"); return -1; }
    if ((policy == 2 || policy == 3) && !probs) { ippm_set_error("ippm_plan_step: policy 2/3 needs probs"); return -1; }
    if ((policy == 1 || policy == 2) && !episode) { ippm_set_error("ippm_plan_step: sampling needs episode ids"); return -1; }
    if (policy < 0 || policy > 3) { ippm_set_error("ippm_plan_step: unknown policy"); return -1; }
  }
  if (n_envs <= 0) return 0;
  const bool plans = (flags & (IPPM_STEP_COMM | IPPM_STEP_GLOBAL)) != 0;
  if (work && plans) {
    if ((flags & (IPPM_STEP_COMM | IPPM_STEP_GLOBAL)) != (IPPM_STEP_COMM | IPPM_STEP_GLOBAL)) {
      ippm_set_error("ippm_plan_step: the work list is built for local and global plans together");
      return -1;
    }
  }
  const bool tile_list = plans && work && (flags & IPPM_STEP_TILES);
  // builders: 3 at large batches (a round of the launch holds 16 wavefronts per CU: a sixth wavefront per env at 1024 envs makes it
  // two rounds); small batches of large teams (config 5's shape: 64 envs x 16 UAVs, 17 plans of up to 17 ops, thousands of items
  // per map) take a builder per map as long as the whole launch stays within one round of the chip
  int builders = ctx->cfg.n_agents >= 7 ? 5 : IPPM_PLAN_BUILDERS_DEFAULT;   // (8 UAVs x 1024 envs: 3 / 5 / 7 / 9 builders -> 59.6 / 51.3 / 50.7 / 53.2 us)
  while (builders < std::min(ctx->cfg.n_agents + 1, IPPM_PLAN_BUILDERS) && (long long)n_envs * (builders + 3) <= 4096) ++builders;
  if (ctx->knob_plan_builders > 0) builders = std::min(ctx->knob_plan_builders, IPPM_PLAN_BUILDERS);
  const int plan_waves = tile_list ? 1 + std::min(ctx->cfg.n_agents + 1, builders) : 1;
  IPPM_LAUNCH(ctx, IPPM_T_PLAN, k_plan_step, dim3(n_envs), dim3(64 * plan_waves), S_(stream), pos, rect, ws, ctx->cfg.n_agents, flags, t, policy,
                     ctx->dcfg, episode, comm_range, draws, comm, probs, action_in, mask, action, fault, rect_next, -1, plans ? work : nullptr,
                     ippm_fuse_wave_rows(ctx, n_envs), (flags & IPPM_STEP_TILES) ? ippm_tile_env_cap(ctx) : ippm_work_env_cap(ctx, n_envs),
                     ctx->dcounters, ctx->n_active, ctx->slabs, ippm_slab_count(ctx),
                     ctx->tl ? -1 : ((ctx->cfg.grid_y % 32 == 0 && ctx->knob_tile_round > 0) ? ippm_round_cells(ctx->knob_tile_round) / 4 - 1 : 0));
  IPPM_LAUNCH_CHECK("plan_step");
  return 0;
}

extern "C" int ippm_action_mask(ippm_ctx* ctx, const int32_t* pos, const int32_t* others, const int32_t* n_others,
                                int32_t max_others, const uint8_t* mask_in, uint8_t* mask_out, int32_t* next_pos, int32_t batch,
                                void* stream) {
  if (!ctx || !pos || !mask_out) { ippm_set_error("ippm_action_mask: null argument"); return -1; }
  if (n_others && !others) { ippm_set_error("ippm_action_mask: n_others without others"); return -1; }
  if (batch <= 0) return 0;
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
  return ippm_plan_step(ctx, episode, pos, nullptr, nullptr, nullptr, nullptr, nullptr, t, IPPM_STEP_MOVE, probs, action_in, policy,
                        mask, action, fault, nullptr, nullptr, n_envs, stream);
}
