// K4 / K5, one-trip tile form (gfx950): the fusion of ippm_fuse_step when no area sums are tracked -- the form the env-only
// step and bench.py run.
//
//   Mapping.fuse_map(..., "local" / "global")   mapping/mappings.py:80-124
//   get_global_reward terms (K5)                 utils/reward.py:68-82, utils/state.py:53-121
//
// The row walker (fuse.hip) gives every wavefront a run of rows of a plan's hull: per item a work-list entry, then the plan,
// then the rows two at a time -- about ten dependent memory round trips with two 16-byte loads in flight per lane, and the
// kernel runs at the speed of those chains (rounds x latency), not of the memory system.  Here the plan kernel
// (step_small.hip, tile_build_map) hands out self-contained ITEMS: a run of consecutive lane-loads of one (slab, column interval)
// of a plan in row-major order, at most 64 * slots of them, with the mask of the ops that meet it.  A wavefront does an item in
// three trips, the first and the last short:
//   1. the item (16 bytes, scalar; the next item is requested while this one is worked on);
//   2. the op records of the mask TOGETHER WITH every map cell of the item: the cells' addresses follow from the item alone
//      -- lane-load t = slot * 64 + lane is element start + t of the region's row-major order, (row (start + t) / W, group
//      (start + t) % W), so every item but a region's last is full whatever the interval's width;
//   3. the measurement-code bytes of every (slot, op), whose addresses need the op records (a small, cache-resident plane);
// then the ordered clamp/add chain in registers and one store per slot.  Nothing is carried from item to item except the
// wavefront's reward and counter sums, so register use is that of one item and the launch is many short independent chains.
#include <algorithm>

#include "ippm_tiles.h"

typedef unsigned ippm_t_u4 __attribute__((ext_vector_type(4)));
#define IPPM_T_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define IPPM_T_OOB 0x7FFFFFF0
#ifndef IPPM_T_LOAD_AUX    // cache policy of the map accesses (variant builds; bit 1 = non-temporal on gfx950)
#define IPPM_T_LOAD_AUX 0
#endif
#ifndef IPPM_T_STORE_AUX
#define IPPM_T_STORE_AUX 0
#endif
#define IPPM_T_FAR (-(1 << 20))   // column of a lane-load past the item's end: no op covers it
// measurement-only variants (make VARIANT=... EXTRA=-DIPPM_X_...; results are wrong on purpose): what the launch takes without the
// reward arithmetic (IPPM_X_NOREWARD), without the per-op clip-and-add (IPPM_X_NOCHAIN), without the code-byte loads (IPPM_X_NOCODE)
#ifdef IPPM_X_NOREWARD
#define IPPM_X_REWARD false
#else
#define IPPM_X_REWARD true
#endif
#ifdef IPPM_X_NOCODE
#define IPPM_X_CODE(load) 0u
#else
#define IPPM_X_CODE(load) (load)
#endif

__device__ __forceinline__ int t_lane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float t_lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

struct TileCtx {
  float* local;
  float* global;
  const uint8_t* code;
  const int32_t* plan;     // ws, read side
  int32_t* ws;
  int n, gx, gy, row_bytes, TB, n_envs;
  float lc, wt;
  int lane;
  // TRACK: the 11 x 11 area sums of the maps (ippm_tiles.h) are kept up to date: a 12 x 12 float64 tile in LDS per wavefront,
  // flushed to the item's map with one global atomic per touched bin when the item is done
  double* s_area;
  double* area;
  float inv_gx, inv_gy;
};

// The item's contribution to the area sums of its map: per lane-load the weighted sigmoid differences of its four cells into
// the (at most) 2 x 2 bins the group meets.  `old4` / `new4`: the cells as loaded and as stored (a lane-load past the item's
// end loaded zeros and stores nothing: zero difference).
// MIS (rows not a multiple of 4 wide): the cells of a row's last group that hang over into the next row were loaded and run through
// the chain but are never stored (another lane owns them): they contribute nothing here either.
template <bool MIS>
__device__ __forceinline__ void tile_area_slot(const TileCtx& w, int x, int y, const float* old4, const float* new4) {
  float d[4], sd = 0.f, cA = 0.f;
  const AreaCols<4> ac = area_cols<4>(y, w.gy, w.inv_gy);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    d[j] = (MIS && y + j >= w.gy) ? 0.f : sigmoid_diff(new4[j], old4[j]);
    sd += d[j];
    cA += ac.wA[j] * d[j];
  }
  const float cB = 11.f * sd - cA;
  const int n11 = 11 * x, rb = area_bin(n11, w.inv_gx);
  const float nA = (float)min((rb + 1) * w.gx - n11, 11), nB = 11.f - nA;
  double* p = w.s_area + rb * IPPM_AREA_LD + ac.cb;
  const float v00 = nA * cA, v01 = nA * cB, v10 = nB * cA, v11 = nB * cB;
  if (v00 != 0.f) atomicAdd(p, (double)v00);                       // ds_add_f64
  if (v01 != 0.f) atomicAdd(p + 1, (double)v01);
  if (v10 != 0.f) atomicAdd(p + IPPM_AREA_LD, (double)v10);
  if (v11 != 0.f) atomicAdd(p + IPPM_AREA_LD + 1, (double)v11);
}
// (a workgroup is one wavefront: __syncthreads() is the LDS ordering point between the atomics above and the reads here)
__device__ __forceinline__ void tile_area_flush(const TileCtx& w, int map_abs) {
  __syncthreads();
  double* dst = w.area + (size_t)map_abs * IPPM_FEAT * IPPM_FEAT;
  for (int k = w.lane; k < IPPM_FEAT * IPPM_AREA_LD; k += 64) {
    const double v = w.s_area[k];
    const int rb = k / IPPM_AREA_LD, cb = k - rb * IPPM_AREA_LD;
    if (v != 0.0) {
      if (cb < IPPM_FEAT) atomicAdd(&dst[rb * IPPM_FEAT + cb], v);
      w.s_area[k] = 0.0;
    }
  }
  __syncthreads();
}

struct TileAcc {   // per wavefront, over all its items (all of one env)
  // sum w(a) (H(b) - H(a)) and sum (w(a) - w(b)) H(b); the increment of T = sum w H is aD - a1.  float64 per lane, fed with the
  // float32 sum of a slot's four cells: where measurements are noise-free the terms are of size 1 with both signs and the sums
  // are what is left after they cancel -- float32 lane sums over a few dozen cells showed at 2e-4 in the returns there
  double a1, aD;
  unsigned cells_l, ops_l, cells_g, ops_g;
};

// One item with at most NA ops (spare slots first) and SLOTS loads in flight per lane.
// The item is a RUN of `cnt` lane-loads of its region (rows from x0 on, groups [g0, g0 + W)) in row-major order, starting at group
// `gs` of row x0: lane-load t = q * 64 + lane (t < cnt) is element gs + t of that order.
// TL (tile storage of the maps, ippm_internal.h): the item is a run of lane-loads of ROWS OF TILES -- x0 a row of tiles, g0 / W / gs lane-loads of it (8 per
// tile: lane-load G of a row of tiles is row (G >> 1) & 3 of tile G >> 3, cells 4 (G & 1) .. + 3 of that row) -- so consecutive lanes cover whole lines.  A row of
// tiles holds 4 map rows: whether an op meets a lane's ROW is then a per-lane question as well (the slabs are cut at rows of tiles).
template <int NA, int SLOTS, bool MIS, bool TRACK, bool TL>
__device__ __forceinline__ void tile_item(const TileCtx& w, TileAcc& acc, int e, int slot, int x0, int cnt, int gs, int g0, int W, unsigned active) {
  const int map_abs = e * (w.n + 1) + slot;
  const bool is_global = slot == w.n;
  const int32_t* plan = w.plan + (size_t)map_abs * IPPM_WS_WORDS;
  const int last_op = plan[WS_PLAN + PL_LAST];
  const __amdgpu_buffer_rsrc_t rmap =
      IPPM_T_RSRC(is_global ? w.global + (size_t)e * IPPM_MAP_PITCH(w.gx, w.gy) : w.local + (size_t)(e * w.n + slot) * IPPM_MAP_PITCH(w.gx, w.gy),
                  (size_t)w.gx * w.gy * 4);
  const __amdgpu_buffer_rsrc_t rcode = IPPM_T_RSRC(w.code, (size_t)w.n_envs * w.n * w.TB);
  // ---- trip 2a: the op records of the mask.  Spare slots come FIRST: an empty slot still clips, like every op of the reference
  // -- a no-op ahead of the first real op, but behind the last it would clip that op's unclamped outputs.
  // Uniform addresses, scalar loads: the fields stay in SGPRs (NA <= 6; items met by more ops: tile_item_long).
  static_assert(NA <= 6, "straight-line chains are compiled for up to six ops");
  const int pad = NA - __popc(active);
  int s_yu[NA], s_yd[NA], cs[NA], s_xl[NA], s_xh[NA];
  float s_lm0[NA], s_lm1[NA];
  int keep_slot = -1;
  {
    unsigned rem = active;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      if (k < pad) { s_yu[k] = 0; s_yd[k] = 0; cs[k] = 0; s_lm0[k] = 0.f; s_lm1[k] = 0.f; s_xl[k] = 0; s_xh[k] = 0; continue; }
      const int idx = __ffs(rem) - 1;
      rem &= rem - 1u;
      const int4 a = *reinterpret_cast<const int4*>(plan + WS_OPS + idx * OP_WORDS);       // {type, src, lm0, yu}
      const int4 b = *reinterpret_cast<const int4*>(plan + WS_OPS + idx * OP_WORDS + 4);   // {yd, xl, xr, lm1}
      const bool isf = a.x != 0;
      s_yu[k] = a.w; s_yd[k] = b.x;
      s_xl[k] = b.y; s_xh[k] = b.z - b.y;      // (TL) the op's rows [xl, xl + xh)
      s_lm0[k] = isf ? __int_as_float(a.z) : 0.f;
      s_lm1[k] = isf ? __int_as_float(b.w) : 0.f;
      // byte of group (row, g) in the source's code tile = (row * row_bytes + g) + cs; a clamp-only op reads some byte of the
      // plane and adds 0 either way
      cs[k] = isf ? (e * w.n + a.y) * w.TB - b.y * w.row_bytes - (a.w >> 2) : 0;
      keep_slot = idx == last_op ? k : keep_slot;
    }
  }
  // ---- trip 2b: every map cell of the item.  Lane-load t = q * 64 + lane -> element tt = gs + t of the region's row-major order
  // -> (row tt / W, group tt % W); tt < 512, W <= 256: floor(tt / W) = (int)((tt + 0.5) * (1 / W)) exactly (ippm_div_small).
  const float inv_w = __builtin_amdgcn_rcpf((float)W);
  CellVec<4> mv[SLOTS];
  int off[SLOTS], coff[SLOTS], ycol[SLOTS], xrow[SLOTS];
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    const int t = q * 64 + w.lane;
    const int r = ippm_div_small(gs + t, inv_w);
    const int gi = gs + t - r * W;
    const bool valid = t < cnt;
    int row, g;
    if (TL) {
      const int G = g0 + gi;                       // lane-load of the row of tiles x0 + r
      row = ((x0 + r) << 2) + ((G >> 1) & 3);
      g = ((G >> 3) << 1) + (G & 1);
      off[q] = valid ? ((x0 + r) * w.gy + G) * 16 : IPPM_T_OOB;
    } else {
      row = x0 + r; g = g0 + gi;
      off[q] = valid ? (row * w.gy + g * 4) * 4 : IPPM_T_OOB;
    }
    xrow[q] = row;
    coff[q] = row * w.row_bytes + g;
    ycol[q] = valid ? g * 4 : IPPM_T_FAR;
    const ippm_t_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rmap, off[q], 0, IPPM_T_LOAD_AUX);
    mv[q].v[0] = __uint_as_float(v.x); mv[q].v[1] = __uint_as_float(v.y); mv[q].v[2] = __uint_as_float(v.z); mv[q].v[3] = __uint_as_float(v.w);
  }
  // ---- trip 3: one measurement-code byte per (slot, op)
  uint32_t cw[SLOTS][NA];
#pragma unroll
  for (int q = 0; q < SLOTS; ++q)
#pragma unroll
    for (int k = 0; k < NA; ++k) cw[q][k] = IPPM_X_CODE(__builtin_amdgcn_raw_buffer_load_b8(rcode, coff[q] + cs[k], 0, 0));
  // ---- the ordered clamp/add chain (mappings.py:80-124 in log-odds): every op clips its input over the whole grid
  // (mappings.py:110-111), then adds the measurement's log-odds inside its footprint; the outputs of the plan's last op stay
  // unclamped (its rectangle is remembered as possibly out of range), every other cell was clipped again by a later op.
  float amax = 0.f;
  unsigned cells = 0, opcells = 0;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    float L[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) L[j] = mv[q].v[j];
    unsigned touched = 0, keepm = 0;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int yu = s_yu[k], yd = s_yd[k];
      const float lm0 = s_lm0[k], lm1 = s_lm1[k];
      // cells y .. y+3 of my group inside [yu, yd): bits [lo, hi)
      const int lo = min(max(yu - ycol[q], 0), 4), hi = min(max(yd - ycol[q], 0), 4);
      unsigned cm = ((1u << (hi - lo)) - 1u) << lo;
      if (TL) cm = (unsigned)(xrow[q] - s_xl[k]) < (unsigned)s_xh[k] ? cm : 0u;   // ... and my row inside the op's rows
      touched |= cm;
      keepm = k == keep_slot ? cm : keepm;
      opcells += (lm0 != 0.f || lm1 != 0.f) ? __popc(cm) : 0;
      const uint32_t cwk = cw[q][k];
#ifndef IPPM_X_NOCHAIN
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lm = ippm_masked(ippm_bitmask(cm, j), ippm_blend(ippm_bitmask(cwk, j), lm1, lm0));
        L[j] = ippm_clampl(L[j], w.lc) + lm;
      }
#else
      L[0] += __uint_as_float(cwk & 1u);   // (the code byte stays live)
#endif
    }
    cells += __popc(touched);
    float out[4];
    if (keep_slot >= 0) {   // (uniform) the item meets the plan's last op: its cells keep their unclamped outputs
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = ippm_blend(ippm_bitmask(keepm, j), L[j], ippm_clampl(L[j], w.lc));
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(out[0]), fabsf(out[1])), fmaxf(fabsf(out[2]), fabsf(out[3]))));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) out[j] = ippm_clampl(L[j], w.lc);   // |out| <= lc: nothing to remember
    }
    {
      ippm_t_u4 v;
      v.x = __float_as_uint(out[0]); v.y = __float_as_uint(out[1]); v.z = __float_as_uint(out[2]); v.w = __float_as_uint(out[3]);
      // a lane-load past the item's end was never in range; a group no op touches cannot occur inside an interval
      if (MIS) {
        // (compile-time) rows are not a multiple of 4 wide: the last group of a row hangs over into the next row -- its cells go out
        // one by one, another lane owns the rest
        const bool tail = ycol[q] + 4 > w.gy;
        __builtin_amdgcn_raw_buffer_store_b128(v, rmap, tail ? IPPM_T_OOB : off[q], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(v.x, rmap, tail ? off[q] : IPPM_T_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(v.y, rmap, tail && ycol[q] + 1 < w.gy ? off[q] + 4 : IPPM_T_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(v.z, rmap, tail && ycol[q] + 2 < w.gy ? off[q] + 8 : IPPM_T_OOB, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(v, rmap, off[q], 0, IPPM_T_STORE_AUX);
      }
    }
    if (TRACK) tile_area_slot<MIS>(w, xrow[q], ycol[q] == IPPM_T_FAR ? 0 : ycol[q], mv[q].v, out);
    if (IPPM_X_REWARD && is_global) {
      // information-gain terms (utils/reward.py:68-82) of the cells the step changed; an untouched cell contributes exact zeros
      // (same weight, same entropy).  Slots whose touched cells all have weight 0 before and after (believed free, still
      // believed free) skip the entropies: wave-uniform on spatially coherent terrain.
      float wa[4], wb[4], wsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t tm = ippm_bitmask(touched, j);
        wa[j] = ippm_masked(tm, ippm_weight_l(out[j], w.wt));
        wb[j] = ippm_masked(tm, ippm_weight_l(mv[q].v[j], w.wt));
        wsum += wa[j] + wb[j];
      }
      if (__any(wsum != 0.f)) {
        float s1 = 0.f, sD = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float hb = ippm_entropy_l(mv[q].v[j], w.lc), ha = ippm_entropy_l(out[j], w.lc);
          s1 += wa[j] * (hb - ha);
          sD += (wa[j] - wb[j]) * hb;
        }
        acc.a1 += (double)s1;
        acc.aD += (double)sD;
      }
    }
  }
  if (__any(amax > w.lc) && w.lane == 0) w.ws[(size_t)map_abs * IPPM_WS_WORDS + WS_FLAG_A] = 1;
  if (TRACK) tile_area_flush(w, map_abs);
  if (is_global) { acc.cells_g += cells; acc.ops_g += opcells; }
  else { acc.cells_l += cells; acc.ops_l += opcells; }
}

// An item met by MORE than six ops (4 % of the lane-loads at config 5's shape, 0.1 % at config 4's): the same run of lane-loads,
// two in flight per lane, but the chain runs one op per iteration of a run-time loop -- the next op's record (scalar) and code
// bytes are requested before the current op's clip-and-add -- with the cells in registers throughout.  The kernel therefore needs
// the registers of a six-op item whatever the team size; round 4 compiled straight-line chains of 8 / 10 / 14 / 18 ops into the
// kernels of larger teams: 94 VGPRs, five wavefronts per SIMD, for every item of config 4 because one item in a thousand meets
// seven ops.  Same arithmetic: the spare slots of a straight-line chain are clips ahead of a real op's own clip.
template <bool MIS, bool TRACK, bool TL>
__device__ __forceinline__ void tile_item_long(const TileCtx& w, TileAcc& acc, int e, int slot, int x0, int cnt, int gs, int g0, int W, unsigned active) {
  constexpr int SLOTS = 2;
  const int map_abs = e * (w.n + 1) + slot;
  const bool is_global = slot == w.n;
  const int32_t* plan = w.plan + (size_t)map_abs * IPPM_WS_WORDS;
  const int last_op = plan[WS_PLAN + PL_LAST];
  const __amdgpu_buffer_rsrc_t rmap =
      IPPM_T_RSRC(is_global ? w.global + (size_t)e * IPPM_MAP_PITCH(w.gx, w.gy) : w.local + (size_t)(e * w.n + slot) * IPPM_MAP_PITCH(w.gx, w.gy),
                  (size_t)w.gx * w.gy * 4);
  const __amdgpu_buffer_rsrc_t rcode = IPPM_T_RSRC(w.code, (size_t)w.n_envs * w.n * w.TB);
  const float inv_w = __builtin_amdgcn_rcpf((float)W);
  CellVec<4> mv[SLOTS];
  float L[SLOTS][4];
  int off[SLOTS], coff[SLOTS], ycol[SLOTS], xrow[SLOTS];
  unsigned touched[SLOTS], keepm[SLOTS];
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    const int t = q * 64 + w.lane;
    const int r = ippm_div_small(gs + t, inv_w);
    const int gi = gs + t - r * W;
    const bool valid = t < cnt;
    int row, g;
    if (TL) {
      const int G = g0 + gi;
      row = ((x0 + r) << 2) + ((G >> 1) & 3);
      g = ((G >> 3) << 1) + (G & 1);
      off[q] = valid ? ((x0 + r) * w.gy + G) * 16 : IPPM_T_OOB;
    } else {
      row = x0 + r; g = g0 + gi;
      off[q] = valid ? (row * w.gy + g * 4) * 4 : IPPM_T_OOB;
    }
    xrow[q] = row;
    coff[q] = row * w.row_bytes + g;
    ycol[q] = valid ? g * 4 : IPPM_T_FAR;
    touched[q] = 0; keepm[q] = 0;
    const ippm_t_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rmap, off[q], 0, IPPM_T_LOAD_AUX);
    mv[q].v[0] = __uint_as_float(v.x); mv[q].v[1] = __uint_as_float(v.y); mv[q].v[2] = __uint_as_float(v.z); mv[q].v[3] = __uint_as_float(v.w);
  }
  // one op: its record (two 16-byte scalar loads) and its code byte per slot
  struct OpIn { int idx, yu, yd, xl, xh; float lm0, lm1; uint32_t cw[SLOTS]; };
  unsigned rem = active;
  auto fetch = [&](OpIn& o) __attribute__((always_inline)) {
    o.idx = __ffs(rem) - 1;
    rem &= rem - 1u;
    const int4 a = *reinterpret_cast<const int4*>(plan + WS_OPS + o.idx * OP_WORDS);       // {type, src, lm0, yu}
    const int4 b = *reinterpret_cast<const int4*>(plan + WS_OPS + o.idx * OP_WORDS + 4);   // {yd, xl, xr, lm1}
    const bool isf = a.x != 0;
    o.yu = a.w; o.yd = b.x;
    o.xl = b.y; o.xh = b.z - b.y;
    o.lm0 = isf ? __int_as_float(a.z) : 0.f;
    o.lm1 = isf ? __int_as_float(b.w) : 0.f;
    const int cs = isf ? (e * w.n + a.y) * w.TB - b.y * w.row_bytes - (a.w >> 2) : 0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) o.cw[q] = IPPM_X_CODE(__builtin_amdgcn_raw_buffer_load_b8(rcode, coff[q] + cs, 0, 0));
  };
  OpIn cur, nxt;
  fetch(cur);
#pragma unroll
  for (int q = 0; q < SLOTS; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) L[q][j] = mv[q].v[j];
  unsigned cells = 0, opcells = 0;
  bool keeps = false;
  for (;;) {
    const bool more = rem != 0;
    if (more) fetch(nxt);
    const bool is_last = cur.idx == last_op;   // (the plan's last op has the highest index: if the item meets it, it ends the chain)
    keeps = keeps || is_last;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      const int lo = min(max(cur.yu - ycol[q], 0), 4), hi = min(max(cur.yd - ycol[q], 0), 4);
      unsigned cm = ((1u << (hi - lo)) - 1u) << lo;
      if (TL) cm = (unsigned)(xrow[q] - cur.xl) < (unsigned)cur.xh ? cm : 0u;
      touched[q] |= cm;
      keepm[q] = is_last ? cm : keepm[q];
      opcells += (cur.lm0 != 0.f || cur.lm1 != 0.f) ? __popc(cm) : 0;
#ifndef IPPM_X_NOCHAIN
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lm = ippm_masked(ippm_bitmask(cm, j), ippm_blend(ippm_bitmask(cur.cw[q], j), cur.lm1, cur.lm0));
        L[q][j] = ippm_clampl(L[q][j], w.lc) + lm;
      }
#else
      L[q][0] += __uint_as_float(cur.cw[q] & 1u);
#endif
    }
    if (!more) break;
    cur = nxt;
  }
  float amax = 0.f;
#pragma unroll
  for (int q = 0; q < SLOTS; ++q) {
    cells += __popc(touched[q]);
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = ippm_blend(ippm_bitmask(keeps ? keepm[q] : 0u, j), L[q][j], ippm_clampl(L[q][j], w.lc));
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(out[0]), fabsf(out[1])), fmaxf(fabsf(out[2]), fabsf(out[3]))));
    {
      ippm_t_u4 v;
      v.x = __float_as_uint(out[0]); v.y = __float_as_uint(out[1]); v.z = __float_as_uint(out[2]); v.w = __float_as_uint(out[3]);
      if (MIS) {
        const bool tail = ycol[q] + 4 > w.gy;
        __builtin_amdgcn_raw_buffer_store_b128(v, rmap, tail ? IPPM_T_OOB : off[q], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(v.x, rmap, tail ? off[q] : IPPM_T_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(v.y, rmap, tail && ycol[q] + 1 < w.gy ? off[q] + 4 : IPPM_T_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(v.z, rmap, tail && ycol[q] + 2 < w.gy ? off[q] + 8 : IPPM_T_OOB, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(v, rmap, off[q], 0, IPPM_T_STORE_AUX);
      }
    }
    if (TRACK) tile_area_slot<MIS>(w, xrow[q], ycol[q] == IPPM_T_FAR ? 0 : ycol[q], mv[q].v, out);
    if (IPPM_X_REWARD && is_global) {   // the reward terms, as in tile_item
      float wa[4], wb[4], wsum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t tm = ippm_bitmask(touched[q], j);
        wa[j] = ippm_masked(tm, ippm_weight_l(out[j], w.wt));
        wb[j] = ippm_masked(tm, ippm_weight_l(mv[q].v[j], w.wt));
        wsum += wa[j] + wb[j];
      }
      if (__any(wsum != 0.f)) {
        float s1 = 0.f, sD = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float hb = ippm_entropy_l(mv[q].v[j], w.lc), ha = ippm_entropy_l(out[j], w.lc);
          s1 += wa[j] * (hb - ha);
          sD += (wa[j] - wb[j]) * hb;
        }
        acc.a1 += (double)s1;
        acc.aD += (double)sD;
      }
    }
  }
  if (__any(amax > w.lc) && w.lane == 0) w.ws[(size_t)map_abs * IPPM_WS_WORDS + WS_FLAG_A] = 1;
  if (TRACK) tile_area_flush(w, map_abs);
  if (is_global) { acc.cells_g += cells; acc.ops_g += opcells; }
  else { acc.cells_l += cells; acc.ops_l += opcells; }
}

// Workgroup = one wavefront; the grid is (envs, wavefronts per env): wavefront `first` = blockIdx.y of an env takes items first,
// first + gridDim.y, ... of the env's list (which env a workgroup serves: blockIdx.x, rotated with `first` -- see the kernel).
// (wavefronts per SIMD: the untracked instantiation fits in 80 VGPRs without scratch and takes 6 -- for every team size since
// round 5, items met by more than six ops run the chain in chunks)
#ifndef IPPM_TILE_WAVES_PER_EU
#define IPPM_TILE_WAVES_PER_EU 5
#endif
// (the untracked tile-storage instantiation needs 84 registers for its six wavefronts' 80: four are spilled (20 bytes of scratch, re-read around the items).
//  Measured alternatives, profiles/r06/tile_storage_ab.txt: five wavefronts per SIMD and no spill 149 against 141 us at 2048 envs x 4 UAVs x 256^2, 865 against 873 at config
//  4's shape; the row packed into the column register: nine spills, 157 us; per-slot row masks built before the chain: twenty-eight.)
#ifndef IPPM_TL_WAVES_PER_EU
#define IPPM_TL_WAVES_PER_EU 6
#endif
template <bool MIS, bool TRACK, bool TL = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(!MIS && !TRACK ? (TL ? IPPM_TL_WAVES_PER_EU : 6) : (TRACK ? 4 : IPPM_TILE_WAVES_PER_EU), 8)))
k_fuse_tiles(const int32_t* __restrict__ work, int n_envs, int env_cap, int rot, int n, int gx, int gy, int row_bytes, int TB, float lc, float wt,
             const int32_t* __restrict__ plan_ro, float* __restrict__ local, float* __restrict__ global,
             const uint8_t* __restrict__ code, int32_t* __restrict__ ws, double* __restrict__ sums,
             unsigned long long* __restrict__ counters, double* __restrict__ area) {
  // (argument order = latency order, as in k_sense_tiles: the work list's address and sizes arrive in SGPRs with the wavefront,
  // every config scalar by value -- the count and the first item are one scalar round trip away, the first item's cells two)
  // Consecutive workgroups = consecutive envs, and the hardware deals consecutive workgroups out to the eight XCDs in turn:
  // wavefront `first` of env e is workgroup e + n_envs * first and runs on XCD (e + (n_envs % 8) * first) % 8.  With an odd batch
  // an env's wavefronts visit all eight XCDs; with a multiple of 8 envs every one of them runs on XCD e % 8 -- each env's whole
  // list on one eighth of the chip.  Harmless while the lists are alike; with per-episode comm ranges or mixed team sizes
  // (config 5) the lists differ fifty-fold and the launch waits for the XCD that drew the long ones (teams dealt out 2, 4, 8, 16
  // by e % 4: every 16-UAV env on XCDs 3 and 7).  `rot` (set for even batches): wavefront `first` of an env sits (first % 8)
  // places further on in its row of workgroups, XCD (e + (n_envs % 8 - 1) * first) % 8 -- an odd coefficient again.  (rot > 1: 2^(rot-1)
  // consecutive wavefronts stay on one XCD -- ((first >> (rot - 1)) % 8) places further on: neighbouring items share lines.)
  const int first = blockIdx.y, step = gridDim.y;
  int env = blockIdx.x + (rot ? ((first >> (rot - 1)) & 7) : 0);   // (rot - 1: consecutive wavefronts per XCD, as a power of two)
  env -= env >= n_envs ? n_envs : 0;
  const int4* __restrict__ items = reinterpret_cast<const int4*>(work + ((n_envs + 3) & ~3)) + (size_t)env * env_cap;
  // count and first item are requested together (the item's address does not depend on the count)
  int tag = work[env];
  int4 it = items[min(first, env_cap - 1)];
  // (pinned: the compiler would sink the item and the pointer arguments behind the test of the tag, a second scalar round trip)
  asm volatile("" : "+s"(tag), "+s"(it.x), "+s"(it.y), "+s"(it.z), "+s"(it.w), "+s"(local), "+s"(global), "+s"(code), "+s"(ws), "+s"(sums),
               "+s"(counters));
  const int lane = threadIdx.x;
  if (!(tag & IPPM_WORK_TILED) || (tag & IPPM_WORK_OVERFLOW)) {  // a list of the other form, or one that did not fit: say so
    if (first == 0 && lane == 0 && counters) atomicAdd(&counters[(env & (IPPM_COUNTER_SLOTS - 1)) * 8 + 6], 1ull);
    return;
  }
  const int count = tag & IPPM_WORK_COUNT;
  if (first >= count) return;
  __shared__ double s_area[TRACK ? IPPM_FEAT * IPPM_AREA_LD + IPPM_AREA_LD : 1];
#ifdef IPPM_X_LDS_PAD   // measurement-only variants (make VARIANT=occN EXTRA=-DIPPM_X_LDS_PAD=bytes): LDS nobody needs, to cap the wavefronts a CU holds (160 KB / bytes)
  __shared__ int s_pad[IPPM_X_LDS_PAD / 4];
  if (n_envs < 0) { s_pad[threadIdx.x] = first; __syncthreads(); if (counters) counters[0] += s_pad[(threadIdx.x + 1) & 63]; }
#endif
  if (TRACK) {
    for (int k = lane; k < IPPM_FEAT * IPPM_AREA_LD + IPPM_AREA_LD; k += 64) s_area[k] = 0.0;
    __syncthreads();
  }
  TileCtx w;
  w.s_area = s_area; w.area = area;
  w.inv_gx = TRACK ? __builtin_amdgcn_rcpf((float)gx) : 0.f;
  w.inv_gy = TRACK ? __builtin_amdgcn_rcpf((float)gy) : 0.f;
  w.local = local; w.global = global; w.code = code; w.plan = plan_ro; w.ws = ws;
  w.n = n; w.gx = gx; w.gy = gy;
  w.row_bytes = row_bytes;
  w.TB = TB;
  w.n_envs = n_envs;
  w.lc = lc; w.wt = wt;
  w.lane = lane;
  TileAcc acc;
  acc.a1 = acc.aD = 0.0;
  acc.cells_l = acc.ops_l = acc.cells_g = acc.ops_g = 0;
  for (int i = first; i < count; i += step) {
    const int4 nx = items[min(i + step, env_cap - 1)];  // the next item travels while this one is worked on
    const int slot = (unsigned)it.w >> 24, gs = it.x & 0xFFFF, cnt = (unsigned)it.x >> 16, x0 = it.y, g0 = it.z & 0xFFFF, W = (unsigned)it.z >> 16;
    const unsigned active = (unsigned)it.w & 0x00FFFFFFu;
    const int na = __popc(active);
    if (na == 1) tile_item<1, 4, MIS, TRACK, TL>(w, acc, env, slot, x0, cnt, gs, g0, W, active);
    else if (na == 2) tile_item<2, 4, MIS, TRACK, TL>(w, acc, env, slot, x0, cnt, gs, g0, W, active);
    else if (na == 3) tile_item<3, 4, MIS, TRACK, TL>(w, acc, env, slot, x0, cnt, gs, g0, W, active);
    else if (na == 4) tile_item<4, 4, MIS, TRACK, TL>(w, acc, env, slot, x0, cnt, gs, g0, W, active);
    else if (na <= 6) tile_item<6, IPPM_X_SLOTS56, MIS, TRACK, TL>(w, acc, env, slot, x0, cnt, gs, g0, W, active);
    else tile_item_long<MIS, TRACK, TL>(w, acc, env, slot, x0, cnt, gs, g0, W, active);
    it = nx;
  }
  // the wavefront's reward terms and work counters: one atomic per quantity
  const float cl = ippm_wave_sum((float)acc.cells_l), ol = ippm_wave_sum((float)acc.ops_l);
  const float cg = ippm_wave_sum((float)acc.cells_g), og = ippm_wave_sum((float)acc.ops_g);
  double a1 = acc.a1, aD = acc.aD;
  if (cg > 0.f) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); aD += __shfl_xor(aD, o, 64); }
  }
  if (lane < 3) {
    // sum (w(a) H(a) - w(b) H(b)) = sum (w(a) - w(b)) H(b) - sum w(a) (H(b) - H(a))
    const double v = lane == 0 ? a1 : (lane == 1 ? aD : aD - a1);
    if (cg > 0.f && sums && v != 0.0) atomicAdd(&sums[(size_t)env * 8 + SUM_ACC1 + lane], v);
  } else if (lane < 7 && counters) {
    const float v = lane == 3 ? cl : (lane == 4 ? ol : (lane == 5 ? cg : og));
    if (v > 0.f) atomicAdd(&counters[((env + first) & (IPPM_COUNTER_SLOTS - 1)) * 8 + 1 + (lane - 3)], (unsigned long long)v);
  }
}

// ippm_fuse_step without area sums on a config that has the tile form (ippm_ctx::tiles); `work` must have been written by
// ippm_plan_step with IPPM_STEP_TILES.
int ippm_launch_fuse_tiles(ippm_ctx* ctx, float* local, float* global, const uint8_t* code, int32_t* ws, double* sums, double* area,
                           const int32_t* work, int n_envs, hipStream_t st) {
  const ippm_config& c = ctx->cfg;
  const int max_ops = c.n_agents + 1;
  const int env_cap = ippm_tile_env_cap(ctx);
  // wavefronts per env: about three items each.  An item is <= 256 lane-loads (1024 cells); a step touches roughly half of the
  // maps over a third of their cells.  Up to 2048 per env: with per-episode comm ranges (config 5) the envs' lists differ fifty-fold
  // in length (484 .. 21 611 items at 64 envs x 16 UAVs x 1024^2) and the wavefronts of a long list are the launch's tail --
  // 256 / 1024 / 2048 per env: 962 / 853 / 850 us there; a wavefront whose env has nothing left for it costs a scalar load.
  // (Since an env's wavefronts go round all XCDs -- `rot` below -- 2048 is worth 3 %: 772 -> 748 us at 64 envs, 934 -> 906 on 256
  // envs of mixed teams; the estimate gives config 5's shape 1844.)
  // (Dealing the wavefronts out in proportion to the lists -- ceil(count / 8 .. 64) per env from a taller grid -- was 6 - 15 % SLOWER
  // than 1024 for everybody -- measured while each env's list still ran on one XCD (see `rot` below), which is what held those
  // launches up, not their longest chains; profiles/r05/c5_wave_distribution.txt)
  const double est_items = 0.5 * (c.n_agents + 1) * (double)c.grid_x * c.grid_y / 3.0 / (256.0 * ippm_tile_slots(max_ops));
  int per_env = ctx->knob_tile_waves > 0 ? ctx->knob_tile_waves : (int)std::max(4.0, std::min(2048.0, est_items / 3.0));
  per_env = std::max(1, std::min(per_env, env_cap));
  // a launch smaller than the chip's wave slots leaves CUs idle: small batches take more wavefronts per env
  while ((long long)per_env * n_envs < 16384 && per_env * 2 <= env_cap && per_env < 256) per_env *= 2;
  // rot - 1 = log2 of the consecutive wavefronts of an env that share an XCD: neighbouring items share code-tile lines and the
  // line at the seam of their runs, so four in a row on one L2 are 1 - 1.5 % faster than one (profiles/r05/tile_rotate_group_ab.txt)
  // -- as long as an env's wavefronts still go round all eight XCDs.  Odd batches spread by themselves.
  int rot = 0;
  if (n_envs >= 8 && n_envs % 2 == 0 && ctx->knob_tile_rotate != 0) {
    rot = 1;
    while (rot < 3 && (per_env >> rot) >= 8) ++rot;
    if (ctx->knob_tile_rotate > 0) rot = std::min(ctx->knob_tile_rotate, 6);
  }
  dim3 grid((unsigned)n_envs, (unsigned)per_env), block(64);
#define IPPM_FT_L(...) IPPM_LAUNCH(ctx, IPPM_T_FUSE, (k_fuse_tiles<__VA_ARGS__>), grid, block, st, work, n_envs, env_cap, rot, c.n_agents, c.grid_x, c.grid_y, c.tile_stride >> 2, \
              (int)ippm_tile_bytes(c.tile_stride, 4), c.logit_clip, c.logit_weight_thr, ws, local, global, code, ws, sums, ctx->dcounters, area)
  // (the kernel's name as written here is what ippm_read_kernel_times reports: <rows only 4-byte aligned, area sums tracked[, tile storage]>)
  const bool mis = (c.grid_y & 3) != 0;   // the instantiation with the cell-by-cell row-tail stores
  if (ctx->tl) { if (area) IPPM_FT_L(false, true, true); else IPPM_FT_L(false, false, true); }
  else if (mis) { if (area) IPPM_FT_L(true, true); else IPPM_FT_L(true, false); }
  else { if (area) IPPM_FT_L(false, true); else IPPM_FT_L(false, false); }
#undef IPPM_FT_L
  IPPM_LAUNCH_CHECK("fuse_tiles");
  return 0;
}
