// K4 / K5: apply the planned fusion ops to the local maps and the global map (gfx950), each touched cell read once and
// written once, in ONE launch for all maps of all envs.
//
//   Mapping.fuse_map(..., "local" / "global")   mapping/mappings.py:80-124
//   get_global_reward terms (K5)                 utils/reward.py:68-82, utils/state.py:53-121
//
// Work decomposition ("row walker").  A WAVEFRONT owns a run of consecutive rows of one map's op hull (a workgroup = four
// consecutive runs of the same map).  The ops of a plan are rectangles, so along x the set of ops covering a row changes
// only at rectangle edges: the rows of a run fall into a few SLABS with a uniform active-op set.  Per slab the wave
//   - finds the active set and the slab's end with a short scalar loop over the plan (the op table sits in VGPR lanes,
//     lane o = op o, fields fetched with v_readlane: dynamic indexing at one instruction per field),
//   - compacts the active ops into NA register slots and jumps to the row loop compiled for that NA (1, 2, 3, 4, 6, ..):
//     straight-line code, no branch inside a row, the ordered clamp/add chain of exactly the ops that are there,
//   - covers the column hull of the active ops with dense 4-cell lane groups (like K3); which cells of a lane's group
//     each op covers is a per-lane bit mask computed once per slab.
//
// History, because the numbers shaped this.  Round 1 decomposed by (map, op): workgroups of op k walked op k's whole
// rectangle and skipped every group a later op also covered (K4 99 us + K5 92 us at 1024 envs x 4 UAVs x 256^2; lanes
// visited sum(A) cells to own the union, one row in flight).  The first row walker kept all NK ops in scalar registers and
// guarded each op's loads and chain step with a scalar branch: 61 branches and ~1200 instructions per two rows whatever
// the slab held, every wave of a workgroup repeating the slab setup -- 285 us, of which 83 us setup and 168 us rows,
// issue-bound at 13 % of the HBM peak.  (It also taught that per-cell selects on loop-invariant masks get hoisted into
// saved-exec branches by the compiler, 5x slower again: IPPM_OPAQUE below.)
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "ippm_tiles.h"

// empty asm: makes a loop-invariant value look loop-variant to the optimiser (no instruction is emitted)
#define IPPM_OPAQUE(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ int lane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

struct OpTable {   // lane o holds op o of the plan (zeros beyond the plan: an empty rectangle that never covers anything)
  int yu, yd;
  int cs;          // uniform part of the op's code byte offsets (see walk_slab); out of range for ops without bits
  float lm0, lm1;  // log-odds of the two measurement values of a fuse op (0, 0 for a clamp-only op)
};
struct SlabTable { // lane s holds slab s of the plan
  int xa, xb, active, hull;
};

typedef unsigned ippm_u4 __attribute__((ext_vector_type(4)));
// raw buffer resources (gfx9 family descriptor word 3 = 0x00020000): uniform base + per-lane 32-bit byte offset, one VALU
// per address instead of a 64-bit multiply-add chain, and loads past the end return 0 instead of faulting
#define IPPM_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)

struct WaveCtx {   // what a wave needs while it walks its rows
  __amdgpu_buffer_rsrc_t map;    // this map: gx * gy floats
  __amdgpu_buffer_rsrc_t code;   // the whole code tensor (lanes outside an op's columns may point anywhere inside it)
  double* s_area;
  double* sums_env;  // this env's reward sums (global map only), nullptr otherwise
  int gx, gy, row_bytes;
  int tl;            // tile storage of the maps (ippm_internal.h): a cell's address goes through ippm_cell_index
  float lc, wt, lp, inv_gx, inv_gy;
  double lp64;       // logit(prior) as the reference holds it (a Python float): the SHIFT chain subtracts it per message
  int lane;
  int last_op;
  unsigned fusemask;
  bool is_global;
};

struct WaveAcc {
  bool exceed;
  // sum w(a) (H(b) - H(a)) and sum (w(a) - w(b)) H(b) (the increment of T = sum w H is aD - a1): float64 per lane, fed with the
  // float32 sum of a row's cells -- with noise-free measurements the terms are +-1 and the sums what is left after they cancel
  double a1, aD;
  unsigned cells, opcells;
};

// Shannon entropy of sigmoid(clamp(l)) in float64 (SHIFT path: the float32 form's 1e-7 absolute error per cell, summed over
// a whole grid of barely changed cells, would show at 1e-4 in the returns)
__device__ __forceinline__ double entropy_l_f64(float l, float lc) {
  const double a = fmin(fabs((double)l), (double)lc);
  const double e = exp(-a), d = 1.0 + e;
  return log2(d) + a * 1.4426950408889634 * (e / d);
}

#ifndef IPPM_FUSE_LOAD_AUX   // cache policy of the map accesses (bit 1 = non-temporal on gfx950)
#define IPPM_FUSE_LOAD_AUX 0
#endif
#ifndef IPPM_FUSE_STORE_AUX
#define IPPM_FUSE_STORE_AUX 0
#endif
template <int VEC>
__device__ __forceinline__ CellVec<VEC> buf_load_cells(__amdgpu_buffer_rsrc_t r, int off) {
  CellVec<VEC> c;
  if (VEC == 4) {
    const ippm_u4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, IPPM_FUSE_LOAD_AUX);
    c.v[0] = __uint_as_float(t.x); c.v[1 % VEC] = __uint_as_float(t.y); c.v[2 % VEC] = __uint_as_float(t.z); c.v[3 % VEC] = __uint_as_float(t.w);
  } else {
    c.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, IPPM_FUSE_LOAD_AUX));
  }
  return c;
}
template <int VEC>
__device__ __forceinline__ void buf_store_cells(__amdgpu_buffer_rsrc_t r, int off, const CellVec<VEC>& c) {
  if (VEC == 4) {
    ippm_u4 t;
    t.x = __float_as_uint(c.v[0]); t.y = __float_as_uint(c.v[1 % VEC]); t.z = __float_as_uint(c.v[2 % VEC]); t.w = __float_as_uint(c.v[3 % VEC]);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, IPPM_FUSE_STORE_AUX);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c.v[0]), r, off, 0, IPPM_FUSE_STORE_AUX);
  }
}

// Rows [x, xe) of one slab with the ops of `set` (bit o = op o), at most NA of them.
// Each of the rpw sub-rows of the wavefront takes a contiguous block of the slab's rows and keeps FU consecutive rows in
// flight (all loads issued before the first use); rows therefore ascend by one per lane, which is what lets the area-sum
// accumulator stay in registers for ~gx/11 rows between two LDS flushes.
// SHIFT (mapping.prior != 0.5, the explicit slow path): apply_update subtracts logit(prior) from EVERY cell of the map for
// every fused message (mappings.py:112-116: the 0.5 padding of map2communicate has logit 0, the prior term does not
// vanish).  There `set` holds every message of the plan, `rowin` says which of them cover these rows, the lanes span the
// whole grid width, every cell runs the whole chain, and the deferred-clamp bookkeeping is moot.
template <int VEC, bool TRACK, bool SHIFT, int NA, int FU>
__device__ __forceinline__ void walk_slab(const WaveCtx& w, const OpTable& t, WaveAcc& acc_out, unsigned set, unsigned rowin, int x,
                                          int xe, int ya, int yb) {
  constexpr unsigned QM = (1u << VEC) - 1u;
  // compact the ops of the set into NA slots (uniform values; spare slots read lane 63 = the empty op)
  int yu[NA], yd[NA];
  int cshift[NA];      // uniform part of a slot's code byte offset; a huge value (-> out of range, reads 0) for slots without bits
  float lm0[NA], lm1[NA];
  double lpk[SHIFT ? NA : 1];   // SHIFT: logit(prior) for the slots that are messages
  // SHIFT: every cell of the grid enters the reward sums with a tiny H(b) - H(a); these are summed in float64 and go out
  // at the end of the slab (kept out of WaveAcc: the common path must not carry six more registers)
  double sd1 = 0.0, sdD = 0.0, sdT = 0.0;
  int keep_slot = -1;  // slot of the plan's last op: its outputs stay unclamped
  // Spare slots come FIRST: an empty slot still clips (like every op of the reference), which is a no-op ahead of the
  // first real op but would wrongly clip the last op's outputs behind it.
  unsigned rem = set;
  const int pad = NA - __popc(set);
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const int idx = k < pad ? 63 : __ffs(rem) - 1;
    rem = k < pad ? rem : (rem & (rem - 1u));
    yu[k] = lane_i(t.yu, idx); yd[k] = lane_i(t.yd, idx);
    lm0[k] = lane_f(t.lm0, idx); lm1[k] = lane_f(t.lm1, idx);
    cshift[k] = lane_i(t.cs, idx);
    const bool rin = !SHIFT || (idx < 32 && ((rowin >> idx) & 1u));
    if (SHIFT && !rin) { yu[k] = 0; yd[k] = 0; cshift[k] = 0x7F000000; }  // a message whose footprint misses these rows: shift only
    if (SHIFT) lpk[SHIFT ? k : 0] = k >= pad ? w.lp64 : 0.0;
    keep_slot = (idx == w.last_op) ? k : keep_slot;
  }
  const RowGeom g = fit_geom<VEC>(ya, yb, xe - x, 1);
  const int sub = w.lane >> g.shift, gl = w.lane & (g.lpr - 1);
  // contiguous row block of my sub-row
  const int block = (xe - x + g.rpw - 1) >> (6 - g.shift);
  const int rs = x + sub * block, re = min(xe, rs + block);
  for (int gi = gl; gi < g.groups; gi += g.lpr) {
    const int y = g.y0 + gi * VEC;
    // column-only part, once per slab: which cells of my group each op covers, which cells the slab touches at all, which
    // of them carry a measurement, which keep their unclamped output
    unsigned cm[NA];
    unsigned touched = 0, keepm = 0, ops_here = 0;
    // cells of my group that exist in this row (a row's last group hangs over into the next row when the grid is not a multiple of
    // VEC wide); only the whole-row walk of SHIFT can meet cells no op's columns exclude
    const unsigned vm = SHIFT ? (1u << min(max(w.gy - y, 0), VEC)) - 1u : QM;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      unsigned mq = 0;
#pragma unroll
      for (int q = 0; q < VEC; ++q) mq |= ((unsigned)(y + q - yu[k]) < (unsigned)(yd[k] - yu[k])) ? (1u << q) : 0u;
      cm[k] = mq;
      const unsigned inm = SHIFT ? (k >= pad ? vm : 0u) : mq;
      touched |= inm;
      ops_here += __popc(inm);
      keepm = (k == keep_slot) ? mq : keepm;
    }
    if (SHIFT) keepm = QM;  // the last message's update reaches every cell: all outputs stay unclamped
    if (touched == 0) continue;  // a gap between side-by-side rectangles: nothing to read or write
    AreaCols<VEC> ac;
    AreaAcc acc;
    if (TRACK) { ac = area_cols<VEC>(y, w.gy, w.inv_gy); acc.init(); }
    const int ycode = VEC == 4 ? (y >> 2) : y;
    const int ybyte = y * 4, gybyte = w.gy * 4;
    float amax = 0.f;
#ifdef IPPM_X_NOROWS
    amax = (float)(cshift[0] + cm[0] + cm[NA - 1]) + lm0[0] + lm1[NA - 1] + (float)keepm;
    for (int row0 = rs; row0 < rs; row0 += FU) {
#else
    for (int row0 = rs; row0 < re; row0 += FU) {
#endif
      // issue every load of FU rows (map cells + one measurement-code byte per slot) before any use
      CellVec<VEC> mvu[FU];
      uint32_t cwu[FU][NA];
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const int row = min(row0 + u, re - 1);  // past the block's end: the last row again (its results are dropped below)
        mvu[u] = buf_load_cells<VEC>(w.map, w.tl ? ippm_cell_index(row, y, w.gy, 1) * 4 : row * gybyte + ybyte);
        const int rowoff = row * w.row_bytes + ycode;
#pragma unroll
        for (int k = 0; k < NA; ++k) cwu[u][k] = __builtin_amdgcn_raw_buffer_load_b8(w.code, rowoff + cshift[k], 0, 0);
      }
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        // no branch on the row's validity: a row past the block's end stores out of range (dropped by the buffer's bounds
        // check) and enters the sums with weight 0
        const bool valid = u == 0 || row0 + u < re;
        const int row = row0 + u;
        CellVec<VEC>& mv = mvu[u];
        // SHIFT: the chain runs in float64 registers and is rounded once, at the store -- there every message of the plan adds
        // to every cell of the grid, and with a float32 rounding per message (7 UAVs x 16 steps = 112 of them) 0.04 % of the cells
        // drifted past 1e-5; one rounding per fusion keeps all of them inside
        using LT = typename std::conditional<SHIFT, double, float>::type;
        LT L[VEC];
        float bsave[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) { L[q] = mv.v[q]; bsave[q] = mv.v[q]; }
        // Ordered clamp/add chain (mappings.py:80-124 in log-odds): every op of the reference clips its input over the WHOLE
        // grid (mappings.py:110-111), then adds the measurement's log-odds inside its footprint.  So every cell this lane
        // holds may be clipped at every op, covered or not (for an uncovered cell that is the reference's own full-grid
        // clip; the deferred-clamp plan guarantees it is a no-op there); only the addend is masked to the footprint.
#ifndef IPPM_X_NOCHAIN
#pragma unroll
        for (int k = 0; k < NA; ++k) {
          const uint32_t cw = cwu[u][k];
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            const float lm = ippm_masked(ippm_bitmask(cm[k], q), ippm_blend(ippm_bitmask(cw, q), lm1[k], lm0[k]));
            if (SHIFT) L[q] = fmin(fmax((double)L[q], -(double)w.lc), (double)w.lc) + ((double)lm - lpk[SHIFT ? k : 0]);
            else L[q] = ippm_clampl((float)L[q], w.lc) + lm;
          }
        }
#else
        for (int k = 0; k < NA; ++k) L[0] += __uint_as_float(cwu[u][k]);
#endif
        // outputs of the plan's last op stay unclamped (its rectangle is remembered as possibly out of range); every other
        // cell was clipped again by a later op of the reference
        float d[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const float a = ippm_blend(ippm_bitmask(keepm, q), (float)L[q], (float)(SHIFT ? fmin(fmax((double)L[q], -(double)w.lc), (double)w.lc)
                                                                                             : (double)ippm_clampl((float)L[q], w.lc)));
          amax = fmaxf(amax, fabsf(a));
          mv.v[q] = a;
          if (TRACK) d[q] = valid && (!SHIFT || ((vm >> q) & 1u)) ? sigmoid_diff(a, bsave[q]) : 0.f;
        }
        {
          const int soff = valid ? (w.tl ? ippm_cell_index(row, y, w.gy, 1) * 4 : row * gybyte + ybyte) : 0x7FFFFFF0;
          if (VEC == 4 && (w.gy & 3) != 0) {
            // (uniform) rows are not a multiple of 4 wide: the last group of a row hangs over into the next row -- its cells go
            // out one by one, another lane owns the rest
            const bool tail = y + 4 > w.gy;
            buf_store_cells<VEC>(w.map, tail ? 0x7FFFFFF0 : soff, mv);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mv.v[0]), w.map, tail ? soff : 0x7FFFFFF0, 0, IPPM_FUSE_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mv.v[1 % VEC]), w.map, tail && y + 1 < w.gy ? soff + 4 : 0x7FFFFFF0, 0, IPPM_FUSE_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mv.v[2 % VEC]), w.map, tail && y + 2 < w.gy ? soff + 8 : 0x7FFFFFF0, 0, IPPM_FUSE_STORE_AUX);
          } else {
            buf_store_cells<VEC>(w.map, soff, mv);
          }
        }
        if (TRACK) area_row<VEC>(acc, w.s_area, ac, min(row, re - 1), w.gx, w.inv_gx, d);
#ifndef IPPM_X_NOREWARD
        if (w.is_global) {
          // information-gain terms (utils/reward.py:68-82); a cell that received no measurement contributes exact zeros
          // (same weight, and the entropy clips its argument).  Rows whose cells all have weight 0 before and after
          // (believed free, still believed free) skip the entropies: wave-uniform on spatially coherent terrain.
          float wa[VEC], wb[VEC], wsum = 0.f;
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            const bool here = valid && (!SHIFT || ((vm >> q) & 1u));
            wa[q] = here ? ippm_weight_l(mv.v[q], w.wt) : 0.f;
            wb[q] = here ? ippm_weight_l(bsave[q], w.wt) : 0.f;
            wsum += wa[q] + wb[q];
          }
          if (__any(wsum != 0.f)) {
            float r1 = 0.f, rD = 0.f;
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
              if (SHIFT) {
                const double hb = entropy_l_f64(bsave[q], w.lc), ha = entropy_l_f64(mv.v[q], w.lc);
                sd1 += (double)wa[q] * (hb - ha);
                sdD += (double)(wa[q] - wb[q]) * hb;
                sdT += (double)wa[q] * ha - (double)wb[q] * hb;
              } else {
                const float hb = ippm_entropy_l(bsave[q], w.lc), ha = ippm_entropy_l(mv.v[q], w.lc);
                r1 += wa[q] * (hb - ha);
                rD += (wa[q] - wb[q]) * hb;
              }
            }
            if (!SHIFT) { acc_out.a1 += (double)r1; acc_out.aD += (double)rD; }
          }
        }
#endif
      }
    }
    acc_out.exceed |= amax > w.lc;
    const unsigned nrows = (unsigned)max(re - rs, 0);
    acc_out.cells += nrows * __popc(touched);
    acc_out.opcells += nrows * ops_here;
    if (TRACK) acc.flush(w.s_area, ac.cb);
  }
  if (SHIFT && w.sums_env) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sd1 += __shfl_xor(sd1, o, 64); sdD += __shfl_xor(sdD, o, 64); sdT += __shfl_xor(sdT, o, 64); }
    if (w.lane < 3) {
      const double v = w.lane == 0 ? sd1 : (w.lane == 1 ? sdD : sdT);
      if (v != 0.0) atomicAdd(&w.sums_env[SUM_ACC1 + w.lane], v);
    }
  }
}

#ifndef IPPM_FU_SMALL
#define IPPM_FU_SMALL 2  // rows in flight per lane in the row loops for 1-2 active ops
#endif
#ifndef IPPM_FU_MID
#define IPPM_FU_MID 2    // ... for 3-4 active ops
#endif
#ifndef IPPM_FU_BIG
#define IPPM_FU_BIG 1    // ... for more
#endif

// One work item = (map, run of `wave_rows` consecutive rows of the map's op hull), done by one wavefront.
// NAMAX = plan-size class of the launch (6 / 10 / 18): the row loops compiled in are those for <= NAMAX active ops.
template <int VEC, bool TRACK, int NAMAX, bool SHIFT>
__device__ __forceinline__ void fuse_item(const ippm_config* __restrict__ c, float* __restrict__ local, float* __restrict__ global,
                                          const uint8_t* __restrict__ code, const int32_t* __restrict__ plan_ro,
                                          int32_t* __restrict__ ws, double* __restrict__ sums, double* __restrict__ area,
                                          unsigned long long* __restrict__ counters, double* s_area, int wave_rows, int min_ops,
                                          int n_envs_total, int e, int slot, int chunk, int cslot, int tl) {
  const int n = c->n_agents;
  const bool is_global = slot == n;
  const size_t wbase = (size_t)(e * (n + 1) + slot) * IPPM_WS_WORDS;
  const int lane = threadIdx.x & 63;
  // Preamble = ONE memory round trip: the header (scalar) and the op table (lane o = op o; lane l also fetches plan edge l)
  // are requested before anything is waited for; nothing here depends on a loaded value.
  const int32_t* __restrict__ hdr = plan_ro + wbase + WS_PLAN;
  const int nops = hdr[PL_NOPS], X0 = hdr[PL_X0], X1 = hdr[PL_X1], last_op = hdr[PL_LAST];
  const int4 oa = *reinterpret_cast<const int4*>(plan_ro + wbase + WS_OPS + min(lane, IPPM_MAX_OPS - 1) * OP_WORDS);
  const int4 ob = *reinterpret_cast<const int4*>(plan_ro + wbase + WS_OPS + min(lane, IPPM_MAX_OPS - 1) * OP_WORDS + 4);
  int edge = plan_ro[wbase + WS_OPS + min(lane >> 1, IPPM_MAX_OPS - 1) * OP_WORDS + ((lane & 1) ? OP_XR : OP_XL)];
  const int gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  constexpr bool shift = SHIFT;  // mapping.prior != 0.5: its own instantiation, so the common path carries none of its code
  if (nops > NAMAX || nops < min_ops) return;  // (another launch handles other plan sizes; 0 ops: nothing to do)
  const int r0 = X0 + chunk * wave_rows, r1 = min(X1, r0 + wave_rows);
  if (r0 >= r1) return;
  const int row_bytes = VEC == 4 ? (S >> 2) : S;
  const int TB = (int)ippm_tile_bytes(S, VEC);
  OpTable t;
  int t_xl, t_xr;
  unsigned fusemask;
  {
    // op record: {type, src, lm0, yu | yd, xl, xr, lm1}
    const bool on = lane < nops;
    const bool isf = on && oa.x != 0;
    t.yu = on ? oa.w : 0; t.yd = on ? ob.x : 0;
    t_xl = on ? ob.y : 0; t_xr = on ? ob.z : 0;
    t.lm0 = isf ? __int_as_float(oa.z) : 0.f;
    t.lm1 = isf ? __int_as_float(ob.w) : 0.f;
    // byte of cell group (row, y) in the op's tile = tile base + (row - xl) * row_bytes + (y - (yu & ~3)) / cells per byte
    //                                              = (row * row_bytes + y / cells per byte)  +  cs
    const int y0 = oa.w & ~3;
    t.cs = isf ? (e * n + oa.y) * TB - ob.y * row_bytes - (VEC == 4 ? (y0 >> 2) : y0) : 0x7F000000;
    fusemask = (unsigned)__ballot(isf);
  }
  // ---- slab table, built by the whole wavefront at once: the ops are rectangles, so along x the set of ops covering a row
  // changes only at rectangle edges.  Lane l holds edge l (xl / xr of op l >> 1; prior != 0.5 adds the grid's borders),
  // ranks it among all edges (ties by lane), the edges are permuted into sorted order, and lane s then owns slab
  // [sorted s, sorted s+1) and collects the ops covering it.
  const int n_edges = 2 * nops + (shift ? 2 : 0);
  if (shift && lane >= 2 * nops) edge = lane == 2 * nops ? 0 : gx;
  int rank = 0;
  for (int j = 0; j < n_edges; ++j) {
    const int ej = lane_i(edge, j);
    rank += (ej < edge || (ej == edge && j < lane)) ? 1 : 0;
  }
  const int sorted = __builtin_amdgcn_ds_permute((lane < n_edges ? rank : lane) << 2, edge);  // lane `rank` receives my edge
  SlabTable st;
  st.xa = sorted;
  st.xb = __builtin_amdgcn_ds_bpermute(min(lane + 1, 63) << 2, sorted);
  if (lane + 1 >= n_edges) st.xb = st.xa;  // no slab behind the last edge
  st.active = 0;
  int hya = gy, hyb = 0;
  for (int o = 0; o < nops; ++o) {
    const int oxl = lane_i(t_xl, o), oxr = lane_i(t_xr, o), oyu = lane_i(t.yu, o), oyd = lane_i(t.yd, o);
    const bool in = oxl <= st.xa && st.xa < oxr && st.xb > st.xa;
    st.active |= in ? (1 << o) : 0;
    hya = in ? min(hya, oyu) : hya;
    hyb = in ? max(hyb, oyd) : hyb;
  }
  st.hull = hya | (hyb << 16);
  const int nslabs = max(n_edges - 1, 0);
  WaveCtx w;
  w.gx = gx; w.gy = gy;
  w.row_bytes = row_bytes;
  w.tl = tl;
  w.map = IPPM_RSRC(is_global ? global + (size_t)e * gx * gy : local + (size_t)(e * n + slot) * gx * gy, (size_t)gx * gy * 4);
  w.code = IPPM_RSRC(code, (size_t)n_envs_total * n * TB);
  w.s_area = s_area;
  w.lc = c->logit_clip; w.wt = c->logit_weight_thr; w.lp = c->logit_prior;
  w.lp64 = c->logit_prior_f64;
  w.inv_gx = w.inv_gy = 0.f;
  w.lane = lane;
  w.last_op = last_op;
  w.fusemask = fusemask;
  w.is_global = is_global;
  w.sums_env = (is_global && sums) ? sums + (size_t)e * 8 : nullptr;
  if (TRACK) {
    area_lds_clear(s_area);
    w.inv_gx = __builtin_amdgcn_rcpf((float)gx);
    w.inv_gy = __builtin_amdgcn_rcpf((float)gy);
    __syncthreads();
  }
  WaveAcc acc;
  acc.exceed = false; acc.a1 = acc.aD = 0.0; acc.cells = acc.opcells = 0;

  // slabs that intersect my rows [r0, r1): the table is sorted, the first one is found with one ballot
  int s = __popcll(__ballot(lane < nslabs && st.xb <= r0));
  for (; s < nslabs; ++s) {
    const int sxa = lane_i(st.xa, s), sxb = lane_i(st.xb, s);
    if (sxa >= r1) break;
    const int x = max(sxa, r0), xe = min(sxb, r1);
    const unsigned active = (unsigned)lane_i(st.active, s);
    if (x >= xe) continue;
    if (shift) {
      // every message of the plan takes part in every row; lanes span the whole width
      const int na = __popc(fusemask);
      if (na == 0) break;
      if (NAMAX <= 6 || na <= 6) walk_slab<VEC, TRACK, true, (NAMAX < 6 ? NAMAX : 6), 1>(w, t, acc, fusemask, active, x, xe, 0, gy);
      else walk_slab<VEC, TRACK, true, NAMAX, 1>(w, t, acc, fusemask, active, x, xe, 0, gy);
    } else if (active != 0) {
      const int hull = lane_i(st.hull, s);
      const int ya = hull & 0xFFFF, yb = hull >> 16;
      // the row loop compiled for this many ops
      const int na = __popc(active);
      if (na == 1) walk_slab<VEC, TRACK, false, 1, IPPM_FU_SMALL>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (na == 2) walk_slab<VEC, TRACK, false, 2, IPPM_FU_SMALL>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (na == 3) walk_slab<VEC, TRACK, false, 3, IPPM_FU_MID>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (na == 4) walk_slab<VEC, TRACK, false, 4, IPPM_FU_MID>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (NAMAX <= 6 || na <= 6) walk_slab<VEC, TRACK, false, (NAMAX < 6 ? NAMAX : 6), IPPM_FU_BIG>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (NAMAX <= 10 || na <= 10) walk_slab<VEC, TRACK, false, (NAMAX < 10 ? NAMAX : 10), IPPM_FU_BIG>(w, t, acc, active, 0u, x, xe, ya, yb);
      else walk_slab<VEC, TRACK, false, NAMAX, IPPM_FU_BIG>(w, t, acc, active, 0u, x, xe, ya, yb);
    }
  }
  if (__any(acc.exceed) && lane == 0) ws[wbase + WS_FLAG_A] = 1;
  // wave reduction of the reward terms and work counters: one atomic per wavefront and quantity
  {
    const float fc = ippm_wave_sum((float)acc.cells), fo = ippm_wave_sum((float)acc.opcells);
    double a1 = acc.a1, aD = acc.aD;
    if (is_global) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o, 64); aD += __shfl_xor(aD, o, 64); }
    }
    if (lane < 3) {
      const double v = lane == 0 ? a1 : (lane == 1 ? aD : aD - a1);
      if (is_global && sums && v != 0.0) atomicAdd(&sums[(size_t)e * 8 + SUM_ACC1 + lane], v);
    } else if (lane < 5 && counters) {
      const float v = lane == 3 ? fc : fo;
      if (v > 0.f) atomicAdd(&counters[(cslot & (IPPM_COUNTER_SLOTS - 1)) * 8 + (is_global ? 3 : 1) + (lane - 3)], (unsigned long long)v);
    }
    if (TRACK) {
      __syncthreads();
      area_lds_commit(s_area, area + (size_t)(e * (n + 1) + slot) * IPPM_FEAT * IPPM_FEAT);
      __syncthreads();
    }
  }
}

// Workgroup = one wavefront.  Two ways to hand out the work items:
//   work == NULL : the grid enumerates (map, run) pairs, most of which turn out empty (maps that received nothing, runs
//                  beyond the hull) -- the stand-alone entry points use this;
//   work != NULL : `work` = per-env {count, items...} written by the plan kernel holds exactly the non-empty items
//                  (item = map << 8 | run); the launch is a fixed number of wavefronts, a few per env, that stride over their
//                  env's items, so hardly a slot is spent on an empty item and no "round" of short-lived workgroups has to
//                  drain before the next.
#ifndef IPPM_FUSE_WAVES
#define IPPM_FUSE_WAVES 4
#endif
template <int VEC, bool TRACK, int NAMAX, bool SHIFT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(IPPM_FUSE_WAVES, 8)))
k_fuse_rows(const ippm_config* __restrict__ c, float* __restrict__ local, float* __restrict__ global,
            const uint8_t* __restrict__ code, const int32_t* __restrict__ plan_ro, int32_t* __restrict__ ws,
            double* __restrict__ sums, double* __restrict__ area, unsigned long long* __restrict__ counters,
            const int32_t* __restrict__ work, int wave_rows, int chunks, int min_ops, int local_units, int agent_sel,
            int n_envs_total, int env_cap, int tl) {
  __shared__ double s_area[TRACK ? (IPPM_FEAT + 1) * IPPM_AREA_LD : 1];
  const int n = c->n_agents;
  if (work) {  // gridDim.x is a multiple of the env count: wavefront b serves an env of row b / E, taking every (gridDim.x / E)-th item
    // (which env: b % E, moved on by one place every fourth row for even batches of 8 or more -- consecutive workgroups go to
    //  consecutive XCDs, and with E a multiple of 8 plain b % E would run each env's whole list on XCD env % 8; fuse_tiles.hip)
    const int row = blockIdx.x / n_envs_total;
    int env = blockIdx.x - row * n_envs_total;
    if (n_envs_total >= 8 && (n_envs_total & 1) == 0) {
      env += (row >> 2) & 7;
      env -= env >= n_envs_total ? n_envs_total : 0;
    }
    const int count = (work[env] & IPPM_WORK_TILED) ? 0 : work[env];  // a list in the tile form is fuse_tiles.hip's
    const int32_t* items = work + n_envs_total + (size_t)env * env_cap;
    const int step = gridDim.x / n_envs_total;
    for (int i = row; i < count; i += step) {
      const int item = items[i];
      const int m = item >> 8;
      fuse_item<VEC, TRACK, NAMAX, SHIFT>(c, local, global, code, plan_ro, ws, sums, area, counters, s_area, wave_rows, min_ops, n_envs_total,
                                   m / (n + 1), m % (n + 1), item & 0xFF, blockIdx.x, tl);
    }
    return;
  }
  const int unit = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  // units [0, local_units) are local maps ((e,i), or (e, agent_sel)), the rest global maps
  int e, slot;
  if (unit >= local_units) { e = unit - local_units; slot = n; }
  else if (agent_sel >= 0) { e = unit; slot = agent_sel; }
  else { e = unit / n; slot = unit % n; }
  fuse_item<VEC, TRACK, NAMAX, SHIFT>(c, local, global, code, plan_ro, ws, sums, area, counters, s_area, wave_rows, min_ops, n_envs_total, e, slot,
                               chunk, blockIdx.x, tl);
}

__global__ void k_reward_finalize(const ippm_config* __restrict__ c, double* __restrict__ sums,
                                  float* __restrict__ reward, int n_envs) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_envs) return;
  ippm_reward_finalize_env(c, sums, reward, e);
}

// ======================================================================================================
// host API
// ======================================================================================================
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }

// Rows per work item; the plan kernel (work list) and the fusion must agree, both derive it from (config, n_envs).
// Sized so that a step yields roughly three items per wavefront slot of the chip (256 CUs x 4 SIMDs x 4 waves): about half
// of the maps take part in a fusion and their op hulls span ~60 % of the grid's rows.  Fewer envs -> shorter runs -> the
// same parallelism with shorter per-wavefront latency chains.
int ippm_fuse_wave_rows(const ippm_ctx* ctx, int n_envs) {
  const int gx = ctx->cfg.grid_x;
  const int forced = ctx->knob_wave_rows;  // IPPM_FUSE_WAVE_ROWS, read at ippm_ctx_create
  int rows = forced;
  if (rows <= 0) {
    const double est_rows = 0.5 * (double)n_envs * (ctx->cfg.n_agents + 1) * 0.6 * gx;
    rows = 8;
    while (rows < 32 && est_rows / rows > 12288.0 * 1.5) rows *= 2;   // (64 rows per item: 5 % slower at 8 UAVs x 512^2)
  }
  return std::min(gx, std::max((gx + 255) / 256, rows));
}

// The kernel instantiation is chosen by the config's largest possible plan (<= 6 ops, <= 10, <= 18).
// local_units / global_units: how many local / global maps; work: the plan kernel's item list (NULL: enumerate).
static int launch_fuse(ippm_ctx* ctx, float* local, float* global, const uint8_t* code, int32_t* ws, double* sums, double* area,
                       const int32_t* work, int local_units, int global_units, int agent_sel, int n_envs_total, hipStream_t st) {
  const ippm_config& c = ctx->cfg;
  const int units = local_units + global_units;
  if (units <= 0) return 0;
  const int max_ops = c.n_agents + 1;  // local: 2 clamp-only ops + N-1 messages; global: 1 clamp-only op + N messages
  const int wave_rows = ippm_fuse_wave_rows(ctx, n_envs_total);
  const int chunks = (c.grid_x + wave_rows - 1) / wave_rows;
  // wavefronts of the persistent form: four rounds of the chip's 4096 slots (256 CUs x 4 SIMDs x 4 waves at 128 VGPRs); measured
  // 8192 / 12288 / 16384 / 24576 -> 122.6 / 118 / 113.5 / 116 us at config 2
  // ... and about three items per wavefront on larger grids / teams (8 UAVs x 512^2 x 1024 envs: 16 / 32 / 64 wavefronts per env
  // -> 1211 / 1173 / 1234 us; 4 UAVs x 1024^2: 16 / 64 -> 1577 / 1477 us)
  const int per_env = ((c.n_agents + 1) * chunks + 2) / 3;
  const int persist = std::max(64, ctx->knob_persist > 0 ? ctx->knob_persist : std::max(16384, per_env * n_envs_total));
  const int pgrid = n_envs_total * std::max(1, std::min(persist / std::max(n_envs_total, 1), (c.n_agents + 1) * chunks));
  dim3 grid(work ? (unsigned)pgrid : (unsigned)units * chunks), block(64);
#define IPPM_FUSE(V, T, NA, SH, MINOPS)                                                                                  \
  IPPM_LAUNCH(ctx, IPPM_T_FUSE, (k_fuse_rows<V, T, NA, SH>), grid, block, st, ctx->dcfg, local, global, code, ws, ws, sums, area, \
              ctx->dcounters, work, wave_rows, chunks, MINOPS, local_units, agent_sel, n_envs_total, ippm_work_env_cap(ctx, n_envs_total), ctx->tl)
#define IPPM_FUSE_ALL(V, T)                                  \
  do {                                                       \
    if (c.logit_prior != 0.f) {                              \
      IPPM_FUSE(V, T, 18, true, 1); /* slow path: one size */ \
      break;                                                 \
    }                                                        \
    /* ONE launch, compiled for the config's largest plan, so that every item is visited once (a launch per plan-size */ \
    /* class -- <= 6, 7..10, 11..18 ops, each skipping the others' items -- was 1.2x slower with 8 UAVs)              */ \
    if (max_ops <= 6) IPPM_FUSE(V, T, 6, false, 1);          \
    else if (max_ops <= 10) IPPM_FUSE(V, T, 10, false, 1);   \
    else IPPM_FUSE(V, T, 18, false, 1);                      \
  } while (0)
  static_assert(IPPM_MAX_OPS <= 18, "largest instantiation of k_fuse_rows");
  if (ctx->vec == 4) { if (area) IPPM_FUSE_ALL(4, true); else IPPM_FUSE_ALL(4, false); }
  else { if (area) IPPM_FUSE_ALL(1, true); else IPPM_FUSE_ALL(1, false); }
#undef IPPM_FUSE_ALL
#undef IPPM_FUSE
  IPPM_LAUNCH_CHECK("fuse_rows");
  return 0;
}

extern "C" int ippm_fuse_local(ippm_ctx* ctx, float* local, const uint8_t* code, const int32_t* rect, const int32_t* pos,
                               const uint8_t* comm, int32_t* ws, int32_t agent_sel, int32_t n_envs, void* stream) {
  if (!ctx || !local || !code || !rect || !pos || !comm || !ws) { ippm_set_error("ippm_fuse_local: null argument"); return -1; }
  if (agent_sel >= ctx->cfg.n_agents) { ippm_set_error("ippm_fuse_local: agent_sel out of range"); return -1; }
  if (int rc = ippm_launch_plan(ctx, rect, pos, comm, ws, 0, n_envs, agent_sel, S_(stream))) return rc;
  const int maps = agent_sel >= 0 ? n_envs : n_envs * ctx->cfg.n_agents;
  return launch_fuse(ctx, local, nullptr, code, ws, nullptr, nullptr, nullptr, maps, 0, agent_sel, n_envs, S_(stream));
}

extern "C" int ippm_comm_fuse_local(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const float* comm_range,
                                    const double* draws, uint8_t* comm, float* local, const uint8_t* code, const int32_t* rect,
                                    int32_t* ws, int32_t t, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !comm || !local || !code || !rect || !ws) { ippm_set_error("ippm_comm_fuse_local: null argument"); return -1; }
  if (!draws && !episode) { ippm_set_error("ippm_comm_fuse_local: Philox draws need the episode ids"); return -1; }
  if (int rc = ippm_plan_step(ctx, episode, const_cast<int32_t*>(pos), comm_range, draws, comm, rect, ws, t, IPPM_STEP_COMM, nullptr,
                              nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, n_envs, stream))
    return rc;
  return launch_fuse(ctx, local, nullptr, code, ws, nullptr, nullptr, nullptr, n_envs * ctx->cfg.n_agents, 0, -1, n_envs, S_(stream));
}

extern "C" int ippm_reward_finalize(ippm_ctx* ctx, double* sums, float* reward, int32_t n_envs, void* stream) {
  if (!ctx || !sums || !reward) { ippm_set_error("ippm_reward_finalize: null argument"); return -1; }
  if (n_envs <= 0) return 0;
  hipLaunchKernelGGL(k_reward_finalize, dim3(grid1(n_envs)), dim3(256), 0, S_(stream), ctx->dcfg, sums, reward, n_envs);
  IPPM_LAUNCH_CHECK("reward_finalize");
  return 0;
}

extern "C" int ippm_fuse_global_reward(ippm_ctx* ctx, float* global, const uint8_t* code, const int32_t* rect,
                                       const int32_t* pos, int32_t* ws, double* sums, float* reward, int32_t n_envs,
                                       void* stream) {
  if (!ctx || !global || !code || !rect || !pos || !ws || !sums || !reward) {
    ippm_set_error("ippm_fuse_global_reward: null argument");
    return -1;
  }
  if (int rc = ippm_launch_plan(ctx, rect, pos, nullptr, ws, 1, n_envs, -1, S_(stream))) return rc;
  if (int rc = launch_fuse(ctx, nullptr, global, code, ws, sums, nullptr, nullptr, 0, n_envs, -1, n_envs, S_(stream))) return rc;
  return ippm_reward_finalize(ctx, sums, reward, n_envs, stream);
}

extern "C" int ippm_fuse_step(ippm_ctx* ctx, float* local, float* global, const uint8_t* code, int32_t* ws, double* sums,
                              double* area, const int32_t* work, int32_t n_envs, void* stream) {
  if (!ctx || !local || !global || !code || !ws || !sums) { ippm_set_error("ippm_fuse_step: null argument"); return -1; }
  // the work list may be in the one-trip tile form (ippm_plan_step with IPPM_STEP_TILES): fuse_tiles.hip; a list of the other
  // form is skipped and counted by whichever kernel is handed it
  if (work && ctx->tiles && !ctx->knob_nowork && !ctx->knob_split)
    return ippm_launch_fuse_tiles(ctx, local, global, code, ws, sums, area, work, n_envs, S_(stream));
  if (ctx->knob_split) {  // measurement aid: K4 and K5 as two launches, so that a kernel trace shows them apart
    if (int rc = launch_fuse(ctx, local, global, code, ws, sums, area, nullptr, n_envs * ctx->cfg.n_agents, 0, -1, n_envs, S_(stream))) return rc;
    return launch_fuse(ctx, local, global, code, ws, sums, area, nullptr, 0, n_envs, -1, n_envs, S_(stream));
  }
  return launch_fuse(ctx, local, global, code, ws, sums, area, ctx->knob_nowork ? nullptr : work,
                     n_envs * ctx->cfg.n_agents, n_envs, -1, n_envs, S_(stream));
}
