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

#include "ippm_tiles.h"

// empty asm: makes a loop-invariant value look loop-variant to the optimiser (no instruction is emitted)
#define IPPM_OPAQUE(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ int lane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

struct OpTable {   // lane o holds op o of the plan (zeros beyond the plan: an empty rectangle that never covers anything)
  int yu, yd, xl, xr, src;
  float lm0, lm1;  // log-odds of the two measurement values of a fuse op (0, 0 for a clamp-only op)
};

typedef unsigned ippm_u4 __attribute__((ext_vector_type(4)));
// raw buffer resources (gfx9 family descriptor word 3 = 0x00020000): uniform base + per-lane 32-bit byte offset, one VALU
// per address instead of a 64-bit multiply-add chain, and loads past the end return 0 instead of faulting
#define IPPM_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)

struct WaveCtx {   // what a wave needs while it walks its rows
  __amdgpu_buffer_rsrc_t map;    // this map: gx * gy floats
  __amdgpu_buffer_rsrc_t code;   // the whole code tensor (lanes outside an op's columns may point anywhere inside it)
  double* s_area;
  int code_env;                  // byte offset of this env's tiles in `code`
  int TB;
  int gx, gy, S, row_bytes;
  float lc, wt, lp, inv_gx, inv_gy;
  int lane;
  int last_op;
  unsigned fusemask;
  bool is_global;
};

struct WaveAcc {
  bool exceed;
  float a1, aD, aT;
  unsigned cells, opcells;
};

template <int VEC>
__device__ __forceinline__ CellVec<VEC> buf_load_cells(__amdgpu_buffer_rsrc_t r, int off) {
  CellVec<VEC> c;
  if (VEC == 4) {
    const ippm_u4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    c.v[0] = __uint_as_float(t.x); c.v[1 % VEC] = __uint_as_float(t.y); c.v[2 % VEC] = __uint_as_float(t.z); c.v[3 % VEC] = __uint_as_float(t.w);
  } else {
    c.v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  }
  return c;
}
template <int VEC>
__device__ __forceinline__ void buf_store_cells(__amdgpu_buffer_rsrc_t r, int off, const CellVec<VEC>& c) {
  if (VEC == 4) {
    ippm_u4 t;
    t.x = __float_as_uint(c.v[0]); t.y = __float_as_uint(c.v[1 % VEC]); t.z = __float_as_uint(c.v[2 % VEC]); t.w = __float_as_uint(c.v[3 % VEC]);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c.v[0]), r, off, 0, 0);
  }
}

// sigmoid(a) - sigmoid(b) = (e_b - e_a) / ((1 + e_a)(1 + e_b)), e = exp(-L): three transcendentals instead of four, exact
// zero for a == b; |L| is capped at 80 so that the product stays finite (sigmoid is saturated to 0 / 1 in float long before)
__device__ __forceinline__ float sigmoid_diff(float a, float b) {
  const float ea = __expf(-fminf(fmaxf(a, -80.f), 80.f)), eb = __expf(-fminf(fmaxf(b, -80.f), 80.f));
  return (eb - ea) * __builtin_amdgcn_rcpf((1.0f + ea) * (1.0f + eb));
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
  float lpk[NA];       // SHIFT: logit(prior) for the slots that are messages
  unsigned fm = 0;     // slots that carry a measurement
  int keep_slot = -1;  // slot of the plan's last op: its outputs stay unclamped
  // Slots beyond the set come FIRST: an empty slot still clips (like every op of the reference), which is a no-op ahead of
  // the first real op but would wrongly clip the last op's outputs behind it.
  unsigned rem = set;
  const int pad = NA - __popc(set);
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const int idx = k < pad ? 63 : __ffs(rem) - 1;
    rem = k < pad ? rem : (rem & (rem - 1u));
    yu[k] = lane_i(t.yu, idx); yd[k] = lane_i(t.yd, idx);
    lm0[k] = lane_f(t.lm0, idx); lm1[k] = lane_f(t.lm1, idx);
    const bool isf = idx < 32 && ((w.fusemask >> idx) & 1u);
    const bool rin = !SHIFT || (idx < 32 && ((rowin >> idx) & 1u));
    if (SHIFT && !rin) { yu[k] = 0; yd[k] = 0; }  // a message whose footprint misses these rows: shift only
    fm |= isf ? (1u << k) : 0u;
    lpk[k] = SHIFT && isf ? w.lp : 0.f;
    // byte of cell group (row, y) in the op's tile: tile base + (row - xl) * row_bytes + (y - (yu & ~3)) / VEC'
    const int y0 = yu[k] & ~3;
    cshift[k] = isf && rin ? w.code_env + lane_i(t.src, idx) * w.TB - lane_i(t.xl, idx) * w.row_bytes - (VEC == 4 ? (y0 >> 2) : y0)
                           : 0x7F000000;
    keep_slot = (idx == w.last_op) ? k : keep_slot;
  }
  const RowGeom g = make_geom<VEC>(ya, yb);
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
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      unsigned mq = 0;
#pragma unroll
      for (int q = 0; q < VEC; ++q) mq |= ((unsigned)(y + q - yu[k]) < (unsigned)(yd[k] - yu[k])) ? (1u << q) : 0u;
      cm[k] = mq;
      const unsigned inm = SHIFT ? (((fm >> k) & 1u) ? QM : 0u) : mq;
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
    for (int row0 = rs; row0 < re; row0 += FU) {
      // issue every load of FU rows (map cells + one measurement-code byte per slot) before any use
      CellVec<VEC> mvu[FU];
      uint32_t cwu[FU][NA];
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const int row = min(row0 + u, re - 1);  // a lane past its block's end re-reads the last row and writes nothing
        mvu[u] = buf_load_cells<VEC>(w.map, row * gybyte + ybyte);
        const int rowoff = row * w.row_bytes + ycode;
#pragma unroll
        for (int k = 0; k < NA; ++k) cwu[u][k] = __builtin_amdgcn_raw_buffer_load_b8(w.code, rowoff + cshift[k], 0, 0);
      }
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const int row = row0 + u;
        if (row >= re) continue;
        CellVec<VEC>& mv = mvu[u];
        float L[VEC], bsave[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) { L[q] = mv.v[q]; bsave[q] = mv.v[q]; }
        // Ordered clamp/add chain (mappings.py:80-124 in log-odds): every op of the reference clips its input over the WHOLE
        // grid (mappings.py:110-111), then adds the measurement's log-odds inside its footprint.  So every cell this lane
        // holds may be clipped at every op, covered or not (for an uncovered cell that is the reference's own full-grid
        // clip; the deferred-clamp plan guarantees it is a no-op there); only the addend is masked to the footprint.
#pragma unroll
        for (int k = 0; k < NA; ++k) {
          const uint32_t cw = cwu[u][k];
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            const float lm = ippm_masked(ippm_bitmask(cm[k], q), ippm_blend(ippm_bitmask(cw, q), lm1[k], lm0[k]));
            L[q] = ippm_clampl(L[q], w.lc) + (SHIFT ? lm - lpk[k] : lm);
          }
        }
        // outputs of the plan's last op stay unclamped (its rectangle is remembered as possibly out of range); every other
        // cell was clipped again by a later op of the reference
        float d[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const float a = ippm_blend(ippm_bitmask(keepm, q), L[q], ippm_clampl(L[q], w.lc));
          amax = fmaxf(amax, fabsf(a));
          mv.v[q] = a;
          if (TRACK) d[q] = sigmoid_diff(a, bsave[q]);
        }
        buf_store_cells<VEC>(w.map, row * gybyte + ybyte, mv);
        if (TRACK) area_row<VEC>(acc, w.s_area, ac, row, w.gx, w.inv_gx, d);
        if (w.is_global) {
          // information-gain terms (utils/reward.py:68-82); a cell that received no measurement contributes exact zeros
          // (same weight, and the entropy clips its argument)
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            const float b = bsave[q], a = mv.v[q];
            const float wa = ippm_weight_l(a, w.wt), wb = ippm_weight_l(b, w.wt);
            const float hb = ippm_entropy_l(b, w.lc), ha = ippm_entropy_l(a, w.lc);
            acc_out.a1 += wa * (hb - ha);
            acc_out.aD += (wa - wb) * hb;
            acc_out.aT += wa * ha - wb * hb;
          }
        }
      }
    }
    acc_out.exceed |= amax > w.lc;
    const unsigned nrows = (unsigned)max(re - rs, 0);
    acc_out.cells += nrows * __popc(touched);
    acc_out.opcells += nrows * ops_here;
    if (TRACK) acc.flush(w.s_area, ac.cb);
  }
}

// NAMAX = plan-size class of the launch (6 / 10 / 18): the row loops compiled in are those for <= NAMAX active ops
template <int VEC, bool TRACK, int NAMAX>
__global__ void __launch_bounds__(256)
k_fuse_rows(const ippm_config* __restrict__ c, float* __restrict__ local, float* __restrict__ global,
            const uint8_t* __restrict__ code, const int32_t* __restrict__ plan_ro, int32_t* __restrict__ ws,
            double* __restrict__ sums, double* __restrict__ area, unsigned long long* __restrict__ counters,
            int wave_rows, int chunks, int min_ops, int local_units, int agent_sel, int n_envs_total) {
  const int n = c->n_agents;
  const int unit = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  // units [0, local_units) are local maps ((e,i), or (e, agent_sel)), the rest global maps
  const bool is_global = unit >= local_units;
  int e, slot;
  if (is_global) { e = unit - local_units; slot = n; }
  else if (agent_sel >= 0) { e = unit; slot = agent_sel; }
  else { e = unit / n; slot = unit % n; }
  const size_t wbase = (size_t)(e * (n + 1) + slot) * IPPM_WS_WORDS;
  const int32_t* __restrict__ hdr = plan_ro + wbase + WS_PLAN;
  const int nops = hdr[PL_NOPS];
  if (nops > NAMAX || nops < min_ops) return;  // (another launch handles other plan sizes; 0 ops: nothing to do)
  const bool shift = c->logit_prior != 0.f;
  const int gx = c->grid_x, gy = c->grid_y;
  const int X0 = shift ? 0 : hdr[PL_X0], X1 = shift ? gx : hdr[PL_X1];
  if (X0 + chunk * 4 * wave_rows >= X1) return;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r0 = X0 + (chunk * 4 + wv) * wave_rows, r1 = min(X1, r0 + wave_rows);
  // op table: lane o loads op o
  OpTable t;
  unsigned fusemask;
  {
    const bool on = lane < nops;
    const int32_t* p = plan_ro + wbase + WS_OPS + (on ? lane : 0) * OP_WORDS;
    const bool isf = on && p[OP_TYPE] != 0;
    const int alt = on ? p[OP_ALT] : 0;
    t.yu = on ? p[OP_YU] : 0; t.yd = on ? p[OP_YD] : 0; t.xl = on ? p[OP_XL] : 0; t.xr = on ? p[OP_XR] : 0;
    t.src = isf ? p[OP_SRC] : 0;
    t.lm0 = isf ? c->logit_meas[alt][0] : 0.f;
    t.lm1 = isf ? c->logit_meas[alt][1] : 0.f;
    fusemask = (unsigned)__ballot(isf);
  }
  __shared__ double s_area[TRACK ? (IPPM_FEAT + 1) * IPPM_AREA_LD : 1];
  WaveCtx w;
  w.gx = gx; w.gy = gy; w.S = c->tile_stride;
  w.row_bytes = VEC == 4 ? (w.S >> 2) : w.S;
  w.TB = (int)ippm_tile_bytes(w.S, VEC);
  w.map = IPPM_RSRC(is_global ? global + (size_t)e * gx * gy : local + (size_t)(e * n + slot) * gx * gy, (size_t)gx * gy * 4);
  w.code = IPPM_RSRC(code, (size_t)n_envs_total * n * w.TB);
  w.code_env = e * n * w.TB;
  w.s_area = s_area;
  w.lc = c->logit_clip; w.wt = c->logit_weight_thr; w.lp = c->logit_prior;
  w.inv_gx = w.inv_gy = 0.f;
  w.lane = lane;
  w.last_op = hdr[PL_LAST];
  w.fusemask = fusemask;
  w.is_global = is_global;
  if (TRACK) {
    area_lds_clear(s_area);
    w.inv_gx = __builtin_amdgcn_rcpf((float)gx);
    w.inv_gy = __builtin_amdgcn_rcpf((float)gy);
    __syncthreads();
  }
  WaveAcc acc;
  acc.exceed = false; acc.a1 = acc.aD = acc.aT = 0.f; acc.cells = acc.opcells = 0;

  int x = r0;
  while (x < r1) {
    // ---- slab [x, xe): the ops covering row x, the first row where that set changes, the column hull (all uniform) ----
    unsigned active = 0;
    int xe = r1, ya = 1 << 30, yb = 0;
    for (int o = 0; o < nops; ++o) {
      const int oxl = lane_i(t.xl, o), oxr = lane_i(t.xr, o);
      const bool inr = oxl <= x && x < oxr;
      if (inr) {
        active |= 1u << o;
        xe = min(xe, oxr);
        ya = min(ya, lane_i(t.yu, o));
        yb = max(yb, lane_i(t.yd, o));
      } else if (oxl > x) {
        xe = min(xe, oxl);
      }
    }
    if (shift) {
      // every message of the plan takes part in every row; lanes span the whole width
      const int na = __popc(fusemask);
      if (na == 0) break;
      if (NAMAX <= 6 || na <= 6) walk_slab<VEC, TRACK, true, (NAMAX < 6 ? NAMAX : 6), 2>(w, t, acc, fusemask, active, x, xe, 0, gy);
      else walk_slab<VEC, TRACK, true, NAMAX, 2>(w, t, acc, fusemask, active, x, xe, 0, gy);
    } else if (active != 0) {
      // the row loop compiled for this many ops; the small ones keep 4 rows in flight per lane
      const int na = __popc(active);
      if (na == 1) walk_slab<VEC, TRACK, false, 1, 4>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (na == 2) walk_slab<VEC, TRACK, false, 2, 4>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (na == 3) walk_slab<VEC, TRACK, false, 3, 2>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (na == 4) walk_slab<VEC, TRACK, false, 4, 2>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (NAMAX <= 6 || na <= 6) walk_slab<VEC, TRACK, false, (NAMAX < 6 ? NAMAX : 6), 2>(w, t, acc, active, 0u, x, xe, ya, yb);
      else if (NAMAX <= 10 || na <= 10) walk_slab<VEC, TRACK, false, (NAMAX < 10 ? NAMAX : 10), 2>(w, t, acc, active, 0u, x, xe, ya, yb);
      else walk_slab<VEC, TRACK, false, NAMAX, 2>(w, t, acc, active, 0u, x, xe, ya, yb);
    }
    x = xe;
  }
  if (__any(acc.exceed) && lane == 0) ws[wbase + WS_FLAG_A] = 1;
  // wave reduction of the reward terms and work counters: one atomic per wavefront and quantity
  {
    const float fc = ippm_wave_sum((float)acc.cells), fo = ippm_wave_sum((float)acc.opcells);
    const float a1 = ippm_wave_sum(acc.a1), aD = ippm_wave_sum(acc.aD), aT = ippm_wave_sum(acc.aT);
    if (lane < 3) {
      const float v = lane == 0 ? a1 : (lane == 1 ? aD : aT);
      if (is_global && sums && v != 0.f) atomicAdd(&sums[(size_t)e * 8 + SUM_ACC1 + lane], (double)v);
    } else if (lane < 5 && counters) {
      const float v = lane == 3 ? fc : fo;
      const int cslot = (blockIdx.x * 4 + wv) & (IPPM_COUNTER_SLOTS - 1);
      if (v > 0.f) atomicAdd(&counters[cslot * 8 + (is_global ? 3 : 1) + (lane - 3)], (unsigned long long)v);
    }
    if (TRACK) {
      __syncthreads();
      area_lds_commit(s_area, area + (size_t)(e * (n + 1) + slot) * IPPM_FEAT * IPPM_FEAT);
    }
  }
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
static int env_int(const char* name, int dflt) {  // tuning knob; the default is the measured best on MI355X
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// One launch per plan-size class (<= 6 ops, 7..10, 11..18); each returns immediately for plans it does not own.
// local_units / global_units: how many local / global maps.
static int launch_fuse(ippm_ctx* ctx, float* local, float* global, const uint8_t* code, int32_t* ws, double* sums, double* area,
                       int local_units, int global_units, int agent_sel, int n_envs_total, hipStream_t st) {
  const ippm_config& c = ctx->cfg;
  const int units = local_units + global_units;
  if (units <= 0) return 0;
  const int max_ops = c.n_agents + 1;  // local: 2 clamp-only ops + N-1 messages; global: 1 clamp-only op + N messages
  const int wave_rows = std::max(1, env_int("IPPM_FUSE_WAVE_ROWS", 16));  // rows per wavefront; a workgroup covers 4x that
  const int chunks = (c.grid_x + 4 * wave_rows - 1) / (4 * wave_rows);
  dim3 grid((unsigned)units * chunks), block(256);
#define IPPM_FUSE(V, T, NA, MINOPS)                                                                                  \
  hipLaunchKernelGGL((k_fuse_rows<V, T, NA>), grid, block, 0, st, ctx->dcfg, local, global, code, ws, ws, sums, area, \
                     ctx->dcounters, wave_rows, chunks, MINOPS, local_units, agent_sel, n_envs_total)
#define IPPM_FUSE_ALL(V, T)                    \
  do {                                         \
    IPPM_FUSE(V, T, 6, 1);                     \
    if (max_ops > 6) IPPM_FUSE(V, T, 10, 7);   \
    if (max_ops > 10) IPPM_FUSE(V, T, 18, 11); \
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
  return launch_fuse(ctx, local, nullptr, code, ws, nullptr, nullptr, maps, 0, agent_sel, n_envs, S_(stream));
}

extern "C" int ippm_comm_fuse_local(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const float* comm_range,
                                    const double* draws, uint8_t* comm, float* local, const uint8_t* code, const int32_t* rect,
                                    int32_t* ws, int32_t t, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !comm || !local || !code || !rect || !ws) { ippm_set_error("ippm_comm_fuse_local: null argument"); return -1; }
  if (!draws && !episode) { ippm_set_error("ippm_comm_fuse_local: Philox draws need the episode ids"); return -1; }
  if (int rc = ippm_plan_step(ctx, episode, const_cast<int32_t*>(pos), comm_range, draws, comm, rect, ws, t, IPPM_STEP_COMM, nullptr,
                              nullptr, 0, nullptr, nullptr, nullptr, nullptr, n_envs, stream))
    return rc;
  return launch_fuse(ctx, local, nullptr, code, ws, nullptr, nullptr, n_envs * ctx->cfg.n_agents, 0, -1, n_envs, S_(stream));
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
  if (int rc = launch_fuse(ctx, nullptr, global, code, ws, sums, nullptr, 0, n_envs, -1, n_envs, S_(stream))) return rc;
  return ippm_reward_finalize(ctx, sums, reward, n_envs, stream);
}

extern "C" int ippm_fuse_step(ippm_ctx* ctx, float* local, float* global, const uint8_t* code, int32_t* ws, double* sums,
                              double* area, int32_t n_envs, void* stream) {
  if (!ctx || !local || !global || !code || !ws || !sums) { ippm_set_error("ippm_fuse_step: null argument"); return -1; }
  return launch_fuse(ctx, local, global, code, ws, sums, area, n_envs * ctx->cfg.n_agents, n_envs, -1, n_envs, S_(stream));
}
