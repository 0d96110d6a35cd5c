// Kernels that stream belief maps: K3 sense+update, K4/K5 fusion (+ information-gain reward), full-grid entropy,
// and the probability <-> log-odds conversions at the API boundary.
//
// Maps are patch-tiled (ippm_internal.h): a 128-byte line holds a 4-row x 8-column patch.  Lane geometry: 8 lanes cover
// one patch (lane = row-in-patch, half), a wavefront covers 8 consecutive patches of a patch row (1 KiB contiguous) or,
// for narrow rectangles, 4/2/1 patches of 2/4/8 patch rows.  Each lane handles four y-consecutive cells ("group") with one
// 16-byte access; the byte planes (truth, measurement codes, flips) ride along as one aligned 32-bit word per lane.
#include <algorithm>
#include <type_traits>

#include "ippm_internal.h"

constexpr int VEC = 4;  // cells per lane group

struct Strip {
  int pc0, npc;  // first patch column / number of patch columns of the rectangle
  int pr0, npr;  // first patch row / number of patch rows
  int shift;     // log2(patch columns per wavefront chunk)
  int ppr;       // patch columns per chunk (1, 2, 4 or 8)
  int spw;       // patch rows per wavefront = 8 / ppr
};
__device__ __forceinline__ Strip make_strip(int yu, int yd, int xl, int xr) {
  Strip s;
  s.pc0 = yu >> 3;
  s.npc = ((yd + 7) >> 3) - s.pc0;
  s.pr0 = xl >> 2;
  s.npr = ((xr + 3) >> 2) - s.pr0;
  const int m = min(s.npc, 8) - 1;
  s.shift = m <= 0 ? 0 : 32 - __clz(m);
  s.ppr = 1 << s.shift;
  s.spw = 8 >> s.shift;
  return s;
}
struct LanePos {
  int sub;   // which patch row of the wavefront's strip set
  int pcl;   // patch column inside the chunk
  int r4;    // row inside the patch
  int half;  // left / right four columns of the patch
};
__device__ __forceinline__ LanePos lane_pos(int lane, const Strip& s) {
  LanePos p;
  const int idx8 = lane >> 3;
  p.sub = idx8 >> s.shift;
  p.pcl = idx8 & (s.ppr - 1);
  p.r4 = (lane >> 1) & 3;
  p.half = lane & 1;
  return p;
}

struct Cells {
  float v[VEC];
};
__device__ __forceinline__ Cells load_cells(const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  Cells r;
  r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  return r;
}
__device__ __forceinline__ void store_cells(float* p, const Cells& r) {
  *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
}
__device__ __forceinline__ uint32_t load_word(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
__device__ __forceinline__ void store_word(uint8_t* p, uint32_t w) { *reinterpret_cast<uint32_t*>(p) = w; }

// ======================================================================================================
// probability <-> log-odds (row-major [n, gx, gy] outside, patch-tiled inside)
// ======================================================================================================
__global__ void k_logodds_to_prob(const ippm_config* __restrict__ c, const float* __restrict__ src, float* __restrict__ dst,
                                  int n_maps) {
  const int gx = c->grid_x, gy = c->grid_y, npc = ippm_gyp(c) >> 3;
  const size_t per = (size_t)ippm_gxp(c) * ippm_gyp(c);
  const size_t total = per * n_maps;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (size_t)gridDim.x * blockDim.x) {
    const size_t m = s / per, w = s - m * per;
    const int patch = (int)(w >> 5), x = (patch / npc) * 4 + (int)((w >> 3) & 3), y = (patch % npc) * 8 + (int)(w & 7);
    if (x < gx && y < gy) dst[(m * gx + x) * gy + y] = ippm_sigmoid(src[s]);
  }
}
__global__ void k_prob_to_logodds(const ippm_config* __restrict__ c, const float* __restrict__ src, float* __restrict__ dst,
                                  int n_maps) {
  const int gx = c->grid_x, gy = c->grid_y, npc = ippm_gyp(c) >> 3;
  const size_t per = (size_t)ippm_gxp(c) * ippm_gyp(c);
  const size_t total = per * n_maps;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (size_t)gridDim.x * blockDim.x) {
    const size_t m = s / per, w = s - m * per;
    const int patch = (int)(w >> 5), x = (patch / npc) * 4 + (int)((w >> 3) & 3), y = (patch % npc) * 8 + (int)(w & 7);
    float l = c->logit_prior;  // padding cells
    if (x < gx && y < gy) {
      const float p = src[(m * gx + x) * gy + y];
      l = __logf(p) - __logf(1.0f - p);  // p in {0,1} gives -inf/+inf: clamped on first use like the reference's clip
    }
    dst[s] = l;
  }
}

// ======================================================================================================
// K3: sense + Bayesian update of the agent's own footprint tile
// ======================================================================================================
template <int UNR>
__global__ void __launch_bounds__(256)
k_sense_update(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode, const int32_t* __restrict__ pos,
               const uint8_t* __restrict__ truth, float* __restrict__ local, const uint8_t* __restrict__ flips,
               uint8_t* __restrict__ code, int32_t* __restrict__ rect_out, int32_t* __restrict__ ws,
               unsigned long long* __restrict__ counters, int stage, int agent_sel, int split) {
  const int n = c->n_agents;
  const int tile = blockIdx.x / split, part = blockIdx.x % split;  // (tile, row part) flattened: grid.x has no 65535 limit
  int e, i;
  if (agent_sel >= 0) { e = tile; i = agent_sel; }
  else { e = tile / n; i = tile % n; }
  const int gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  const int gyp = ippm_gyp(c), npcT = gyp >> 3;
  const size_t map_stride = (size_t)ippm_gxp(c) * gyp;
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
  const Strip st = make_strip(yu, yd, xl, xr);
  const int per = (st.npr + split - 1) / split;
  const int a0 = part * per, a1 = min(st.npr, a0 + per);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const LanePos lp = lane_pos(lane, st);
  float* map = local + (size_t)(e * n + i) * map_stride;
  const uint8_t* tr = truth + (size_t)e * gx * gyp;
  uint8_t* cd = code + (size_t)(e * n + i) * S * S;
  const uint8_t* fl = flips ? flips + (size_t)(e * n + i) * S * S : nullptr;
  const int64_t ep = episode ? episode[e] : 0;
  const uint32_t sw = ippm_stream_word((uint32_t)i, (uint32_t)stage, IPPM_DOMAIN_FLIP);
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
  const int tile_y0 = yu & ~7;
  const int stride = 4 * st.spw;  // patch rows advanced per iteration of the workgroup
  bool exceed = false;
  for (int pcc = lp.pcl; pcc < st.npc; pcc += st.ppr) {
    const int y = (st.pc0 + pcc) * 8 + lp.half * 4;
    unsigned inm = 0;  // which of my four cells lie inside the footprint's columns
#pragma unroll
    for (int q = 0; q < VEC; ++q) inm |= ((unsigned)(y + q - yu) < (unsigned)w) ? (1u << q) : 0u;
    if (inm == 0) continue;
    for (int prr = a0 + wv * st.spw + lp.sub; prr < a1; prr += stride * UNR) {
      // UNR independent rows per lane: all their loads are in flight before the first use
      Cells m[UNR];
      uint32_t tw[UNR], fw[UNR];
      bool on[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int x = (st.pr0 + prr + u * stride) * 4 + lp.r4;
        on[u] = (prr + u * stride) < a1 && (unsigned)(x - xl) < (unsigned)h;
        tw[u] = 0; fw[u] = 0;
        if (on[u]) {
          m[u] = load_cells(map + ippm_cell_off(x, y, npcT));
          tw[u] = load_word(tr + (size_t)x * gyp + y);
          if (fl) fw[u] = load_word(fl + (size_t)(x - xl) * S + (y - tile_y0));
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (!on[u]) continue;
        const int x = (st.pr0 + prr + u * stride) * 4 + lp.r4;
        const size_t lin = (size_t)x * gy + y;  // Philox counters are defined on the row-major cell index
        Philox4 ph;
        uint32_t rnd[VEC];
        if (!fl) {
          if ((gy & 3) == 0) {
            ph = ippm_philox((uint32_t)(lin >> 2), (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1);
#pragma unroll
            for (int q = 0; q < VEC; ++q) rnd[q] = ph.v[q];
          } else {  // grids not a multiple of 4 wide: the four cells straddle two counters
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
              const Philox4 p1 = ippm_philox((uint32_t)((lin + q) >> 2), (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1);
              rnd[q] = p1.v[(lin + q) & 3];
            }
          }
        }
        uint32_t cw = 0;
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          // branch-free: cells of an edge group that lie outside the footprint keep their value
          const bool in = (inm >> q) & 1u;
          const uint32_t flip = fl ? ((fw[u] >> (8 * q)) & 1u) : (rnd[q] < thr ? 1u : 0u);
          const uint32_t obs = ((tw[u] >> (8 * q)) & 1u) ^ flip;
          // mappings.py:109-124 in log-odds: clip the prior belief, add the measurement's log-odds
          const float l = ippm_clampl(m[u].v[q], lc) + (obs ? lm1 : lm0);
          exceed |= in & (fabsf(l) > lc);
          m[u].v[q] = in ? l : m[u].v[q];
          cw |= (in ? obs : 0u) << (8 * q);
        }
        store_cells(map + ippm_cell_off(x, y, npcT), m[u]);
        store_word(cd + (size_t)(x - xl) * S + (y - tile_y0), cw);
      }
    }
  }
  if (ws && __any(exceed) && lane == 0) ws[(size_t)(e * (n + 1) + i) * IPPM_WS_WORDS + WS_FLAG_S] = 1;
  if (counters && part == 0 && threadIdx.x == 0)
    atomicAdd(&counters[(tile & (IPPM_COUNTER_SLOTS - 1)) * 8 + 0], (unsigned long long)h * w);
}

// ======================================================================================================
// K4 / K5: apply the planned ops to a map, each touched cell read once and written once.
// REWARD: also accumulate the information-gain reward terms of K5 (utils/reward.py:68-82).
//
// Work decomposition: one workgroup column per (map, op).  The workgroups of op k walk the patches of ITS rectangle and
// own every 4-cell group that no later op touches; an owned group gets the complete ordered chain of all ops covering
// each of its cells.  Every group of the union is therefore read and written exactly once, by exactly one workgroup,
// whatever the overlap pattern.  A lane keeps its column group while it walks down the rows, so everything that depends
// on columns only is folded into bit masks once per column chunk.  NK = ops held in registers (scalar loads, unrolled).
// ======================================================================================================
struct OpRec {
  int info;  // type | src << 8 | alt << 16
  int yu, yd, xl, xr;
};

template <bool REWARD, int NK>
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
  int kyu = 0, kyd = 0, kxl = 0, kxr = 0, kinfo = 0;
#pragma unroll
  for (int o = 0; o < NK; ++o)
    if (o == k) { kyu = op[o].yu; kyd = op[o].yd; kxl = op[o].xl; kxr = op[o].xr; kinfo = op[o].info; }
  const int S = c->tile_stride;
  const int gyp = ippm_gyp(c), npcT = gyp >> 3;
  const size_t map_stride = (size_t)ippm_gxp(c) * gyp;
  const bool k_is_last = hdr[PL_LAST] == k;
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  const Strip st = make_strip(kyu, kyd, kxl, kxr);
  const int per = (st.npr + split - 1) / split;
  const int a0 = part * per, a1 = min(st.npr, a0 + per);
  const int x_lo = max(kxl, (st.pr0 + a0) * 4), x_hi = min(kxr, (st.pr0 + a1) * 4);  // rows this workgroup walks
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const LanePos lp = lane_pos(lane, st);
  float* map = maps + (size_t)m * map_stride;
  const uint8_t* code_e = code + (size_t)e * n * S * S;
  bool exceed = false;
  float a1s = 0.f, aD = 0.f, aT = 0.f;
  unsigned cells = 0, opcells = 0;
  const int h = kxr - kxl, wdt = kyd - kyu;
  // Does any other op's rectangle intersect the rows/columns this workgroup walks?  If not (the common case) every
  // group is covered by op k alone: a short branch-free loop does the job.
  bool alone = true;
  unsigned hitmask = 0;  // ops (including k) that can touch a group this workgroup walks: all others are skipped wholesale
#pragma unroll
  for (int o = 0; o < NK; ++o) {
    // column ranges widened to whole 4-cell groups: ownership is decided per group, so two rectangles that merely share
    // an edge group already interact
    const bool hit = op[o].xl < x_hi && op[o].xr > x_lo && (op[o].yu & ~(VEC - 1)) < ((kyd + VEC - 1) & ~(VEC - 1)) &&
                     ((op[o].yd + VEC - 1) & ~(VEC - 1)) > (kyu & ~(VEC - 1));
    alone &= (o == k) || !hit;
    hitmask |= (hit || o == k) ? (1u << o) : 0u;
  }
  if (alone) {
    const bool isf = (kinfo & 0xFF) != 0;
    const int alt = (kinfo >> 16) & 0xFF;
    const float lm0 = isf ? c->logit_meas[alt][0] : 0.f, lm1 = isf ? c->logit_meas[alt][1] : 0.f;
    const uint8_t* ctile = code_e + (size_t)((kinfo >> 8) & 0xFF) * S * S - (kyu & ~7);
    for (int pcc = lp.pcl; pcc < st.npc; pcc += st.ppr) {
      const int y = (st.pc0 + pcc) * 8 + lp.half * 4;
      unsigned inm = 0;
#pragma unroll
      for (int q = 0; q < VEC; ++q) inm |= ((unsigned)(y + q - kyu) < (unsigned)wdt) ? (1u << q) : 0u;
      if (inm == 0) continue;
      for (int prr = a0 + wv * st.spw + lp.sub; prr < a1; prr += 4 * st.spw) {
        const int x = (st.pr0 + prr) * 4 + lp.r4;
        if ((unsigned)(x - kxl) >= (unsigned)h) continue;
        const size_t off = ippm_cell_off(x, y, npcT);
        Cells mv = load_cells(map + off);
        uint32_t cw = 0;
        if (isf) cw = load_word(ctile + (size_t)(x - kxl) * S + y);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const float b = mv.v[q];
          float a = ippm_clampl(b, lc) + (((cw >> (8 * q)) & 1u) ? lm1 : lm0);
          a = k_is_last ? a : ippm_clampl(a, lc);
          const bool in = (inm >> q) & 1u;
          a = in ? a : b;
          exceed |= fabsf(a) > lc && in;
          mv.v[q] = a;
          if (REWARD) {
            const float sel = (in && isf) ? 1.f : 0.f;
            const float wa = ippm_weight_l(a, wt), wb = ippm_weight_l(b, wt);
            const float hb = ippm_entropy_l(b, lc), ha = ippm_entropy_l(a, lc);
            a1s += sel * (wa * (hb - ha));
            aD += sel * ((wa - wb) * hb);
            aT += sel * (wa * ha - wb * hb);
          }
        }
        cells += __popc(inm);
        store_cells(map + off, mv);
      }
    }
    opcells = cells;
  } else {
    using Mask = typename std::conditional<(NK * VEC > 32), unsigned long long, unsigned>::type;
    static_assert(NK * VEC <= 64, "op masks are at most 64 bits");
    constexpr unsigned QM = (1u << VEC) - 1u;
    for (int pcc = lp.pcl; pcc < st.npc; pcc += st.ppr) {
      const int y = (st.pc0 + pcc) * 8 + lp.half * 4;
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
      if (((unsigned)(cmask >> (k * VEC)) & QM) == 0u) continue;  // my group lies outside op k's columns
      for (int prr = a0 + wv * st.spw + lp.sub; prr < a1; prr += 4 * st.spw) {
        const int x = (st.pr0 + prr) * 4 + lp.r4;
        if ((unsigned)(x - kxl) >= (unsigned)h) continue;
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
        const size_t off = ippm_cell_off(x, y, npcT);
        // issue every load of this group (map cells + the measurement codes of all covering ops) before any use
        Cells mv = load_cells(map + off);
        uint32_t cw[NK];
#pragma unroll
        for (int o = 0; o < NK; ++o) {
          cw[o] = 0;
          if (!((hitmask >> o) & 1u)) continue;
          if (o <= k && (op[o].info & 0xFF) && ((unsigned)(act >> (o * VEC)) & QM))
            cw[o] = load_word(code_e + (size_t)((op[o].info >> 8) & 0xFF) * S * S + (size_t)(x - op[o].xl) * S +
                              (y - (op[o].yu & ~7)));
        }
        const Cells old = mv;
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
            const float l = ippm_clampl(L[q], lc) + (((cw[o] >> (8 * q)) & 1u) ? lm1 : lm0);
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
            a1s += sel * (wa * (hb - ha));
            aD += sel * ((wa - wb) * hb);
            aT += sel * (wa * ha - wb * hb);
          }
        }
        store_cells(map + off, mv);
      }
    }
  }
  if (__any(exceed) && lane == 0) ws[wbase + WS_FLAG_A] = 1;
  // block reduction of the reward terms and work counters: one atomic per workgroup and quantity
  {
    const float fc = ippm_wave_sum((float)cells), fo = ippm_wave_sum((float)opcells);
    if (REWARD) { a1s = ippm_wave_sum(a1s); aD = ippm_wave_sum(aD); aT = ippm_wave_sum(aT); }
    if (lane == 0) { s_red[wv][0] = a1s; s_red[wv][1] = aD; s_red[wv][2] = aT; s_red[wv][3] = fc; s_red[wv][4] = fo; }
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
template <bool REWARD>
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
  const int S = c->tile_stride;
  const int gyp = ippm_gyp(c), npcT = gyp >> 3;
  const size_t map_stride = (size_t)ippm_gxp(c) * gyp;
  const int X0 = hdr[PL_X0], X1 = hdr[PL_X1], last_op = hdr[PL_LAST];
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  const Strip st = make_strip(hdr[PL_Y0], hdr[PL_Y1], X0, X1);
  const int per = (st.npr + split - 1) / split;
  const int a0 = part * per, a1 = min(st.npr, a0 + per);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const LanePos lp = lane_pos(lane, st);
  float* map = maps + (size_t)m * map_stride;
  const uint8_t* code_e = code + (size_t)e * n * S * S;
  bool exceed = false;
  float a1s = 0.f, aD = 0.f, aT = 0.f;
  unsigned cells = 0, opcells = 0;
  for (int pcc = lp.pcl; pcc < st.npc; pcc += st.ppr) {
    const int y = (st.pc0 + pcc) * 8 + lp.half * 4;
    for (int prr = a0 + wv * st.spw + lp.sub; prr < a1; prr += 4 * st.spw) {
      const int x = (st.pr0 + prr) * 4 + lp.r4;
      bool need = false;
      for (int o = 0; o < nops; ++o) {
        const int32_t* op = s_ops + o * OP_WORDS;
        need |= (x >= op[OP_XL] && x < op[OP_XR] && y + VEC > op[OP_YU] && y < op[OP_YD]);
      }
      if (!need) continue;
      const size_t off = ippm_cell_off(x, y, npcT);
      Cells mv = load_cells(map + off);
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
          cw = load_word(code_e + (size_t)op[OP_SRC] * S * S + (size_t)(x - op[OP_XL]) * S + (y - (op[OP_YU] & ~7)));
          lm0 = c->logit_meas[op[OP_ALT]][0];
          lm1 = c->logit_meas[op[OP_ALT]][1];
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          const int yy = y + q;
          if (yy >= op[OP_YU] && yy < op[OP_YD]) {
            L[q] = ippm_clampl(L[q], lc);
            if (op[OP_TYPE]) { L[q] += ((cw >> (8 * q)) & 1u) ? lm1 : lm0; fused[q] = true; }
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
          a1s += wa * (hb - ha);
          aD += (wa - wb) * hb;
          aT += wa * ha - wb * hb;
        }
      }
      store_cells(map + off, mv);
    }
  }
  if (__any(exceed) && lane == 0) w[WS_FLAG_A] = 1;
  {
    const float fc = ippm_wave_sum((float)cells), fo = ippm_wave_sum((float)opcells);
    if (REWARD) { a1s = ippm_wave_sum(a1s); aD = ippm_wave_sum(aD); aT = ippm_wave_sum(aT); }
    if (lane == 0) { s_red[wv][0] = a1s; s_red[wv][1] = aD; s_red[wv][2] = aT; s_red[wv][3] = fc; s_red[wv][4] = fo; }
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

// full-grid weighted entropy per map (initialisation of T, evaluation metrics); walks the tiled storage linearly
__global__ void __launch_bounds__(256)
k_weighted_entropy(const ippm_config* __restrict__ c, const float* __restrict__ maps, const uint8_t* __restrict__ truth,
                   double* __restrict__ out, int maps_per_truth) {
  const int m = blockIdx.y;
  const int gx = c->grid_x, gy = c->grid_y, gyp = ippm_gyp(c), npc = gyp >> 3;
  const size_t per = (size_t)ippm_gxp(c) * gyp;
  const float* p = maps + (size_t)m * per;
  const uint8_t* t = truth ? truth + (size_t)(m / maps_per_truth) * gx * gyp : nullptr;
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  float acc = 0.f;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < per; s += (size_t)gridDim.x * blockDim.x) {
    const int patch = (int)(s >> 5), x = (patch / npc) * 4 + (int)((s >> 3) & 3), y = (patch % npc) * 8 + (int)(s & 7);
    if (x >= gx || y >= gy) continue;
    const float v = p[s];
    const float wgt = t ? (float)t[(size_t)x * gyp + y] : ippm_weight_l(v, wt);
    acc += wgt * ippm_entropy_l(v, lc);
  }
  acc = ippm_wave_sum(acc);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[m], (double)(sh[0] + sh[1] + sh[2] + sh[3]));
}

// ======================================================================================================
// host API
// ======================================================================================================
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }

extern "C" int ippm_logodds_to_prob(ippm_ctx* ctx, const float* src, float* dst, int64_t n_maps, void* stream) {
  if (!ctx || !src || !dst) { ippm_set_error("ippm_logodds_to_prob: null argument"); return -1; }
  const size_t total = (size_t)ippm_host_gxp(ctx->cfg) * ippm_host_gyp(ctx->cfg) * n_maps;
  hipLaunchKernelGGL(k_logodds_to_prob, dim3(std::min(8192, grid1(total))), dim3(256), 0, S_(stream), ctx->dcfg, src, dst, (int)n_maps);
  IPPM_LAUNCH_CHECK("logodds_to_prob");
  return 0;
}

extern "C" int ippm_prob_to_logodds(ippm_ctx* ctx, const float* src, float* dst, int64_t n_maps, void* stream) {
  if (!ctx || !src || !dst) { ippm_set_error("ippm_prob_to_logodds: null argument"); return -1; }
  const size_t total = (size_t)ippm_host_gxp(ctx->cfg) * ippm_host_gyp(ctx->cfg) * n_maps;
  hipLaunchKernelGGL(k_prob_to_logodds, dim3(std::min(8192, grid1(total))), dim3(256), 0, S_(stream), ctx->dcfg, src, dst, (int)n_maps);
  IPPM_LAUNCH_CHECK("prob_to_logodds");
  return 0;
}

extern "C" int ippm_sense_update(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const uint8_t* truth,
                                 float* local, const uint8_t* flips, uint8_t* code, int32_t* rect, int32_t* ws,
                                 int32_t stage, int32_t agent_sel, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !truth || !local || !code || !rect) { ippm_set_error("ippm_sense_update: null argument"); return -1; }
  if (!flips && !episode) { ippm_set_error("ippm_sense_update: Philox flips need the episode ids"); return -1; }
  if (agent_sel >= ctx->cfg.n_agents) { ippm_set_error("ippm_sense_update: agent_sel out of range"); return -1; }
  const int maps = agent_sel >= 0 ? n_envs : n_envs * ctx->cfg.n_agents;
  const int split = std::max(1, ippm_env_int("IPPM_SPLIT_K3", 2));
  dim3 grid((unsigned)maps * split), block(256);
  const int unr = ippm_env_int("IPPM_UNROLL_K3", 2);
#define IPPM_K3_LAUNCH(U)                                                                                                 \
  hipLaunchKernelGGL((k_sense_update<U>), grid, block, 0, S_(stream), ctx->dcfg, episode, pos, truth, local, flips, code, rect, \
                     ws, ctx->dcounters, stage, agent_sel, split)
  if (unr >= 4) IPPM_K3_LAUNCH(4);
  else if (unr >= 2) IPPM_K3_LAUNCH(2);
  else IPPM_K3_LAUNCH(1);
#undef IPPM_K3_LAUNCH
  IPPM_LAUNCH_CHECK("sense_update");
  return 0;
}

// The three instantiations share one plan: <= 6 ops and 7..10 ops take the register paths (workgroup column per op),
// larger plans the generic path.  Each launch returns immediately for plans it does not own.
template <bool REWARD>
static void launch_apply_t(ippm_ctx* ctx, float* maps, const uint8_t* code, int32_t* ws, double* sums, int n_maps, int split,
                           hipStream_t st, int agent_sel) {
  const int max_ops = ctx->cfg.n_agents + 1;
  dim3 block(256);
  hipLaunchKernelGGL((k_apply_ops<REWARD, 6>), dim3((unsigned)n_maps * split, std::min(max_ops, 6)), block, 0, st, ctx->dcfg, maps,
                     code, ws, ws, sums, ctx->dcounters, split, 1, agent_sel);
  if (max_ops > 6)
    hipLaunchKernelGGL((k_apply_ops<REWARD, 10>), dim3((unsigned)n_maps * split, std::min(max_ops, 10)), block, 0, st, ctx->dcfg,
                       maps, code, ws, ws, sums, ctx->dcounters, split, 7, agent_sel);
  if (max_ops > 10)
    hipLaunchKernelGGL((k_apply_ops_generic<REWARD>), dim3((unsigned)n_maps * 8), block, 0, st, ctx->dcfg, maps, code, ws, sums,
                       ctx->dcounters, 8, 11, agent_sel);
}

void ippm_launch_apply(bool reward, ippm_ctx* ctx, float* maps, const uint8_t* code, int32_t* ws, double* sums, int n_maps,
                       int split, hipStream_t st, int agent_sel) {
  if (reward) launch_apply_t<true>(ctx, maps, code, ws, sums, n_maps, split, st, agent_sel);
  else launch_apply_t<false>(ctx, maps, code, ws, sums, n_maps, split, st, agent_sel);
}

extern "C" int ippm_weighted_entropy(ippm_ctx* ctx, const float* maps, const uint8_t* truth, int32_t maps_per_truth,
                                     double* out, int32_t n_maps, void* stream) {
  if (!ctx || !maps || !out) { ippm_set_error("ippm_weighted_entropy: null argument"); return -1; }
  IPPM_HIP(hipMemsetAsync(out, 0, sizeof(double) * n_maps, S_(stream)));
  const size_t cells = (size_t)ippm_host_gxp(ctx->cfg) * ippm_host_gyp(ctx->cfg);
  hipLaunchKernelGGL(k_weighted_entropy, dim3(std::min(32, grid1(cells)), n_maps), dim3(256), 0, S_(stream), ctx->dcfg, maps, truth,
                     out, maps_per_truth > 0 ? maps_per_truth : 1);
  IPPM_LAUNCH_CHECK("weighted_entropy");
  return 0;
}
