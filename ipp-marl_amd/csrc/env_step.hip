// Environment-step kernels for gfx950 that touch maps outside the fusion: reset (device MT19937), K2 footprint,
// K3 sense + Bayes update, boundary conversions, full-grid entropy.  (Small-state kernels: step_small.hip; K4/K5: fuse.hip.)
//
// All of this is HBM-bound byte/float streaming over map tiles: no MFMA.  The layout rules are
//   - a map row (y contiguous) is covered by lanes holding 4 grid-aligned cells each (one 16-byte access),
//     truth is bit-packed, measurement codes / flips are one nibble per lane group: one byte load per lane each;
//   - narrow footprints pack several rows into one 64-lane wavefront (lanes-per-row = next pow2);
//   - every map cell is read and written at most once per kernel, whatever the number of fused measurements.
#include <algorithm>
#include <cstdlib>

#include "ippm_tiles.h"

// ======================================================================================================
// reset: legacy NumPy MT19937 streams regenerated on the device
// ======================================================================================================
__global__ void k_reset_scalars(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                                int32_t* __restrict__ pos, int32_t* __restrict__ split_pct,
                                float* __restrict__ comm_range, int32_t* __restrict__ ws, double* __restrict__ sums,
                                double* __restrict__ area, int n_envs) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int n = c->n_agents;
  int per = n + 1;
  if (tid >= n_envs * per) return;
  int e = tid / per, k = tid % per;
  int64_t ep = episode[e];
  // clear this map's workspace (deferred clamp state + plan)
  int32_t* w = ws + (size_t)(e * per + k) * IPPM_WS_WORDS;
  // the written-cells box of the finished episode moves to the (now idle) first op record, where every workgroup of
  // ippm_reset_maps finds it unchanged while the first one already writes the new episode's box
  const int32_t box_x = w[WS_BBOX_X], box_y = w[WS_BBOX_Y], sbox_x = w[WS_SBOX_X], sbox_y = w[WS_SBOX_Y];
  for (int i = 0; i < WS_OPS; ++i) w[i] = 0;
  {   // the union of the fused-cells box and the sensed-cells box (either may be empty: x1 <= x0)
    int ax0 = box_x & 0xFFFF, ax1 = (unsigned)box_x >> 16, ay0 = box_y & 0xFFFF, ay1 = (unsigned)box_y >> 16;
    const int bx0 = sbox_x & 0xFFFF, bx1 = (unsigned)sbox_x >> 16, by0 = sbox_y & 0xFFFF, by1 = (unsigned)sbox_y >> 16;
    if (bx1 > bx0 && by1 > by0) {
      if (ax1 <= ax0 || ay1 <= ay0) { ax0 = bx0; ax1 = bx1; ay0 = by0; ay1 = by1; }
      else { ax0 = min(ax0, bx0); ax1 = max(ax1, bx1); ay0 = min(ay0, by0); ay1 = max(ay1, by1); }
    }
    w[WS_OPS + 0] = ax0 | (ax1 << 16); w[WS_OPS + 1] = ay0 | (ay1 << 16);
  }
  if (area) {  // area sums of the all-prior map: sigmoid(0) = 0.5 times the bin's weight total gx*gy
    double* a = area + (size_t)(e * per + k) * IPPM_FEAT * IPPM_FEAT;
    const double v = (double)ippm_sigmoid(c->logit_prior) * (double)c->grid_x * (double)c->grid_y;
    for (int q = 0; q < IPPM_FEAT * IPPM_FEAT; ++q) a[q] = v;
  }
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
      const float lp = c->logit_prior;
      const double wh = lp == 0.f ? 0.5 : (double)(ippm_weight_l(lp, c->logit_weight_thr) * ippm_entropy_l(lp, c->logit_clip));
      s[SUM_T] = wh * (double)c->grid_x * (double)c->grid_y;
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

// One 16-byte access per lane, one trip per workgroup, non-temporal: tools/probe/copy_probe.cpp measures 6.5 TB/s for this
// copy on a 1 GiB buffer against 4.6-5.0 TB/s for grid-stride forms (and for hipMemcpyDtoD) and 3.6 TB/s when a workgroup
// moves 32 KiB: on this device many short workgroups stream better than few long ones.
typedef float ippm_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_fill_f32x4(float4* __restrict__ p, float v, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const ippm_f4 val = {v, v, v, v};
  if (i < n4) __builtin_nontemporal_store(val, reinterpret_cast<ippm_f4*>(p) + i);
}
// plain device-to-device copy: the streaming-rate yardstick of bench.py
__global__ void __launch_bounds__(256) k_stream_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    const ippm_f4 v = __builtin_nontemporal_load(reinterpret_cast<const ippm_f4*>(src) + i);
    __builtin_nontemporal_store(v, reinterpret_cast<ippm_f4*>(dst) + i);
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
// K3: sense + Bayesian update of the agent's own footprint tile
//   Mapping.update_grid_map = Simulation.get_measurement + apply_update (mappings.py:32-78,109-124; simulations.py:42-65)
// TRACK: also add the cells' change into the map's 11x11 area sums (ippm_tiles.h).
// The launch carries ceil(E/256) extra workgroups that complete the reward of the step's global fusion
// (k_reward_finalize's job: K3 is the kernel that closes an env step, the sums are final when it starts).
// ======================================================================================================
template <int VEC, int UNR, bool TRACK>
__global__ void __launch_bounds__(256)
k_sense_update(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
               const int32_t* __restrict__ pos, const uint8_t* __restrict__ truth, float* __restrict__ local,
               const uint8_t* __restrict__ flips, uint8_t* __restrict__ code, const int32_t* __restrict__ rect_in,
               int32_t* __restrict__ rect_out, int32_t* __restrict__ ws, double* __restrict__ area,
               double* __restrict__ sums, float* __restrict__ reward, unsigned long long* __restrict__ counters, int stage,
               int agent_sel, int split, int n_tiles, int n_envs, const int32_t* __restrict__ n_active) {
  if ((int)blockIdx.x >= n_tiles * split) {  // reward-finalize tail
    const int e = ((int)blockIdx.x - n_tiles * split) * 256 + (int)threadIdx.x;
    if (e < n_envs) ippm_reward_finalize_env(c, sums, reward, e);
    return;
  }
  const int n = c->n_agents;
  const int tile = blockIdx.x / split, part = blockIdx.x % split;  // (tile, row part) flattened: grid.x has no 65535 limit
  int e, i;
  if (agent_sel >= 0) { e = tile; i = agent_sel; }
  else { e = tile / n; i = tile % n; }
  if (n_active && i >= n_active[e]) return;   // not flying in this env (ippm_set_team_sizes)
  const int gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  const int32_t* p = pos + (size_t)(e * n + i) * 3;
  int r[4];
  if (rect_in) {  // projected by the kernel that moved the agents (k_plan_step): no pos -> index -> table chain here
    const int32_t* ri = rect_in + (size_t)(e * n + i) * IPPM_SENSE_REC_WORDS;   // the footprint words of K1's sense record
    r[0] = ri[0]; r[1] = ri[1]; r[2] = ri[2]; r[3] = ri[3];
  } else {
    ippm_footprint_rect(c, p[0], p[1], p[2], r, nullptr);
  }
  const int yu = r[0], yd = r[1], xl = r[2], xr = r[3];
  if (rect_out && part == 0 && threadIdx.x < 4) rect_out[(size_t)(e * n + i) * 4 + threadIdx.x] = r[threadIdx.x];
  const int h = xr - xl, w = yd - yu;
  if (h <= 0 || w <= 0) return;
  const int k = ippm_alt_index(c, p[2]);
  const float lm0 = c->logit_meas[k][0], lm1 = c->logit_meas[k][1];
  const uint32_t thr = c->flip_threshold[k];
  const float lc = c->logit_clip;
  const float lp = c->logit_prior;  // 0 unless mapping.prior != 0.5 (mappings.py:112-116: l_x + l_y - l_p)
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
  __shared__ double s_area[TRACK ? (IPPM_FEAT + 1) * IPPM_AREA_LD : 1];
  float inv_gx = 0.f, inv_gy = 0.f;
  if (TRACK) {
    area_lds_clear(s_area);
    inv_gx = __builtin_amdgcn_rcpf((float)gx);
    inv_gy = __builtin_amdgcn_rcpf((float)gy);
    __syncthreads();
  }
  bool exceed = false;
  for (int gi = gl; gi < g.groups; gi += g.lpr) {
    const int y = g.y0 + gi * VEC;
    AreaCols<VEC> ac;
    AreaAcc acc;
    if (TRACK) { ac = area_cols<VEC>(y, gy, inv_gy); acc.init(); }
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
          m[u] = load_cells_row<VEC>(map + (size_t)(xl + rr) * gy, y, gy);
          tw[u] = VEC == 4 ? ippm_truth4(tr, cell, ippm_truth_bytes(gx, gy)) : ippm_truth1(tr, cell);
          if (fl) fw[u] = load_bits<VEC>(fl, rr, y - tile_y0, S);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int rr = row + u * stride;
        if (rr >= r1) continue;
        const size_t cell = (size_t)(xl + rr) * gy + y;
        uint32_t phbits = 0;
        if (!fl && VEC == 4) phbits = philox_flip_bits4((uint32_t)cell, (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1, thr, (gy & 3) != 0);
        uint32_t cw = 0;
        float d[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
          // branch-free: cells of an edge group that lie outside the footprint keep their value
          const bool in = (unsigned)(y + q - yu) < (unsigned)w;
          uint32_t flip;
          if (fl) flip = (fw[u] >> q) & 1u;
          else if (VEC == 4) flip = (phbits >> q) & 1u;
          else {
            Philox4 p1 = ippm_philox((uint32_t)((cell + q) >> 2), (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1);
            flip = p1.v[(cell + q) & 3] < thr ? 1u : 0u;
          }
          const uint32_t obs = ((tw[u] >> q) & 1u) ^ flip;
          // mappings.py:109-124 in log-odds: clip the prior belief, add the measurement's log-odds
          const float old = m[u].v[q];
          const float l = ippm_clampl(old, lc) + ((obs ? lm1 : lm0) - lp);
          exceed |= in & (fabsf(l) > lc);
          m[u].v[q] = in ? l : old;
          cw |= (in ? obs : 0u) << q;
          if (TRACK) d[q] = in ? sigmoid_diff(l, old) : 0.f;
        }
        store_cells_row<VEC>(map + (size_t)(xl + rr) * gy, y, gy, m[u]);
        store_bits<VEC>(cd, rr, y - tile_y0, S, cw);
        if (TRACK) area_row<VEC>(acc, s_area, ac, xl + rr, gx, inv_gx, d);
      }
    }
    if (TRACK) acc.flush(s_area, ac.cb);
  }
  if (ws && __any(exceed) && lane == 0) ws[(size_t)(e * (n + 1) + i) * IPPM_WS_WORDS + WS_FLAG_S] = 1;
  if (counters && part == 0 && threadIdx.x == 0)
    atomicAdd(&counters[(tile & (IPPM_COUNTER_SLOTS - 1)) * 8 + 0], (unsigned long long)h * w);
  if (TRACK) {
    __syncthreads();
    area_lds_commit(s_area, area + (size_t)(e * (n + 1) + i) * IPPM_FEAT * IPPM_FEAT);
  }
}

// ------------------------------------------------------------------------------------------------------
// K3, tile form (no area tracking): the form the env-only step and bench.py's roofline run.
// A footprint at 15 m is 90 cells = 23-24 four-cell groups: as one 32-lane row it fills 72-75 % of the lanes, and every
// lane pays the full Philox round chain.  Here a row is covered by CH = 3 passes of `lpr` lanes (8 lanes x 3 for the 90-cell
// footprint: 96-100 % of the lanes busy, 128-byte row segments), a wavefront covers 64/lpr rows, a workgroup 4 x that in ONE
// trip -- no row loop in the common case -- with the CH loads of a lane's row in flight together.  Loads and stores go
// through raw buffer resources (one VALU per address; a lane without work points out of range: loads return 0, stores
// are dropped), per-cell selects are bit-mask blends, the clip is one v_med3.
// ------------------------------------------------------------------------------------------------------
typedef unsigned ippm_k3_u4 __attribute__((ext_vector_type(4)));
#define IPPM_K3_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define IPPM_K3_OOB 0x7FFFFFF0
#ifndef IPPM_K3_WAVES      // wavefronts per workgroup of k_sense_tiles (variant builds: 2 / 8)
#define IPPM_K3_WAVES 4
#endif
#ifndef IPPM_K3_CH         // loads in flight per lane of k_sense_tiles (variant builds)
#define IPPM_K3_CH 3
#endif
#ifndef IPPM_K3_GRID_ORDER // default workgroup order of k_sense_tiles (template parameter GO)
#define IPPM_K3_GRID_ORDER 0
#endif
#ifndef IPPM_K3_LOAD_AUX   // cache policy of the map accesses (bit 1 = non-temporal on gfx950)
#define IPPM_K3_LOAD_AUX 0
#endif
#ifndef IPPM_K3_STORE_AUX
#define IPPM_K3_STORE_AUX 0
#endif

// MIS: the grid is not a multiple of 4 wide (rows only 4-byte aligned); FLIPS: explicit flip tiles (parity mode) instead of Philox;
// REC: rect_in holds K1's sense records (the closing kernel of a batched step) -- without it the footprint and the sensor constants
// come from pos and the config tables.  All
// are compile-time so that the production instantiation <4, false, false> carries neither the second Philox call and the
// cell-by-cell tail stores nor the flips resource (the kernel sits at the SGPR limit: every uniform it holds less is a
// v_writelane / v_readlane pair less in its instruction stream).
// WPG x CH: wavefronts per workgroup and loads in flight per lane (the production shape is chosen per grid width, ippm_sense_step)
// GO: the launch's fastest-varying workgroup index -- 0: a footprint's parts, 1: the agents of an env (a map's parts n workgroups apart)
// TL: the maps are stored as 128-byte TILES of 4 rows x 8 cells (ippm_internal.h "tile storage"; dense 16-byte form only).  The footprint's
// lane-loads are then dealt out tile by tile -- 8 consecutive lanes take one tile = one line -- over the tiles its rows and columns meet, so
// every line the footprint touches is read once and written whole; the lanes of an edge tile whose row or cells lie outside the footprint
// load and store them unchanged (`inm` = 0).  Truth bits, Philox counters and code bytes keep their (row, column) addressing.
template <int VEC, bool MIS, bool FLIPS, bool REC, bool DENSE, bool TRACK, bool TL = false, int WPG = IPPM_K3_WAVES, int CHN = IPPM_K3_CH, int GO = IPPM_K3_GRID_ORDER>
__global__ void __launch_bounds__(64 * WPG)
k_sense_tiles(const int32_t* __restrict__ rect_in, int n, int agent_sel, int stage, int rows_per_part, int gy, int gx,
              float* __restrict__ local, const uint8_t* __restrict__ truth, const int64_t* __restrict__ episode,
              uint8_t* __restrict__ code, int S_arg, float lc_arg, uint32_t k0_arg, uint32_t k1_arg,
              const ippm_config* __restrict__ c, const int32_t* __restrict__ pos,
              const uint8_t* __restrict__ flips, int32_t* __restrict__ rect_out, int32_t* __restrict__ ws,
              double* __restrict__ sums, float* __restrict__ reward, unsigned long long* __restrict__ counters,
              double* __restrict__ area, const int32_t* __restrict__ n_active, int col_round) {
  // Argument order = latency order.  A workgroup lives for one trip, so what stands in front of its map loads is paid by every
  // wavefront: with the config fields behind the config pointer behind the kernel-argument load, the footprint came in three
  // dependent scalar round trips.  The first 14 argument words arrive in SGPRs with the wavefront (kernel-argument preload,
  // csrc/Makefile): the address of the agent's sense record (K1: footprint + the measurement constants of its altitude) needs
  // nothing else, and every config scalar the kernel uses is passed by value -- one scalar round trip, then the map loads.
  constexpr int CH = CHN;
  // grid = (row parts, agents, envs): no index arithmetic to undo
  // (GO == 2, variant builds: the parts slowest -- all first parts, then all second parts, ...: 37 us at config 2 against 34)
  // (GO == 3, round 5, removed: every workgroup of an env on ONE XCD -- x = (env % 8) + 8 * agent, y = part, z = env / 8, so that the
  //  parts of a footprint share one L2's truth and code-tile lines, on the XCD the fusion's wavefronts of that env run on: 36.2 us
  //  against 34.7 at 256^2, 68.5 / 62.8 at 512^2, 110 / 94 at 1024^2 -- spreading a footprint over the XCDs is what it wants)
  const int part = GO == 1 ? blockIdx.y : (GO == 2 ? blockIdx.z : blockIdx.x), e = GO == 2 ? blockIdx.y : blockIdx.z;
  const int agent_blk = GO == 0 ? blockIdx.y : blockIdx.x;
  const int i = agent_sel >= 0 ? agent_sel : agent_blk;
  const int tile = e * n + agent_blk;
  int r[4];
  float lm0, lm1;
  uint32_t thr;
  int64_t ep = episode ? episode[e] : 0;
  int S = S_arg;
  float lc = lc_arg;
  uint32_t k0 = k0_arg, k1 = k1_arg;
  if (REC) {   // K1's sense record: the footprint and the measurement constants of its altitude in one 32-byte scalar load
    const int4* ri = reinterpret_cast<const int4*>(rect_in + (size_t)(e * n + i) * IPPM_SENSE_REC_WORDS);
    const int4 a = ri[0], b = ri[1];
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
    lm0 = __int_as_float(b.x); lm1 = __int_as_float(b.y); thr = (uint32_t)b.z;
  } else {
    const int32_t* p0 = pos + (size_t)(e * n + i) * 3;
    ippm_footprint_rect(c, p0[0], p0[1], p0[2], r, nullptr);
    const int k = ippm_alt_index(c, p0[2]);
    const float lp = c->logit_prior;
    lm0 = c->logit_meas[k][0] - lp; lm1 = c->logit_meas[k][1] - lp;
    thr = c->flip_threshold[k];
  }
  // every scalar the kernel will use is on its way; left alone the compiler sinks the later ones to their first use, behind a
  // second (third) scalar wait in front of the map loads
  {
    int b0 = __builtin_amdgcn_readfirstlane(__float_as_int(lm0)), b1 = __builtin_amdgcn_readfirstlane(__float_as_int(lm1));
    int b2 = __builtin_amdgcn_readfirstlane(__float_as_int(lc)), e0 = (int)(uint32_t)ep, e1 = (int)(uint32_t)(ep >> 32);
    asm volatile("" : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]), "+s"(thr), "+s"(b0), "+s"(b1), "+s"(b2), "+s"(e0), "+s"(e1), "+s"(S),
                 "+s"(k0), "+s"(k1), "+s"(code), "+s"(ws), "+s"(counters), "+s"(sums), "+s"(rect_out));
    lm0 = __int_as_float(b0); lm1 = __int_as_float(b1); lc = __int_as_float(b2);
    ep = (int64_t)(((uint64_t)(uint32_t)e1 << 32) | (uint32_t)e0);
  }
  // (an agent that does not fly in this env, ippm_set_team_sizes: K1 left it an empty sense record; without records the team
  //  size decides)
  if (!REC && n_active && i >= n_active[e]) { r[0] = r[1] = r[2] = r[3] = 0; }
  const int yu = r[0], yd = r[1], xl = r[2], xr = r[3];
  if (rect_out && part == 0 && threadIdx.x < 4) rect_out[(size_t)(e * n + i) * 4 + threadIdx.x] = r[threadIdx.x];
  const int h = xr - xl, w = yd - yu;
  // col_round (cells; VEC = off, 32 = a 128-byte line; dense 16-byte form only): the row segments the workgroups walk are rounded
  // OUTWARDS to whole lines.  The groups this adds lie outside the footprint's columns: loaded, left as they are (`inm` below is 0 for
  // their cells) and stored back, so that every line of the footprint is written whole -- no truth, no code byte, no flag for them.
  static_assert(!TL || (DENSE && VEC == 4 && !MIS), "tile storage: dense 16-byte form");
  const int cr = TL ? 8 : ((DENSE && VEC == 4 && !MIS) ? col_round : VEC);
  const int y0 = yu & ~(cr - 1), tile_y0 = yu & ~3;
  // (TL: `groups` = the lane-loads of one ROW OF TILES of the footprint, 8 per tile; `tr0` = its first row of tiles)
  const int tr0 = xl >> 2;
  const int groups = TL ? (((yd + 7) >> 3) - (yu >> 3)) * 8 : (min(gy, (yd + cr - 1) & ~(cr - 1)) - y0 + VEC - 1) / VEC;
  // DENSE: the footprint's 4-cell groups in ROW-MAJOR order, T = row * W + group (W = groups per row), are dealt out in runs: a
  // wavefront takes CH * 64 consecutive ones, its lane's loads are T = base + q * 64 + lane -- a load instruction covers 64
  // consecutive groups (2.7 whole 368-byte row segments of a 15 m footprint at 256^2) and every wavefront but a footprint's last
  // is full whatever the width.  (Round 4 gave a wavefront floor(CH * 64 / W) whole rows: 184 of 192 lane-loads at 256^2 and 512^2,
  // and no shape with fewer than two rows per wavefront -- 1024^2 could not run with two loads in flight.)
  float dense_invw = 0.f;
  int dense_base = 0, dense_total = 0;
  int part_rows = rows_per_part;
  if (DENSE) {
    dense_invw = __builtin_amdgcn_rcpf((float)max(groups, 1));   // floor(T / W) = (int)((T + 0.5) / W) exactly for T < 2^20 (ippm_div_small)
    dense_total = (TL ? ((xr + 3) >> 2) - tr0 : h) * groups;
    dense_base = (part * WPG + (int)(threadIdx.x >> 6)) * (CH * 64);
    part_rows = h;      // (r0, r1 below only decide whether the workgroup has work: part * WPG * CH * 64 < total)
  }
  const int r0 = DENSE ? (part * WPG * CH * 64 < dense_total ? 0 : h) : part * part_rows, r1 = min(h, r0 + part_rows);
  // the reward of the step whose global fusion ran in the launch before this one: any one thread per env completes it (last:
  // nothing of this launch waits for it)
  if (w <= 0 || r0 >= r1) {
    if (sums && part == 0 && agent_blk == 0 && threadIdx.x == 0) ippm_reward_finalize_env(c, sums, reward, e);
    return;
  }
  // TRACK: the map's 11 x 11 area sums (K6's view of it, ippm_tiles.h) follow the cells this workgroup changes: weighted sigmoid
  // differences into a 12 x 12 float64 tile in LDS, one global atomic per touched bin at the end
  __shared__ double s_area[TRACK ? (IPPM_FEAT + 1) * IPPM_AREA_LD : 1];
  float inv_gx = 0.f, inv_gy = 0.f;
  if (TRACK) {
    area_lds_clear(s_area);
    inv_gx = __builtin_amdgcn_rcpf((float)gx);
    inv_gy = __builtin_amdgcn_rcpf((float)gy);
    __syncthreads();
  }
  constexpr bool mis = MIS;   // rows start at addresses that are only 4-byte aligned
  int shift = 3;
  while (shift < 6 && ((groups + (1 << shift) - 1) >> shift) > CH) ++shift;
  const int lpr = 1 << shift, rpw = 64 >> shift;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane >> shift, gl = lane & (lpr - 1);
  const size_t TB = ippm_tile_bytes(S, VEC);
  const __amdgpu_buffer_rsrc_t rmap = IPPM_K3_RSRC(local + (size_t)(e * n + i) * IPPM_MAP_PITCH(gx, gy), (size_t)gx * gy * 4);
  const __amdgpu_buffer_rsrc_t rtruth = IPPM_K3_RSRC(truth + (size_t)e * ippm_truth_bytes(gx, gy), ippm_truth_bytes(gx, gy));
  const __amdgpu_buffer_rsrc_t rcode = IPPM_K3_RSRC(code + (size_t)(e * n + i) * TB, TB);
  const __amdgpu_buffer_rsrc_t rflip = IPPM_K3_RSRC(FLIPS ? flips + (size_t)(e * n + i) * TB : code, FLIPS ? TB : 0);
  const uint32_t sw = ippm_stream_word((uint32_t)i, (uint32_t)stage, IPPM_DOMAIN_FLIP);
  float amax = 0.f;
  for (int gbase = 0; gbase < (DENSE ? 1 : groups); gbase += CH * lpr) {     // one trip unless the footprint is wider than 3 x 64 groups
    for (int row0 = r0 + (DENSE ? 0 : wv * rpw + sub); row0 < r1; row0 += WPG * rpw) {  // one trip for the common footprints
      CellVec<VEC> m[CH];
      uint32_t tw[CH], fw[CH];
      int cellv[CH], rowv[CH], yv[CH], poff[CH];
      bool on[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        int row, y;
        if (DENSE && TL) {
          // lane-load T of the footprint's tiles in row-major order of TILES: row of tiles rr, lane-load gi of its 8 per tile --
          // (gi >> 3) the tile, (gi >> 1) & 3 the row inside it, gi & 1 which half of that row's 8 cells.  One line = 8 consecutive T.
          const int T = dense_base + q * 64 + lane;
          const int rr = ippm_div_small(T, dense_invw);
          const int gi = T - rr * groups;
          row = ((tr0 + rr) << 2) + ((gi >> 1) & 3) - xl;      // (may lie outside [0, h): an edge tile's rows beyond the footprint)
          on[q] = T < dense_total;
          y = y0 + ((gi >> 3) << 3) + ((gi & 1) << 2);
          poff[q] = ((tr0 + rr) * gy + y0 + gi) * 16;   // a row of tiles is 16 gy bytes, tile c of it at 128 c = 16 (8 c) bytes, lane-load g of the row of tiles at 16 g
        } else if (DENSE) {
          const int T = dense_base + q * 64 + lane;
          const int rr = ippm_div_small(T, dense_invw);
          const int gi = T - rr * groups;
          row = rr;
          on[q] = T < dense_total;
          y = y0 + gi * VEC;
        } else {
          const int gidx = gbase + gl + q * lpr;
          on[q] = gidx < groups;
          row = row0;
          y = y0 + gidx * VEC;
        }
        const int x = xl + row;
        const int cell = x * gy + y;
        cellv[q] = cell; rowv[q] = row; yv[q] = y;
        if (!TL) poff[q] = cell * 4;
        if (VEC == 4) {
          const ippm_k3_u4 t = __builtin_amdgcn_raw_buffer_load_b128(rmap, on[q] ? poff[q] : IPPM_K3_OOB, 0, IPPM_K3_LOAD_AUX);
          m[q].v[0] = __uint_as_float(t.x); m[q].v[1 % VEC] = __uint_as_float(t.y);
          m[q].v[2 % VEC] = __uint_as_float(t.z); m[q].v[3 % VEC] = __uint_as_float(t.w);
        } else {
          m[q].v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmap, on[q] ? cell * 4 : IPPM_K3_OOB, 0, IPPM_K3_LOAD_AUX));
        }
        // (grids not a multiple of 4 wide: a group's four truth bits may straddle a byte -- two bytes at any byte address)
        tw[q] = mis ? (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rtruth, on[q] ? (cell >> 3) : IPPM_K3_OOB, 0, 0)
                    : (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rtruth, on[q] ? (cell >> 3) : IPPM_K3_OOB, 0, 0);
        fw[q] = FLIPS ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rflip, on[q] && (!TL || (unsigned)row < (unsigned)h) ? (int)tile_index<VEC>(row, y - tile_y0, S) : IPPM_K3_OOB, 0, 0)
                      : 0u;
      }
#pragma unroll
      for (int q = 0; q < CH; ++q) {
        // wave-uniform: narrow footprints use one or two of the three passes
        if (DENSE ? (dense_base + q * 64 >= dense_total) : (gbase + q * lpr >= groups)) continue;
        int y = yv[q];
        // (the column masks below do not depend on the row: left alone they are hoisted in front of the loads, ~60 instructions
        // between the wavefront's start and its first memory request)
        asm volatile("" : "+v"(y));
        const int row = rowv[q];
        const int cell = cellv[q];
        const uint32_t tbits = VEC == 4 ? (tw[q] >> (cell & 7)) & 0xFu : (tw[q] >> (cell & 7)) & 1u;
        uint32_t flipbits;
        if (FLIPS) {
          flipbits = fw[q] & (VEC == 4 ? 0xFu : 1u);
        } else if (VEC == 4) {
          flipbits = philox_flip_bits4((uint32_t)cell, (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1, thr, mis);
        } else {
          const Philox4 ph = ippm_philox((uint32_t)(cell >> 2), (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1);
          const int s4 = cell & 3;
          const uint32_t word = s4 == 0 ? ph.v[0] : s4 == 1 ? ph.v[1] : s4 == 2 ? ph.v[2] : ph.v[3];
          flipbits = word < thr ? 1u : 0u;
        }
        uint32_t inm = 0;  // cells of the group inside the footprint's columns (edge groups)
#pragma unroll
        for (int j = 0; j < VEC; ++j) inm |= ((unsigned)(y + j - yu) < (unsigned)w) ? (1u << j) : 0u;
        if (TL) inm = (unsigned)row < (unsigned)h ? inm : 0u;   // ... and rows (an edge tile's rows above / below the footprint)
        const uint32_t obs = (tbits ^ flipbits) & inm;
        // mappings.py:109-124 in log-odds: clip the prior belief, add the measurement's log-odds (minus logit(prior))
        float dsig[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float old = m[q].v[j];
          const float l = ippm_clampl(old, lc) + ippm_blend(ippm_bitmask(obs, j), lm1, lm0);
          const uint32_t im = ippm_bitmask(inm, j);
          m[q].v[j] = ippm_blend(im, l, old);
          amax = fmaxf(amax, fabsf(ippm_masked(im, l)));
          if (TRACK) dsig[j] = sigmoid_diff(m[q].v[j], old);   // (exactly 0 for the cells outside the footprint's columns)
        }
        if (TRACK && VEC == 4 && on[q]) {
          float sd = 0.f, cA = 0.f;
          const AreaCols<VEC> ac = area_cols<VEC>(y, gy, inv_gy);
#pragma unroll
          for (int j = 0; j < VEC; ++j) { sd += dsig[j]; cA += ac.wA[j] * dsig[j]; }
          const float cB = 11.f * sd - cA;
          const int n11 = 11 * (xl + row), rb = area_bin(n11, inv_gx);
          const float nA = (float)min((rb + 1) * gx - n11, 11), nB = 11.f - nA;
          double* pa = s_area + rb * IPPM_AREA_LD + ac.cb;
          const float v00 = nA * cA, v01 = nA * cB, v10 = nB * cA, v11 = nB * cB;
          if (v00 != 0.f) atomicAdd(pa, (double)v00);
          if (v01 != 0.f) atomicAdd(pa + 1, (double)v01);
          if (v10 != 0.f) atomicAdd(pa + IPPM_AREA_LD, (double)v10);
          if (v11 != 0.f) atomicAdd(pa + IPPM_AREA_LD + 1, (double)v11);
        }
        const int off = on[q] ? poff[q] : IPPM_K3_OOB;
        if (VEC == 4) {
          ippm_k3_u4 t;
          t.x = __float_as_uint(m[q].v[0]); t.y = __float_as_uint(m[q].v[1 % VEC]);
          t.z = __float_as_uint(m[q].v[2 % VEC]); t.w = __float_as_uint(m[q].v[3 % VEC]);
          // the last group of a row that is not a multiple of 4 wide hangs over into the next row: its cells go out one by one
          const bool tail = mis && y + 4 > gy;
          __builtin_amdgcn_raw_buffer_store_b128(t, rmap, tail ? IPPM_K3_OOB : off, 0, IPPM_K3_STORE_AUX);
          if (mis) {
            __builtin_amdgcn_raw_buffer_store_b32(t.x, rmap, tail ? off : IPPM_K3_OOB, 0, IPPM_K3_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(t.y, rmap, tail && y + 1 < gy ? off + 4 : IPPM_K3_OOB, 0, IPPM_K3_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(t.z, rmap, tail && y + 2 < gy ? off + 8 : IPPM_K3_OOB, 0, IPPM_K3_STORE_AUX);
          }
        } else {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(m[q].v[0]), rmap, off, 0, IPPM_K3_STORE_AUX);
        }
        // (a group outside the footprint's columns -- rounded row segments -- has no byte in the code tile)
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)obs, rcode, on[q] && inm != 0 ? (int)tile_index<VEC>(row, y - tile_y0, S) : IPPM_K3_OOB, 0, 0);
      }
      if (DENSE) break;   // (a part is exactly one trip of its four wavefronts)
    }
  }
  if (ws && __any(amax > lc) && lane == 0) ws[(size_t)(e * (n + 1) + i) * IPPM_WS_WORDS + WS_FLAG_S] = 1;
  if (counters && part == 0 && threadIdx.x == 0)
    atomicAdd(&counters[(tile & (IPPM_COUNTER_SLOTS - 1)) * 8 + 0], (unsigned long long)h * w);
  if (sums && part == 0 && agent_blk == 0 && threadIdx.x == 0) ippm_reward_finalize_env(c, sums, reward, e);
  if (TRACK) {
    __syncthreads();
    area_lds_commit(s_area, area + (size_t)(e * (n + 1) + i) * IPPM_FEAT * IPPM_FEAT);
  }
}


// ------------------------------------------------------------------------------------------------------
// Episode reset of the maps in one launch: prior fill + start-position sensing (Mapping.init_priors, mappings.py:126-132, and the
// t = 0 Mapping.update_grid_map of agent.py:43-49), restricted to what the last episode wrote.
//   - Every map carries the bounding box of the cells written since its last reset (ws words WS_BBOX_*, kept by k_plan_step).
//     At config 2 that box is 54 % of a local map and 85 % of a global map on average: the fills were 200 us of a 580 us reset,
//     1.6 GB written to store one constant.  full != 0 (first use, or maps written behind the planner's back): whole maps.
//   - A local map's start footprint gets its first measurement right here (K3's arithmetic on a prior cell: clamp(prior) + the
//     measurement's log-odds) instead of fill -> read -> modify -> write by a K3 launch of its own.
// Two kinds of workgroup share the launch and write disjoint cells.  FILL workgroups (blockIdx.x < fill_chunks): 32 rows of one
// map, 8 rows per wavefront, one 16-byte store per lane and row; they skip the 4-cell groups that meet the map's start footprint,
// and rows outside the box cost an early exit.  SENSE workgroups: a 32-row part of one agent's
// start footprint in K3's dense lane geometry (a 90-cell footprint row = 3 passes of 8 lanes; a row-per-wavefront layout would
// run the Philox rounds on 23 of 64 lanes), writing whole groups -- measured cells and the prior cells that share their groups.
// ------------------------------------------------------------------------------------------------------
#ifndef IPPM_RESET_ROWS
#define IPPM_RESET_ROWS 32   // rows per FILL workgroup = 4 wavefronts x 8 rows.  One row per wavefront ran at the pace of wavefront
#endif                       // launches (1.4 M of them: 310 us for 0.9 GB); 8 / 16 / 32 rows per workgroup: 201 / 180 / 148 us
__global__ void __launch_bounds__(256)
k_reset_maps(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode, const int32_t* __restrict__ pos,
             const uint8_t* __restrict__ truth, float* __restrict__ local, float* __restrict__ global, const uint8_t* __restrict__ flips,
             uint8_t* __restrict__ code, const int32_t* __restrict__ rect, int32_t* __restrict__ ws, int full, int fill_chunks,
             const int32_t* __restrict__ n_active, int col_align, int32_t* __restrict__ slabs, int n_slabs, int tl) {
  const int n = c->n_agents, gx = c->grid_x, gy = c->grid_y, S = c->tile_stride;
  const int e = blockIdx.z, m = blockIdx.y;
  const bool is_global = m == n;
  const bool flying = is_global || !n_active || m < n_active[e];   // (ippm_set_team_sizes: the others neither sense nor publish)
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int32_t* w = ws + (size_t)(e * (n + 1) + m) * IPPM_WS_WORDS;
  // everything a workgroup needs comes in ONE round trip: the start footprint was projected by the launch before this one
  // (343 k workgroups of one 16-byte store per lane: with the position -> lattice index -> centre table chain in front of the
  // stores the launch ran at the pace of that chain, 214 us)
  int yu = 0, yd = 0, xl = 0, xr = 0;
  if (!is_global && flying) {
    const int4 r = *reinterpret_cast<const int4*>(rect + (size_t)(e * n + m) * 4);
    yu = r.x; yd = r.y; xl = r.z; xr = r.w;
    if (xr <= xl || yd <= yu) { yu = yd = xl = xr = 0; }
  }
  const float lp = c->logit_prior;
  float* map = is_global ? global + (size_t)e * IPPM_MAP_PITCH(gx, gy) : local + (size_t)(e * n + m) * IPPM_MAP_PITCH(gx, gy);
  const __amdgpu_buffer_rsrc_t rmap = IPPM_K3_RSRC(map, (size_t)gx * gy * 4);
  const bool mis = (gy & 3) != 0;
  const int fg0 = yu >> 2, fg1 = (yd + 3) >> 2;   // groups that meet the footprint's columns
  if ((int)blockIdx.x < fill_chunks) {
    // ---- FILL
    int bx0 = 0, bx1 = gx, by0 = 0, by1 = gy;
    if (!full) {
      const int bx = w[WS_OPS + 0], by = w[WS_OPS + 1];   // parked there by k_reset_scalars
      bx0 = bx & 0xFFFF; bx1 = (unsigned)bx >> 16; by0 = by & 0xFFFF; by1 = (unsigned)by >> 16;
      // the box's columns rounded outwards to whole 128-byte lines (col_align = 32 cells; rows are a multiple of that long): the
      // cells this adds hold the prior already, and a row segment that starts and ends on line boundaries is written as whole
      // lines -- a partial line at each end of each of a box's ~200 rows is a masked write the memory side pays more for than for
      // the 48 extra bytes (round 6: footprints shifted onto line boundaries made this launch 11 % faster, profiles/r06)
      if (col_align > 0 && by1 > by0) { by0 &= ~(col_align - 1); by1 = min(gy, (by1 + col_align - 1) & ~(col_align - 1)); }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the new episode's box: what this launch writes that is not the prior
      w[WS_BBOX_X] = xl | (xr << 16);
      w[WS_BBOX_Y] = yu | (yd << 16);
    }
    if (slabs) {
      // Dirty slabs (ippm_set_dirty_slabs): instead of the map's one box, every 16-row slab has its own column interval -- what the
      // plans and the sense records of the finished episode marked in it.  A wavefront (8 rows) reads its slab's interval; after a
      // workgroup barrier (both wavefronts of a slab have read it) the slab is re-armed for the new episode: the start footprint's
      // columns if it meets these rows, else empty.  This workgroup is the only one that touches these words in this launch.
      static_assert(IPPM_RESET_ROWS == 2 * IPPM_SLAB_ROWS && IPPM_RESET_ROWS / 4 * 2 == IPPM_SLAB_ROWS, "two wavefronts of 8 rows per 16-row slab");
      int32_t* sl = slabs + (size_t)(e * (n + 1) + m) * 2 * n_slabs;
      const int s_idx = (int)blockIdx.x * 2 + (wv >> 1);
      int lo = 0, hi = 0;
      if (s_idx < n_slabs) { lo = sl[s_idx]; hi = sl[n_slabs + s_idx]; }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
      __syncthreads();
      if ((wv & 1) == 0 && lane == 0 && s_idx < n_slabs) {
        const int r0 = s_idx * IPPM_SLAB_ROWS, r1 = r0 + IPPM_SLAB_ROWS;
        const bool meets = xr > xl && yd > yu && xl < r1 && xr > r0;
        sl[s_idx] = meets ? yu : IPPM_SLAB_EMPTY_LO;
        sl[n_slabs + s_idx] = meets ? yd : 0;
      }
      if (!full) {
        by0 = lo; by1 = hi;
        if (col_align > 0 && by1 > by0) { by0 &= ~(col_align - 1); by1 = min(gy, (by1 + col_align - 1) & ~(col_align - 1)); }
        bx0 = 0; bx1 = gx;      // (rows: the slab's own 16, i.e. this wavefront's 8)
      }
    }
    if (by1 <= by0) return;
    const ippm_k3_u4 t = {__float_as_uint(lp), __float_as_uint(lp), __float_as_uint(lp), __float_as_uint(lp)};
    // a wavefront takes IPPM_RESET_ROWS / 4 consecutive rows, a lane one 16-byte store in each of them (all in flight together)
    constexpr int RPW = IPPM_RESET_ROWS / 4;
    const int xw = blockIdx.x * IPPM_RESET_ROWS + wv * RPW;
    if (xw >= bx1 || xw + RPW <= bx0) return;
    if (tl) {
      // tile storage: the wavefront's 8 rows are two rows of tiles; it writes every TILE that meets the box (whole lines: an edge tile's cells
      // outside the box hold the prior already) except the tiles that meet the start footprint -- those belong to the SENSE workgroups, whole.
      const int G0 = (by0 >> 3) << 3, G1 = ((by1 + 7) >> 3) << 3;                        // lane-loads of a row of tiles: 8 per tile
      const int fR0 = xl >> 2, fR1 = (xr + 3) >> 2, fG0 = (yu >> 3) << 3, fG1 = ((yd + 7) >> 3) << 3;   // the footprint's tiles (empty footprint: fR1 <= fR0 or fG1 <= fG0)
      const bool fp_some = xr > xl && yd > yu;
#pragma unroll
      for (int u = 0; u < RPW / 4; ++u) {
        const int R = (xw >> 2) + u;
        if (R * 4 >= bx1 || R * 4 + 4 <= bx0 || R * 4 >= gx) continue;
        const bool fp_row = fp_some && R >= fR0 && R < fR1;
        for (int G = G0 + lane; G < G1; G += 64) {
          const bool skip = fp_row && G >= fG0 && G < fG1;
          __builtin_amdgcn_raw_buffer_store_b128(t, rmap, skip ? IPPM_K3_OOB : (R * gy + G) * 16, 0, 2);
        }
      }
      return;
    }
    for (int g = (by0 >> 2) + lane; g < ((by1 + 3) >> 2); g += 64) {
      const int y = g * 4;
      const bool fp_col = g >= fg0 && g < fg1;
#pragma unroll
      for (int u = 0; u < RPW; ++u) {
        const int x = xw + u;
        // outside the box, or a group a SENSE workgroup writes: the store goes nowhere
        const bool skip = x < bx0 || x >= bx1 || x >= gx || (fp_col && x >= xl && x < xr);
        const int off = (x * gy + y) * 4;
        if (mis && y + 4 > gy) {   // a row's last group hangs over into the next row: cell by cell
          if (!skip) {
            __builtin_amdgcn_raw_buffer_store_b32(t.x, rmap, off, 0, 0);
            if (y + 1 < gy) __builtin_amdgcn_raw_buffer_store_b32(t.x, rmap, off + 4, 0, 0);
            if (y + 2 < gy) __builtin_amdgcn_raw_buffer_store_b32(t.x, rmap, off + 8, 0, 0);
          }
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(t, rmap, skip ? IPPM_K3_OOB : off, 0, 2);   // non-temporal: read a step later at the earliest
        }
      }
    }
    return;
  }
  // ---- SENSE: rows [r0, r1) of the footprint, K3's geometry (k_sense_tiles), on prior cells
  if (is_global) return;
  const int part = (int)blockIdx.x - fill_chunks;
  const int h = xr - xl, wdt = yd - yu;
  // (tile storage: a part is 8 rows of the footprint's TILES, rows [r0, r1) relative to xl may start above the footprint)
  const int r0 = tl ? ((xl >> 2) + part * 8) * 4 - xl : part * 32, r1 = tl ? min(((xr + 3) >> 2) * 4 - xl, r0 + 32) : min(h, r0 + 32);
  if (wdt <= 0 || h <= 0 || r0 >= r1) return;
  const int32_t* p = pos + (size_t)(e * n + m) * 3;
  const int k = ippm_alt_index(c, p[2]);
  const float lc = c->logit_clip;
  const float lm0 = c->logit_meas[k][0] - lp, lm1 = c->logit_meas[k][1] - lp;
  const uint32_t thr = c->flip_threshold[k];
  const uint32_t sw = ippm_stream_word((uint32_t)m, 0u, IPPM_DOMAIN_FLIP);
  const int64_t ep = episode ? episode[e] : 0;
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
  const __amdgpu_buffer_rsrc_t rtruth = IPPM_K3_RSRC(truth + (size_t)e * ippm_truth_bytes(gx, gy), ippm_truth_bytes(gx, gy));
  const size_t TB = ippm_tile_bytes(S, 4);
  const __amdgpu_buffer_rsrc_t rcode = IPPM_K3_RSRC(code + (size_t)(e * n + m) * TB, TB);
  const __amdgpu_buffer_rsrc_t rflip = IPPM_K3_RSRC(flips ? flips + (size_t)(e * n + m) * TB : code, flips ? TB : 0);
  if (tl) {
    // every tile the footprint meets, whole: a wavefront takes rows of tiles, its lanes their lane-loads; cells outside the footprint get the prior
    const int G0 = (yu >> 3) << 3, G1 = ((yd + 7) >> 3) << 3, y0c = yu & ~3;
    for (int R = ((xl + r0) >> 2) + wv; R * 4 < xl + r1; R += 4) {
      for (int G = G0 + lane; G < G1; G += 64) {
        const int x = (R << 2) + ((G >> 1) & 3), y = ((G >> 3) << 3) + ((G & 1) << 2);
        const int row = x - xl;
        const int cell = x * gy + y;
        const uint32_t tw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rtruth, cell >> 3, 0, 0);
        const uint32_t tbits = (tw >> (cell & 7)) & 0xFu;
        uint32_t inm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) inm |= ((unsigned)(y + j - yu) < (unsigned)wdt) ? (1u << j) : 0u;
        inm = (unsigned)row < (unsigned)h ? inm : 0u;
        uint32_t flipbits;
        if (flips) flipbits = inm ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rflip, (int)tile_index<4>(row, y - y0c, S), 0, 0) & 0xFu : 0u;
        else flipbits = philox_flip_bits4((uint32_t)cell, (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1, thr, false);
        const uint32_t obs = (tbits ^ flipbits) & inm;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((inm >> j) & 1u) ? ippm_clampl(lp, lc) + (((obs >> j) & 1u) ? lm1 : lm0) : lp;
        ippm_k3_u4 t;
        t.x = __float_as_uint(v[0]); t.y = __float_as_uint(v[1]); t.z = __float_as_uint(v[2]); t.w = __float_as_uint(v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(t, rmap, (R * gy + G) * 16, 0, 0);
        if (inm) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)obs, rcode, (int)tile_index<4>(row, y - y0c, S), 0, 0);
      }
    }
    return;
  }
  const int y0 = yu & ~3, groups = fg1 - fg0;
  int shift = 3;
  while (shift < 6 && ((groups + (1 << shift) - 1) >> shift) > 3) ++shift;
  const int lpr = 1 << shift, rpw = 64 >> shift;
  const int sub = lane >> shift, gl = lane & (lpr - 1);
  for (int gbase = 0; gbase < groups; gbase += 3 * lpr) {
    for (int row = r0 + wv * rpw + sub; row < r1; row += 4 * rpw) {
      const int x = xl + row;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int gidx = gbase + gl + q * lpr;
        if (gidx >= groups) continue;
        const int y = y0 + gidx * 4;
        const int cell = x * gy + y;
        const uint32_t tw = mis ? (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rtruth, cell >> 3, 0, 0)
                                : (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rtruth, cell >> 3, 0, 0);
        const uint32_t tbits = (tw >> (cell & 7)) & 0xFu;
        uint32_t flipbits;
        if (flips) flipbits = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rflip, (int)tile_index<4>(row, y - y0, S), 0, 0) & 0xFu;
        else flipbits = philox_flip_bits4((uint32_t)cell, (uint32_t)ep, sw, (uint32_t)(ep >> 32), k0, k1, thr, mis);
        uint32_t inm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) inm |= ((unsigned)(y + j - yu) < (unsigned)wdt) ? (1u << j) : 0u;
        const uint32_t obs = (tbits ^ flipbits) & inm;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)   // mappings.py:109-124 on a prior cell, exactly as K3 forms it; the group's other cells: prior
          v[j] = ((inm >> j) & 1u) ? ippm_clampl(lp, lc) + (((obs >> j) & 1u) ? lm1 : lm0) : lp;
        ippm_k3_u4 t;
        t.x = __float_as_uint(v[0]); t.y = __float_as_uint(v[1]); t.z = __float_as_uint(v[2]); t.w = __float_as_uint(v[3]);
        const int off = cell * 4;
        if (mis && y + 4 > gy) {
          __builtin_amdgcn_raw_buffer_store_b32(t.x, rmap, off, 0, 0);
          if (y + 1 < gy) __builtin_amdgcn_raw_buffer_store_b32(t.y, rmap, off + 4, 0, 0);
          if (y + 2 < gy) __builtin_amdgcn_raw_buffer_store_b32(t.z, rmap, off + 8, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(t, rmap, off, 0, 0);
        }
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)obs, rcode, (int)tile_index<4>(row, y - y0, S), 0, 0);
      }
    }
  }
}

// full-grid weighted entropy per map (initialisation of T, evaluation metrics)
__global__ void __launch_bounds__(256)
k_weighted_entropy(const ippm_config* __restrict__ c, const float* __restrict__ maps, const uint8_t* __restrict__ truth,
                   double* __restrict__ out, int maps_per_truth, int tl) {
  const int m = blockIdx.y;
  const size_t total = (size_t)c->grid_x * c->grid_y;
  const float* p = maps + (size_t)m * total;
  const uint8_t* t = truth ? truth + (size_t)(m / maps_per_truth) * ippm_truth_bytes(c->grid_x, c->grid_y) : nullptr;
  const float lc = c->logit_clip, wt = c->logit_weight_thr;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float v = p[i];
    const float wgt = t ? (float)ippm_truth1(t, ippm_stored_cell(i, c->grid_y, tl)) : ippm_weight_l(v, wt);
    acc += wgt * ippm_entropy_l(v, lc);
  }
  acc = ippm_wave_sum(acc);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[m], (double)(s[0] + s[1] + s[2] + s[3]));
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

static int fill_f32(ippm_ctx* ctx, float* p, float v, size_t n, hipStream_t st) {
  if (n == 0) return 0;
  if (n % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    IPPM_LAUNCH(ctx, IPPM_T_RESET, k_fill_f32x4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), st, reinterpret_cast<float4*>(p), v, n / 4);
  } else {
    IPPM_LAUNCH(ctx, IPPM_T_RESET, k_fill_f32, dim3(std::min(4096, grid1(n))), dim3(256), st, p, v, n);
  }
  IPPM_LAUNCH_CHECK("fill");
  return 0;
}

extern "C" int ippm_reset_episode(ippm_ctx* ctx, const int64_t* episode, int32_t* pos, uint8_t* truth, float* local,
                                  float* global, int32_t* split_pct, float* comm_range_out, int32_t* ws, double* sums,
                                  double* area, int32_t n_envs, void* stream) {
  if (!ctx || !episode || !pos || !ws) { ippm_set_error("ippm_reset_episode: null argument"); return -1; }
  if (truth && !split_pct) { ippm_set_error("ippm_reset_episode: truth generation needs split_pct scratch"); return -1; }
  if (n_envs <= 0) return 0;
  const ippm_config& c = ctx->cfg;
  const int per = c.n_agents + 1;
  IPPM_LAUNCH(ctx, IPPM_T_RESET, k_reset_scalars, dim3(grid1((size_t)n_envs * per, 64)), dim3(64), S_(stream), ctx->dcfg, episode, pos,
                     split_pct, comm_range_out, ws, sums, area, n_envs);
  IPPM_LAUNCH_CHECK("reset_scalars");
  const size_t cells = (size_t)c.grid_x * c.grid_y;
  if (truth) {
    IPPM_LAUNCH(ctx, IPPM_T_RESET, k_fill_truth, dim3(std::min(64, grid1(cells)), n_envs), dim3(256), S_(stream), ctx->dcfg, split_pct,
                       truth, n_envs);
    IPPM_LAUNCH_CHECK("fill_truth");
  }
  if (local) if (int rc = fill_f32(ctx, local, c.logit_prior, cells * n_envs * c.n_agents, S_(stream))) return rc;
  if (global) if (int rc = fill_f32(ctx, global, c.logit_prior, cells * n_envs, S_(stream))) return rc;
  return 0;
}

extern "C" int ippm_reset_maps(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const uint8_t* truth, float* local,
                               float* global, const uint8_t* flips, uint8_t* code, int32_t* rect, int32_t* ws, int32_t full,
                               int32_t n_envs, void* stream) {
  if (!ctx || !pos || !truth || !local || !global || !code || !rect || !ws) { ippm_set_error("ippm_reset_maps: null argument"); return -1; }
  if (!flips && !episode) { ippm_set_error("ippm_reset_maps: Philox flips need the episode ids"); return -1; }
  if (ctx->vec != 4) { ippm_set_error("ippm_reset_maps: needs the 16-byte layout (grid_y >= 44); use ippm_reset_episode + ippm_sense_update"); return -2; }
  if (n_envs <= 0) return 0;
  if (n_envs > 65535) { ippm_set_error("ippm_reset_maps: more than 65535 envs per launch"); return -1; }
  const ippm_config& c = ctx->cfg;
  // K2 first: the start footprints into `rect` (k_reset_maps then starts from them instead of re-deriving them per workgroup)
  hipLaunchKernelGGL(k_footprint, dim3(grid1(n_envs * c.n_agents)), dim3(256), 0, S_(stream), ctx->dcfg, pos, rect, (int32_t*)nullptr,
                     n_envs * c.n_agents);
  IPPM_LAUNCH_CHECK("footprint");
  int h_max = 1;
  for (int k = 0; k < c.space_z; ++k) h_max = std::max(h_max, 2 * c.radius_x[k]);
  // (tile storage: a SENSE part is 8 rows of tiles, and a footprint of h rows meets up to h / 4 + 1 of them)
  const int fill_chunks = (c.grid_x + IPPM_RESET_ROWS - 1) / IPPM_RESET_ROWS, sense_parts = ctx->tl ? ((h_max + 3) / 4 + 1 + 7) / 8 : (h_max + 31) / 32;
  dim3 grid((unsigned)(fill_chunks + sense_parts), (unsigned)(c.n_agents + 1), (unsigned)n_envs);
  IPPM_LAUNCH(ctx, IPPM_T_RESET_MAPS, k_reset_maps, grid, dim3(256), S_(stream), ctx->dcfg, episode, pos, truth, local, global, flips, code, rect,
              ws, full ? 1 : 0, fill_chunks, ctx->n_active, (c.grid_y % 32 == 0 && !ctx->tl) ? ctx->knob_reset_align : 0, ctx->slabs, ippm_slab_count(ctx), ctx->tl);
  IPPM_LAUNCH_CHECK("reset_maps");
  return 0;
}

extern "C" int ippm_logodds_to_prob(ippm_ctx* ctx, const float* src, float* dst, int64_t n, void* stream) {
  if (!ctx || !src || !dst) { ippm_set_error("ippm_logodds_to_prob: null argument"); return -1; }
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_logodds_to_prob, dim3(std::min(4096, grid1((size_t)n))), dim3(256), 0, S_(stream), src, dst, (size_t)n);
  IPPM_LAUNCH_CHECK("logodds_to_prob");
  return 0;
}

extern "C" int ippm_prob_to_logodds(ippm_ctx* ctx, const float* src, float* dst, int64_t n, void* stream) {
  if (!ctx || !src || !dst) { ippm_set_error("ippm_prob_to_logodds: null argument"); return -1; }
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_prob_to_logodds, dim3(std::min(4096, grid1((size_t)n))), dim3(256), 0, S_(stream), src, dst, (size_t)n);
  IPPM_LAUNCH_CHECK("prob_to_logodds");
  return 0;
}

// row-major <-> tile storage, 16 bytes per lane: lane-load g of the tile-storage order is cells (x, y .. y + 3)
__global__ void __launch_bounds__(256) k_maps_relayout(const float4* __restrict__ src, float4* __restrict__ dst, int gy, size_t groups_per_map,
                                                       size_t total, int to_tiled) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const size_t m = i / groups_per_map, g = i - m * groups_per_map;     // g: 16-byte slot of the map in tile-storage order
  const size_t cell = ippm_stored_cell(g * 4, gy, 1);                  // its first cell's row-major number
  const size_t rm = m * groups_per_map + cell / 4;
  if (to_tiled) dst[i] = src[rm]; else dst[rm] = src[i];
}

extern "C" int ippm_maps_relayout(ippm_ctx* ctx, const float* src, float* dst, int32_t n_maps, int32_t to_tiled, void* stream) {
  if (!ctx || !src || !dst || src == dst) { ippm_set_error("ippm_maps_relayout: null argument or src == dst"); return -1; }
  const ippm_config& c = ctx->cfg;
  if (c.grid_x % 4 || c.grid_y % 8) { ippm_set_error("ippm_maps_relayout: the grid is not made of whole tiles (grid_x % 4, grid_y % 8)"); return -2; }
  if (n_maps <= 0) return 0;
  const size_t gpm = (size_t)c.grid_x * c.grid_y / 4, total = gpm * (size_t)n_maps;
  hipLaunchKernelGGL(k_maps_relayout, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S_(stream), reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), c.grid_y, gpm, total, to_tiled ? 1 : 0);
  IPPM_LAUNCH_CHECK("maps_relayout");
  return 0;
}

extern "C" int ippm_clamp_logodds(ippm_ctx* ctx, float* maps, int64_t n, void* stream) {
  if (!ctx || !maps) { ippm_set_error("ippm_clamp_logodds: null argument"); return -1; }
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_clamp_logodds, dim3(std::min(4096, grid1((size_t)n))), dim3(256), 0, S_(stream), maps, ctx->cfg.logit_clip, (size_t)n);
  IPPM_LAUNCH_CHECK("clamp_logodds");
  return 0;
}

extern "C" int ippm_stream_copy(ippm_ctx* ctx, const void* src, void* dst, int64_t n_bytes, void* stream) {
  if (!ctx || !src || !dst) { ippm_set_error("ippm_stream_copy: null argument"); return -1; }
  if (n_bytes % 16 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) {
    ippm_set_error("ippm_stream_copy: needs 16-byte aligned buffers and size");
    return -1;
  }
  if (n_bytes <= 0) return 0;
  const size_t n4 = (size_t)n_bytes / 16;
  hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, S_(stream),
                     reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n4);
  IPPM_LAUNCH_CHECK("stream_copy");
  return 0;
}

extern "C" int ippm_footprint(ippm_ctx* ctx, const int32_t* pos, int32_t* rect, int32_t* rect_unclipped, int32_t n_envs,
                              void* stream) {
  if (!ctx || !pos || !rect) { ippm_set_error("ippm_footprint: null argument"); return -1; }
  const int n = n_envs * ctx->cfg.n_agents;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_footprint, dim3(grid1(n)), dim3(256), 0, S_(stream), ctx->dcfg, pos, rect, rect_unclipped, n);
  IPPM_LAUNCH_CHECK("footprint");
  return 0;
}

extern "C" int ippm_sense_step(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const uint8_t* truth, float* local,
                               const uint8_t* flips, uint8_t* code, const int32_t* rect_in, int32_t* rect, int32_t* ws,
                               double* area, double* sums, float* reward, int32_t stage, int32_t agent_sel, int32_t n_envs,
                               void* stream) {
  if (!ctx || !pos || !truth || !local || !code || !rect) { ippm_set_error("ippm_sense_update: null argument"); return -1; }
  if (!flips && !episode) { ippm_set_error("ippm_sense_update: Philox flips need the episode ids"); return -1; }
  if (agent_sel >= ctx->cfg.n_agents) { ippm_set_error("ippm_sense_update: agent_sel out of range"); return -1; }
  if ((sums == nullptr) != (reward == nullptr)) { ippm_set_error("ippm_sense_step: sums and reward go together"); return -1; }
  if (n_envs <= 0) return 0;
  const int maps = agent_sel >= 0 ? n_envs : n_envs * ctx->cfg.n_agents;
  const int tail = sums ? grid1(n_envs) : 0;
  dim3 block(256);
  if ((!area || ctx->vec == 4) && (ctx->tl || !env_int("IPPM_K3_CLASSIC", 0))) {
    // tile form: one trip per workgroup for the common footprints (rows_per_part = 4 wavefronts x 8 rows); with the area sums
    // tracked in its TRACK instantiation (16-byte layout only: k_sense_update below keeps the narrow grids)
    const ippm_config& c = ctx->cfg;
    int h_max = 1;
    for (int k = 0; k < c.space_z; ++k) h_max = std::max(h_max, 2 * c.radius_x[k]);
    // workgroup shape: wavefronts per workgroup x loads in flight per lane.  (4, 3) unless the production combination below runs
    // with another one (knobs IPPM_K3_WPG / IPPM_K3_CHN, or the per-grid-width choice of ippm_ctx_create)
    int wpg = IPPM_K3_WAVES, chn = IPPM_K3_CH;
    const bool shaped = ctx->vec == 4 && !flips && rect_in && !area && (c.grid_y & 3) == 0 && ctx->knob_k3_dense != 0;
    if (shaped) { wpg = ctx->k3_wpg; chn = ctx->k3_chn; }
    const int rows_per_part = std::max(4, env_int("IPPM_K3_ROWS", 8 * wpg));
    block = dim3(64 * wpg);
    int parts = (h_max + rows_per_part - 1) / rows_per_part;
    // dense lane mapping (k_sense_tiles<..., DENSE>): a part is a run of wpg * chn * 64 of the footprint's row-major 4-cell groups
    // (W per row: one more than its width needs when it starts off a group boundary)
    const bool tl = ctx->tl != 0;     // tile storage (ippm_set_map_layout): always the dense form
    const bool dense = ctx->vec == 4 && (ctx->knob_k3_dense != 0 || tl);
    // row segments rounded outwards to whole 128-byte lines (k_sense_tiles): dense form, rows a multiple of 32 cells long; on by default for
    // rows of at least 512 cells (IPPM_K3_ROUND forces it on / off) -- profiles/r06/tile_round_ab.txt
    const int col_round = (dense && !tl && (c.grid_y % 32) == 0 && ctx->knob_k3_round > 0) ? ippm_round_cells(ctx->knob_k3_round) : 4;
    if (dense && tl) {
      int need = 1;    // lane-loads of the largest footprint's tiles: rows of tiles x 8 per tile, one more of either than its size needs
      for (int k = 0; k < c.space_z; ++k) {
        const int ht = std::min(c.grid_x / 4, (2 * c.radius_x[k] + 3) / 4 + 1), wt = std::min(c.grid_y / 8, (2 * c.radius_y[k] + 7) / 8 + 1);
        need = std::max(need, (ht * wt * 8 + wpg * chn * 64 - 1) / (wpg * chn * 64));
      }
      parts = need;
    } else if (dense) {
      int need = 1;
      for (int k = 0; k < c.space_z; ++k) {
        int wmax = (2 * c.radius_y[k] + 3) / 4 + 1;
        if (col_round > 4) wmax = std::min((c.grid_y + 3) / 4, (wmax + 2 * (col_round / 4 - 1) + col_round / 4 - 1) / (col_round / 4) * (col_round / 4));   // rounded row segments
        need = std::max(need, (2 * c.radius_x[k] * wmax + wpg * chn * 64 - 1) / (wpg * chn * 64));
      }
      parts = need;
    }
    if (n_envs > 65535) { ippm_set_error("ippm_sense_step: more than 65535 envs per launch"); return -1; }
    const unsigned g_agents = (unsigned)(agent_sel >= 0 ? 1 : c.n_agents);
    const int go = shaped ? ctx->k3_go : IPPM_K3_GRID_ORDER;
    dim3 grid = go == 1 ? dim3(g_agents, (unsigned)parts, (unsigned)n_envs)
              : (go == 2 ? dim3(g_agents, (unsigned)n_envs, (unsigned)parts) : dim3((unsigned)parts, g_agents, (unsigned)n_envs));
    int32_t* rect_out = rect_in == rect ? nullptr : rect;
#define IPPM_K3T____(V, M, F, R, D, T, L, ...)                                                                                       \
  IPPM_LAUNCH(ctx, IPPM_T_SENSE, (k_sense_tiles<V, M, F, R, D, T, L __VA_OPT__(,) __VA_ARGS__>), grid, block, S_(stream), rect_in, c.n_agents, agent_sel, stage, rows_per_part, \
              c.grid_y, c.grid_x, local, truth, episode, code, c.tile_stride, c.logit_clip, (uint32_t)c.philox_seed,                 \
              (uint32_t)(c.philox_seed >> 32), ctx->dcfg, pos, flips, rect_out, ws, sums, reward, ctx->dcounters, area, ctx->n_active, col_round)
#define IPPM_K3T___(V, M, F, R, D, T, ...) IPPM_K3T____(V, M, F, R, D, T, false __VA_OPT__(,) __VA_ARGS__)
#define IPPM_K3T__(V, M, F, R, D) do { if (area) IPPM_K3T___(V, M, F, R, D, true); else IPPM_K3T___(V, M, F, R, D, false); } while (0)
#define IPPM_K3TL_(F, R) do { if (area) IPPM_K3T____(4, false, F, R, true, true, true); else IPPM_K3T____(4, false, F, R, true, false, true); } while (0)
#define IPPM_K3TL(F) do { if (rect_in) IPPM_K3TL_(F, true); else IPPM_K3TL_(F, false); } while (0)
#define IPPM_K3T_(V, M, F, R) do { if (dense) IPPM_K3T__(V, M, F, R, true); else IPPM_K3T__(V, M, F, R, false); } while (0)
#define IPPM_K3T(V, M, F) do { if (rect_in) IPPM_K3T_(V, M, F, true); else IPPM_K3T_(V, M, F, false); } while (0)
    if (shaped && !(wpg == IPPM_K3_WAVES && chn == IPPM_K3_CH && go == IPPM_K3_GRID_ORDER)) {   // the closing K3 of the env-only step in another workgroup shape / order
#define IPPM_K3S(W_, C_, G_) if (wpg == W_ && chn == C_ && go == G_) { \
        if (tl) IPPM_K3T____(4, false, false, true, true, false, true, W_, C_, G_); else IPPM_K3T____(4, false, false, true, true, false, false, W_, C_, G_); \
        IPPM_LAUNCH_CHECK("sense_tiles"); return 0; }
      IPPM_K3S(2, 2, 1) IPPM_K3S(2, 2, 0) IPPM_K3S(1, 3, 0) IPPM_K3S(2, 3, 0) IPPM_K3S(1, 4, 0) IPPM_K3S(2, 4, 0) IPPM_K3S(4, 4, 0) IPPM_K3S(4, 2, 0) IPPM_K3S(4, 3, 1)
#undef IPPM_K3S
      ippm_set_error("ippm_sense_step: no instantiation for this K3 shape (IPPM_K3_WPG x IPPM_K3_CHN)");
      return -1;
    }
    const bool mis = (c.grid_y & 3) != 0;
    if (tl) {
      if (flips) IPPM_K3TL(true); else IPPM_K3TL(false);
    } else if (ctx->vec == 4) {
      if (flips) { if (mis) IPPM_K3T(4, true, true); else IPPM_K3T(4, false, true); }
      else { if (mis) IPPM_K3T(4, true, false); else IPPM_K3T(4, false, false); }
    } else {
      if (flips) IPPM_K3T___(1, false, true, false, false, false); else IPPM_K3T___(1, false, false, false, false, false);
    }
#undef IPPM_K3T____
#undef IPPM_K3T___
#undef IPPM_K3TL_
#undef IPPM_K3TL
#undef IPPM_K3T__
#undef IPPM_K3T_
#undef IPPM_K3T
    IPPM_LAUNCH_CHECK("sense_tiles");
    return 0;
  }
  const int split = std::max(1, env_int("IPPM_SPLIT_K3", 2));
  dim3 grid((unsigned)maps * split + tail);
  const int unr = env_int("IPPM_UNROLL_K3", 2);
  int32_t* rect_out = rect_in == rect ? nullptr : rect;
#define IPPM_K3_LAUNCH(V, U, T)                                                                                               \
  IPPM_LAUNCH(ctx, IPPM_T_SENSE, (k_sense_update<V, U, T>), grid, block, S_(stream), ctx->dcfg, episode, pos, truth, local, flips, code, \
              rect_in, rect_out, ws, area, sums, reward, ctx->dcounters, stage, agent_sel, split, maps, n_envs, ctx->n_active)
  if (ctx->vec == 4) {
    if (area) { if (unr >= 2) IPPM_K3_LAUNCH(4, 2, true); else IPPM_K3_LAUNCH(4, 1, true); }
    else if (unr >= 4) IPPM_K3_LAUNCH(4, 4, false);
    else if (unr >= 2) IPPM_K3_LAUNCH(4, 2, false);
    else IPPM_K3_LAUNCH(4, 1, false);
  } else {
    if (area) IPPM_K3_LAUNCH(1, 1, true); else IPPM_K3_LAUNCH(1, 1, false);
  }
#undef IPPM_K3_LAUNCH
  IPPM_LAUNCH_CHECK("sense_update");
  return 0;
}

extern "C" int ippm_sense_update(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const uint8_t* truth,
                                 float* local, const uint8_t* flips, uint8_t* code, int32_t* rect, int32_t* ws,
                                 int32_t stage, int32_t agent_sel, int32_t n_envs, void* stream) {
  return ippm_sense_step(ctx, episode, pos, truth, local, flips, code, nullptr, rect, ws, nullptr, nullptr, nullptr, stage, agent_sel,
                         n_envs, stream);
}

extern "C" int ippm_weighted_entropy(ippm_ctx* ctx, const float* maps, const uint8_t* truth, int32_t maps_per_truth,
                                     double* out, int32_t n_maps, void* stream) {
  if (!ctx || !maps || !out) { ippm_set_error("ippm_weighted_entropy: null argument"); return -1; }
  if (n_maps <= 0) return 0;
  IPPM_HIP(hipMemsetAsync(out, 0, sizeof(double) * n_maps, S_(stream)));
  const size_t cells = (size_t)ctx->cfg.grid_x * ctx->cfg.grid_y;
  hipLaunchKernelGGL(k_weighted_entropy, dim3(std::min(32, grid1(cells)), n_maps), dim3(256), 0, S_(stream), ctx->dcfg, maps, truth,
                     out, maps_per_truth > 0 ? maps_per_truth : 1, ctx->tl);
  IPPM_LAUNCH_CHECK("weighted_entropy");
  return 0;
}
