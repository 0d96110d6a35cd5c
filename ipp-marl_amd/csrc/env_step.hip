// Environment-step kernels for gfx950: reset (device MT19937), K1 mask/act/move, K2 footprint,
// K3 sense+update, comm matrix, K4 local fusion, K5 global fusion + information-gain reward.
//
// All of this is HBM-bound byte/float streaming over map tiles: no MFMA.  The layout rules are
//   - a map row (y contiguous) is covered by lanes holding 4 grid-aligned cells each (one 16-byte access),
//     the 1-byte truth/code/flip planes ride along as one aligned 32-bit word per lane;
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
  const int gys = ippm_gyp(c);  // truth rows are padded to the patch width like the maps
  uint8_t* t = truth + (size_t)e * gx * gys;
  int total = gx * gys;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int x = i / gys, y = i - x * gys;
    int v = (split < 2) ? x : y;
    t[i] = (y < gy && v >= lo && v < hi) ? 1 : 0;
  }
}

__global__ void k_fill_f32(float* __restrict__ p, float v, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
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
// comm matrix
// ======================================================================================================
__global__ void k_comm(const ippm_config* __restrict__ c, const int64_t* __restrict__ episode,
                       const int32_t* __restrict__ pos, const float* __restrict__ comm_range,
                       const double* __restrict__ draws, uint8_t* __restrict__ comm, int t, int n_envs) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int n = c->n_agents;
  if (tid >= n_envs * n) return;
  int e = tid / n, i = tid % n;
  const int32_t* pi = pos + (size_t)(e * n + i) * 3;
  const double range = comm_range ? (double)comm_range[e] : c->comm_range;
  const int64_t ep = episode ? episode[e] : 0;
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
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
  }
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

__global__ void k_plan(const ippm_config* __restrict__ c, const int32_t* __restrict__ rect,
                       const int32_t* __restrict__ pos, const uint8_t* __restrict__ comm, int32_t* __restrict__ ws,
                       int global_maps, int n_envs, int agent_sel) {
  int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = c->n_agents;
  const int per = (global_maps || agent_sel >= 0) ? 1 : n;
  if (tid >= n_envs * per) return;
  const int e = tid / per;
  const int i = global_maps ? n : (agent_sel >= 0 ? agent_sel : tid % n);
  int32_t* w = ws + (size_t)(e * (n + 1) + i) * IPPM_WS_WORDS;
  int nops = 0, x0 = 1 << 30, x1 = -1, y0 = 1 << 30, y1 = -1;
  int last_src = -1;
  for (int j = 0; j < n; ++j) {
    bool take = global_maps ? true : (j != i && comm[(size_t)(e * n + i) * n + j] != 0);
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
    bool take = global_maps ? true : (j != i && comm[(size_t)(e * n + i) * n + j] != 0);
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

// ======================================================================================================
// K4 / K5: apply the planned ops to a map, each touched cell read once and written once.
// REWARD: also accumulate the information-gain reward terms of K5 (utils/reward.py:68-82).
// ======================================================================================================
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
  int32_t* pe = pos + (size_t)e * n * 3;
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
  if (fault) fault[e] = flt;
}

// ======================================================================================================
// host API
// ======================================================================================================
static inline hipStream_t S_(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int grid1(size_t n, int b = 256) { return (int)((n + b - 1) / b); }
int ippm_env_int(const char* name, int dflt) {
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
  const size_t cells = (size_t)ippm_host_gxp(c) * ippm_host_gyp(c);  // padded, patch-tiled map storage
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

extern "C" int ippm_comm_matrix(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const float* comm_range,
                                const double* draws, uint8_t* comm, int32_t t, int32_t n_envs, void* stream) {
  if (!ctx || !pos || !comm) { ippm_set_error("ippm_comm_matrix: null argument"); return -1; }
  if (!draws && !episode) { ippm_set_error("ippm_comm_matrix: Philox draws need the episode ids"); return -1; }
  hipLaunchKernelGGL(k_comm, dim3(grid1((size_t)n_envs * ctx->cfg.n_agents)), dim3(256), 0, S_(stream), ctx->dcfg, episode,
                     pos, comm_range, draws, comm, t, n_envs);
  IPPM_LAUNCH_CHECK("comm");
  return 0;
}

extern "C" int ippm_fuse_local(ippm_ctx* ctx, float* local, const uint8_t* code, const int32_t* rect, const int32_t* pos,
                               const uint8_t* comm, int32_t* ws, int32_t agent_sel, int32_t n_envs, void* stream) {
  if (!ctx || !local || !code || !rect || !pos || !comm || !ws) { ippm_set_error("ippm_fuse_local: null argument"); return -1; }
  if (agent_sel >= ctx->cfg.n_agents) { ippm_set_error("ippm_fuse_local: agent_sel out of range"); return -1; }
  const int maps = agent_sel >= 0 ? n_envs : n_envs * ctx->cfg.n_agents;
  hipLaunchKernelGGL(k_plan, dim3(grid1(maps, 64)), dim3(64), 0, S_(stream), ctx->dcfg, rect, pos, comm, ws, 0, n_envs, agent_sel);
  IPPM_LAUNCH_CHECK("plan_local");
  ippm_launch_apply(false, ctx, local, code, ws, nullptr, maps, std::max(1, ippm_env_int("IPPM_SPLIT_K4", 1)), S_(stream), agent_sel);
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
  ippm_launch_apply(true, ctx, global, code, ws, sums, n_envs, std::max(1, ippm_env_int("IPPM_SPLIT_K5", 1)), S_(stream), -1);
  IPPM_LAUNCH_CHECK("fuse_global");
  hipLaunchKernelGGL(k_reward_finalize, dim3(grid1(n_envs)), dim3(256), 0, S_(stream), ctx->dcfg, sums, reward, n_envs);
  IPPM_LAUNCH_CHECK("reward_finalize");
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
