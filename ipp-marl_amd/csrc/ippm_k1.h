// K1 (action mask + collision mask + action choice + move) for one env by one wavefront, shared by the plan kernel
// (step_small.hip: the learned-policy step, where K1 follows the actor) and the tile fusion's launch (fuse_tiles.hip: the env-only
// step, where K1 rides along as one extra wavefront per env instead of sitting on the plan kernel's critical path).
//   AgentActionSpace.get_action_mask / apply_collision_mask / action_to_position   agent/action_space.py:25-589
//   ActorNetwork.do_eps_exploration                                                actor/network.py:90-96
#pragma once
#include "ippm_internal.h"

#ifdef __HIPCC__
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

// AgentActionSpace.get_action_mask for one action (action_space.py:25-196)
__device__ __forceinline__ bool action_in_bounds(const ippm_config* c, int a, int px, int py, int pz) {
  const int A = c->n_actions, s = c->spacing;
  const int max_alt = c->min_altitude + (c->space_z - 1) * s;
  int dx, dy, dz;
  action_offset(A, a, s, dx, dy, dz);
  const int nx = px + dx, ny = py + dy, nz = pz + dz;
  bool ok = nx >= 0 && nx <= c->x_dim_m && ny >= 0 && ny <= c->y_dim_m;
  if (A == 6 || A == 27) ok = ok && nz >= c->min_altitude && nz <= max_alt;
  if ((A == 9 || A == 27) && dx == 0 && dy == 0 && dz == 0) ok = false;
  return ok;
}
__device__ __forceinline__ uint32_t boundary_mask(const ippm_config* c, int px, int py, int pz) {
  uint32_t m = 0;
  for (int a = 0; a < c->n_actions; ++a) m |= action_in_bounds(c, a, px, py, pz) ? (1u << a) : 0u;
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

// the order-dependent zeroing rules of apply_collision_mask for one moved agent
__device__ __forceinline__ uint32_t collide(int A, uint32_t m, uint32_t z) {
  if (!z) return m;
  if (A == 6) return __popc(m) > 1 ? (m & ~z) : m;
  if (A == 9) { m &= ~z; return m == 0 ? z : m; }
  return m & ~z;
}


// LDS hand-over between the lanes of ONE wavefront (K1 runs on wavefront 0 of the plan kernel's workgroup while the others build
// tile items, so a workgroup barrier is not available here): DS operations of a wavefront execute in program order, the fence
// only keeps the compiler from moving them.
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// K1 for one env by one wavefront (lanes 0..63 of wavefront 0).  s_pos: the env's positions in LDS, updated in place.
// get_action_mask -> apply_collision_mask -> action choice -> action_to_position (action_space.py:25-589,
// actor/network.py:90-96, coma_wrapper.py:97-104)
__device__ void k1_env(const ippm_config* __restrict__ c, int64_t ep, int32_t* s_pos, const float* __restrict__ probs_e,
                       const int32_t* __restrict__ action_in_e, int policy, int t, uint8_t* __restrict__ mask_e,
                       int32_t* __restrict__ action_e, int32_t* __restrict__ fault_e, int n_act = -1) {
  // n = the agents that fly in this env (ippm_set_team_sizes; all of them by default): the loops below never look at the others
  const int n = n_act >= 0 ? n_act : c->n_agents, A = c->n_actions, s = c->spacing;
  const int lane = threadIdx.x & 63;
  const uint32_t k0 = (uint32_t)c->philox_seed, k1 = (uint32_t)(c->philox_seed >> 32);
  int flt = 0;
  // Off the serial chain, up front: every agent's boundary mask (its position does not change before its own move; lanes =
  // actions, one ballot each, kept in lane i) and its Philox word (lane i draws for agent i).
  __shared__ float s_pr[IPPM_MAX_AGENTS * IPPM_MAX_ACTIONS];  // the policy's probabilities: fetched once, not per agent in the chain
  if (policy >= 2)
    for (int q = lane; q < n * A; q += 64) s_pr[q] = probs_e[q];
  uint32_t bmask_mine = 0, word_mine = 0;
  for (int i = 0; i < n; ++i) {
    const uint32_t b = (uint32_t)__ballot(lane < A && action_in_bounds(c, lane, s_pos[i * 3], s_pos[i * 3 + 1], s_pos[i * 3 + 2]));
    bmask_mine = lane == i ? b : bmask_mine;
  }
  if ((policy == 1 || policy == 2) && lane < n)
    word_mine = ippm_philox(0u, (uint32_t)ep, ippm_stream_word((uint32_t)lane, (uint32_t)t, IPPM_DOMAIN_ACTION), (uint32_t)(ep >> 32), k0, k1).v[0];
  wave_sync_lds();
  for (int i = 0; i < n; ++i) {
    const int px = s_pos[i * 3], py = s_pos[i * 3 + 1], pz = s_pos[i * 3 + 2];
    const uint32_t bmask = (uint32_t)__builtin_amdgcn_readlane((int)bmask_mine, i);
    const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)word_mine, i);
    uint32_t m = bmask;
    int ix, iy, iz;
    ippm_pos_to_index(c, px, py, pz, ix, iy, iz);
    for (int j = 0; j < i; ++j) {  // s_pos[j] already holds agent j's post-move position
      int jx, jy, jz;
      ippm_pos_to_index(c, s_pos[j * 3], s_pos[j * 3 + 1], s_pos[j * 3 + 2], jx, jy, jz);
      m = collide(A, m, collision_bits(A, jx - ix, jy - iy, jz - iz));
    }
    int a = -1;
    if (m == 0) {
      flt |= 1 << i;  // the reference's torch.multinomial raises on an all-zero distribution
    } else if (policy == 0) {
      a = action_in_e[i];
    } else if (policy == 1) {
      const int kth = (int)__umulhi(word, (uint32_t)__popc(m));
      // the kth valid action: the lane whose bit is set and has kth set bits below it
      const bool mine = lane < A && ((m >> lane) & 1u) && __popc(m & ((1u << (lane & 31)) - 1u)) == kth;
      a = __ffsll((unsigned long long)__ballot(mine)) - 1;
    } else {
      const float* pr = s_pr + i * A;
      if (policy == 3) {  // eval: argmax of probs*mask (first maximum)
        float best = -1.f;
        for (int q = 0; q < A; ++q) {
          float v = ((m >> q) & 1u) ? pr[q] : 0.f;
          if (v > best) { best = v; a = q; }
        }
      } else {  // train: inverse CDF over probs*mask, sequential float32 sums without FMA contraction
        float total = 0.f;
        for (int q = 0; q < A; ++q) total = __fadd_rn(total, ((m >> q) & 1u) ? pr[q] : 0.f);
        const float u = (float)(word >> 8) * (1.0f / 16777216.0f);
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
    if (a < 0 || a >= A) a = bmask ? __ffs(bmask) - 1 : 0;  // keep the state sane: first boundary-valid action
    int dx, dy, dz;
    action_offset(A, a, s, dx, dy, dz);
    wave_sync_lds();  // every lane has read agent i's old position
    if (lane == 0) {
      s_pos[i * 3] = px + dx; s_pos[i * 3 + 1] = py + dy; s_pos[i * 3 + 2] = pz + dz;
      action_e[i] = a;
    }
    if (lane < A) mask_e[(size_t)i * A + lane] = (m >> lane) & 1u;
    wave_sync_lds();
  }
  // this step's empty-mask bits replace the last step's; IPPM_FAULT_WORK_OVERFLOW (set by a builder wavefront of the same launch,
  // possibly at this very moment) is sticky
  if (fault_e && lane == 0) {
    atomicAnd(fault_e, IPPM_FAULT_WORK_OVERFLOW);
    if (flt) atomicOr(fault_e, flt);
  }
}


// written-cells box of a map (ws words WS_BBOX_*): union with rows [x0, x1) x columns [y0, y1)
__device__ __forceinline__ void box_union(int32_t* wm, int x0, int x1, int y0, int y1, int wx = WS_BBOX_X, int wy = WS_BBOX_Y) {
  if (x1 <= x0 || y1 <= y0) return;
  const int bx = wm[wx], by = wm[wy];
  int ax0 = bx & 0xFFFF, ax1 = (unsigned)bx >> 16, ay0 = by & 0xFFFF, ay1 = (unsigned)by >> 16;
  if (ax1 <= ax0 || ay1 <= ay0) { ax0 = x0; ax1 = x1; ay0 = y0; ay1 = y1; }
  else { ax0 = min(ax0, x0); ax1 = max(ax1, x1); ay0 = min(ay0, y0); ay1 = max(ay1, y1); }
  wm[wx] = ax0 | (ax1 << 16);
  wm[wy] = ay0 | (ay1 << 16);
}

// dirty slabs of a map (ippm_set_dirty_slabs): rows [x0, x1) x columns [y0, y1) have been (or are about to be) written.  sl = the map's
// [2, ns] record.  No return values are used: the atomics are issued and forgotten.
__device__ __forceinline__ void slab_mark(int32_t* sl, int ns, int x0, int x1, int y0, int y1) {
  if (x1 <= x0 || y1 <= y0) return;
  for (int s = x0 / IPPM_SLAB_ROWS; s <= (x1 - 1) / IPPM_SLAB_ROWS && s < ns; ++s) {
    atomicMin(sl + s, y0);
    atomicMax(sl + ns + s, y1);
  }
}

#endif  // __HIPCC__
