// Internal declarations shared by the HIP translation units of libippmarl.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <string>

#include "ippmarl.h"

// ---- workspace layout: int32 [E, N+1, IPPM_WS_WORDS]; slot N of an env is its global map ------------
// words 0..7     : persistent deferred-clamp state (DESIGN.md "deferred clamp")
// words 8..15    : plan header written by the plan kernels
// words 16..159  : op list, 8 words per op
// IPPM_WS_WORDS (160) comes from ippmarl.h
#define WS_FLAG_A 0   // region A may hold values outside [clip_lo, clip_hi]
#define WS_RECT_A 1   // .. 1..4 = [yu,yd,xl,xr]
#define WS_FLAG_S 5   // the agent's current footprint rect may hold out-of-range values (set by K3)
#define WS_BBOX_X 6   // x0 | x1 << 16, WS_BBOX_Y y0 | y1 << 16: bounding box of every cell of the map written since the episode's
#define WS_BBOX_Y 7   // reset by fusions (the plans' hulls, kept by the planning wavefront of k_plan_step); ippm_reset_maps fills only
#define WS_SBOX_X 14  // the union of this box and the next: the same for the footprints K3 sensed (kept by the K1 wavefront, which runs
#define WS_SBOX_Y 15  // beside the planning wavefront: two boxes, two writers, no read-modify-write race)
#define WS_PLAN 8
#define PL_NOPS 0
#define PL_X0 1
#define PL_X1 2
#define PL_Y0 3
#define PL_Y1 4
#define PL_LAST 5     // index of the last op (its outputs stay unclamped)
#define WS_OPS 16
// ---- dirty slabs (ippm_set_dirty_slabs): int32 [E, N+1, 2, NS], NS = ceil(grid_x / IPPM_SLAB_ROWS); [.., 0, s] = lowest column written in rows
// [16 s, 16 s + 16) of the map since the episode's reset, [.., 1, s] = one past the highest; empty: (IPPM_SLAB_EMPTY_LO, 0).  Marked with
// fire-and-forget atomicMin / atomicMax by the plan kernel (the step's plans: planning wavefront; the footprints K3 will sense: K1 wavefront), consumed
// and re-armed slab by slab by ippm_reset_maps' FILL workgroups.
#define IPPM_SLAB_ROWS 16
#define IPPM_SLAB_EMPTY_LO 0x7FFFFFFF
#define OP_WORDS 8
#define OP_TYPE 0     // 0 = clamp only, 1 = fuse measurement
#define OP_SRC 1      // source agent j of the measurement
#define OP_LM0 2      // float bits: log-odds of the two measurement values at j's altitude (0 for a clamp-only op)
#define OP_YU 3
#define OP_YD 4
#define OP_XL 5
#define OP_XR 6
#define OP_LM1 7
#define IPPM_MAX_OPS (IPPM_MAX_AGENTS + 2)
// Variant builds only (make VARIANT=skewN EXTRA=-DIPPM_MAP_SKEW=N; tools/alloc_skew_sample.py): N floats of padding behind every map, honoured
// by K3's tile form, the tile fusion and k_reset_maps -- the env-only step -- and by nothing else.  0 in the product library.
#ifndef IPPM_MAP_SKEW
#define IPPM_MAP_SKEW 0
#endif
#define IPPM_MAP_PITCH(gx, gy) ((size_t)(gx) * (size_t)(gy) + IPPM_MAP_SKEW)
#define IPPM_COUNTER_SLOTS 64  // work counters are spread over 64 slots to keep atomics off one address

// sums layout: double [E, 8]
#define SUM_S1 0
#define SUM_S2 1
#define SUM_T 2
#define SUM_ACC1 3
#define SUM_ACCD 4
#define SUM_ACCT 5

#define IPPM_TIMED_CAP 4096   // event pairs per kernel class between two reads; launches beyond it go untimed
struct ippm_ctx {
  ippm_config cfg;           // host copy
  ippm_config* dcfg;         // device copy
  unsigned long long* dcounters;  // device, IPPM_COUNTER_SLOTS x 8 words (summed into ippm_counters on read)
  int vec;                   // 4: 16-byte lane groups (grid_y >= 44), else 1
  // tuning knobs, resolved ONCE at ippm_ctx_create (the work buffer's size, the plan kernel's item layout and the fusion launch
  // all derive from them and must agree for the context's lifetime)
  int knob_wave_rows, knob_persist, knob_nowork, knob_split, knob_tile_waves, knob_tile_rotate, knob_plan_builders, knob_k3_dense, knob_terrain_one_launch, knob_reset_align, knob_tile_round, knob_k3_round;
  int32_t* slabs = nullptr;    // ippm_set_dirty_slabs: per (env, map, 16-row slab) the column interval written since the episode's reset (device, caller-owned)
  const int32_t* n_active;   // device int32 [E] or nullptr: agents flying in each env (ippm_set_team_sizes)
  int k3_wpg, k3_chn, k3_go;        // workgroup shape of the env-only step's K3 (wavefronts per workgroup, loads in flight per lane)
  float2* d_roots;           // e^{2 pi i k / 1024}, k = 0..1023: the twiddle table of the terrain transforms (terrain.hip)
  int tiles;                 // the config can take the one-trip tile form of the fusion (16-byte lane groups, prior 0.5)
  int tl = 0;                // TILE STORAGE of the maps (ippm_set_map_layout): 128-byte tiles of 4 rows x 8 cells instead of row-major rows
  // kernel timing (ippm_kernel_timing): per kernel class a pool of event pairs attached to the dispatches themselves
  int timing;
  hipEvent_t* ev[IPPM_TIMED_CLASSES];
  int ev_made[IPPM_TIMED_CLASSES], ev_used[IPPM_TIMED_CLASSES];
  const char* ev_name[IPPM_TIMED_CLASSES];
};
// start/stop events for the next launch of class `cls` (nullptr, nullptr when timing is off or the pool is exhausted)
void ippm_timing_events(ippm_ctx* ctx, int cls, const char* name, hipEvent_t* a, hipEvent_t* b);
// Launch with the kernel's own begin/end timestamps recorded when timing is on: hipExtLaunchKernelGGL binds the two events to
// the dispatch packet (no extra barrier packets on the stream), which is what rocprofv3's kernel trace reports as its duration.
#define IPPM_LAUNCH_SH(ctx, cls, kern, grid, block, shmem, st, ...)                             \
  do {                                                                                          \
    hipEvent_t _ea = nullptr, _eb = nullptr;                                                    \
    if ((ctx)->timing) ippm_timing_events((ctx), (cls), #kern, &_ea, &_eb);                     \
    if (_ea) hipExtLaunchKernelGGL(kern, grid, block, shmem, st, _ea, _eb, 0, __VA_ARGS__);     \
    else hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);                         \
  } while (0)
#define IPPM_LAUNCH(ctx, cls, kern, grid, block, st, ...) IPPM_LAUNCH_SH(ctx, cls, kern, grid, block, 0, st, __VA_ARGS__)

void ippm_set_error(const std::string& msg);
// k_plan for local (global_maps == 0) or global fusion plans; step_small.hip
// Fusion work list (int32, caller-owned, ippm_work_words() long): [E] item counts, then [E][cap] items (map << 8 | run of
// rows), cap = (N+1) * runs per map.  Every env owns its slice: the plan wavefront of env e WRITES count and items (no atomics,
// nothing to clear between steps), the fusion's wavefronts each serve ONE env and stride through its list.
int ippm_fuse_wave_rows(const ippm_ctx* ctx, int n_envs);  // rows per work item (fuse.hip)
int ippm_work_env_cap(const ippm_ctx* ctx, int n_envs);    // items an env's slice can hold
// The tile form of the list (IPPM_STEP_TILES; fuse_tiles.hip): [E] counts tagged IPPM_WORK_TILED, then from word (E + 3) & ~3 on
// [E][cap] items of 4 words {run's first group | lane-loads << 16, first row, region's first group | groups per row << 16,
// op mask | map slot << 24}, cap = ippm_tile_env_cap().
int ippm_tile_env_cap(const ippm_ctx* ctx);
// IPPM_TILE_ROUND / IPPM_K3_ROUND: 1 = whole 128-byte lines (32 cells); 8 / 16 / 32 = that many cells (measurement: half and quarter lines)
static inline int ippm_round_cells(int knob) { return (knob == 8 || knob == 16 || knob == 32) ? knob : 32; }
static inline int ippm_slab_count(const ippm_ctx* ctx) { return (ctx->cfg.grid_x + IPPM_SLAB_ROWS - 1) / IPPM_SLAB_ROWS; }   // dirty slabs per map
int ippm_launch_fuse_tiles(ippm_ctx* ctx, float* local, float* global, const uint8_t* code, int32_t* ws, double* sums, double* area,
                           const int32_t* work, int n_envs, hipStream_t st);   // fuse_tiles.hip (area != nullptr: area sums tracked)
#define IPPM_WORK_TILED 0x40000000     // tag of a count written in the tile form (a kernel of the other form skips the list)
#define IPPM_WORK_OVERFLOW 0x20000000  // the env's items did not fit (never, by the bound of ippm_tile_env_cap)
#define IPPM_WORK_COUNT 0x0FFFFFFF
// loads in flight per lane of a tile item, by the number of ops that meet it (the code bytes of every op are in flight too)
// (4 for up to four ops, 2 beyond: items of more than six ops run their chain six ops at a time, fuse_tiles.hip)
#ifndef IPPM_X_SLOTS56     // measurement-only variants (make VARIANT=slots4 EXTRA=-DIPPM_X_SLOTS56=4): loads in flight per lane of an item met by five or six ops
#define IPPM_X_SLOTS56 2
#endif
__host__ __device__ inline int ippm_tile_slots(int na) { return na <= 4 ? 4 : (na <= 6 ? IPPM_X_SLOTS56 : 2); }
int ippm_launch_plan(ippm_ctx* ctx, const int32_t* rect, const int32_t* pos, const uint8_t* comm, int32_t* ws, int global_maps,
                     int n_envs, int agent_sel, hipStream_t st);
// ---- TILE STORAGE of the maps (ippm_ctx::tl, ippm_set_map_layout) ---------------------------------------------------------------------
// A map of gx x gy float32 cells is stored as 128-byte tiles of 4 rows x 8 cells, the tiles in row-major order: cell (x, y) is float
//   (x >> 2) * 4 gy + (y >> 3) * 32 + (x & 3) * 8 + (y & 7)
// of its map -- a row of tiles (4 map rows) is 4 gy floats, a tile one 128-byte line, a row's 8 cells inside a tile 32 contiguous bytes, so a
// grid-aligned 4-cell group stays one 16-byte access.  Why: a footprint of 90 x 90 cells at an arbitrary position touches 23-24 x 12-13 whole lines
// instead of 90 row segments of 3-4 lines with a partial line at either end; a bare read-modify-write of that shape runs 15-28 % faster
// (tools/probe/rmw_ceiling.cpp, profiles/r06/rmw_ceiling.txt).  Same bytes per map, nothing else changes: truth bits, code bytes, Philox counters, plans, boxes
// and area sums keep their (row, column) meaning.  Needs whole tiles (gx % 4 == 0, gy % 8 == 0) and the tile form of the fusion.
static inline bool ippm_tile_storage_ok(const ippm_ctx* ctx) {
  return ctx->tiles && ctx->vec == 4 && ctx->cfg.grid_x % 4 == 0 && ctx->cfg.grid_y % 8 == 0;
}
int ippm_check_hip(hipError_t err, const char* what);
#define IPPM_HIP(call)                                       \
  do {                                                       \
    int _rc = ippm_check_hip((call), #call);                 \
    if (_rc) return _rc;                                     \
  } while (0)
#define IPPM_LAUNCH_CHECK(name) IPPM_HIP(hipGetLastError())

// ---- device helpers ------------------------------------------------------------------------------------
#ifdef __HIPCC__

// float index of cell (x, y) inside its map, row-major (tl == 0) or tile storage (tl != 0); y .. y + 3 of a grid-aligned 4-cell group are
// contiguous in both
__host__ __device__ __forceinline__ int ippm_cell_index(int x, int y, int gy, int tl) {
  return tl ? ((x >> 2) * gy << 2) + ((y >> 3) << 5) + ((x & 3) << 3) + (y & 7) : x * gy + y;
}
// the (row-major) linear cell number x * gy + y of the cell stored at float index i of a map
__host__ __device__ __forceinline__ size_t ippm_stored_cell(size_t i, int gy, int tl) {
  if (!tl) return i;
  const size_t rt = i / ((size_t)gy * 4), rem = i - rt * (size_t)gy * 4;
  const int within = (int)(rem & 31);
  return (rt * 4 + (size_t)(within >> 3)) * (size_t)gy + (rem >> 5) * 8 + (size_t)(within & 7);
}

// ---- bit-packed byte planes ------------------------------------------------------------------------------------
// truth: one bit per cell, cell (x,y) -> bit (x*gy + y) of the env's bit string (bytes = ceil(gx*gy/32)*4).
// code / flips tiles: the 4 observation bits of a lane's 4-cell group share one byte (low nibble) when the grid is a
// multiple of 4 wide (row stride S/4 bytes); otherwise one byte per cell (row stride S).  Measured on MI355X: the
// 1-byte-per-cell planes cost 27 % of K3's time for 20 % of its bytes; packed they cost 7 % (tools/probe).
__device__ __forceinline__ uint32_t ippm_truth4(const uint8_t* tr, size_t lin, size_t nbytes) {  // cells lin..lin+3, any alignment
  const size_t b = lin >> 3;
  uint32_t w = tr[b];
  if ((lin & 7) > 4 && b + 1 < nbytes) w |= (uint32_t)tr[b + 1] << 8;   // the group straddles a byte (grids not a multiple of 4 wide)
  return (w >> (lin & 7)) & 0xFu;
}
__device__ __forceinline__ uint32_t ippm_truth1(const uint8_t* tr, size_t lin) { return (tr[lin >> 3] >> (lin & 7)) & 1u; }
// (grids that are not a multiple of 4 wide read a group's four bits with a 2-byte load at any byte address: the plane then holds one
//  byte beyond the last cell's, so that the load at the last byte stays inside the plane -- a partly out-of-range buffer load returns 0)
__host__ __device__ __forceinline__ size_t ippm_truth_bytes(int gx, int gy) {
  return (((size_t)gx * gy + ((gy & 3) ? 8 : 0) + 31) / 32) * 4;
}
__host__ __device__ __forceinline__ size_t ippm_tile_bytes(int S, int vec) { return vec == 4 ? (size_t)S * (S / 4) : (size_t)S * S; }

__device__ __forceinline__ float ippm_clipf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// Maps are stored as float32 LOG-ODDS L = ln(p/(1-p)) (DESIGN.md "log-odds storage"): the reference's
// clip(p, 1e-4, 0.9999) is clamp(L, -lc, +lc) with lc = ln(0.9999/0.0001), its Bayes update is an add.
// (v_med3_f32: one instruction; fminf(fmaxf()) costs five, three of them NaN canonicalisations)
__device__ __forceinline__ float ippm_clampl(float l, float lc) { return __builtin_amdgcn_fmed3f(l, -lc, lc); }
// Per-cell selects on bit masks without compare/select pairs and without ever becoming branches: ippm_bitmask = v_bfe_i32
// (0 or ~0, hidden from the optimiser so that it is not folded back into a select or hoisted into a saved-exec branch),
// ippm_blend = v_bfi_b32.
__device__ __forceinline__ uint32_t ippm_bitmask(uint32_t bits, int q) {
  uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)bits, q, 1);
  asm volatile("" : "+v"(m));
  return m;
}
__device__ __forceinline__ float ippm_blend(uint32_t m, float a, float b) {  // m ? a : b, m = 0 or ~0
  return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}
__device__ __forceinline__ float ippm_masked(uint32_t m, float a) { return __uint_as_float(__float_as_uint(a) & m); }  // m ? a : +0

// p = 1 - 1/(1+e^L) evaluated as 1/(1+e^-L): accurate relative to p (and e^L-small p) in float32
__device__ __forceinline__ float ippm_sigmoid(float l) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-l));
}
__device__ __forceinline__ float ippm_logit(float x) { return __logf(x * __builtin_amdgcn_rcpf(1.0f - x)); }

// Shannon entropy in bits of p = sigmoid(clamp(L)) (utils/state.py:118-121), symmetric in the sign of L;
// the small side q = e/(1+e) keeps full relative precision
__device__ __forceinline__ float ippm_entropy_l(float l, float lc) {
  // with a = min(|L|, lc), e = exp(-a), q = e/(1+e):  H = log2(1+e) + a*log2(e_)*q   (3 transcendentals)
  const float a = fminf(fabsf(l), lc);
  const float e = __expf(-a);
  const float d = 1.0f + e;
  return __log2f(d) + (a * 1.44269504f) * (e * __builtin_amdgcn_rcpf(d));
}
// entropy of a probability (used on the 11x11 resized planes)
__device__ __forceinline__ float ippm_entropy(float p, float lo, float hi) {
  p = ippm_clipf(p, lo, hi);
  float q = 1.0f - p;
  return -p * __log2f(p) - q * __log2f(q);
}

// class weight (utils/state.py:60-73 with class_weighting [0,1]): thresholds 0.501 / 0.499 on the unclipped value
__device__ __forceinline__ float ippm_weight(float p) { return p > 0.501f ? 1.0f : (p < 0.499f ? 0.0f : 0.5f); }
// same test on log-odds: wt = ln(0.501/0.499)
__device__ __forceinline__ float ippm_weight_l(float l, float wt) { return l > wt ? 1.0f : (l < -wt ? 0.0f : 0.5f); }

__device__ __forceinline__ float ippm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Philox4x32-10 (Salmon et al., SC'11); mirrored in oracle/ipp_oracle.py::philox4x32
struct Philox4 {
  uint32_t v[4];
};
__host__ __device__ __forceinline__ Philox4 ippm_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                        uint32_t k1) {
#ifndef IPPM_X_PHILOX_ROUNDS      // measurement-only variants (make VARIANT=ph2 EXTRA=-DIPPM_X_PHILOX_ROUNDS=2): what K3 takes with a cheaper generator
#define IPPM_X_PHILOX_ROUNDS 10
#endif
#pragma unroll
  for (int r = 0; r < IPPM_X_PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#ifdef __HIP_DEVICE_COMPILE__
    // three-input xor in one instruction (gfx950's v_bitop3_b32, truth table 0x96): the compiler leaves these as two v_xor each,
    // and the ten rounds of a call are most of K3's instruction stream
    uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
#else
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
#endif
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}
#define IPPM_DOMAIN_FLIP 0u
#define IPPM_DOMAIN_ACTION 1u
#define IPPM_DOMAIN_COMM 2u
__host__ __device__ __forceinline__ uint32_t ippm_stream_word(uint32_t agent, uint32_t stage, uint32_t domain) {
  return (agent & 0xFFu) | ((stage & 0xFFFFu) << 8) | ((domain & 0xFFu) << 24);
}

// lattice index of a position (agent/state_space.py:53-57)
// exact floor(n / d) for 0 <= n < 2^20, 0 < d <= 2^14 through one float reciprocal ((n + 1/2) / d is never within 1e-6 of an
// integer); an integer division by a run-time divisor is a ~35-instruction sequence, and the serial agent loop of K1 holds thirty
__device__ __forceinline__ int ippm_div_small(int n, float inv_d) { return (int)(((float)n + 0.5f) * inv_d); }
__device__ __forceinline__ void ippm_pos_to_index(const ippm_config* c, int px, int py, int pz, int& ix, int& iy,
                                                  int& iz) {
  const float inv = __builtin_amdgcn_rcpf((float)c->spacing);
  ix = ippm_div_small(px, inv);
  iy = ippm_div_small(py, inv);
  iz = ippm_div_small(pz, inv) - 1;
}
__device__ __forceinline__ int ippm_alt_index(const ippm_config* c, int pz) {
  const int d = pz - c->min_altitude;
  int k = d < 0 ? -1 : ippm_div_small(d, __builtin_amdgcn_rcpf((float)c->spacing));
  return k < 0 ? 0 : (k >= c->space_z ? c->space_z - 1 : k);
}

// Camera.project_field_of_view with host-tabulated centre cells / radii (sensors/cameras.py:62-77)
__device__ __forceinline__ void ippm_footprint_rect(const ippm_config* c, int px, int py, int pz, int* clipped,
                                                    int* full) {
  const float inv = __builtin_amdgcn_rcpf((float)c->spacing);
  // (positions off the lattice only occur for candidates that are masked out: clamped, never used)
  int ix = min(max(ippm_div_small(max(px, 0), inv), 0), IPPM_MAX_LATTICE - 1), iy = min(max(ippm_div_small(max(py, 0), inv), 0), IPPM_MAX_LATTICE - 1);
  const int k = ippm_alt_index(c, pz);
  int xl = c->centre_x[ix] - c->radius_x[k], xr = c->centre_x[ix] + c->radius_x[k];
  int yu = c->centre_y[iy] - c->radius_y[k], yd = c->centre_y[iy] + c->radius_y[k];
  if (full) { full[0] = yu; full[1] = yd; full[2] = xl; full[3] = xr; }
  int gx1 = c->grid_x - 1, gy1 = c->grid_y - 1;
  clipped[0] = min(max(yu, 0), gy1);
  clipped[1] = min(max(yd, 0), gy1);
  clipped[2] = min(max(xl, 0), gx1);
  clipped[3] = min(max(xr, 0), gx1);
#ifdef IPPM_X_ALIGN_FOOTPRINTS   // measurement-only variant (make VARIANT=alignfp EXTRA=-DIPPM_X_ALIGN_FOOTPRINTS=32): every footprint is
  {                              // shifted left onto a multiple of that many cells (32 cells = one 128-byte line), same size -- what the
    const int sh = clipped[0] % IPPM_X_ALIGN_FOOTPRINTS;   // map kernels would take if footprint rows started on line boundaries
    clipped[0] -= sh; clipped[1] -= sh;
  }
#endif
}


// reward of one env from the sums K5 accumulated (utils/reward.py:25-40,74-82); rolls the running weighted entropy T forward
__device__ __forceinline__ void ippm_reward_finalize_env(const ippm_config* __restrict__ c, double* __restrict__ sums,
                                                         float* __restrict__ reward, int e) {
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

// ---- legacy NumPy MT19937: first outputs of RandomState(seed) and the masked-rejection bounded draw ----
#define MT_NOUT 16
__host__ __device__ inline void ippm_mt19937_first_outputs(uint32_t seed, uint32_t* out) {
  // init_genrand recurrence; output k of the first twist needs old[k], old[k+1], old[k+397]
  uint32_t lo[MT_NOUT + 1], hi[MT_NOUT];
  uint32_t s = seed;
#pragma unroll
  for (int pos = 0; pos <= MT_NOUT; ++pos) {
    lo[pos] = s;
    s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)pos + 1u;
  }
  for (int pos = MT_NOUT + 1; pos < 397; ++pos) s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)pos + 1u;
#pragma unroll
  for (int k = 0; k < MT_NOUT; ++k) {
    hi[k] = s;
    s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(397 + k) + 1u;
  }
#pragma unroll
  for (int k = 0; k < MT_NOUT; ++k) {
    uint32_t y = (lo[k] & 0x80000000u) | (lo[k + 1] & 0x7fffffffu);
    uint32_t v = hi[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    v ^= v >> 11;
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= v >> 18;
    out[k] = v;
  }
}

// RandomState.randint(low, low+rng+1): masked rejection on 32-bit outputs (numpy legacy bounded integers)
__host__ __device__ inline uint32_t ippm_mt_bounded(const uint32_t* out, int& cursor, uint32_t rng) {
  uint32_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t v = 0;
  while (cursor < MT_NOUT) {
    v = out[cursor++] & mask;
    if (v <= rng) return v;
  }
  return v <= rng ? v : rng;  // unreachable in practice: 16 rejections in a row
}

// agent/state_space.py:29-32
__host__ __device__ inline void ippm_start_state(int env_seed, int64_t episode, int agent, int spacing, int space_x,
                                                 int space_y, int* out3) {
  uint32_t out[MT_NOUT];
  ippm_mt19937_first_outputs((uint32_t)((int64_t)env_seed * episode * (int64_t)agent), out);
  int cur = 0;
  out3[0] = spacing * (int)ippm_mt_bounded(out, cur, (uint32_t)(space_x - 1));
  out3[1] = spacing * (int)ippm_mt_bounded(out, cur, (uint32_t)(space_y - 1));
  out3[2] = 15;
}

// mapping/ground_truths.py:43-48: np.random.seed(episode); randint(4); randint(30, 61)
__host__ __device__ inline void ippm_truth_params(int64_t episode, int* split, int* pct) {
  uint32_t out[MT_NOUT];
  ippm_mt19937_first_outputs((uint32_t)episode, out);
  int cur = 0;
  *split = (int)ippm_mt_bounded(out, cur, 3u);
  *pct = 30 + (int)ippm_mt_bounded(out, cur, 30u);
}

#endif  // __HIPCC__
