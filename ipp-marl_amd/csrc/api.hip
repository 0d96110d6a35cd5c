// Context management, error reporting and host-side helpers of libippmarl.so.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ippm_internal.h"

static thread_local std::string g_last_error;
void ippm_set_error(const std::string& msg) { g_last_error = msg; }
int ippm_check_hip(hipError_t err, const char* what) {
  if (err == hipSuccess) return 0;
  g_last_error = std::string(what) + ": " + hipGetErrorString(err);
  return -100 - (int)err;
}

extern "C" const char* ippm_last_error(void) { return g_last_error.c_str(); }
extern "C" int ippm_version(void) { return IPPM_VERSION; }
extern "C" int ippm_config_size(void) { return (int)sizeof(ippm_config); }

// Exact area-average weights, the definition used for cv2.resize(INTER_AREA) (utils/state.py:22-41):
// output bin o covers source interval [o*s, (o+1)*s), s = n_src/n_dst; weight = overlap / s.
// Same float64 expressions as oracle/ipp_oracle.py::area_weights.  Per source index: first bin, weight
// into it and into the next bin (n_src >= n_dst => at most two bins).
extern "C" int ippm_area_weights(int32_t n_src, int32_t n_dst, int32_t* bin0, float* w0, float* w1) {
  if (n_src < n_dst || n_dst <= 0) { ippm_set_error("ippm_area_weights: needs n_src >= n_dst > 0"); return -1; }
  const double s = (double)n_src / (double)n_dst;
  std::vector<double> W((size_t)n_dst * n_src, 0.0);
  for (int o = 0; o < n_dst; ++o) {
    const double lo = o * s, hi = (o + 1) * s;
    const int i1 = std::min(n_src, (int)std::ceil(hi));
    for (int i = (int)std::floor(lo); i < i1; ++i)
      W[(size_t)o * n_src + i] = std::max(0.0, std::min(hi, (double)i + 1) - std::max(lo, (double)i)) / s;
  }
  for (int i = 0; i < n_src; ++i) {
    int b = -1;
    for (int o = 0; o < n_dst; ++o)
      if (W[(size_t)o * n_src + i] > 1e-12) { b = o; break; }
    if (b < 0) b = std::min(n_dst - 1, (int)(i / s));
    bin0[i] = b;
    w0[i] = (float)W[(size_t)b * n_src + i];
    w1[i] = b + 1 < n_dst ? (float)W[(size_t)(b + 1) * n_src + i] : 0.f;
  }
  return 0;
}

// host-callable mirrors of device RNG pieces so that CPU-only tests can pin them to NumPy / the oracle
extern "C" int ippm_host_philox(const uint32_t* ctr_key6, uint32_t* out4) {
  Philox4 r = ippm_philox(ctr_key6[0], ctr_key6[1], ctr_key6[2], ctr_key6[3], ctr_key6[4], ctr_key6[5]);
  for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
  return 0;
}
extern "C" int ippm_host_start_state(int32_t env_seed, int64_t episode, int32_t agent, int32_t spacing, int32_t space_x,
                                     int32_t space_y, int32_t* out3) {
  ippm_start_state(env_seed, episode, agent, spacing, space_x, space_y, out3);
  return 0;
}
extern "C" int ippm_host_truth_params(int64_t episode, int32_t* out2) {
  ippm_truth_params(episode, &out2[0], &out2[1]);
  return 0;
}

// Workgroup shape of the env-only step's K3 (wavefronts per workgroup x loads in flight per lane) and the order of its workgroups, by the width of the widest
// footprint row in 4-cell groups and the maps' storage layout; called at ippm_ctx_create and whenever the layout changes (ippm_set_map_layout).
static void ippm_resolve_k3_shape(ippm_ctx* ctx) {
  const ippm_config& c = ctx->cfg;
  auto knob = [](const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; };
  int wmax = 1;
  for (int k = 0; k < c.space_z; ++k) wmax = std::max(wmax, (2 * c.radius_y[k] + 3) / 4 + 1);
  const bool narrow = wmax <= 64;
  // Row-major, measured on one allocation per grid, alternating episodes (tools/ab_knobs.py, profiles/r05/k3_workgroup_shapes.txt): 256^2 (W = 23): (4,3) 35.2 us,
  // (2,2) 34.4; 512^2 (W = 46): (4,3) 67.6, (2,2) 64.2, (1,3) 69.9; 1024^2 (W = 91): (4,3) 94.5, (2,3) 93.8, (2,2) 100.4, (4,4) 162.7.  Short rows: many small
  // workgroups with two loads in flight; long rows (a load instruction no longer spans a row segment): fewer, with three.
  // ... and the order of its workgroups: with rows of up to 32 groups (256^2) consecutive workgroups take the agents of an env (a map's parts n workgroups
  // apart): 33.94 - 34.04 -> 33.54 - 33.70 us in four alternating processes; 512^2 63.0 -> 63.5, 1024^2 91.3 -> 96.6: the parts of a footprint first there.
  int wpg = narrow ? 2 : 4, chn = narrow ? 2 : 3, go = wmax <= 32 ? 1 : 0;
  // Tile storage on 256^2-class grids: whole lines cost the memory side less, and a third load in flight per lane pays (profiles/r06/tile_storage_ab.txt:
  // 2048 envs x 4 UAVs (2,2,1) 66.2 us, (1,3,0) 57.7, (2,3,0) 57.7, (2,4,0) 60.8; 1024 envs x 8 UAVs 66.2 / 58.1 / 61.6 / 61.3); at 512^2 (2,2,0) stays (239 us; (1,3,0) 245).
  if (ctx->tl && wmax <= 32) { wpg = 1; chn = 3; go = 0; }
  ctx->k3_wpg = knob("IPPM_K3_WPG", wpg);
  ctx->k3_chn = knob("IPPM_K3_CHN", chn);
  ctx->k3_go = knob("IPPM_K3_GO", go);
}

extern "C" int ippm_ctx_create(const ippm_config* cfg, ippm_ctx** out) {
  if (!cfg || !out) { ippm_set_error("ippm_ctx_create: null argument"); return -1; }
  const ippm_config& c = *cfg;
  auto bad = [](const char* m) { ippm_set_error(std::string("ippm_ctx_create: ") + m); return -2; };
  if (c.n_agents < 1 || c.n_agents > IPPM_MAX_AGENTS) return bad("n_agents out of range");
  if (c.grid_x < 1 || c.grid_y < 1) return bad("empty grid");
  if (c.space_x < 1 || c.space_x > IPPM_MAX_LATTICE || c.space_y < 1 || c.space_y > IPPM_MAX_LATTICE) return bad("lattice too large");
  if (c.space_z < 1 || c.space_z > IPPM_MAX_Z) return bad("too many altitude levels");
  if (!(c.n_actions == 4 || c.n_actions == 6 || c.n_actions == 9 || c.n_actions == 27)) return bad("num_actions must be 4, 6, 9 or 27");
  if (!(c.logit_clip > 0.f) || !(c.logit_weight_thr > 0.f)) return bad("logit_clip / logit_weight_thr not set");
  if (!(c.prior > 0.f && c.prior < 1.f)) return bad("mapping.prior must lie in (0, 1)");
  // prior != 0.5 is the explicit slow path: every fusion then walks the whole grid (fuse.hip, SHIFT)
  if (c.tile_stride % 4 != 0) return bad("tile_stride must be a multiple of 4");
  for (int k = 0; k < c.space_z; ++k)
    if (2 * c.radius_x[k] > c.tile_stride || 2 * c.radius_y[k] + 3 > c.tile_stride) return bad("tile_stride too small for the footprint");
  if (c.spacing <= 0 || c.budget <= 0) return bad("spacing and budget must be positive");
  // lattice indices come from ippm_div_small (one float reciprocal): exact for 0 <= metres < 2^20 and spacing <= 2^14 only
  if (c.spacing > (1 << 14)) return bad("spacing above 16384 m");
  if (c.x_dim_m < 0 || c.y_dim_m < 0 || c.x_dim_m >= (1 << 20) || c.y_dim_m >= (1 << 20)) return bad("x_dim / y_dim must lie in [0, 2^20) metres");
  if (c.min_altitude < 0 || (long long)c.min_altitude + (long long)c.space_z * c.spacing >= (1 << 20)) return bad("altitudes must lie in [0, 2^20) metres");
  ippm_ctx* ctx = new ippm_ctx();
  std::memset(ctx, 0, sizeof(*ctx));
  ctx->cfg = c;
  // 16-byte lane groups (4 grid-aligned cells of a row) for every grid whose 4-cell groups span at most two of the 11 column
  // bins of the area sums (grid_y >= 44); narrower grids take the one-cell-per-lane instantiations.  The grid need not be a
  // multiple of 4 wide: rows then start at addresses that are only 4-byte aligned (gfx950 serves 16-byte accesses there,
  // tools/probe/unaligned_probe.cpp), the last group of a row hangs over into the next row and is stored cell by cell, and a
  // group's four Philox words / truth bits straddle two counter values / two bytes (the reference's default 493 x 493).
  ctx->vec = (c.grid_y >= 4 * IPPM_FEAT) ? 4 : 1;
  // tuning knobs are read once, here: the size of the caller's work buffer, the plan kernel's item layout and the fusion launch
  // all follow from them, and a value that changed between two calls would make them disagree
  auto knob = [](const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; };
  ctx->knob_wave_rows = knob("IPPM_FUSE_WAVE_ROWS", 0);
  ctx->knob_persist = knob("IPPM_FUSE_PERSIST", 0);
  ctx->knob_nowork = knob("IPPM_FUSE_NOWORK", 0);
  ctx->knob_split = knob("IPPM_FUSE_SPLIT", 0);
  ctx->knob_tile_waves = knob("IPPM_TILE_WAVES", 0);
  ctx->knob_tile_rotate = knob("IPPM_TILE_ROTATE", -1);   // -1: by the launch (fuse_tiles.hip); 0: off; k: groups of 2^(k-1) wavefronts
  ctx->knob_plan_builders = knob("IPPM_PLAN_BUILDERS", 0);
  // Workgroup shape of the env-only step's K3 (wavefronts per workgroup x loads in flight per lane), by the width of the widest
  // footprint row in 4-cell groups.  Measured on one allocation per grid, alternating episodes (tools/ab_knobs.py,
  // profiles/r05/k3_workgroup_shapes.txt): 256^2 (W = 23): (4,3) 35.2 us, (2,2) 34.4; 512^2 (W = 46): (4,3) 67.6, (2,2) 64.2, (1,3)
  // 69.9; 1024^2 (W = 91): (4,3) 94.5, (2,3) 93.8, (2,2) 100.4, (4,4) 162.7.  Short rows: many small workgroups with two loads in
  // flight; long rows (a load instruction no longer spans a row segment): fewer, with three.
  ippm_resolve_k3_shape(ctx);
  // the tile fusion's column intervals rounded outwards to whole 128-byte lines (step_small.hip, tile_build_map): 1 on, 0 off, default: on
  // for rows of at least 512 cells.  Measured (round 6, profiles/r06/tile_round_ab.txt): 512^2 x 8 UAVs fusion 1045 -> 1024 us and the K3
  // behind it 286 -> 275; 256^2 x 4 UAVs fusion 74.7 -> 83-87 us (a 90-cell row grows from 3.7 to 4.7 lines' worth of lane-loads there)
  ctx->knob_tile_round = knob("IPPM_TILE_ROUND", -1);
  if (ctx->knob_tile_round < 0) ctx->knob_tile_round = ctx->cfg.grid_y >= 512 ? 1 : 0;
  ctx->knob_k3_round = knob("IPPM_K3_ROUND", -1);        // the same for K3's row segments (env_step.hip)
  if (ctx->knob_k3_round < 0) ctx->knob_k3_round = ctx->cfg.grid_y >= 512 ? 1 : 0;
  ctx->knob_reset_align = knob("IPPM_RESET_ALIGN", 32);   // cells the reset's fill boxes are rounded outwards to (32 = a 128-byte line; 0: not)
  if (ctx->knob_reset_align & (ctx->knob_reset_align - 1)) ctx->knob_reset_align = 32;
  ctx->knob_terrain_one_launch = knob("IPPM_TERRAIN_ONE_LAUNCH", 0);   // 1: ippm_terrain_truth's second pass as one launch (terrain.hip: measured, no gain)
  ctx->knob_k3_dense = knob("IPPM_K3_DENSE", 1);   // 0: the power-of-two lane layout of round 3 (A/B: tools/ab_knobs.py)
  ctx->tiles = (ctx->vec == 4 && c.logit_prior == 0.f && c.grid_x < 32768 && c.grid_y <= 1024 && !knob("IPPM_NO_TILES", 0)) ? 1 : 0;
  // tile storage of the maps (ippm_set_map_layout); IPPM_MAP_TILED=1 turns it on at creation where the configuration can take it
  ctx->tl = (knob("IPPM_MAP_TILED", 0) > 0 && ippm_tile_storage_ok(ctx)) ? 1 : 0;
  ippm_resolve_k3_shape(ctx);
  int rc = ippm_check_hip(hipMalloc(&ctx->dcfg, sizeof(ippm_config)), "hipMalloc(cfg)");
  if (!rc) rc = ippm_check_hip(hipMemcpy(ctx->dcfg, cfg, sizeof(ippm_config), hipMemcpyHostToDevice), "hipMemcpy(cfg)");
  if (!rc) rc = ippm_check_hip(hipMalloc(&ctx->dcounters, IPPM_COUNTER_SLOTS * 8 * sizeof(unsigned long long)), "hipMalloc(counters)");
  if (!rc) rc = ippm_check_hip(hipMemset(ctx->dcounters, 0, IPPM_COUNTER_SLOTS * 8 * sizeof(unsigned long long)), "hipMemset(counters)");
  if (!rc) {   // the 1024-th roots of unity for the terrain transforms, float64 rounded once
    std::vector<float2> roots(1024);
    for (int k = 0; k < 1024; ++k) {
      const double a = 2.0 * M_PI * (double)k / 1024.0;
      roots[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (int q = 0; q < 4; ++q) {   // exact values on the axes
      roots[256 * q] = make_float2(q == 0 ? 1.f : (q == 2 ? -1.f : 0.f), q == 1 ? 1.f : (q == 3 ? -1.f : 0.f));
    }
    rc = ippm_check_hip(hipMalloc(&ctx->d_roots, sizeof(float2) * 1024), "hipMalloc(roots)");
    if (!rc) rc = ippm_check_hip(hipMemcpy(ctx->d_roots, roots.data(), sizeof(float2) * 1024, hipMemcpyHostToDevice), "hipMemcpy(roots)");
  }
  if (rc) { ippm_ctx_destroy(ctx); return rc; }
  *out = ctx;
  return 0;
}

extern "C" int ippm_set_map_layout(ippm_ctx* ctx, int32_t tiled) {
  if (!ctx) { ippm_set_error("ippm_set_map_layout: null context"); return -1; }
  if (tiled && !ippm_tile_storage_ok(ctx)) {
    ippm_set_error("ippm_set_map_layout: tile storage needs the tile form of the fusion (prior 0.5, grid_y >= 44) and a grid of whole tiles (grid_x % 4 == 0, grid_y % 8 == 0)");
    return -2;
  }
  ctx->tl = tiled ? 1 : 0;
  ippm_resolve_k3_shape(ctx);   // (K3's workgroup shape goes by the layout on 256^2-class grids)
  return 0;
}

// Where tile storage has been measured to pay (profiles/r06/tile_storage_ab.txt; MI355X, env-only step, alternating processes on one box; step = steady state):
//   256^2 x 4 UAVs:  512 envs (0.7 GB of maps) equal; 1024 envs (1.3 GB) K3 33.3 -> 36.2 us, fusion 74.3 -> 77.6: step +3 %; 2048 envs (2.7 GB) K3 71 -> 66, fusion 172 -> 141: -7 %;
//                    4096 envs (5.4 GB) K3 152 -> 133, fusion 367 -> 292: -15 %
//   256^2 x 8 UAVs x 1024 envs (2.4 GB)   K3 73 -> 66, fusion 285 -> 220, reset fill 330 -> 255: step -10 %
//   512^2 x 4 UAVs x 1024 envs (5.2 GB)   K3 121 -> 114, fusion 291 -> 259: -6 %
//   512^2 x 8 UAVs: 256 envs (2.4 GB) -3 %; 1024 envs (9.4 GB, BASELINE config 4's per-GPU shape) K3 256 -> 239, fusion 970 -> 873, fill 1300 -> 1100: -8 .. -16 %
//   1024^2 x 16 UAVs x 64 envs (4.6 GB)   K3 101.5 -> 100.1, fusion 592 -> 602: +3 % (rows of up to 360 cells = 11-12 lines, written whole by the rounded row segments already)
// Row-major rows cost the more per cell the more maps a launch ranges over -- per env the fusion takes 0.073 us at 1024 envs of config 2's shape, 0.084 at 2048, 0.090 at
// 4096 -- while the tile walk gets cheaper (0.076, 0.069, 0.071): whole lines keep their price, partial lines do not.  Below ~2 GB of maps the 8-11 % more lane-loads of a tile walk
// on small footprints are what shows.  Hence: tiles when the batch's maps take 2 GB or more and footprint rows are at most 256 cells.
extern "C" int ippm_map_layout_advice(ippm_ctx* ctx, int32_t n_envs, int32_t* tiled) {
  if (!ctx || !tiled) { ippm_set_error("ippm_map_layout_advice: null argument"); return -1; }
  int wmax = 0;
  for (int k = 0; k < ctx->cfg.space_z; ++k) wmax = std::max(wmax, 2 * ctx->cfg.radius_y[k]);
  const double map_bytes = 4.0 * ctx->cfg.grid_x * ctx->cfg.grid_y * (ctx->cfg.n_agents + 1.0) * std::max(n_envs, 0);
  *tiled = (ippm_tile_storage_ok(ctx) && wmax <= 256 && map_bytes >= 2147483648.0) ? 1 : 0;
  return 0;
}

extern "C" int ippm_map_layout(ippm_ctx* ctx, int32_t* tiled) {
  if (!ctx || !tiled) { ippm_set_error("ippm_map_layout: null argument"); return -1; }
  *tiled = ctx->tl;
  return 0;
}

extern "C" int ippm_set_team_sizes(ippm_ctx* ctx, const int32_t* n_active) {
  if (!ctx) { ippm_set_error("ippm_set_team_sizes: null context"); return -1; }
  ctx->n_active = n_active;
  return 0;
}

extern "C" int ippm_dirty_slab_words(ippm_ctx* ctx, int32_t n_envs, int64_t* words) {
  if (!ctx || !words || n_envs < 0) { ippm_set_error("ippm_dirty_slab_words: bad argument"); return -1; }
  *words = (int64_t)n_envs * (ctx->cfg.n_agents + 1) * 2 * ippm_slab_count(ctx);
  return 0;
}

extern "C" int ippm_set_dirty_slabs(ippm_ctx* ctx, int32_t* slabs) {
  if (!ctx) { ippm_set_error("ippm_set_dirty_slabs: null context"); return -1; }
  if (slabs && ctx->cfg.grid_y > 0xFFFF) { ippm_set_error("ippm_set_dirty_slabs: grid too wide"); return -1; }
  ctx->slabs = slabs;
  return 0;
}

extern "C" int ippm_ctx_destroy(ippm_ctx* ctx) {
  if (!ctx) return 0;
  if (ctx->dcfg) (void)hipFree(ctx->dcfg);
  if (ctx->dcounters) (void)hipFree(ctx->dcounters);
  if (ctx->d_roots) (void)hipFree(ctx->d_roots);
  for (int k = 0; k < IPPM_TIMED_CLASSES; ++k) {
    for (int i = 0; i < 2 * ctx->ev_made[k]; ++i) (void)hipEventDestroy(ctx->ev[k][i]);
    delete[] ctx->ev[k];
  }
  delete ctx;
  return 0;
}

extern "C" int ippm_sync(ippm_ctx* ctx, void* stream) {
  (void)ctx;
  IPPM_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  return 0;
}

// ---- kernel timing ----------------------------------------------------------------------------------------------------
void ippm_timing_events(ippm_ctx* ctx, int cls, const char* name, hipEvent_t* a, hipEvent_t* b) {
  *a = *b = nullptr;
  if (cls < 0 || cls >= IPPM_TIMED_CLASSES || ctx->ev_used[cls] >= IPPM_TIMED_CAP) return;
  if (!ctx->ev[cls]) {
    ctx->ev[cls] = new hipEvent_t[2 * IPPM_TIMED_CAP];
    ctx->ev_made[cls] = 0;
  }
  const int k = ctx->ev_used[cls];
  if (k >= ctx->ev_made[cls]) {
    hipEvent_t ea, eb;
    if (hipEventCreate(&ea) != hipSuccess) return;
    if (hipEventCreate(&eb) != hipSuccess) { (void)hipEventDestroy(ea); return; }
    ctx->ev[cls][2 * k] = ea; ctx->ev[cls][2 * k + 1] = eb;
    ctx->ev_made[cls] = k + 1;
  }
  *a = ctx->ev[cls][2 * k]; *b = ctx->ev[cls][2 * k + 1];
  ctx->ev_used[cls] = k + 1;
  ctx->ev_name[cls] = name;
}

extern "C" int ippm_tile_form(ippm_ctx* ctx, int32_t* yes) {
  if (!ctx || !yes) { ippm_set_error("ippm_tile_form: null argument"); return -1; }
  *yes = ctx->tiles;
  return 0;
}

extern "C" int ippm_kernel_timing(ippm_ctx* ctx, int32_t enable) {
  if (!ctx) { ippm_set_error("ippm_kernel_timing: null context"); return -1; }
  ctx->timing = enable ? 1 : 0;
  return 0;
}

extern "C" int ippm_read_kernel_times(ippm_ctx* ctx, int32_t cls, int32_t reset, int64_t* launches, double* total_us, double* min_us,
                                      char* name, int32_t name_len, void* stream) {
  if (!ctx || !launches || !total_us) { ippm_set_error("ippm_read_kernel_times: null argument"); return -1; }
  if (cls < 0 || cls >= IPPM_TIMED_CLASSES) { ippm_set_error("ippm_read_kernel_times: unknown kernel class"); return -1; }
  IPPM_HIP(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
  double sum = 0.0, mn = 0.0;
  const int n = ctx->ev_used[cls];
  for (int k = 0; k < n; ++k) {
    float ms = 0.f;
    IPPM_HIP(hipEventElapsedTime(&ms, ctx->ev[cls][2 * k], ctx->ev[cls][2 * k + 1]));
    sum += 1e3 * (double)ms;
    mn = (k == 0 || 1e3 * (double)ms < mn) ? 1e3 * (double)ms : mn;
  }
  *launches = n; *total_us = sum;
  if (min_us) *min_us = mn;
  if (name && name_len > 0) {
    // "(k_fuse_rows<4, false, 6, false>)" as written at the launch site -> without the macro's parentheses
    std::string s = ctx->ev_name[cls] ? ctx->ev_name[cls] : "";
    if (s.size() >= 2 && s.front() == '(' && s.back() == ')') s = s.substr(1, s.size() - 2);
    for (size_t at; (at = s.find(" >")) != std::string::npos;) s.erase(at, 1);   // (an empty variadic tail of the launch macro leaves "false >",
    for (size_t at; (at = s.find(" ,")) != std::string::npos;) s.erase(at, 1);   //  a non-empty one "false , 2, 2>")
    std::strncpy(name, s.c_str(), (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (reset) ctx->ev_used[cls] = 0;
  return 0;
}

extern "C" int ippm_read_counters(ippm_ctx* ctx, ippm_counters* out, int reset, void* stream) {
  if (!ctx || !out) { ippm_set_error("ippm_read_counters: null argument"); return -1; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  IPPM_HIP(hipStreamSynchronize(s));
  unsigned long long raw[IPPM_COUNTER_SLOTS * 8], host[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  IPPM_HIP(hipMemcpy(raw, ctx->dcounters, sizeof(raw), hipMemcpyDeviceToHost));
  for (int sl = 0; sl < IPPM_COUNTER_SLOTS; ++sl)
    for (int k = 0; k < 8; ++k) host[k] += raw[sl * 8 + k];
  out->sense_cells = host[0];
  out->fuse_local_cells = host[1];
  out->fuse_local_ops = host[2];
  out->fuse_global_cells = host[3];
  out->fuse_global_ops = host[4];
  out->feature_cells = host[5];
  out->reserved[0] = host[6];
  out->reserved[1] = host[7];
  if (reset) IPPM_HIP(hipMemset(ctx->dcounters, 0, sizeof(raw)));
  return 0;
}

#ifdef IPPM_PLAN_STAMPS
// variant builds only: the raw counter slots (word 7 of each slot holds a phase stamp of k_plan_step's env 0)
extern "C" int ippm_debug_raw_counters(ippm_ctx* ctx, unsigned long long* out512) {
  IPPM_HIP(hipDeviceSynchronize());
  IPPM_HIP(hipMemcpy(out512, ctx->dcounters, IPPM_COUNTER_SLOTS * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}
#endif
