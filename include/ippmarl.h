/*
 * ippmarl.h -- C-ABI of libippmarl.so: the MI355X (gfx950) implementation of the ipp-marl hot path
 * (multi-UAV environment step + COMA target/advantage arithmetic).
 *
 * The reference (dmar-bonn/ipp-marl) is pure Python and has no FFI; its seam is the Python object
 * surface (COMAWrapper / Agent / Mapping / ...).  This header is the boundary a maintainer binds with
 * ctypes; every entry point cites the reference function(s) it replaces, paths
 * relative to marl_framework/.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; ippm_last_error() gives a thread-local message;
 *   - the CALLER owns every buffer: plain device pointers (e.g. torch.Tensor.data_ptr()), contiguous,
 *     dtype/shape as documented; the library allocates only the opaque context (a few KB of tables);
 *   - every launch takes an explicit hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     nothing synchronises except ippm_sync and ippm_read_counters;
 *   - no global mutable state: contexts are independent and may be used from different threads;
 *   - batched over E independent environments ("envs"); N = agents per env; maps are gx rows x gy columns,
 *     row index = x-cell, contiguous index = y-cell (reference: map[xl:xr, yu:yd], mappings.py:46-49).
 *
 * Array layouts (E envs, N agents, A actions, S = cfg.tile_stride)
 *   episode   int64  [E]          episode number of each env (seeds truth, start states, Philox streams)
 *   pos       int32  [E,N,3]      UAV position in metres (x,y,z)
 *   rect      int32  [E,N,4]      clipped footprint [yu,yd,xl,xr], half-open when sliced (cameras.py:62-77)
 *   truth     uint8  [E,TRB]      ground truth, BIT-PACKED: cell (x,y) = bit (x*gy+y) of the env's little-endian bit string,
 *                                 TRB = ceil(gx*gy/32)*4 bytes (ceil((gx*gy+8)/32)*4 when gy is not a multiple of 4: one spare byte
 *                                 behind the last cell's, for the 2-byte loads of groups that straddle a byte)
 *   local     float  [E,N,gx,gy]  per-agent occupancy belief, stored as LOG-ODDS ln(p/(1-p)) (0 = prior 0.5);
 *   global    float  [E,gx,gy]    fused team belief, log-odds.  ippm_logodds_to_prob / ippm_prob_to_logodds
 *                                 convert at the boundary (DESIGN.md "log-odds storage")
 *   code      uint8  [E,N,TB]     last measurement of each agent, 1 bit per cell (1 = observed occupied).  With col =
 *                                 y-(yu & ~3): when gy % 4 == 0 the four grid-aligned cells of a lane group share the low
 *                                 nibble of byte [x-xl][col/4] (row stride S/4, TB = S*S/4); otherwise one byte per cell at
 *                                 [x-xl][col] (row stride S, TB = S*S).  (1-byte-per-cell planes cost K3 27 % of its time.)
 *   flips     uint8  same layout as code; 1 = this cell's observation is flipped (parity mode)
 *   comm      uint8  [E,N,N]      comm[e,i,j] = 1 iff agent i receives agent j's message (diagonal = 1)
 *   mask      uint8  [E,N,A]      action mask after boundary + collision masking
 *   action    int32  [E,N]
 *   ws        int32  [E,N+1,IPPM_WS_WORDS]  fusion workspace (deferred-clamp state + per-step op plan);
 *                                 initialised by ippm_reset_episode, otherwise opaque (zero-fill to start)
 *   sums      double [E,8]        reward accumulators: [0]=S1 [1]=S2 [2]=T (running weighted entropy of the
 *                                 global map), [3..5] scratch; initialised by ippm_reset_episode
 *   area      double [E,N+1,121]  11x11 area sums of every belief map (slot N = the global map): area[b] = sum over cells of
 *                                 (11*overlap of the cell with row bin bx) * (same for by) * p(cell); area / (gx*gy) is the
 *                                 exact area average cv2.resize(map, (11,11), INTER_AREA) of utils/state.py:22-41.  Optional
 *                                 (NULL = not tracked): when given, every kernel that writes a map keeps it up to date, so
 *                                 the K6 feature builders never stream a map; ippm_area_sums rebuilds it from scratch.
 */
#ifndef IPPMARL_H
#define IPPMARL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPPM_VERSION 500
#define IPPM_MAX_AGENTS 16
#define IPPM_MAX_LATTICE 64 /* lattice points per horizontal axis */
#define IPPM_MAX_Z 8        /* altitude levels */
#define IPPM_MAX_ACTIONS 27
#define IPPM_FEAT 11        /* the reference's networks hard-wire an 11x11 lattice (actor/network.py:19-21) */
#define IPPM_ACTOR_PLANES 7
#define IPPM_CRITIC_PLANES 12
#define IPPM_WS_WORDS 160
/* ippm_plan_step flags */
#define IPPM_STEP_COMM 1   /* comm matrix + local-fusion plans */
#define IPPM_STEP_GLOBAL 2 /* global-fusion plan */
#define IPPM_STEP_MOVE 4   /* K1 mask/act/move (+ footprints of the new positions) */
#define IPPM_STEP_TILES 8  /* accepted, implied: the work list's form follows the context (ippm_tile_form), see ippm_plan_step */
/* fault[e] (ippm_plan_step): bits 0..n_agents-1 = agents whose action mask became empty at the LAST step; this bit is sticky
 * (the caller clears it): the env's tile work list did not fit its slice at some step (ruled out by the capacity bound of the
 * list; an env that shows it is no longer fused: ippm_fuse_step skips an overflowed list and counts it in counters.reserved[0]) */
#define IPPM_FAULT_WORK_OVERFLOW 0x40000000
#define IPPM_SENSE_REC_WORDS 8  /* words per agent of ippm_plan_step's rect_next / ippm_sense_step's rect_in */

/* Derived constants, computed on the host in float64 with the reference's expression order
 * (ippmarl/derived.py) and handed over as plain integers/floats. */
typedef struct ippm_config {
  int32_t n_agents;
  int32_t grid_x, grid_y;             /* cells: GridMap.x_dim / y_dim (grid_maps.py:17-50) */
  int32_t space_x, space_y, space_z;  /* lattice dims (state_space.py:16-21) */
  int32_t spacing, min_altitude;      /* metres */
  int32_t x_dim_m, y_dim_m;           /* world size in metres */
  int32_t n_actions;                  /* 4 | 6 | 9 | 27 (action_space.py) */
  int32_t budget;                     /* episode has budget+1 steps */
  int32_t env_seed;                   /* params.environment.seed (start states: state_space.py:29) */
  int32_t tile_stride;                /* S: row stride (and row count) of code/flips tiles, multiple of 4 */
  int32_t fix_range;                  /* 0: per-episode range from {0,15,25,100} (communication_log.py:22-31) */
  int32_t reserved0;
  int32_t centre_x[IPPM_MAX_LATTICE]; /* floor(x_m / res_x) per lattice index (cameras.py:66) */
  int32_t centre_y[IPPM_MAX_LATTICE]; /* floor(y_m / res_x) -- the reference uses res_x for both axes */
  int32_t radius_x[IPPM_MAX_Z];       /* floor(0.5*floor(2 z tan(ax/2)/res_x)) per altitude index */
  int32_t radius_y[IPPM_MAX_Z];
  float logit_meas[IPPM_MAX_Z][2];    /* ln(y/(1-y)) in float32 for y = f32(round(noise,3)), f32(round(1-noise,3)) */
  float meas_value[IPPM_MAX_Z][2];    /* the two measurement values themselves (simulations.py:47-51) */
  uint32_t flip_threshold[IPPM_MAX_Z];/* observation flipped iff philox word < threshold = floor(noise*2^32) */
  float prior;                        /* mapping.prior; != 0.5 takes the explicit full-grid path of the fusion (every message shifts
                                         every cell; chain in float64 registers, one rounding per fusion: maps and returns at 1e-5
                                         like the default) */
  float clip_lo, clip_hi;             /* 1e-4, 0.9999 (mappings.py:110-111, state.py:119-120) */
  float logit_prior;                  /* ln(prior/(1-prior)); 0 for the default prior 0.5 */
  float logit_clip;                   /* ln(clip_hi/(1-clip_hi)) = 9.21024...; the clip is symmetric in log-odds */
  float logit_weight_thr;             /* ln(0.501/0.499): class-weight thresholds of utils/state.py:65-66 */
  double comm_range;                  /* metres, used when fix_range != 0 */
  double failure_rate;                /* link drop probability (communication_log.py:46-54) */
  uint64_t philox_seed;               /* key of the counter-based RNG (production randomness) */
  double gamma, lambda_;              /* TD(lambda) (batch_memory.py:17-21) */
  float logit_noise[IPPM_MAX_Z];      /* ln((1-noise)/noise) from the float64 noise level: the hypothetical updates of
                                         IG_baseline.get_individual_ig use the scalar noise, not a float32 measurement */
  double logit_prior_f64;             /* ln(prior/(1-prior)) in float64, as mappings.py:114 forms it from the Python float: the
                                         whole-grid path of prior != 0.5 runs its chain in float64 registers */
} ippm_config;

typedef struct ippm_ctx ippm_ctx;

/* Algorithmic work counters accumulated by the kernels (cells, not bytes; see DESIGN.md for bytes/cell). */
typedef struct ippm_counters {
  uint64_t sense_cells;        /* K3: footprint cells sensed+updated */
  uint64_t fuse_local_cells;   /* K4: cells read+written by local fusion (union per map) */
  uint64_t fuse_local_ops;     /* K4: sum over (cell, op) pairs = the reference's per-neighbour work */
  uint64_t fuse_global_cells;  /* K5: cells read+written by global fusion */
  uint64_t fuse_global_ops;
  uint64_t feature_cells;      /* K6: map cells streamed by the feature builders */
  uint64_t reserved[2];        /* [0]: fusion launches that were handed a work list of the other form and skipped it (0 in a
                                  correct call sequence) */
} ippm_counters;

const char* ippm_last_error(void);
int ippm_version(void);
int ippm_config_size(void); /* sizeof(ippm_config): lets a binding verify its struct mirror */

int ippm_ctx_create(const ippm_config* cfg, ippm_ctx** out);
int ippm_ctx_destroy(ippm_ctx* ctx);
int ippm_sync(ippm_ctx* ctx, void* stream);
int ippm_read_counters(ippm_ctx* ctx, ippm_counters* out, int reset, void* stream); /* synchronises */

/* Kernel timing (measurement aid; no reference counterpart).  While enabled, every launch of the classes below carries a
 * HIP event pair bound to the dispatch itself (hipExtLaunchKernelGGL start/stop events): their difference is the kernel's own
 * begin-to-end duration -- the figure rocprofv3's kernel trace reports -- with no barrier packet added to the stream.
 * ippm_read_kernel_times synchronises the stream and returns, for one class, the number of timed launches since the last
 * reset, their summed and shortest duration in microseconds and the name of the last kernel launched in that class as the
 * compiler (and rocprofv3) spells it.  At most IPPM_TIMED_CAP launches per class are held between two resetting reads. */
#define IPPM_T_SENSE 0        /* K3: k_sense_tiles / k_sense_update */
#define IPPM_T_FUSE 1         /* K4 + K5: k_fuse_rows */
#define IPPM_T_PLAN 2         /* k_plan_step */
#define IPPM_T_ACTOR_FEAT 3   /* K6 actor */
#define IPPM_T_CRITIC_FEAT 4  /* K6 critic */
#define IPPM_T_RESET 5        /* reset: scalars, truth split, prior fills */
#define IPPM_T_TERRAIN 6      /* random-field synthesis passes */
#define IPPM_T_RESET_MAPS 7   /* ippm_reset_maps: box-limited prior fill + start-position sensing */
#define IPPM_TIMED_CLASSES 8
int ippm_kernel_timing(ippm_ctx* ctx, int32_t enable);
int ippm_read_kernel_times(ippm_ctx* ctx, int32_t cls, int32_t reset, int64_t* launches, double* total_us, double* min_us,
                           char* name, int32_t name_len, void* stream);

/* ---- reset ---------------------------------------------------------------------------------------
 * ippm_reset_episode replaces Mapping.__init__ -> Simulation.simulate_map -> gaussian_random_field
 * (mapping/simulations.py:34-40, mapping/ground_truths.py:42-56: half-plane split drawn from the legacy
 * MT19937 stream np.random.seed(episode)), AgentStateSpace.get_random_agent_state
 * (agent/state_space.py:28-51: RandomState(seed*episode*agent_id)), Mapping.init_priors
 * (mappings.py:126-132) and CommunicationLog.__init__'s per-episode range (communication_log.py:22-31).
 * The MT19937 streams are regenerated on the device, bit-exactly.
 * truth may be NULL (caller supplies its own terrain; split_pct int32 [E,2] is required otherwise);
 * local, global, comm_range_out (float [E]), sums and area may be NULL. */
int ippm_reset_episode(ippm_ctx* ctx, const int64_t* episode, int32_t* pos, uint8_t* truth, float* local,
                       float* global, int32_t* split_pct, float* comm_range_out, int32_t* ws, double* sums,
                       double* area, int32_t n_envs, void* stream);

/* Reset of the maps in one pass for the batched step (16-byte layout, grid_y >= 44): prior fill of local and global maps
 * (Mapping.init_priors, mappings.py:126-132) restricted to the bounding box of what the last episode wrote into each map (kept in
 * `ws` by ippm_plan_step; full != 0: whole maps -- first use, or after maps were written by other entry points), fused with the
 * start-position sensing of every agent (the stage-0 ippm_sense_update: agent.py:43-49 -> mappings.py:32-78), which needs
 * `truth` in place.  Call after ippm_reset_episode (with local = global = NULL there) and after the terrain is generated;
 * writes local, global, code, rect and the boxes in ws.  flips as for ippm_sense_update (NULL: Philox). */
int ippm_reset_maps(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const uint8_t* truth, float* local, float* global,
                    const uint8_t* flips, uint8_t* code, int32_t* rect, int32_t* ws, int32_t full, int32_t n_envs, void* stream);

/* Elementwise conversions between the stored log-odds and the reference's probabilities (n floats). */
int ippm_logodds_to_prob(ippm_ctx* ctx, const float* src, float* dst, int64_t n, void* stream);
int ippm_prob_to_logodds(ippm_ctx* ctx, const float* src, float* dst, int64_t n, void* stream);
/* clip(p, 1e-4, 0.9999) of n stored log-odds in place: the full-grid input clip of one stand-alone fuse_map call. */
int ippm_clamp_logodds(ippm_ctx* ctx, float* maps, int64_t n, void* stream);

/* Plain device-to-device copy with 16-byte lane accesses (n_bytes and both pointers multiples of 16): the streaming-rate
 * denominator bench.py quotes next to the HBM peak. */
int ippm_stream_copy(ippm_ctx* ctx, const void* src, void* dst, int64_t n_bytes, void* stream);

/* ---- K2: Camera.project_field_of_view (sensors/cameras.py:46-79) ------------------------------------ */
int ippm_footprint(ippm_ctx* ctx, const int32_t* pos, int32_t* rect, int32_t* rect_unclipped, int32_t n_envs,
                   void* stream);

/* ---- K3: Mapping.update_grid_map = Simulation.get_measurement + apply_update ---------------------------
 * (mapping/mappings.py:32-78,109-124; mapping/simulations.py:42-65).  Computes rect from pos, senses the
 * footprint (flip source: explicit `flips` tiles, or Philox(seed; episode, agent, stage) when NULL),
 * Bayes-updates local[e,i] in place and stores the measurement codes.  `stage` = 0 for the start-position
 * sensing, t+1 for step t.  agent_sel >= 0 restricts the call to one agent (drop-in single-agent API). */
int ippm_sense_update(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const uint8_t* truth,
                      float* local, const uint8_t* flips, uint8_t* code, int32_t* rect, int32_t* ws,
                      int32_t stage, int32_t agent_sel, int32_t n_envs, void* stream);
/* K3 as the closing kernel of a batched env step (COMAWrapper.steps' `agent.step` calls, coma_wrapper.py:106-134):
 * rect_in (optional) = the sense records ippm_plan_step wrote for the new positions, int32 [E,N,IPPM_SENSE_REC_WORDS] (the
 * footprint and the measurement constants of its altitude: saves the dependent pos -> lattice index -> centre-table and
 * pos -> altitude -> sensor-table loads in front of the map accesses); area (optional) = tracked area sums, updated for local[e,i];
 * sums + reward (optional, together) = complete the reward of the step's global fusion in the same launch
 * (what ippm_reward_finalize does; the fusion of ippm_fuse_step leaves it open). */
int ippm_sense_step(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const uint8_t* truth, float* local,
                    const uint8_t* flips, uint8_t* code, const int32_t* rect_in, int32_t* rect, int32_t* ws, double* area,
                    double* sums, float* reward, int32_t stage, int32_t agent_sel, int32_t n_envs, void* stream);

/* ---- comm: CommunicationLog.get_messages (agent/communication_log.py:39-58) ---------------------------
 * draws: float64 [E,N,N] replacing np.random.random_sample() per ordered pair, or NULL -> Philox.
 * comm_range: float [E] per-env range or NULL -> cfg.comm_range. */
int ippm_comm_matrix(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const float* comm_range,
                     const double* draws, uint8_t* comm, int32_t t, int32_t n_envs, void* stream);

/* ---- K4: Agent.receive_messages -> Mapping.fuse_map(..., "local") (agent/agent.py:62-71,
 * mappings.py:82-89).  For each agent i the measurements of the received agents j != i are fused into
 * local[e,i] in ascending j; every map cell is read and written at most once.  `ws` carries the deferred
 * full-grid input clip of the reference (DESIGN.md "deferred clamp").  agent_sel >= 0 fuses only that agent's map
 * (drop-in Agent.receive_messages). */
int ippm_fuse_local(ippm_ctx* ctx, float* local, const uint8_t* code, const int32_t* rect, const int32_t* pos,
                    const uint8_t* comm, int32_t* ws, int32_t agent_sel, int32_t n_envs, void* stream);
/* ippm_comm_matrix followed by ippm_fuse_local for all agents, with the comm matrix and the fusion plans built by ONE
 * small kernel (both sit on the critical path of every step).  Same results as the two calls. */
int ippm_comm_fuse_local(ippm_ctx* ctx, const int64_t* episode, const int32_t* pos, const float* comm_range,
                         const double* draws, uint8_t* comm, float* local, const uint8_t* code, const int32_t* rect,
                         int32_t* ws, int32_t t, int32_t n_envs, void* stream);

/* ---- K5: Mapping.fuse_map(..., "global") + get_global_reward (mappings.py:91-102, utils/reward.py:11-82,
 * utils/state.py:53-121).  Fuses all N measurements into global[e] in place and returns
 * reward[e] = (22*S1/S2 - 0.5, 10*S1/(gx*gy) - 0.17) and sums[e] = (S1, S2, T, ...) in float64, where T is
 * the running weighted entropy of the map (initialised by ippm_reset_episode; re-seed it with
 * ippm_weighted_entropy if the map is replaced from outside). */
int ippm_fuse_global_reward(ippm_ctx* ctx, float* global, const uint8_t* code, const int32_t* rect,
                            const int32_t* pos, int32_t* ws, double* sums, float* reward, int32_t n_envs,
                            void* stream);

/* ---- the batched step in three launches: ippm_plan_step -> ippm_fuse_step -> ippm_sense_step ---------------------------
 * ippm_plan_step: everything of an env step that touches no map, one wavefront per env, selected by `flags`:
 *   IPPM_STEP_COMM    CommunicationLog.get_messages for every agent + the local-fusion plans (as ippm_comm_fuse_local)
 *   IPPM_STEP_GLOBAL  the global-fusion plan (as ippm_fuse_global_reward's first stage)
 *   IPPM_STEP_MOVE    K1 = ippm_mask_act_move on the same positions (comm and the plans see the pre-move ones); also writes
 *                     rect_next int32 [E,N,IPPM_SENSE_REC_WORDS] (optional) = the sense records of the NEW positions for
 *                     ippm_sense_step: {yu, yd, xl, xr (clipped footprint), float bits of the two measurement log-odds minus
 *                     logit(prior) at the new altitude, its flip threshold, 0}.
 *                     Only policies that do not depend on this step's observations (0 explicit, 1 uniform) can share a call
 *                     with COMM/GLOBAL; a learned policy calls MOVE separately after the actor.
 *   work (optional, int32 [ippm_work_words()]): with COMM | GLOBAL the kernel also lists the non-empty work items of the
 *   step's fusion for ippm_fuse_step, every env into its own slice (count + items: written, never accumulated, so there is
 *   nothing to clear between steps).  The list has one of two forms, chosen by the CONTEXT (ippm_tile_form()), not by the
 *   caller: self-contained one-trip tile items (rows x column interval x op mask, each at most 1024 cells) on contexts that have
 *   the tile form -- 16-byte lane groups (grid_y >= 44) and mapping.prior == 0.5 --, runs of rows of a plan's hull (map, run)
 *   on all others.  ippm_fuse_step of the same context consumes whichever form ippm_plan_step wrote, with or without area
 *   sums.  IPPM_STEP_TILES is accepted for source compatibility and changes nothing (it is implied where the tile form exists
 *   and ignored where it does not).  A list that a kernel cannot read (written by another context's plan step, or overflowed)
 *   fuses nothing and is counted in ippm_counters.reserved[0]; an overflowed one also sets IPPM_FAULT_WORK_OVERFLOW in fault[e].
 * ippm_fuse_step: K4 for all local maps and K5 for all global maps from the plans above, in one launch; keeps `area`
 *   (optional) up to date; leaves the reward sums open (ippm_sense_step or ippm_reward_finalize completes them).  With the
 *   `work` list of the same step's ippm_plan_step a fixed number of resident wavefronts strides over exactly the non-empty
 *   items; without it (NULL) the grid enumerates every (map, run) pair.
 * ippm_reward_finalize: reward[e] = (22*S1/S2 - 0.5, 10*S1/(gx*gy) - 0.17) from the accumulated sums (utils/reward.py:25-40). */
/* Mixed team sizes in one batch (BASELINE config 5: "mixed team sizes 2-16 UAVs").  The reference's team size is a per-run
 * parameter (coma_wrapper.py:25-26, missions/episode_generator.py:99-102); here env e of a batch may fly n_active[e] <= n_agents
 * UAVs: it then evolves exactly like a run of the reference with n_agents = n_active[e] (same episode number): agents
 * >= n_active[e] are heard by nobody, hear nobody, do not move, sense or publish, and the agent-id plane of the network inputs is
 * (i + 1) / n_active[e].  n_active: DEVICE int32 [n_envs], caller-owned, read at every launch of the batched step
 * (ippm_plan_step, ippm_sense_step, ippm_reset_maps, ippm_actor_features, ippm_critic_features; ippm_fuse_step does not read it:
 * it executes the plans ippm_plan_step wrote, and an agent that does not fly gets an EMPTY plan at every plan step, so team sizes
 * may change between any two steps); NULL (the default): every env flies n_agents.  Arrays keep their [E, n_agents, ...] strides;
 * rows of inactive agents are zero (observations, critic states) or not meaningful.
 * The single-purpose entry points (ippm_comm_matrix, ippm_fuse_local, ippm_ig_*, ...) do not look at it. */
int ippm_set_team_sizes(ippm_ctx* ctx, const int32_t* n_active);
/* Dirty slabs: what the episode reset has to fill.  Without them (the default) ippm_plan_step keeps ONE bounding box per map of
 * everything fusions and sensing wrote since the episode's reset (ws words 6-7, 14-15) and ippm_reset_maps writes the prior into that
 * box: 51 % of a local map and 87 % of the global one for 38 % / 58 % written at BASELINE config 2.  With a registered slab array --
 * slabs: DEVICE int32 [ippm_dirty_slab_words(n_envs)] = [n_envs, n_agents + 1, 2, ceil(grid_x / 16)], caller-owned, any contents (the
 * next ippm_reset_maps with full = 1 initialises it) -- the plan kernel marks, per 16-row slab of a map, the column interval the step's
 * plans and sense records touch (fire-and-forget atomic min / max), and ippm_reset_maps fills every slab's own interval (42 % / 69 %) and
 * re-arms it with the new episode's start footprint.  Maps written by other entry points (ippm_sense_update, ippm_fuse_local, host
 * copies) are not tracked either way: pass full = 1 to the next ippm_reset_maps.  NULL: back to the boxes.  Measured (round 6,
 * BASELINE config 2): the marks cost the plan kernel 3.8 us per step and the fill gains nothing from its fewer bytes, so the Python
 * host registers a slab array only on request (IPPM_DIRTY_SLABS=1).  Mirrors nothing in the
 * reference (mapping/mappings.py:126-132 allocates fresh prior maps per episode). */
int ippm_dirty_slab_words(ippm_ctx* ctx, int32_t n_envs, int64_t* words);
int ippm_set_dirty_slabs(ippm_ctx* ctx, int32_t* slabs);
/* Storage layout of the belief maps (`local`, `global` of every entry point of this context).  0 (the default): row-major, cell (x, y) at
 * float x * grid_y + y -- the reference's numpy layout (mapping/mappings.py:18-25).  1: TILE STORAGE -- 128-byte tiles of 4 rows x 8 cells, the
 * tiles in row-major order: cell (x, y) at float (x >> 2) * 4 grid_y + (y >> 3) * 32 + (x & 3) * 8 + (y & 7).  Same size, same values; every
 * entry point that takes maps reads and writes them in the context's layout (ippm_maps_relayout converts between the two), everything
 * else -- truth bits, code tiles, footprints, plans, area sums, features -- is unchanged.  A footprint then touches whole lines only, which
 * keeps its price however many maps a launch ranges over (DESIGN.md "tile storage").  Needs the tile form (ippm_tile_form) and grid_x % 4 ==
 * 0, grid_y % 8 == 0: rc -2 otherwise.  Set it BEFORE the maps are first written; switching with live maps needs ippm_maps_relayout. */
int ippm_set_map_layout(ippm_ctx* ctx, int32_t tiled);
int ippm_map_layout(ippm_ctx* ctx, int32_t* tiled);
/* 1 where the configuration can take tile storage AND it has been measured to pay for a batch of n_envs envs: the batch's maps take 2 GB or more and
 * footprint rows are at most 256 cells (row-major rows cost the more per cell the more maps a launch ranges over, whole lines keep their price: BASELINE
 * config 4's per-GPU shape -8 .. -16 % per env step, config 2's shape -7 % at 2048 envs and -15 % at 4096 but +3 % at its 1024, config 5's 1024 x 1024
 * grid +3 %) -- what the Python host's map_layout="auto" follows for envs WITHOUT tracked area sums.  The advice is about the env-only kernels
 * (area == NULL): with area sums tracked, a tile walk's lanes fall into a third as many area bins per instruction as a row walk's and their LDS atomics
 * queue up (K3 85 -> 190 us, fusion 198 -> 232 at 2048 envs of config 2's shape), so a training rollout keeps rows. */
int ippm_map_layout_advice(ippm_ctx* ctx, int32_t n_envs, int32_t* tiled);
/* n_maps maps of grid_x * grid_y floats from `src` to `dst` (src != dst): to_tiled = 1 row-major -> tile storage, 0 the other way
 * (whatever the context's own layout is). */
int ippm_maps_relayout(ippm_ctx* ctx, const float* src, float* dst, int32_t n_maps, int32_t to_tiled, void* stream);
int ippm_plan_step(ippm_ctx* ctx, const int64_t* episode, int32_t* pos, const float* comm_range, const double* draws,
                   uint8_t* comm, const int32_t* rect, int32_t* ws, int32_t t, int32_t flags, const float* probs,
                   const int32_t* action_in, int32_t policy, uint8_t* mask, int32_t* action, int32_t* fault,
                   int32_t* rect_next, int32_t* work, int32_t n_envs, void* stream);
int ippm_fuse_step(ippm_ctx* ctx, float* local, float* global, const uint8_t* code, int32_t* ws, double* sums, double* area,
                   const int32_t* work, int32_t n_envs, void* stream);
int ippm_work_words(ippm_ctx* ctx, int32_t n_envs, int64_t* words); /* length of `work` in int32 words for n_envs envs */
int ippm_tile_form(ippm_ctx* ctx, int32_t* yes); /* 1: this configuration has the tile form (IPPM_STEP_TILES) */
int ippm_reward_finalize(ippm_ctx* ctx, double* sums, float* reward, int32_t n_envs, void* stream);

/* Full-grid weighted entropy sum(w(p) H(p)) per map (utils/state.py:53-121, "reward" mode); n_maps maps of
 * gx*gy floats; out float64 [n_maps].  truth != NULL: weights from ground truth ("eval" mode), map m uses
 * truth[m / maps_per_truth]. */
int ippm_weighted_entropy(ippm_ctx* ctx, const float* maps, const uint8_t* truth, int32_t maps_per_truth,
                          double* out, int32_t n_maps, void* stream);

/* get_global_reward on two explicit maps (utils/reward.py:11-82): sums float64 [n_maps,2] = (S1,S2);
 * reward float [n_maps,2] = (relative, absolute) or NULL. */
int ippm_reward_from_maps(ippm_ctx* ctx, const float* before, const float* after, double* sums, float* reward,
                          int32_t n_maps, void* stream);

/* ---- K1: AgentActionSpace.get_action_mask + apply_collision_mask + ActorNetwork.do_eps_exploration +
 * action_to_position (agent/action_space.py:25-589, actor/network.py:90-96, coma_wrapper.py:97-104).
 * Sequential over the agents of an env (agent i is masked against the already-moved j < i).
 * policy: 0 = explicit `action_in`; 1 = uniform over the valid mask (Philox); 2 = sample from
 * probs*mask (Philox, "train"); 3 = argmax of probs*mask ("eval").  probs float [E,N,A] (policy 2/3).
 * fault[e] != 0 when an agent's mask became empty (the reference's torch.multinomial raises there): bit i = agent i, this step's
 * bits replace the last step's (IPPM_FAULT_WORK_OVERFLOW is kept). */
/* AgentActionSpace.get_action_mask (mask_in == NULL) / apply_collision_mask (mask_in = the mask to refine) for a batch
 * of single agents: pos int32 [B,3]; others int32 [B,max_others,3] = positions of already-moved agents, n_others
 * int32 [B] (NULL: none); masks uint8 [B,A] (agent/action_space.py:25-196,309-589). */
int ippm_action_mask(ippm_ctx* ctx, const int32_t* pos, const int32_t* others, const int32_t* n_others,
                     int32_t max_others, const uint8_t* mask_in, uint8_t* mask_out, int32_t* next_pos, int32_t batch,
                     void* stream); /* next_pos int32 [B,A,3] or NULL: action_to_position for every action */

int ippm_mask_act_move(ippm_ctx* ctx, const int64_t* episode, int32_t* pos, const float* probs,
                       const int32_t* action_in, int32_t policy, int32_t t, uint8_t* mask, int32_t* action,
                       int32_t* fault, int32_t n_envs, void* stream);

/* ---- K6: network inputs (actor/transformations.py:14-176, critic/transformations.py:17-132,
 * utils/state.py:22-41).  obs float [E,N,11,11,7]; state float [E,N,11,11,12].
 * The maps enter through their area sums `area` double [E,N+1,121] (tracked by the map kernels, or rebuilt by
 * ippm_area_sums): the feature builders never stream a map.
 * ippm_actor_features must run after the local fusion of step t (uses the fused local maps' sums, the published
 * measurements/rects/positions and comm); ippm_critic_features after the global fusion and K1 of step t with
 * pos_pre = the pre-move positions.
 * ippm_area_sums: area slots of n_maps log-odds maps from scratch (a streaming pass: 16-byte loads, 2 rows in flight);
 *   map m belongs to env m / maps_per_env, slot slot0 + m % maps_per_env (local maps: maps_per_env = N, slot0 = 0; global
 *   maps: maps_per_env = 1, slot0 = N).
 * ippm_area_resize: cv2.resize(src, (11,11), INTER_AREA) of n_arrays plain float arrays [rows,cols] -> dst float
 *   [n_arrays,11,11]; scratch double [n_arrays,121]. */
int ippm_actor_features(ippm_ctx* ctx, const double* area, const uint8_t* code, const int32_t* rect,
                        const int32_t* pos, const uint8_t* comm, int32_t t, float* obs, int32_t n_envs,
                        void* stream);
int ippm_critic_features(ippm_ctx* ctx, const double* area, const int32_t* rect, const int32_t* pos_pre,
                         const int32_t* action, const float* obs, float* state, int32_t n_envs, void* stream);
int ippm_area_sums(ippm_ctx* ctx, const float* maps, double* area, int32_t n_maps, int32_t maps_per_env, int32_t slot0,
                   void* stream);
int ippm_area_resize(ippm_ctx* ctx, const float* src, int32_t rows, int32_t cols, float* dst, double* scratch,
                     int32_t n_arrays, void* stream);
/* calculate_w_entropy on n explicit probabilities (utils/state.py:53-121): grid = clip(p, 1e-4, 0.9999), se = H(grid),
 * weightings = class weight of `target` (NULL: of p itself -- "reward"/"actor"/"global"; the ground truth for "eval"),
 * w_entropy = weightings * se.  Every output may be NULL. */
int ippm_entropy_maps(ippm_ctx* ctx, const float* prob, const float* target, float* w_entropy, float* weightings,
                      float* se, float* grid, int64_t n, void* stream);

/* ---- K7: COMA counterfactual advantage (actor/learner.py:55-95).  probs/q float [B,A], mask uint8 [B,A],
 * action int32 [B] -> advantage float [B], pi_tilde float [B,A] (may be NULL). */
int ippm_coma_advantage(ippm_ctx* ctx, const float* probs, const float* q, const uint8_t* mask,
                        const int32_t* action, float* advantage, float* pi_tilde, int32_t batch, void* stream);

/* col2im of the input gradient of a stride-1, unpadded kernel x kernel convolution evaluated as a GEMM (the learners' conv2:
 * actor/network.py:19-21, critic/network.py:21-23; MIOpen's float32 backward-data kernel runs it at a third of the forward's
 * rate).  cols float [batch*out_h*out_w, kernel*kernel*channels] (tap-major, channel-minor) -> grad_x float
 * [batch, out_h+kernel-1, out_w+kernel-1, channels] (channels-last).  No context needed. */
int ippm_col2im_nhwc(const float* cols, float* grad_x, int32_t batch, int32_t out_h, int32_t out_w, int32_t kernel,
                     int32_t channels, void* stream);

/* activation(conv(x)) of the learners' convnets (actor/network.py:72-80, critic/network.py:33-41) with the convolution run
 * bias-free by the library: y = max(x + bias, 0) IN PLACE on channels-last rows x float [rows, channels]; and its backward
 * pass grad_x = grad_y * (y > 0), grad_bias[c] += sum over rows of grad_x[., c] (grad_bias must be zeroed by the caller).
 * channels % 4 == 0 and channels / 4 divides 256; 16-byte aligned buffers.  No context needed. */
int ippm_bias_relu_nhwc(float* x, const float* bias, int64_t rows, int32_t channels, void* stream);
int ippm_bias_relu_backward_nhwc(const float* grad_y, const float* y, float* grad_x, float* grad_bias, int64_t rows,
                                 int32_t channels, void* stream);

/* ---- K8: BatchMemory.build_td_targets (batch_memory.py:120-162) over `chains` independent transition
 * lists of length `len` (row-major [chains,len]): reward float, done uint8, q_sel float = target critic
 * Q(s_t)[a_t] -> td_target, discounted_return float. */
int ippm_td_lambda(ippm_ctx* ctx, const float* reward, const uint8_t* done, const float* q_sel, float* td_target,
                   float* disc_return, int32_t chains, int32_t len, void* stream);

/* ---- greedy information-gain planner (IG_baseline.py:222-325) and evaluation metrics -------------------------
 * ippm_ig_candidates (K9): gains float [E,N,A] = expected weighted entropy reduction over the footprint each valid action
 * leads to, / 1000 (get_individual_ig); masked actions get 0.  ippm_ig_select (K10): get_relative_ig +
 * get_cell_utilities (when communication != 0) + argmax -> action int32 [E,N]; utilities float [E,N,A] optional.
 * An agent whose candidates all have zero gain gets 0 / 0 = nan relative gains, and np.argmax's rule applies (the first nan
 * is the maximum), as in the reference.
 * ippm_f1_counts: int64 [n_maps,3] = (tp, fp, fn) of the map thresholded at log-odds > logodds_threshold (0 <=> p > 0.5)
 * against the truth (utils/utils.py:64-76: sklearn f1_score(...)[1] = 2tp / (2tp + fp + fn)).  Cells whose evidence
 * cancels exactly sit at p = 0.5 +- rounding noise in the reference, which decides their class there; the threshold lets
 * a caller bracket that. */
int ippm_ig_candidates(ippm_ctx* ctx, const float* local, const int32_t* pos, const uint8_t* mask, float* gains,
                       int32_t n_envs, void* stream);
int ippm_ig_select(ippm_ctx* ctx, const int32_t* pos, const uint8_t* mask, const float* gains, int32_t communication,
                   int32_t* action, float* utilities, int32_t n_envs, void* stream);
int ippm_f1_counts(ippm_ctx* ctx, const float* maps, const uint8_t* truth, int32_t maps_per_truth, float logodds_threshold,
                   int64_t* out, int32_t n_maps, void* stream);

/* Synthetic random-field terrain: the device ends of the spectral synthesis the reference performs for every episode
 * (mapping/ground_truths.py:16-40; mapping/simulations.py:34-40) and then overwrites with the half-plane split.  The
 * caller runs the two FFTs and the sqrt(P(k)) product between the calls (ippmarl/terrain.py uses rocFFT via torch.fft).
 * ippm_terrain_noise: standard-normal white noise float [E,gx,gy] from Philox(seed; episode, cell), so an episode's
 *   terrain is the same in every batch and on every rank.
 * Power-of-two grids (sides 8..1024) do the whole synthesis in the library: the spectrum is drawn directly (the FFT of
 *   real N(0,1) noise is Hermitian complex white noise) and inverted in two LDS passes.
 *   ippm_terrain_spectrum: spec complex64 [E,gx,gy/2+1] = amp[gx,gy/2+1] * bin noise, Philox(seed; episode, bin).
 *   ippm_terrain_field: field float [E,gx,gy] = unnormalised inverse real transform of `spec`, or, when spec is NULL,
 *     of the spectrum ippm_terrain_spectrum would have written for (episode, amp) (drawn in-kernel, never stored);
 *     work = complex64 [E,gy/2+1,gx] scratch.  range_keys (uint32 [E,2], optional) receives each field's (min, max)
 *     as order-preserving keys (key = ~bits for negative floats, bits | 0x80000000 otherwise) for ippm_terrain_pack.
 * ippm_terrain_pack: truth bit = (f - min f)/(max f - min f) >= 0.5 per env (ground_truths.py:32-40), written in the
 *   bit-packed truth layout above; min/max are taken from range_keys when given, else reduced here. */
int ippm_terrain_noise(ippm_ctx* ctx, const int64_t* episode, float* noise, int32_t n_envs, void* stream);
int ippm_terrain_spectrum(ippm_ctx* ctx, const int64_t* episode, const float* amp, float* spec, int32_t n_envs, void* stream);
int ippm_terrain_field(ippm_ctx* ctx, const int64_t* episode, const float* amp, const float* spec, float* work, float* field,
                       uint32_t* range_keys, int32_t n_envs, void* stream);
int ippm_terrain_pack(ippm_ctx* ctx, const float* field, const uint32_t* range_keys, uint8_t* truth, int32_t n_envs, void* stream);
/* ippm_terrain_truth = ippm_terrain_field(spec = NULL) + ippm_terrain_pack without ever storing the field: the second transform
 * pass runs twice (once for the field's min / max, once more for the threshold bits, same arithmetic), which moves half the bytes
 * of writing the field and reading it back.  work: complex64 [E,gy/2+1,gx] scratch; range_keys: uint32 [4*E + 4] scratch (per env:
 * min key, max key, and two words + a trailing record for the one-launch form of the pass, IPPM_TERRAIN_ONE_LAUNCH=1: arrivals, a
 * fault word that stays 0 unless an in-launch wait gave up, the launch's ticket counter). */
int ippm_terrain_truth(ippm_ctx* ctx, const int64_t* episode, const float* amp, float* work, uint32_t* range_keys, uint8_t* truth,
                       int32_t n_envs, void* stream);

/* Host helpers (no GPU needed): exported so that CPU-only tests can pin the device's integer streams and
 * resize weights to NumPy / the oracle. */
int ippm_area_weights(int32_t n_src, int32_t n_dst, int32_t* bin0, float* w0, float* w1);
int ippm_host_philox(const uint32_t* ctr_key6, uint32_t* out4);           /* Philox4x32-10(c0..c3,k0,k1) */
int ippm_host_start_state(int32_t env_seed, int64_t episode, int32_t agent, int32_t spacing, int32_t space_x,
                          int32_t space_y, int32_t* out3);                /* state_space.py:28-51 */
int ippm_host_truth_params(int64_t episode, int32_t* out2);               /* ground_truths.py:43-48 */

#ifdef __cplusplus
}
#endif
#endif /* IPPMARL_H */
