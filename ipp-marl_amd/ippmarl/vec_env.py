"""Batched, device-resident multi-UAV environment: E independent episodes stepped by HIP kernels.

Struct-of-arrays state lives in torch tensors on one GPU (torch is only the allocator/stream provider);
every arithmetic step is a kernel of libippmarl.so.  One ``build_observations`` + ``steps`` pair is what
the reference's COMAWrapper.build_observations / COMAWrapper.steps (coma_wrapper.py:37-183) do for ONE
environment, here for all E at once:

    reset(episodes)                       Mapping/Agent construction + t=0 start sensing  (episode_generator.py:39-47,
                                          agent.py:43-49)
    build_observations(t)                 publish -> comm-range receive/fuse (K4) -> actor features (K6)
    steps(t, ...)                         global fuse + reward (K5), sequential mask/act/move (K1),
                                          critic features (K6), sense + update at the new positions (K3)

Randomness: explicit inputs (flip tiles, actions, comm draws) in parity mode, otherwise the counter-based
Philox4x32-10 keyed by (philox_seed; episode, agent, step) -- independent of how envs are sharded over GPUs.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import os
import time

import numpy as np
import torch

from . import _ffi
from .derived import DerivedConstants

POLICY_EXPLICIT, POLICY_UNIFORM, POLICY_SAMPLE, POLICY_ARGMAX = 0, 1, 2, 3


def hot_layout(shapes, align: int = 2 << 20):
    """Byte spans [(offset, bytes)] of the planes ``shapes`` = ((name, shape, torch dtype), ...) inside one allocation, every plane
    on an ``align`` boundary, and the allocation's size."""
    spans, total = [], 0
    for _, shape, dt in shapes:
        n = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
        spans.append((total, n))
        total += (n + align - 1) // align * align
    return spans, total


PLACEMENT_SLACK_MB = 66      # the k-th candidate of VecEnv.tune_placement asks for 66 * (k % 16) MB more than the planes need


def placement_alive_cap(free_bytes: int, arena_bytes: int) -> int:
    """How many candidate arenas VecEnv.tune_placement keeps allocated at once: what fits in half of the free device memory
    (each with the largest slack), at least the arena in use and one candidate."""
    return max(2, int(0.5 * free_bytes // (arena_bytes + ((PLACEMENT_SLACK_MB * 15) << 20))) + 1)


def rows_view(t: torch.Tensor, tiled: bool) -> torch.Tensor:
    """The [.., gx, gy] row-major picture of maps stored as `t` [.., gx, gy] (a copy when `tiled`: tile (R, C) of a map = its rows 4R..4R+3,
    columns 8C..8C+7, stored as 32 consecutive floats, tiles in row-major order)."""
    if not tiled:
        return t
    gx, gy = t.shape[-2:]
    lead = t.shape[:-2]
    k = len(lead)
    return t.reshape(*lead, gx // 4, gy // 8, 4, 8).permute(*range(k), k, k + 2, k + 1, k + 3).reshape(*lead, gx, gy)


def tiles_view(t: torch.Tensor, tiled: bool) -> torch.Tensor:
    """Inverse of rows_view: the storage form of row-major maps `t` [.., gx, gy]."""
    if not tiled:
        return t
    gx, gy = t.shape[-2:]
    lead = t.shape[:-2]
    k = len(lead)
    return t.reshape(*lead, gx // 4, 4, gy // 8, 8).permute(*range(k), k, k + 2, k + 1, k + 3).reshape(*lead, gx, gy)


def placement_stop_reason(scores, jumps: int = 0) -> Optional[str]:
    """Why VecEnv.tune_placement may stop after the draws ``scores`` (us per step of the two map kernels), or None to go on.
    The two kinds of allocation are 7-8 % apart and each is sharp to ~1 %:
      * a draw well below the MEDIAN of the draws is a fast one among slow ones (below the worst is not enough: a slow outlier among
        slow draws -- 121, 121, 126 -- would end the search on a slow one);
      * when fast draws are the majority the median is a fast score and that rule cannot fire: then BOTH kinds must have shown twice --
        two draws within 2 % of the best, and two draws more than 6 % above it that agree with each other to 2 % (113, 114, 122, 122.5:
        stop at the fourth draw; 121, 121.5, 129 is one slow outlier among slow draws, not a second kind: go on);
      * twelve draws in a row within 2 % of each other -- and, when the search jumps (tune_placement), at least two jumps among them: the
        box has one kind only, nothing to search for."""
    k = len(scores)
    if k < 2:
        return None
    best = min(scores)
    if best < 0.96 * float(np.median(scores)):
        return "a fast allocation found"
    slow = sorted(v for v in scores if v > 1.06 * best)
    if sum(1 for v in scores if v <= 1.02 * best) >= 2 and any(b <= 1.02 * a for a, b in zip(slow, slow[1:])):
        return "a fast allocation found"
    if k >= 12 and max(scores) < 1.02 * best and (jumps == 0 or jumps >= 2):
        return "no spread between the first draws"
    return None


class VecEnv:
    def __init__(self, params: Dict, n_envs: int, device: str = "cuda:0", philox_seed: int = 3, terrain: str = "split",
                 track_area: bool = True, team_sizes=None, map_layout: str = "auto", layout_envs: Optional[int] = None):
        if not torch.cuda.is_available():
            raise _ffi.IppmError("VecEnv needs an AMD GPU (HIP): there is no CPU path for the env step")
        self.params = params
        self.d = DerivedConstants(params, philox_seed=philox_seed)
        self.E = int(n_envs)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.ctx = _ffi.Context(self.d)
        # STORAGE LAYOUT of the belief maps (ippm_set_map_layout): "rows" = row-major [gx, gy], the reference's numpy layout; "tiles" = 128-byte
        # tiles of 4 rows x 8 cells (a footprint then touches whole lines only, DESIGN.md "tile storage"); "auto" = tiles where the library
        # says they pay (ippm_map_layout_advice: the batch's maps take 2 GB or more and footprint rows are at most 256 cells -- BASELINE
        # config 4's per-GPU shape, not config 2's 1024 envs; `layout_envs` = the batch the question is asked for when this env is one
        # sub-batch of it) -- or wherever the configuration can take them with IPPM_MAP_TILED=1, nowhere with IPPM_MAP_TILED=0 (the GPU
        # suite runs under both).  `local` / `glob` ARE the storage: rows_view() gives the [.., gx, gy] picture of either layout
        # (posterior_local / posterior_global go through it).
        if map_layout not in ("auto", "rows", "tiles"):
            raise ValueError(f"map_layout: 'auto', 'rows' or 'tiles', not {map_layout!r}")
        self.tiled = False
        want = map_layout == "tiles"
        if map_layout == "auto":
            forced = os.environ.get("IPPM_MAP_TILED", "")
            if forced in ("0", "1"):
                want = forced == "1"
            else:
                # (the advice is for the env-only kernels.  With tracked area sums -- the training rollout -- tiles are SLOWER: the lanes of a tile
                #  walk fall into a third as many area bins per instruction as the lanes of a row walk, and their LDS atomics queue up: K3 85 -> 190 us,
                #  fusion 198 -> 232 at 2048 envs x 4 UAVs x 256^2, profiles/r06/tile_storage_ab.txt -- so a tracked env keeps rows unless forced)
                advice = np.zeros(1, dtype=np.int32)
                self.ctx.call("ippm_map_layout_advice", int(layout_envs if layout_envs is not None else self.E), advice.ctypes.data)
                want = bool(advice[0]) and not track_area
        if want:
            try:
                self.ctx.call("ippm_set_map_layout", 1)
                self.tiled = True
            except _ffi.IppmError:
                if map_layout == "tiles":
                    raise
        else:
            self.ctx.call("ippm_set_map_layout", 0)
        d, E, dev = self.d, self.E, self.device
        N, A, S = d.n_agents, d.n_actions, d.tile_stride
        z = lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        self.episode = z(E, dtype=torch.int64)
        self.pos = z(E, N, 3, dtype=torch.int32)
        self.pos_pre = z(E, N, 3, dtype=torch.int32)
        self.rect = z(E, N, 4, dtype=torch.int32)
        self.rect_next = z(E, N, _ffi.SENSE_REC_WORDS, dtype=torch.int32)   # sense records of the post-move positions (K1 -> K3)
        # The step's hot planes live in ONE allocation (see tune_placement):
        #   truth  bit-packed ground truth (1 bit per cell)
        #   local, glob  beliefs as float32 log-odds (0 = prior 0.5); posterior_local()/posterior_global() export p
        #   code   packed measurement codes (a nibble per 4-cell group)
        self._hot_shapes = (("local", (E, N, d.grid_x, d.grid_y), torch.float32), ("glob", (E, d.grid_x, d.grid_y), torch.float32),
                            ("code", (E, N, d.tile_bytes), torch.uint8), ("truth", (E, d.truth_bytes), torch.uint8))
        self._place_hot()
        # Mixed team sizes in one batch (BASELINE config 5): env e flies team_sizes[e] <= n_agents UAVs and evolves exactly like a
        # run of the reference with that n_agents (its team size is a per-run parameter: coma_wrapper.py:25-26,
        # missions/episode_generator.py:99-102).  Arrays keep the [E, n_agents, ...] shapes; rows of agents that do not fly are
        # not meaningful (their observations are zero, their sense records empty).  None: every env flies n_agents.
        self.n_active = None
        if team_sizes is not None:
            ts = torch.as_tensor(team_sizes, dtype=torch.int32).reshape(-1)
            if ts.numel() != E or int(ts.min()) < 1 or int(ts.max()) > N:
                raise ValueError(f"team_sizes: {E} values in 1..{N} (params' n_agents is the capacity)")
            self.n_active = ts.to(dev)
            self.ctx.call("ippm_set_team_sizes", self._p(self.n_active))
        # Dirty slabs (IPPM_DIRTY_SLABS=1; env-only form, 16-byte layout): per 16-row slab of every map the column interval written since
        # the episode's reset, marked by the plan kernel and consumed by ippm_reset_maps -- a finer account of what the reset has to fill
        # than one box per map (42 % / 69 % of a local / the global map against 51 % / 87 %; 38 % / 58 % are written).  Built, tested and
        # measured in round 6, and OFF by default: the marks (~70 fire-and-forget atomics per map and step) cost the plan kernel 3.8 us a
        # step (14.8 -> 18.5) and the fill did not get faster for its 18 % fewer bytes (161 against 162 us on the box it was timed on:
        # it issues a store per row and wavefront whatever the interval's width) -- profiles/r06/dirty_slabs_ab.txt.
        self.slabs = None
        if not track_area and d.vec == 4 and os.environ.get("IPPM_DIRTY_SLABS", "0") == "1":
            words = np.zeros(1, dtype=np.int64)
            self.ctx.call("ippm_dirty_slab_words", E, words.ctypes.data)
            self.slabs = z(int(words[0]), dtype=torch.int32)
            self.ctx.call("ippm_set_dirty_slabs", self._p(self.slabs))
        self.comm = z(E, N, N, dtype=torch.uint8)
        self.comm_range = z(E, dtype=torch.float32)
        self.mask = z(E, N, A, dtype=torch.uint8)
        self.action = z(E, N, dtype=torch.int32)
        self.fault = z(E, dtype=torch.int32)
        self.ws = z(E, N + 1, _ffi.WS_WORDS, dtype=torch.int32)
        self.sums = z(E, 8, dtype=torch.float64)
        self.reward = z(E, 2, dtype=torch.float32)
        self.split_pct = z(E, 2, dtype=torch.int32)
        # work list of a step's fusion: the plan kernel lists the non-empty (map, run of rows) items, the fusion kernel's
        # resident wavefronts stride over them
        words = np.zeros(1, dtype=np.int64)
        self.ctx.call("ippm_work_words", E, words.ctypes.data)
        self.work = z(int(words[0]), dtype=torch.int32)
        yes = np.zeros(1, dtype=np.int32)
        self.ctx.call("ippm_tile_form", yes.ctypes.data)
        self._tile_form = bool(yes[0])   # the fusion runs in one-trip tile items (16-byte layout, prior 0.5), area sums tracked or not
        # 11x11 area sums of every map (slot N = global): the input of the K6 feature builders.  track_area=True: K3 / K4 /
        # K5 keep them up to date as they write maps (the batched training path); False: rebuilt by a streaming pass right
        # before the features are needed (env-only stepping never needs them; the single-env drop-in engine, whose maps
        # can be replaced from outside, uses this mode).
        self.track_area = bool(track_area)
        self.area = z(E, N + 1, _ffi.FEAT * _ffi.FEAT, dtype=torch.float64)
        self.obs = None
        self._obs_t = None       # step whose actor observations `obs` holds
        self.state = None
        self.t = 0
        # "split": the half-plane truth the reference flies over (ground_truths.py:42-56); "random_field": the
        # power-law random field it synthesises first (ground_truths.py:25-40), generated on the device (terrain.py)
        if terrain not in ("split", "random_field"):
            raise ValueError(f"unknown terrain {terrain!r}")
        self.terrain = terrain
        self._field = None
        self._pending_t = None   # step whose fusion (K4 + K5) build_observations has already launched
        self._boxes_valid = False  # ws holds, per map, the box of everything written since the last reset (kept by the plan kernel)
        self._profile = False    # time the kernels with events bound to their dispatches (bench.py's roofline legs)
        self._ep_host = None     # pinned staging buffer of reset()'s episode ids, and the event of its last copy
        self._ep_copied = None
        # terrain of the NEXT episodes, synthesised on a side stream while the current episodes are stepped (prefetch_terrain)
        self._side = None
        self._next_ids = None    # host copy of the episode ids whose terrain `_truth_next` holds (or will hold: `_next_ready`)
        self._truth_next = None
        self._episode_next = None
        self._next_host = None
        self._next_ready = None

    # ------------------------------------------------------------------------------------------------
    def _place_hot(self, slack_mb: int = 0):
        """(Re)allocates the hot planes as views of one zeroed device allocation, each plane on a 2 MB boundary."""
        _, total = hot_layout(self._hot_shapes)
        self._use_arena(torch.zeros(total + (slack_mb << 20), dtype=torch.uint8, device=self.device))

    def _use_arena(self, arena: torch.Tensor):
        spans, total = hot_layout(self._hot_shapes)
        if arena.numel() < total or arena.dtype != torch.uint8:
            raise ValueError(f"an arena for this env holds at least {total} bytes (uint8)")
        self._arena = arena
        for (name, shape, dt), (off, n) in zip(self._hot_shapes, spans):
            setattr(self, name, arena[off:off + n].view(dt).view(shape))
        self._boxes_valid = False          # the maps of another allocation: the next reset fills them whole

    @property
    def profile(self) -> bool:
        return self._profile

    @profile.setter
    def profile(self, on: bool):
        """While set, every launch of the timed kernel classes (K3, fusion, plan, K6, reset, terrain) carries a HIP event pair
        bound to the dispatch itself (ippm_kernel_timing): kernel-only durations, no barrier packets on the stream."""
        if bool(on) != self._profile:
            self.ctx.kernel_timing(bool(on))
            self._profile = bool(on)

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _p(self, t):
        return _ffi.ptr(t)

    @property
    def _area_arg(self):
        return self._p(self.area) if self.track_area else None

    def state_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in vars(self).values() if isinstance(t, torch.Tensor))

    def rows_view(self, maps: torch.Tensor) -> torch.Tensor:
        """Row-major [.., gx, gy] picture of maps held in this env's storage layout (the tensor itself when the layout is row-major)."""
        return rows_view(maps, self.tiled)

    def tiles_view(self, maps: torch.Tensor) -> torch.Tensor:
        """Storage form of row-major maps [.., gx, gy] (what may be copied into `local` / `glob`)."""
        return tiles_view(maps, self.tiled)

    def _to_prob(self, logodds: torch.Tensor) -> torch.Tensor:
        logodds = self.rows_view(logodds).contiguous()
        out = torch.empty_like(logodds)
        self.ctx.call("ippm_logodds_to_prob", self._p(logodds), self._p(out), logodds.numel(), self.stream)
        return out

    def posterior_local(self) -> torch.Tensor:
        """Occupancy probabilities of the agents' local maps, float32 [E,N,gx,gy] (the reference's local_map)."""
        return self._to_prob(self.local)

    def posterior_global(self) -> torch.Tensor:
        return self._to_prob(self.glob)

    @property
    def truth_map(self) -> torch.Tensor:
        """Ground truth as row-major uint8 [E, gx, gy] (unpacked on the host: inspection / tests only)."""
        return torch.from_numpy(self.d.unpack_truth(self.truth.cpu().numpy()))

    def footprints(self, pos: Optional[torch.Tensor] = None):
        pos = self.pos if pos is None else pos
        rect = torch.empty(self.E, self.d.n_agents, 4, dtype=torch.int32, device=self.device)
        full = torch.empty_like(rect)
        self.ctx.call("ippm_footprint", self._p(pos), self._p(rect), self._p(full), self.E, self.stream)
        return rect, full

    def rebuild_area(self, local: bool = True, glob: bool = True):
        """Area sums from scratch (streaming pass over the maps): after maps were written from outside, or when the env
        does not track them."""
        N = self.d.n_agents
        if local:
            self.ctx.call("ippm_area_sums", self._p(self.local), self._p(self.area), self.E * N, N, 0, self.stream)
        if glob:
            self.ctx.call("ippm_area_sums", self._p(self.glob), self._p(self.area), self.E, 1, N, self.stream)

    # ------------------------------------------------------------------------------------------------
    def reset(self, episodes, truth: Optional[torch.Tensor] = None, start_positions: Optional[torch.Tensor] = None,
              flips: Optional[torch.Tensor] = None, terrain: Optional[str] = None):
        """Starts episode ``episodes[e]`` in env e (all envs at once) and performs the t=0 start-position sensing.
        ``truth`` (explicit [E,gx,gy] field) overrides ``terrain`` ("split" | "random_field", default: the env's)."""
        d = self.d
        terrain = self.terrain if terrain is None else terrain
        ep = torch.as_tensor(episodes, dtype=torch.int64).reshape(self.E)
        if int(ep.max()) * d.env_seed * max(d.n_agents - 1, 1) >= 2 ** 32 or int(ep.min()) < 0:
            raise ValueError("episode * seed * agent_id must stay below 2**32 (NumPy legacy seeding limit)")
        # episode ids go through a pinned staging buffer: a pageable host-to-device copy stalls the stream for ~40 us per reset
        if self._ep_host is None:
            self._ep_host = torch.empty(self.E, dtype=torch.int64).pin_memory()
        if self._ep_copied is not None:
            self._ep_copied.synchronize()      # the previous reset's copy has read the buffer
        self._ep_host.copy_(ep)
        self.episode.copy_(self._ep_host, non_blocking=True)
        self._ep_copied = torch.cuda.Event()
        self._ep_copied.record()
        # Env-only form: the prior fill of the maps and the start-position sensing are ONE pass (ippm_reset_maps, below, once the
        # terrain is in place) over the box each map was written in; with tracked area sums: full fills here, then a K3 launch.
        one_pass = not self.track_area and d.vec == 4
        self.ctx.call("ippm_reset_episode", self._p(self.episode), self._p(self.pos),
                      self._p(self.truth) if truth is None and terrain == "split" else None,
                      None if one_pass else self._p(self.local), None if one_pass else self._p(self.glob),
                      self._p(self.split_pct), self._p(self.comm_range), self._p(self.ws), self._p(self.sums), self._area_arg,
                      self.E, self.stream)
        if truth is not None:
            packed = d.pack_truth(torch.as_tensor(truth).cpu().numpy().reshape(self.E, d.grid_x, d.grid_y))
            self.truth.copy_(torch.from_numpy(packed).to(self.device))
        elif terrain == "random_field":
            if self._next_ids is not None and np.array_equal(self._next_ids, ep.numpy()):
                # synthesised ahead of time on the side stream (prefetch_terrain): the packed truth planes are 8 MB at config 2,
                # copied rather than swapped so that every recorded launch keeps reading `truth`
                torch.cuda.current_stream(self.device).wait_event(self._next_ready)
                self.truth.copy_(self._truth_next)
            else:
                if self._next_ready is not None:   # the side stream may still be using the generator's scratch buffers
                    torch.cuda.current_stream(self.device).wait_event(self._next_ready)
                self._terrain().generate(self.episode, self.truth, self.stream)
            self._next_ids = None
        elif terrain != "split":
            raise ValueError(f"unknown terrain {terrain!r}")
        if start_positions is not None:
            self.pos.copy_(torch.as_tensor(start_positions).to(self.device, torch.int32))
        self.t = 0
        self._pending_t = None
        self._obs_t = None
        if one_pass:
            self.ctx.call("ippm_reset_maps", self._p(self.episode), self._p(self.pos), self._p(self.truth), self._p(self.local),
                          self._p(self.glob), self._p(flips), self._p(self.code), self._p(self.rect), self._p(self.ws),
                          0 if self._boxes_valid else 1, self.E, self.stream)
            self._boxes_valid = True
        else:
            self._sense(stage=0, flips=flips)

    def _terrain(self):
        if self._field is None:
            from .terrain import RandomFieldTerrain
            self._field = RandomFieldTerrain(self.d, self.ctx, self.device, self.params["sensor"]["simulation"]["cluster_radius"])
        return self._field

    def prefetch_terrain(self, episodes) -> None:
        """Starts synthesising the random-field terrain of ``episodes`` (the ids the NEXT reset will be given) on a side stream,
        so that it runs beside the current episodes' steps instead of in front of the next reset: the synthesis (spectrum, two
        FFT passes, threshold: ~300 us for 1024 fields of 256^2) is latency-bound at 16 wavefronts per CU and needs nothing but
        the episode numbers (mapping/ground_truths.py:16-40 draws the field from the episode's own stream).  A reset that is
        given exactly these ids takes the prefetched planes; any other reset ignores them and synthesises in line."""
        if self.terrain != "random_field":
            return
        d = self.d
        ep = torch.as_tensor(episodes, dtype=torch.int64).reshape(self.E)
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._truth_next = torch.zeros_like(self.truth)
            self._episode_next = torch.zeros_like(self.episode)
            self._next_host = torch.empty(self.E, dtype=torch.int64).pin_memory()
        if self._next_ready is not None:
            self._next_ready.synchronize()     # the previous prefetch has read the staging buffer (and the scratch is free)
        self._next_host.copy_(ep)
        started = torch.cuda.Event()
        started.record(main)                   # after everything queued so far: an in-line synthesis of reset() included
        with torch.cuda.stream(self._side):
            self._side.wait_event(started)
            self._episode_next.copy_(self._next_host, non_blocking=True)
            self._terrain().generate(self._episode_next, self._truth_next, self._side.cuda_stream)
            self._next_ready = torch.cuda.Event()
            self._next_ready.record(self._side)
        self._next_ids = ep.numpy().copy()

    def sense(self, stage: int, flips: Optional[torch.Tensor] = None, agent: int = -1, close_step: bool = False):
        """K3 at the current positions, called from outside the batched step (drop-in Agent / Mapping): the maps are then
        written without the plan kernel's knowledge, so the next reset fills them whole."""
        self._boxes_valid = False
        self._sense(stage, flips, agent, close_step)

    def _sense(self, stage: int, flips: Optional[torch.Tensor] = None, agent: int = -1, close_step: bool = False):
        """K3 at the current positions (stage 0 = start sensing, t+1 = sensing of step t).  ``close_step``: this is the K3
        that ends a batched step -- it takes the footprints K1 projected (rect_next) and completes the step's reward."""
        self.ctx.call("ippm_sense_step", self._p(self.episode), self._p(self.pos), self._p(self.truth), self._p(self.local),
                      self._p(flips), self._p(self.code), self._p(self.rect_next) if close_step else None, self._p(self.rect),
                      self._p(self.ws), self._area_arg, self._p(self.sums) if close_step else None,
                      self._p(self.reward) if close_step else None, stage, agent, self.E, self.stream)

    def comm_matrix(self, t: int, comm_draws: Optional[torch.Tensor] = None):
        self.ctx.call("ippm_comm_matrix", self._p(self.episode), self._p(self.pos), self._p(self.comm_range),
                      self._p(comm_draws), self._p(self.comm), t, self.E, self.stream)

    def fuse_local(self, agent: int = -1):
        """Stand-alone K4 (drop-in Agent.receive_messages); does not track the area sums."""
        if self.n_active is not None:
            raise _ffi.IppmError("fuse_local: the single-purpose entry points know one team size per context (team_sizes is set)")
        self._boxes_valid = False
        self.ctx.call("ippm_fuse_local", self._p(self.local), self._p(self.code), self._p(self.rect), self._p(self.pos),
                      self._p(self.comm), self._p(self.ws), agent, self.E, self.stream)

    def _plan_step(self, t: int, flags: int, comm_draws=None, policy: int = 0, probs=None, actions=None):
        if self._tile_form and flags & (_ffi.STEP_COMM | _ffi.STEP_GLOBAL):
            flags |= _ffi.STEP_TILES   # the fusion takes the work list as one-trip tile items (16-byte layout, prior 0.5)
        self.ctx.call("ippm_plan_step", self._p(self.episode), self._p(self.pos), self._p(self.comm_range), self._p(comm_draws),
                      self._p(self.comm), self._p(self.rect), self._p(self.ws), t, flags, self._p(probs), self._p(actions), policy,
                      self._p(self.mask), self._p(self.action), self._p(self.fault), self._p(self.rect_next), self._p(self.work), self.E,
                      self.stream)

    def _fuse_step(self):
        self.ctx.call("ippm_fuse_step", self._p(self.local), self._p(self.glob), self._p(self.code), self._p(self.ws),
                      self._p(self.sums), self._area_arg, self._p(self.work), self.E, self.stream)

    def _actor_features(self, t: int):
        if self.obs is None:
            self.obs = torch.empty(self.E, self.d.n_agents, _ffi.FEAT, _ffi.FEAT, _ffi.ACTOR_PLANES, dtype=torch.float32,
                                   device=self.device)
        if not self.track_area:
            self.rebuild_area(local=True, glob=False)
        self.ctx.call("ippm_actor_features", self._p(self.area), self._p(self.code), self._p(self.rect), self._p(self.pos),
                      self._p(self.comm), t, self._p(self.obs), self.E, self.stream)
        self._obs_t = t
        return self.obs

    def build_observations(self, t: int, comm_draws: Optional[torch.Tensor] = None, features: bool = True):
        """comm matrix + fusion plans -> local fusion (K4) and global fusion (K5) of the published measurements in one
        launch -> actor observation [E,N,11,11,7] (K6).  The global fusion is logically the first thing steps() does; it
        only reads what the previous K3 wrote and nothing here reads the global map, so it shares K4's launch."""
        if self._pending_t is not None:
            raise _ffi.IppmError(f"build_observations({t}) called twice without steps({self._pending_t}): the measurements of a "
                                 "step can be fused only once")
        self._plan_step(t, _ffi.STEP_COMM | _ffi.STEP_GLOBAL, comm_draws)
        self._fuse_step()
        self._pending_t = t
        return self._actor_features(t) if features else None

    def build_features_only(self, t: int):
        """K6 actor features from the current device state (used by the drop-in transformations)."""
        return self._actor_features(t)

    def steps(self, t: int, policy: int = POLICY_UNIFORM, probs: Optional[torch.Tensor] = None,
              actions: Optional[torch.Tensor] = None, flips: Optional[torch.Tensor] = None, features: bool = True,
              comm_draws: Optional[torch.Tensor] = None):
        """Global fusion + reward of the measurements published this step (K5; already launched by build_observations),
        K1, critic features, K3.

        Returns (reward [E,2] = (relative, absolute), done: bool, state [E,N,11,11,12] or None)."""
        d = self.d
        if features:
            self.pos_pre.copy_(self.pos)
        if probs is not None:
            probs = probs.to(torch.float32).contiguous()
        if actions is not None:
            actions = actions.to(self.device, torch.int32).contiguous()
        if self._pending_t is None:   # stepping without observations (env-only): plans + K1 share one launch
            if policy in (POLICY_SAMPLE, POLICY_ARGMAX):
                raise _ffi.IppmError("a learned policy needs build_observations() before steps()")
            self._plan_step(t, _ffi.STEP_COMM | _ffi.STEP_GLOBAL | _ffi.STEP_MOVE, comm_draws, policy, probs, actions)
            self._fuse_step()
        else:
            if self._pending_t != t:
                raise _ffi.IppmError(f"steps({t}) after build_observations({self._pending_t})")
            self._plan_step(t, _ffi.STEP_MOVE, None, policy, probs, actions)
        self._pending_t = None
        state = None
        if features:
            if self.obs is None or self._obs_t != t:   # the critic state embeds this step's actor observations
                raise _ffi.IppmError(f"steps({t}, features=True) needs build_observations({t}, features=True) first "
                                     f"(observations held: step {self._obs_t})")
            if self.state is None:
                self.state = torch.empty(self.E, d.n_agents, _ffi.FEAT, _ffi.FEAT, _ffi.CRITIC_PLANES, dtype=torch.float32,
                                         device=self.device)
            if not self.track_area:
                self.rebuild_area(local=False, glob=True)
            # rect still holds the pre-move (published) footprints: K3 below overwrites it
            self.ctx.call("ippm_critic_features", self._p(self.area), self._p(self.rect), self._p(self.pos_pre),
                          self._p(self.action), self._p(self.obs), self._p(self.state), self.E, self.stream)
            state = self.state
        self._sense(stage=t + 1, flips=flips, close_step=True)
        self.t = t + 1
        return self.reward, t == d.budget, state

    # ---- greedy information-gain policy (IG_baseline.py:127-148, 222-325) for the whole batch ---------------------
    def ig_actions(self, communication: bool = True) -> torch.Tensor:
        """Actions int32 [E,N] of the greedy expected-information-gain planner on the current local maps: masks against the
        current positions of the agents before each one (the reference's quirk), K9 candidate gains, K10 selection."""
        if self.n_active is not None:
            raise _ffi.IppmError("ig_actions: the greedy planner's kernels know one team size per context (team_sizes is set)")
        d, E, N, A = self.d, self.E, self.d.n_agents, self.d.n_actions
        others = self.pos.view(E, 1, N, 3).expand(E, N, N, 3).contiguous()
        n_others = torch.arange(N, dtype=torch.int32, device=self.device).view(1, N).expand(E, N).contiguous()
        mask = torch.empty(E, N, A, dtype=torch.uint8, device=self.device)
        self.ctx.call("ippm_action_mask", self._p(self.pos), self._p(others), self._p(n_others), N, None, self._p(mask), None, E * N,
                      self.stream)
        gains = torch.empty(E, N, A, dtype=torch.float32, device=self.device)
        self.ctx.call("ippm_ig_candidates", self._p(self.local), self._p(self.pos), self._p(mask), self._p(gains), E, self.stream)
        actions = torch.empty(E, N, dtype=torch.int32, device=self.device)
        self.ctx.call("ippm_ig_select", self._p(self.pos), self._p(mask), self._p(gains), 1 if communication else 0, self._p(actions), None,
                      E, self.stream)
        self.ig_gains, self.ig_mask = gains, mask
        return actions

    # ---- hipGraph replay of the launch-bound part of a random-policy step ------------------------------------
    def capture_step_graphs(self, policy: int = POLICY_UNIFORM):
        """Captures, for every t of an episode, the fixed launch pair {comm + plans + K1} -> {K4 + K5} into a hipGraph (the
        per-step arguments t / stage are baked in, all arrays are fixed device buffers).  The sensing kernel K3 stays an
        ordinary launch so that callers can bracket it with events."""
        graphs = []
        torch.cuda.synchronize(self.device)
        saved = {k: getattr(self, k).clone() for k in ("local", "glob", "ws", "sums", "pos", "comm", "mask", "action", "fault", "reward",
                                                       "area", "rect_next")}
        capture_stream = torch.cuda.Stream(device=self.device)
        for t in range(self.d.budget + 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=capture_stream):
                self._plan_step(t, _ffi.STEP_COMM | _ffi.STEP_GLOBAL | _ffi.STEP_MOVE, None, policy, None, None)
                self._fuse_step()
            graphs.append(g)
        torch.cuda.synchronize(self.device)
        for k, v in saved.items():   # capture does not execute, but keep the state untouched in any case
            getattr(self, k).copy_(v)
        self._graphs = graphs
        return graphs

    def step_graphed(self, t: int):
        """One random-policy env step: graph replay (comm, plans, K1, K4, K5) + K3."""
        self._graphs[t].replay()
        self._sense(stage=t + 1, close_step=True)
        self.t = t + 1
        return self.reward, t == self.d.budget

    # ------------------------------------------------------------------------------------------------
    def tune_placement(self, draws: int = 24) -> Optional[dict]:
        """Draws the allocation of the env's hot planes (maps, code and truth planes: one device allocation) up to ``draws``
        times -- until one is clearly of the fast kind -- and keeps the one on which the step's two map kernels run fastest.

        Measured (tools/placement_probe*.py; config 2, MI355X, ROCm 7.2): a device allocation is either a good or a bad place
        for these planes, for as long as it lives.  With all of them on good allocations the fusion kernel takes 76.5 us and
        K3 35.8; each large plane on a bad one costs the fusion about 8 us (both maps: 90 us) and K3 about 1.5 -- same sizes,
        same relative virtual addresses, same streaming-copy rate (6.7 TB/s either way), any offset inside an allocation behaves
        like the allocation; only replacing the allocation changes it, and about every second fresh allocation is a bad one
        (the two "kinds of box", 81 vs 86 us, of the round's earlier bench lines were this lottery, drawn once per buffer and
        process).  The accesses that suffer are the scattered ones (90-cell row segments, code bytes), which points at the
        translation reach of physically fragmented allocations rather than at DRAM; nothing in user space shows it directly.
        The remedy that needs no knowledge of the cause: one allocation for everything hot (one draw decides, instead of four
        independent ones), timed with one episode per candidate by the kernels' own dispatch-bound events; the best is kept,
        the others are released.  Costs ~3 ms per draw at construction; the env must be reset afterwards.  Returns the trace
        of the search (None for batches too small to matter)."""
        d = self.d
        if draws < 2 or self.E * d.grid_x * d.grid_y < (1 << 24):
            return None
        if getattr(self, "_graphs", None):
            raise _ffi.IppmError("tune_placement moves the maps: call it before capture_step_graphs (captured launches keep the old addresses)")
        T = d.budget + 1
        ids = list(range(1, self.E + 1))

        def episode():
            self._boxes_valid = False
            self.reset(ids)
            for t in range(T):
                if self.track_area:
                    self.build_observations(t, features=False)
                self.steps(t, policy=POLICY_UNIFORM, features=False)

        def score():
            episode()                      # first touch of the candidate
            was = self.profile
            self.event_times_us()
            self.profile = True
            episode()
            self.profile = was
            tm = self.event_times_us()
            return sum(tm[k]["avg_us"] * tm[k]["launches"] for k in ("sense", "fuse") if k in tm) / T

        score()                            # (the process's first episodes run 2-4 % slow whatever the allocation: not a sample)
        # Rejected candidates stay allocated while the search runs (a released block is what the next request would get back), but
        # never more of them than fit in HALF of the memory that is free now: beyond that the worst ones are handed back to the
        # driver.  At config 4's per-GPU shape the arena is ~10 GB: 24 candidates would be 240 GB next to a trainer's activations.
        _, total = hot_layout(self._hot_shapes)
        slack = lambda k: PLACEMENT_SLACK_MB * (k % 16)    # noqa: E731  (varies the request so that no cached block fits it exactly)
        free_b, _ = torch.cuda.mem_get_info(self.device)
        max_alive = placement_alive_cap(free_b, total)     # (the arena in use counts as one)
        scores, arenas = [score()], [self._arena]      # arenas[k] is None once released
        stopped = "draws exhausted"
        # JUMPS (round 6): the kind of an allocation comes in streaks -- neighbouring requests land in the same region of physical memory --
        # and a streak can outlast the search (24 slow draws in a row for one sub-batch, a fast one at the second draw of the next, on one
        # box).  After every `jump_after` draws that look alike (within 2 % of each other: one kind so far) a block of ballast is taken
        # and held until the search ends, so that the next draws come from somewhere else.  IPPM_PLACEMENT_JUMP_GB (default 12; 0: no jumps).
        ballast, jump_after, since_jump = [], 4, 0
        jump_gb = float(os.environ.get("IPPM_PLACEMENT_JUMP_GB", "12"))
        jumps = 0
        for k in range(1, draws):
            if not os.environ.get("IPPM_PLACEMENT_NO_EARLY"):
                why = placement_stop_reason(scores, jumps)
                if why:
                    stopped = why
                    break
            since_jump += 1
            if jump_gb > 0 and since_jump >= jump_after and max(scores) < 1.02 * min(scores):
                free_now, _ = torch.cuda.mem_get_info(self.device)
                want = int(min(jump_gb * 2 ** 30, 0.25 * free_now))
                if want > (1 << 30):
                    try:
                        ballast.append(torch.empty(want, dtype=torch.uint8, device=self.device))
                        jumps += 1
                        since_jump = 0
                    except torch.cuda.OutOfMemoryError:
                        pass
            alive = [i for i, a in enumerate(arenas) if a is not None]
            if len(alive) >= max_alive:    # hand the slowest candidates back (never the best one)
                best_now = min(alive, key=scores.__getitem__)
                drop = sorted((i for i in alive if i != best_now), key=scores.__getitem__, reverse=True)[:len(alive) - max_alive + 1]
                self._use_arena(arenas[best_now])
                for i in drop:
                    arenas[i] = None
                torch.cuda.empty_cache()
            try:
                self._place_hot(slack_mb=slack(k))
            except torch.cuda.OutOfMemoryError:
                stopped = "out of memory"
                break
            arenas.append(self._arena)
            scores.append(score())
            if os.environ.get("IPPM_PLACEMENT_TRACE"):
                print(f"placement draw {k}: arena {self._arena.data_ptr():#x} + {self._arena.numel() / 2 ** 20:.0f} MB -> {scores[-1]:.1f} us", flush=True)
        best = min((i for i, a in enumerate(arenas) if a is not None), key=scores.__getitem__)
        self._use_arena(arenas[best])
        del arenas, ballast
        torch.cuda.empty_cache()           # the rejected allocations go back to the driver
        self._boxes_valid = False
        self._pending_t = None
        self._obs_t = None
        return {"draws": len(scores), "max_draws": draws, "kept": best, "map_kernels_us_per_step": [round(v, 1) for v in scores],
                "stopped": stopped, "max_candidates_alive": max_alive, "jumps": jumps}

    def event_times_us(self, clear: bool = True) -> Dict[str, Dict[str, float]]:
        """{kernel class: {"launches", "avg_us", "min_us", "kernel"}} of the launches made while ``profile`` was set
        (synchronises the stream, and the side stream of prefetch_terrain)."""
        if self._side is not None:
            self._side.synchronize()
        return self.ctx.kernel_times(self.stream, reset=clear)

    def counters(self, reset: bool = False) -> dict:
        return self.ctx.counters(self.stream, reset)

    def check_faults(self) -> None:
        """Raises if the device has flagged something that "cannot happen" (synchronises): an env whose tile work list did not fit
        its slice (IPPM_FAULT_WORK_OVERFLOW: that env's maps are no longer fused), a fusion launch that rejected a work list, or a
        terrain workgroup whose in-launch wait for its env's (min, max) gave up."""
        bad = (self.fault & _ffi.FAULT_WORK_OVERFLOW).ne(0).nonzero().view(-1)
        if bad.numel():
            raise _ffi.IppmError(f"work list overflow in envs {bad[:8].tolist()} (of {bad.numel()}): the tile list's capacity bound is wrong")
        rejects = self.counters()["work_list_rejects"]
        if rejects:
            raise _ffi.IppmError(f"{rejects} fusion launches were handed a work list they could not read")
        if self._field is not None and getattr(self._field, "_keys", None) is not None:
            words = self._field._keys.view(-1)[:-4].view(-1, 4)[:, 3]
            if bool(words.ne(0).any()):
                raise _ffi.IppmError("terrain synthesis: a workgroup gave up waiting for its env's (min, max) -- truth planes are wrong")

    def pack_flips(self, tiles, rects: np.ndarray) -> torch.Tensor:
        """tiles[e][i]: uint8 [h,w] (1 = flipped) for clipped rect rects[e,i] = [yu,yd,xl,xr] -> device tile layout."""
        N = self.d.n_agents
        out = np.zeros((self.E, N, self.d.tile_bytes), dtype=np.uint8)
        for e in range(self.E):
            for i in range(N):
                if tiles[e][i] is not None:
                    out[e, i] = self.d.pack_tile(rects[e, i], tiles[e][i])
        return torch.from_numpy(out).to(self.device)


class SplitVecEnv:
    """A batch of envs stepped as ``parts`` sub-batches, each a ``VecEnv`` on its own HIP stream (env-only stepping).

    The envs of a batch share nothing, so any split of them computes the same episodes; what the split buys is overlap: an env step
    is three launches -- a latency-bound plan kernel (a few hundred bytes per env, 15 us at config 2 with the device nearly idle),
    then two bandwidth-bound map kernels -- and on ONE stream nothing else can run under the plan kernel.  With two half-batches on
    two streams the plan kernel (and the reset's issue-bound terrain passes) of one half runs beside the map kernels of the other:
    0.1526 -> 0.140-0.1425 ms per step of 1024 envs x 4 UAVs x 256^2 (26.8 -> 28.7-29.3 M agent-env steps/s, profiles/r05/
    two_streams_probe.txt; three parts 0.1425-0.1446, four 0.1475: smaller launches cost more than the extra overlap gives).
    Round 4 tried overlap INSIDE one batch's step -- the global fusion beside the local one, the next wave's terrain beside the
    steps -- and lost: two bandwidth-bound launches only get in each other's way.  Here each stream's own sequence stays as it is.

    Part k owns envs [k * E / parts, (k + 1) * E / parts) in the order of ``episodes`` / ``team_sizes``; streams are ordered against
    the caller's current stream at the start of every call, and ``join()`` makes the current stream wait for all parts (call it
    before reading state; ``torch.cuda.synchronize()`` also does).  Rollouts that feed a network step all envs at the same ``t``
    through one actor batch and use ``VecEnv`` directly."""

    def __init__(self, params: Dict, n_envs: int, parts: int = 2, device: str = "cuda:0", philox_seed: int = 3, terrain: str = "split",
                 track_area: bool = False, team_sizes=None, map_layout: str = "auto"):
        if parts < 1 or n_envs < parts:
            raise ValueError("SplitVecEnv: 1 <= parts <= n_envs")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        base, extra = divmod(int(n_envs), parts)
        self.sizes = [base + (1 if k < extra else 0) for k in range(parts)]
        self.offsets = [sum(self.sizes[:k]) for k in range(parts)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(parts)]
        self.stream_check_error = None     # why the side-by-side check of the streams could not run (None: it ran, or was not needed)
        self.stream_redraws, self.stream_probe = self._spread_streams()
        ts = None if team_sizes is None else [int(v) for v in team_sizes]
        self.parts = []
        for k, (n, off) in enumerate(zip(self.sizes, self.offsets)):
            with torch.cuda.stream(self.streams[k]):
                self.parts.append(VecEnv(params, n, device=device, philox_seed=philox_seed, terrain=terrain, track_area=track_area,
                                         team_sizes=None if ts is None else ts[off:off + n], map_layout=map_layout, layout_envs=int(n_envs)))
        self.E = int(n_envs)
        self.d = self.parts[0].d
        self.tiled = self.parts[0].tiled
        self.params = params

    # -- stream plumbing ------------------------------------------------------------------------------
    @staticmethod
    def _side_by_side(a, b, cycles: int) -> Tuple[float, float]:
        """(wall time of one spin kernel on each of two streams) / (two on one stream), and the latter in seconds: about 0.5 when
        the two streams run side by side, about 1 when the runtime serves both from one hardware queue."""
        def run(s0, s1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(s0):
                torch.cuda._sleep(cycles)
            with torch.cuda.stream(s1):
                torch.cuda._sleep(cycles)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        run(a, b)
        serial = run(a, a)
        return run(a, b) / max(serial, 1e-9), serial

    def _spread_streams(self, tries: int = 8):
        """HIP serves a process's streams from a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and torch hands streams
        out of a pool of 32: two sub-batches' streams can land on ONE queue, where their kernels run one after the other and the
        split buys nothing (one process in eight of profiles/r05/two_streams_probe.txt; a config 5 line at the one-stream figure).
        Two spin kernels tell: a stream that does not run beside every earlier one is swapped for the pool's next.
        -> (streams swapped, the last ratios measured); skipped where the spin kernel is not to be had."""
        if len(self.streams) < 2 or not hasattr(torch.cuda, "_sleep"):
            return 0, []
        redraws, ratios = 0, []
        try:
            cycles = 1 << 20
            for _ in range(4):          # a spin long enough to time from the host (>= 0.4 ms for the pair)
                if self._side_by_side(self.streams[0], self.streams[0], cycles)[1] >= 4e-4:
                    break
                cycles <<= 2
            else:
                return 0, []
            for k in range(1, len(self.streams)):
                for _ in range(tries):
                    ratios = [round(self._side_by_side(self.streams[j], self.streams[k], cycles)[0], 2) for j in range(k)]
                    if max(ratios) < 0.8:
                        break
                    self.streams[k] = torch.cuda.Stream(device=self.device)
                    redraws += 1
        except RuntimeError as exc:     # a HIP error of the spin kernel / the timing: the env works without the check, but says so --
            self.stream_check_error = f"{type(exc).__name__}: {exc}"   # its streams may share a queue (bench: config.stream_check.error)
        return redraws, ratios

    def _each(self, ordered: bool = True):
        """(part, its slice of the batch) with the part's stream current and (``ordered``) behind the caller's stream."""
        if ordered:
            cur = torch.cuda.current_stream(self.device)
            entered = torch.cuda.Event()
            entered.record(cur)
        for k, env in enumerate(self.parts):
            if ordered:
                self.streams[k].wait_event(entered)
            with torch.cuda.stream(self.streams[k]):
                yield env, slice(self.offsets[k], self.offsets[k] + self.sizes[k])

    def join(self):
        """The caller's current stream waits for everything queued on the parts' streams."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            done = torch.cuda.Event()
            done.record(s)
            cur.wait_event(done)

    # -- the VecEnv surface the env-only loop uses -------------------------------------------------------
    def reset(self, episodes, **kw):
        ep = torch.as_tensor(episodes, dtype=torch.int64).reshape(self.E)
        for env, sl in self._each():
            env.reset(ep[sl], **{k: (v[sl] if v is not None and hasattr(v, "__getitem__") and not isinstance(v, str) else v) for k, v in kw.items()})

    def steps(self, t: int, policy: int = POLICY_UNIFORM, actions: Optional[torch.Tensor] = None):
        """One env step of every part (plan -> fusion -> K3 on its stream); nothing is returned: ``join()``, then read ``reward``."""
        for env, sl in self._each():
            env.steps(t, policy=policy, actions=None if actions is None else actions[sl], features=False)

    # -- the episode loop, owned by the split: every part in its own phase of the episode -----------------------------
    def start(self, episodes_of_wave, stagger: bool = False, policy: int = POLICY_UNIFORM):
        """Begins an endless run of whole episodes: ``episodes_of_wave(w)`` -> the E episode ids of wave w (part k flies its slice of
        every wave), then ``advance()`` steps every part once and resets a part that has finished its episode to its next wave.
        ``stagger``: part k starts k * T / parts steps AHEAD of part 0 (those steps are taken here), so that from then on at most one
        part resets at any step and its reset -- a write-only fill and the issue-bound terrain passes, 387 us at config 2 -- runs
        beside the other parts' map kernels instead of beside the other parts' resets.  Every episode is the same episode as in any
        other batching (the streams of an episode are keyed by its number): only WHEN it is flown changes.
        Measured (round 6, config 2, two and three parts, alternating processes on one box): no gain -- 0.1386-0.1394 ms per step
        staggered against 0.1369-0.1382 in lock step; the parts' resets side by side cost no more than one after the other, and a
        short window holds more part-resets than its share.  Hence off by default."""
        T = self.d.budget + 1
        self._episodes_of_wave = episodes_of_wave
        self._phase = [0] * len(self.parts)        # the step each part takes next
        self._wave = [0] * len(self.parts)         # the wave each part is flying
        self.part_resets = 0                       # resets of single parts so far (the whole batch has reset part_resets / parts times)
        for k, (env, sl) in enumerate(self._each()):
            env.reset(torch.as_tensor(episodes_of_wave(0), dtype=torch.int64).reshape(self.E)[sl])
            self.part_resets += 1
            if stagger:
                for _ in range(k * T // len(self.parts)):
                    env.steps(self._phase[k], policy=policy, features=False)
                    self._phase[k] += 1

    def advance(self, policy: int = POLICY_UNIFORM):
        """One env step of every part at its own step of the episode; a part whose episode is over starts its next wave."""
        T = self.d.budget + 1
        # (not ordered behind the caller's stream again at every step: start() was, and nothing but the parts' own streams touches
        #  their state in between -- an event and a wait per part and step are barrier packets in front of every plan kernel)
        for k, (env, sl) in enumerate(self._each(ordered=False)):
            env.steps(self._phase[k], policy=policy, features=False)
            self._phase[k] += 1
            if self._phase[k] == T:
                self._wave[k] += 1
                env.reset(torch.as_tensor(self._episodes_of_wave(self._wave[k]), dtype=torch.int64).reshape(self.E)[sl])
                self._phase[k] = 0
                self.part_resets += 1

    def tune_placement(self, draws: int = 24):
        return [env.tune_placement(draws) for env, _ in self._each()]

    @property
    def profile(self) -> bool:
        return self.parts[0].profile

    @profile.setter
    def profile(self, on: bool):
        for env in self.parts:
            env.profile = on

    def check_faults(self) -> None:
        for env, _ in self._each():
            env.check_faults()

    def counters(self, reset: bool = False) -> dict:
        total: Dict[str, int] = {}
        for env, _ in self._each():
            for k, v in env.counters(reset).items():
                total[k] = total.get(k, 0) + v
        return total

    def event_times_us(self, clear: bool = True) -> Dict[str, Dict[str, float]]:
        """Launch-weighted merge of the parts' dispatch-bound kernel times (a launch = one part's kernel)."""
        out: Dict[str, Dict[str, float]] = {}
        for env, _ in self._each():
            for cls, rec in env.event_times_us(clear).items():
                acc = out.setdefault(cls, {"launches": 0, "avg_us": 0.0, "min_us": rec["min_us"], "kernel": rec["kernel"]})
                acc["avg_us"] = (acc["avg_us"] * acc["launches"] + rec["avg_us"] * rec["launches"]) / (acc["launches"] + rec["launches"])
                acc["launches"] += rec["launches"]
                acc["min_us"] = min(acc["min_us"], rec["min_us"])
        return out

    def rows_view(self, maps: torch.Tensor) -> torch.Tensor:
        return rows_view(maps, self.tiled)

    def _cat(self, name):
        self.join()
        return torch.cat([getattr(env, name) for env in self.parts], dim=0)

    pos = property(lambda self: self._cat("pos"))
    local = property(lambda self: self._cat("local"))
    glob = property(lambda self: self._cat("glob"))
    reward = property(lambda self: self._cat("reward"))
    action = property(lambda self: self._cat("action"))
    mask = property(lambda self: self._cat("mask"))
    fault = property(lambda self: self._cat("fault"))
    sums = property(lambda self: self._cat("sums"))
