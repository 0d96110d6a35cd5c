"""Host-side derived constants handed to the device as plain integers / float32 (include/ippmarl.h).

The footprint arithmetic sits on float64 knife edges (default params: 170.99999999999997 -> 170 cells), so
the centre-cell and half-width tables are evaluated here with NumPy in the reference's expression order
(sensors/cameras.py:31-77, mapping/grid_maps.py:17-66) and only integers cross the C-ABI.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

MAX_AGENTS, MAX_LATTICE, MAX_Z = 16, 64, 8
CLIP_LO, CLIP_HI = 0.0001, 0.9999


START_ALTITUDE = 15   # agent/state_space.py:32: state_z = 15


def _noise(altitude: int) -> float:
    # sensors/models/sensor_models.py:13-22 (coeff_a/coeff_b are never used by the reference)
    return {5: 0.01, 10: 0.265, 15: 0.375}.get(int(altitude), 0)


class DerivedConstants:
    def __init__(self, params: Dict, philox_seed: int = 3):
        env, exp = params["environment"], params["experiment"]
        con, uav, mis = exp["constraints"], exp["uav"], exp["missions"]
        fov, pix = params["sensor"]["field_of_view"], params["sensor"]["pixel"]
        for section, key in (("environment", "x_dim"), ("environment", "y_dim")):
            if key not in params[section]:
                raise ValueError(f"Cannot find {section}.{key} in config")  # grid_maps.py:19-28
        self.params = params
        self.n_agents = int(mis["n_agents"])
        self.n_actions = int(con["num_actions"])
        self.budget = int(con["budget"])
        self.spacing = int(con["spacing"])
        self.min_altitude = int(con["min_altitude"])
        self.max_altitude = int(con["max_altitude"])
        self.x_dim_m, self.y_dim_m = int(env["x_dim"]), int(env["y_dim"])
        self.env_seed = int(env["seed"])
        self.prior = float(params["mapping"]["prior"])
        self.comm_range = float(uav["communication_range"])
        self.fix_range = bool(uav["fix_range"])
        self.failure_rate = float(uav["failure_rate"])
        self.gamma = float(params["networks"]["gamma"])
        self.lam = float(params["networks"]["lambda"])
        self.philox_seed = int(philox_seed)
        self.angle_x, self.angle_y = fov["angle_x"], fov["angle_y"]
        # cell size and grid dims (grid_maps.py:29-32,47-66)
        self.res_x = (2 * self.min_altitude * math.tan(math.radians(self.angle_x) * 0.5)) / pix["number_x"]
        self.res_y = (2 * self.min_altitude * math.tan(math.radians(self.angle_y) * 0.5)) / pix["number_y"]
        self.grid_x = int(self.x_dim_m / self.res_x)
        self.grid_y = int(self.y_dim_m / self.res_y)
        # lattice (state_space.py:16-21)
        self.space_x = self.x_dim_m // self.spacing + 1
        self.space_y = self.y_dim_m // self.spacing + 1
        self.space_z = (self.max_altitude - self.min_altitude) // self.spacing + 1
        self.altitudes = [self.min_altitude + k * self.spacing for k in range(self.space_z)]
        if self.n_agents > MAX_AGENTS or max(self.space_x, self.space_y) > MAX_LATTICE or self.space_z > MAX_Z:
            raise ValueError("configuration exceeds the compiled limits of libippmarl")
        # The reference starts every UAV at z = 15 m whatever the altitude bounds say (agent/state_space.py:32) and projects its
        # camera from the true altitude; the device tabulates footprints and sensor noise per lattice level, so the start level
        # has to be one of them (otherwise the UAVs would fly at an altitude the tables do not hold)
        if START_ALTITUDE not in self.altitudes:
            raise ValueError(f"experiment.constraints min/max_altitude = {self.min_altitude}/{self.max_altitude} with spacing "
                             f"{self.spacing} leave out {START_ALTITUDE} m, the altitude every UAV starts at (agent/state_space.py:32)")
        # centre cell of every lattice coordinate: floor(pos / res_x) for BOTH axes (cameras.py:66)
        xs = np.arange(self.space_x) * self.spacing
        ys = np.arange(self.space_y) * self.spacing
        self.centre_x = np.floor(xs / self.res_x).astype(np.int64)
        self.centre_y = np.floor(ys / self.res_x).astype(np.int64)
        # half widths per altitude (cameras.py:62-67), same operation order on NumPy scalars
        rx, ry = [], []
        for z in self.altitudes:
            z = np.int64(z)
            x_range_m = 2 * z * np.tan(0.5 * np.radians(self.angle_x))
            y_range_m = 2 * z * np.tan(0.5 * np.radians(self.angle_y))
            cells = np.array([np.floor(x_range_m / self.res_x), np.floor(y_range_m / self.res_y)])
            r = np.floor(0.5 * cells)
            rx.append(int(r[0]))
            ry.append(int(r[1]))
        self.radius_x, self.radius_y = rx, ry
        # measurement values / logits exactly as the reference forms them in float32
        # (simulations.py:47-51: round(.,3) -> float32; mappings.py:113: np.log(y / (1 - y)) on float32)
        self.meas_value = np.zeros((self.space_z, 2), dtype=np.float32)
        self.logit_meas = np.zeros((self.space_z, 2), dtype=np.float32)
        self.flip_threshold = np.zeros(self.space_z, dtype=np.uint64)
        self.logit_noise = np.zeros(self.space_z, dtype=np.float64)   # IG planner: scalar float64 noise levels
        for k, z in enumerate(self.altitudes):
            nz = _noise(z)
            with np.errstate(divide="ignore"):
                self.logit_noise[k] = np.log((1 - nz) / nz) if nz > 0 else np.inf
            acc = 1 - nz
            y = np.float32(np.round(np.array([1 - acc, acc * 1.0]), 3))
            self.meas_value[k] = y
            with np.errstate(divide="ignore"):
                self.logit_meas[k] = np.log(y / (1 - y))
            self.flip_threshold[k] = int(math.floor(nz * 4294967296.0))
        # altitudes outside the sensor model's table (sensor_models.py:13-22) are noise-free: logit_noise = inf there (the greedy
        # planner of the reference divides by that noise: IG_baseline raises there, as does the reference)
        self.noise_free_altitudes = [z for z in self.altitudes if _noise(z) == 0]
        # log-odds constants of the device representation (maps are stored as ln(p/(1-p)))
        self.logit_prior = float(np.log(self.prior / (1 - self.prior))) if 0 < self.prior < 1 else 0.0
        self.logit_clip = float(np.log(CLIP_HI / (1 - CLIP_HI)))
        self.logit_weight_thr = float(np.log(0.501 / 0.499))
        # code/flip tile stride: widest footprint + 3 cells of alignment slack, multiple of 4
        need = max(max(2 * r for r in rx), max(2 * r for r in ry) + 3, 4)
        self.tile_stride = (need + 3) // 4 * 4

    # -- helpers shared by the host-side mirrors ------------------------------------------------------
    def position_to_index(self, position):
        return np.array([position[0] // self.spacing, position[1] // self.spacing, position[2] // self.spacing - 1])

    def index_to_position(self, state):
        return np.array([state[0] * self.spacing, state[1] * self.spacing, self.spacing + state[2] * self.spacing])

    def footprint(self, position):
        """([yu,yd,xl,xr] unclipped, clipped) from the tables -- integers only."""
        ix, iy = int(position[0]) // self.spacing, int(position[1]) // self.spacing
        k = min(max((int(position[2]) - self.min_altitude) // self.spacing, 0), self.space_z - 1)
        xl, xr = int(self.centre_x[ix]) - self.radius_x[k], int(self.centre_x[ix]) + self.radius_x[k]
        yu, yd = int(self.centre_y[iy]) - self.radius_y[k], int(self.centre_y[iy]) + self.radius_y[k]
        full = [yu, yd, xl, xr]
        cy = lambda v: min(max(v, 0), self.grid_y - 1)  # noqa: E731
        cx = lambda v: min(max(v, 0), self.grid_x - 1)  # noqa: E731
        return full, [cy(yu), cy(yd), cx(xl), cx(xr)]

    # ---- packed byte planes (mirror of ippm_internal.h) ---------------------------------------------------------
    @property
    def vec(self) -> int:
        # mirror of ippm_ctx_create: 16-byte lane groups need a multiple-of-4 width and at least 4 cells per feature bin
        return 4 if self.grid_y >= 44 else 1   # (as ippm_ctx::vec: the grid need not be a multiple of 4 wide)

    @property
    def truth_bytes(self) -> int:
        # (grids not a multiple of 4 wide: room for the 2-byte load at the last cell's byte, as ippm_truth_bytes)
        return (self.grid_x * self.grid_y + (8 if self.grid_y % 4 else 0) + 31) // 32 * 4

    @property
    def tile_bytes(self) -> int:
        s = self.tile_stride
        return s * (s // 4) if self.vec == 4 else s * s

    def pack_tile(self, rect, bits) -> np.ndarray:
        """bits uint8 [h,w] in {0,1} of clipped rect [yu,yd,xl,xr] -> one device code/flips tile (uint8 [tile_bytes])."""
        yu, yd, xl, xr = (int(v) for v in rect)
        s = self.tile_stride
        full = np.zeros((s, s), dtype=np.uint8)
        off = yu & 3
        full[: xr - xl, off: off + yd - yu] = np.asarray(bits, dtype=np.uint8).reshape(xr - xl, yd - yu)
        if self.vec == 1:
            return full.reshape(-1)
        g = full.reshape(s, s // 4, 4)
        return (g[..., 0] | (g[..., 1] << 1) | (g[..., 2] << 2) | (g[..., 3] << 3)).astype(np.uint8).reshape(-1)

    def unpack_tile(self, rect, tile) -> np.ndarray:
        yu, yd, xl, xr = (int(v) for v in rect)
        s = self.tile_stride
        tile = np.asarray(tile, dtype=np.uint8)
        if self.vec == 1:
            full = tile.reshape(s, s)
        else:
            t = tile.reshape(s, s // 4)
            full = np.stack([(t >> q) & 1 for q in range(4)], axis=-1).reshape(s, s)
        off = yu & 3
        return full[: xr - xl, off: off + yd - yu].copy()

    def pack_truth(self, truth) -> np.ndarray:
        """uint8/bool [..., gx, gy] -> bit-packed uint8 [..., truth_bytes] (cell x*gy+y = bit of the little-endian string)."""
        t = np.asarray(truth).astype(np.uint8).reshape(*np.shape(truth)[:-2], -1)
        packed = np.packbits(t, axis=-1, bitorder="little")
        out = np.zeros((*packed.shape[:-1], self.truth_bytes), dtype=np.uint8)
        out[..., : packed.shape[-1]] = packed
        return out

    def unpack_truth(self, packed) -> np.ndarray:
        p = np.asarray(packed, dtype=np.uint8)
        bits = np.unpackbits(p, axis=-1, bitorder="little")[..., : self.grid_x * self.grid_y]
        return bits.reshape(*p.shape[:-1], self.grid_x, self.grid_y)

    def max_start_seed(self, episode: int) -> int:
        return self.env_seed * int(episode) * max(self.n_agents - 1, 0)
