"""ctypes binding of libippmarl.so (include/ippmarl.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch  # noqa: F401  -- must be imported BEFORE libippmarl.so is dlopen'ed: both link libamdhip64.so.7 and the process
#                            must end up with ONE HIP runtime (torch's), otherwise stream handles and device pointers
#                            handed over by torch belong to a different runtime ("no ROCm-capable device is detected")

from .derived import DerivedConstants, MAX_LATTICE, MAX_Z, CLIP_LO, CLIP_HI

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPPMARL_LIB", os.path.join(_HERE, "..", "lib", "libippmarl.so"))
WS_WORDS = 160
FAULT_WORK_OVERFLOW = 0x40000000   # IPPM_FAULT_WORK_OVERFLOW (ippmarl.h): sticky bit of fault[e]
FEAT, ACTOR_PLANES, CRITIC_PLANES = 11, 7, 12
STEP_COMM, STEP_GLOBAL, STEP_MOVE, STEP_TILES = 1, 2, 4, 8   # ippm_plan_step flags
SENSE_REC_WORDS = 8   # words per agent of ippm_plan_step's rect_next / ippm_sense_step's rect_in (IPPM_SENSE_REC_WORDS)
# kernel classes of ippm_read_kernel_times (IPPM_T_*)
TIMED = {"sense": 0, "fuse": 1, "plan": 2, "actor_features": 3, "critic_features": 4, "reset": 5, "terrain": 6, "reset_maps": 7}


class IppmConfig(C.Structure):
    _fields_ = [
        ("n_agents", C.c_int32), ("grid_x", C.c_int32), ("grid_y", C.c_int32),
        ("space_x", C.c_int32), ("space_y", C.c_int32), ("space_z", C.c_int32),
        ("spacing", C.c_int32), ("min_altitude", C.c_int32),
        ("x_dim_m", C.c_int32), ("y_dim_m", C.c_int32),
        ("n_actions", C.c_int32), ("budget", C.c_int32), ("env_seed", C.c_int32),
        ("tile_stride", C.c_int32), ("fix_range", C.c_int32), ("reserved0", C.c_int32),
        ("centre_x", C.c_int32 * MAX_LATTICE), ("centre_y", C.c_int32 * MAX_LATTICE),
        ("radius_x", C.c_int32 * MAX_Z), ("radius_y", C.c_int32 * MAX_Z),
        ("logit_meas", (C.c_float * 2) * MAX_Z), ("meas_value", (C.c_float * 2) * MAX_Z),
        ("flip_threshold", C.c_uint32 * MAX_Z),
        ("prior", C.c_float), ("clip_lo", C.c_float), ("clip_hi", C.c_float),
        ("logit_prior", C.c_float), ("logit_clip", C.c_float), ("logit_weight_thr", C.c_float),
        ("comm_range", C.c_double), ("failure_rate", C.c_double),
        ("philox_seed", C.c_uint64), ("gamma", C.c_double), ("lambda_", C.c_double),
        ("logit_noise", C.c_float * MAX_Z),
        ("logit_prior_f64", C.c_double),
    ]


class IppmCounters(C.Structure):
    _fields_ = [("sense_cells", C.c_uint64), ("fuse_local_cells", C.c_uint64), ("fuse_local_ops", C.c_uint64),
                ("fuse_global_cells", C.c_uint64), ("fuse_global_ops", C.c_uint64), ("feature_cells", C.c_uint64),
                ("reserved", C.c_uint64 * 2)]


P = C.c_void_p
I32, I64 = C.c_int32, C.c_int64

# name -> argtypes (all return int); mirrors include/ippmarl.h one to one
PROTOTYPES = {
    "ippm_ctx_create": [C.POINTER(IppmConfig), C.POINTER(P)],
    "ippm_ctx_destroy": [P],
    "ippm_sync": [P, P],
    "ippm_read_counters": [P, C.POINTER(IppmCounters), C.c_int, P],
    "ippm_kernel_timing": [P, I32],
    "ippm_read_kernel_times": [P, I32, I32, P, P, P, P, I32, P],
    "ippm_reset_episode": [P, P, P, P, P, P, P, P, P, P, P, I32, P],
    "ippm_reset_maps": [P, P, P, P, P, P, P, P, P, P, I32, I32, P],
    "ippm_logodds_to_prob": [P, P, P, I64, P],
    "ippm_prob_to_logodds": [P, P, P, I64, P],
    "ippm_footprint": [P, P, P, P, I32, P],
    "ippm_stream_copy": [P, P, P, I64, P],
    "ippm_sense_update": [P, P, P, P, P, P, P, P, P, I32, I32, I32, P],
    "ippm_sense_step": [P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, P],
    "ippm_set_team_sizes": [P, P],
    "ippm_dirty_slab_words": [P, I32, P],
    "ippm_set_dirty_slabs": [P, P],
    "ippm_set_map_layout": [P, I32],
    "ippm_map_layout": [P, P],
    "ippm_map_layout_advice": [P, I32, P],
    "ippm_maps_relayout": [P, P, P, I32, I32, P],
    "ippm_plan_step": [P, P, P, P, P, P, P, P, I32, I32, P, P, I32, P, P, P, P, P, I32, P],
    "ippm_fuse_step": [P, P, P, P, P, P, P, P, I32, P],
    "ippm_reward_finalize": [P, P, P, I32, P],
    "ippm_work_words": [P, I32, P],
    "ippm_tile_form": [P, P],
    "ippm_area_sums": [P, P, P, I32, I32, I32, P],
    "ippm_area_resize": [P, P, I32, I32, P, P, I32, P],
    "ippm_entropy_maps": [P, P, P, P, P, P, P, I64, P],
    "ippm_comm_matrix": [P, P, P, P, P, P, I32, I32, P],
    "ippm_fuse_local": [P, P, P, P, P, P, P, I32, I32, P],
    "ippm_comm_fuse_local": [P, P, P, P, P, P, P, P, P, P, I32, I32, P],
    "ippm_action_mask": [P, P, P, P, I32, P, P, P, I32, P],
    "ippm_clamp_logodds": [P, P, I64, P],
    "ippm_fuse_global_reward": [P, P, P, P, P, P, P, P, I32, P],
    "ippm_weighted_entropy": [P, P, P, I32, P, I32, P],
    "ippm_reward_from_maps": [P, P, P, P, P, I32, P],
    "ippm_mask_act_move": [P, P, P, P, P, I32, I32, P, P, P, I32, P],
    "ippm_actor_features": [P, P, P, P, P, P, I32, P, I32, P],
    "ippm_critic_features": [P, P, P, P, P, P, P, I32, P],
    "ippm_coma_advantage": [P, P, P, P, P, P, P, I32, P],
    "ippm_td_lambda": [P, P, P, P, P, P, I32, I32, P],
    "ippm_col2im_nhwc": [P, P, I32, I32, I32, I32, I32, P],
    "ippm_bias_relu_nhwc": [P, P, I64, I32, P],
    "ippm_bias_relu_backward_nhwc": [P, P, P, P, I64, I32, P],
    "ippm_ig_candidates": [P, P, P, P, P, I32, P],
    "ippm_ig_select": [P, P, P, P, I32, P, P, I32, P],
    "ippm_f1_counts": [P, P, P, I32, C.c_float, P, I32, P],
    "ippm_terrain_noise": [P, P, P, I32, P],
    "ippm_terrain_spectrum": [P, P, P, P, I32, P],
    "ippm_terrain_field": [P, P, P, P, P, P, P, I32, P],
    "ippm_terrain_pack": [P, P, P, P, I32, P],
    "ippm_terrain_truth": [P, P, P, P, P, P, I32, P],
    "ippm_area_weights": [I32, I32, P, P, P],
    "ippm_host_philox": [P, P],
    "ippm_host_start_state": [I32, I64, I32, I32, I32, I32, P],
    "ippm_host_truth_params": [I64, P],
}

_lib: Optional[C.CDLL] = None


class IppmError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Loads libippmarl.so; raises if it has not been built (``python __graft_entry__.py`` builds it)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.abspath(LIB_PATH)
    if not os.path.isfile(path):
        raise IppmError(f"{path} not found: build the HIP library first (make -C ipp-marl_amd/csrc). "
                        "There is no CPU fallback for the ipp-marl hot path.")
    lib = C.CDLL(path)
    lib.ippm_last_error.restype = C.c_char_p
    lib.ippm_last_error.argtypes = []
    lib.ippm_version.restype = C.c_int
    lib.ippm_version.argtypes = []
    lib.ippm_config_size.restype = C.c_int
    lib.ippm_config_size.argtypes = []
    if lib.ippm_config_size() != C.sizeof(IppmConfig):
        raise IppmError(f"ippm_config layout mismatch: library {lib.ippm_config_size()} B, binding {C.sizeof(IppmConfig)} B")
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load_library().ippm_last_error().decode()
        raise IppmError(f"{what} failed ({rc}): {msg}")


def make_config(d: DerivedConstants) -> IppmConfig:
    c = IppmConfig()
    c.n_agents, c.grid_x, c.grid_y = d.n_agents, d.grid_x, d.grid_y
    c.space_x, c.space_y, c.space_z = d.space_x, d.space_y, d.space_z
    c.spacing, c.min_altitude = d.spacing, d.min_altitude
    c.x_dim_m, c.y_dim_m = d.x_dim_m, d.y_dim_m
    c.n_actions, c.budget, c.env_seed = d.n_actions, d.budget, d.env_seed
    c.tile_stride, c.fix_range = d.tile_stride, 1 if d.fix_range else 0
    for i in range(d.space_x):
        c.centre_x[i] = int(d.centre_x[i])
    for i in range(d.space_y):
        c.centre_y[i] = int(d.centre_y[i])
    for k in range(d.space_z):
        c.radius_x[k], c.radius_y[k] = d.radius_x[k], d.radius_y[k]
        for o in range(2):
            c.logit_meas[k][o] = float(d.logit_meas[k, o])
            c.meas_value[k][o] = float(d.meas_value[k, o])
        c.flip_threshold[k] = int(d.flip_threshold[k])
    c.prior, c.clip_lo, c.clip_hi = d.prior, CLIP_LO, CLIP_HI
    c.logit_prior, c.logit_clip, c.logit_weight_thr = d.logit_prior, d.logit_clip, d.logit_weight_thr
    c.comm_range, c.failure_rate = d.comm_range, d.failure_rate
    c.philox_seed = d.philox_seed & 0xFFFFFFFFFFFFFFFF
    c.gamma, c.lambda_ = d.gamma, d.lam
    for k in range(d.space_z):
        c.logit_noise[k] = float(d.logit_noise[k])
    c.logit_prior_f64 = float(d.logit_prior)
    return c


def ptr(t) -> Optional[int]:
    """Device (or host) pointer of a contiguous torch tensor / NumPy array; None passes NULL."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return t.ctypes.data
    assert t.is_contiguous(), "libippmarl needs contiguous tensors"
    return t.data_ptr()


class Context:
    """Owns one ippm_ctx.  Methods are thin, checked wrappers; tensors are validated by the callers."""

    def __init__(self, derived: DerivedConstants):
        self.lib = load_library()
        self.derived = derived
        self.cfg = make_config(derived)
        self.handle = P()
        check(self.lib.ippm_ctx_create(C.byref(self.cfg), C.byref(self.handle)), "ippm_ctx_create")

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.ippm_ctx_destroy(self.handle)
            self.handle = P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name: str, *args):
        check(getattr(self.lib, name)(self.handle, *args), name)

    def kernel_timing(self, enable: bool):
        """Attach begin/end events to every launch of the timed kernel classes (ippm_kernel_timing)."""
        self.call("ippm_kernel_timing", 1 if enable else 0)

    def kernel_times(self, stream, reset: bool = True) -> dict:
        """{class: {"launches", "avg_us", "min_us", "kernel"}} of the launches timed since the last reset (synchronises)."""
        out = {}
        for name, cls in TIMED.items():
            n, tot, mn = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
            buf = C.create_string_buffer(128)
            self.call("ippm_read_kernel_times", cls, 1 if reset else 0, C.addressof(n), C.addressof(tot), C.addressof(mn),
                      C.addressof(buf), 128, stream)
            if n.value:
                out[name] = {"launches": int(n.value), "avg_us": tot.value / n.value, "min_us": mn.value,
                             "kernel": buf.value.decode()}
        return out

    def counters(self, stream, reset: bool = False) -> dict:
        out = IppmCounters()
        check(self.lib.ippm_read_counters(self.handle, C.byref(out), 1 if reset else 0, stream), "ippm_read_counters")
        res = {k: int(getattr(out, k)) for k, _ in IppmCounters._fields_ if k != "reserved"}
        res["work_list_rejects"] = int(out.reserved[0])   # fusion launches handed a work list of the other form (must stay 0)
        return res


# ---- host helpers (usable without a GPU) ------------------------------------------------------------
def host_philox(c0, c1, c2, c3, k0, k1):
    lib = load_library()
    inp = (C.c_uint32 * 6)(c0 & 0xFFFFFFFF, c1 & 0xFFFFFFFF, c2 & 0xFFFFFFFF, c3 & 0xFFFFFFFF, k0 & 0xFFFFFFFF, k1 & 0xFFFFFFFF)
    out = (C.c_uint32 * 4)()
    check(lib.ippm_host_philox(inp, out), "ippm_host_philox")
    return [int(v) for v in out]


def host_start_state(env_seed, episode, agent, spacing, space_x, space_y):
    lib = load_library()
    out = (C.c_int32 * 3)()
    check(lib.ippm_host_start_state(env_seed, episode, agent, spacing, space_x, space_y, out), "ippm_host_start_state")
    return [int(v) for v in out]


def host_truth_params(episode):
    lib = load_library()
    out = (C.c_int32 * 2)()
    check(lib.ippm_host_truth_params(episode, out), "ippm_host_truth_params")
    return int(out[0]), int(out[1])


def host_area_weights(n_src, n_dst=FEAT):
    lib = load_library()
    b = np.zeros(n_src, dtype=np.int32)
    w0 = np.zeros(n_src, dtype=np.float32)
    w1 = np.zeros(n_src, dtype=np.float32)
    check(lib.ippm_area_weights(n_src, n_dst, b.ctypes.data, w0.ctypes.data, w1.ctypes.data), "ippm_area_weights")
    return b, w0, w1
