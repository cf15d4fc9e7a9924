"""ctypes binding of ``libcosim_hip.so`` (the C ABI of ``include/cosim.h``).

PyTorch is used here only as plumbing: device buffers, streams and (in ``distributed.py``)
``torch.distributed``.  All arithmetic of the hot path happens in the HIP kernels.  There is no
CPU fallback: if the shared library is missing or no GPU is present, construction fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

from .model import CosimModel

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "libcosim_hip.so")
CSRC = os.path.join(_DIR, "csrc")

CS_MAXFIELD = 16
CS_MAXCMD = 6
OBS_IDS = {"dof_pos": 0, "dof_vel": 1, "ang_vel": 2, "lin_vel": 3, "projected_gravity": 4, "last_action": 5,
           "height_map": 6, "command": 7}


class ObsConfig(ctypes.Structure):
    """``cosim_obs_config_t`` (include/cosim.h)."""
    _fields_ = [
        ("stack_size", ctypes.c_int), ("command_dim", ctypes.c_int),
        ("n_stacked", ctypes.c_int), ("n_non_stacked", ctypes.c_int),
        ("stacked_field", ctypes.c_int * CS_MAXFIELD), ("non_stacked_field", ctypes.c_int * CS_MAXFIELD),
        ("field_dim", ctypes.c_int * 8), ("field_interval", ctypes.c_int * 8), ("field_scale", ctypes.c_float * 8),
        ("noise_mean", ctypes.c_float * 8), ("noise_std", ctypes.c_float * 8),
        ("noise_lower", ctypes.c_float * 8), ("noise_upper", ctypes.c_float * 8),
        ("noise_enabled", ctypes.c_int), ("position_command", ctypes.c_int),
        ("command_scales", ctypes.c_float * CS_MAXCMD),
        ("max_sim_step", ctypes.c_int), ("action_delay_prob", ctypes.c_float), ("init_noise", ctypes.c_float),
        ("auto_reset", ctypes.c_int),
        ("hm_res_x", ctypes.c_int), ("hm_res_y", ctypes.c_int), ("hm_size_x", ctypes.c_float), ("hm_size_y", ctypes.c_float),
    ]


# Packed fp32 (v_pk_fma_f32) pays in the solver's long FMA chains, but the SLP vectoriser's default cost model also packs the
# quaternion / cross-product code, where the v_mov shuffles around each packed op cost more than the op saves (kinematics: 243
# moves for 237 packed ops).  Measured on MI355X, 1000-step benches, threshold 0 (default) / 1 / 2 / 3 / 4 / off: headline
# 13.00 / 13.17 / 13.23 / 13.36 / 11.87 / 11.5 M env-steps/s, p_v3_flat 7.9 / 8.3 / 8.8 / 8.3 / 7.4, w4_rocky 2.73 / 2.76 / 2.86 / 2.87 / 2.85.
HIPCC_TUNING = ["-mllvm", "-slp-threshold=2"]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    # every file the translation unit pulls in (cosim_engine.hip includes the other .hip / .h files of csrc/)
    inc = os.path.join(_DIR, "..", "include")
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    if not force and os.path.isfile(LIB_PATH):
        newest = max(os.path.getmtime(p) for p in deps)
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", *HIPCC_TUNING,
           "-o", LIB_PATH, os.path.join(CSRC, "cosim_engine.hip")]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load_library():
    """Load ``libcosim_hip.so``; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libcosim_hip.so must bind to the HIP runtime torch has loaded (torch ships its own libamdhip64);
    # two HIP runtimes in one process do not see each other's devices, streams or allocations.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP engine has no CPU fallback)")
    L = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.cosim_create.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ctypes.c_uint64, ctypes.c_int64, ctypes.POINTER(vp)]
    L.cosim_destroy.argtypes = [vp]
    L.cosim_query.argtypes = [vp, ctypes.c_char_p]
    L.cosim_set_param.argtypes = [vp, ctypes.c_char_p, vp, ci]
    L.cosim_reset.argtypes = [vp, vp, vp, vp, vp]
    L.cosim_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.cosim_step_range.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    L.cosim_join.argtypes = [vp, vp]
    L.cosim_debug_support.argtypes = [vp, ci, vp, ci, vp, ci]
    L.cosim_hull_support_check.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp, vp]
    L.cosim_rollout.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    L.cosim_range.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(vp)]
    L.cosim_range_mark.argtypes = [vp, ci]
    L.cosim_get.argtypes = [vp, ctypes.c_char_p, vp, vp]
    L.cosim_set.argtypes = [vp, ctypes.c_char_p, vp, vp]
    L.cosim_event_push.argtypes = [vp, vp, vp, vp]
    L.cosim_debug_forward.argtypes = [vp, ci, ctypes.c_char_p, vp, ci]
    L.cosim_kernel_time.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ci)]
    L.cosim_set_timing.argtypes = [vp, ci]
    L.cosim_profile_step.argtypes = [vp] * 7
    L.cosim_last_error.restype = ctypes.c_char_p
    for fn in ("cosim_create", "cosim_destroy", "cosim_query", "cosim_set_param", "cosim_reset", "cosim_step", "cosim_step_range", "cosim_get",
               "cosim_join", "cosim_range", "cosim_range_mark", "cosim_debug_counters", "cosim_rollout", "cosim_rollout",
               "cosim_set", "cosim_event_push", "cosim_debug_forward", "cosim_kernel_time", "cosim_set_timing",
               "cosim_profile_step", "cosim_model_sizeof", "cosim_obs_config_sizeof"):
        getattr(L, fn).restype = ci
    if L.cosim_model_sizeof() != ctypes.sizeof(CosimModel):
        raise RuntimeError("cosim_model_t layout mismatch between include/cosim_model.h and libcosim_hip.so: rebuild")
    if L.cosim_obs_config_sizeof() != ctypes.sizeof(ObsConfig):
        raise RuntimeError("cosim_obs_config_t layout mismatch: rebuild libcosim_hip.so")
    _lib = L
    return L


EXPORTS = ["cosim_create", "cosim_destroy", "cosim_query", "cosim_set_param", "cosim_reset", "cosim_step", "cosim_step_range", "cosim_get",
           "cosim_join", "cosim_range", "cosim_range_mark", "cosim_debug_counters", "cosim_rollout", "cosim_hull_support_check", "cosim_debug_support",
           "cosim_set", "cosim_event_push", "cosim_debug_forward", "cosim_kernel_time", "cosim_set_timing",
           "cosim_profile_step", "cosim_mlp_forward", "cosim_lstm_cell", "cosim_fleet_stats", "cosim_fleet_hist", "cosim_last_error", "cosim_model_sizeof", "cosim_obs_config_sizeof"]


def make_obs_config(config: dict, obs_to_dim: Dict[str, int], control_freq: float, auto_reset: bool) -> ObsConfig:
    """Flatten config["observation"/"env"/"random"] into ``cosim_obs_config_t``.

    Mirrors the constructor-time checks of the reference wrappers: unknown observation names raise ``KeyError``
    (wrappers.py:116-117), non-positive frequencies raise ``ValueError`` (:186-187).
    """
    ob = config["observation"]
    c = ObsConfig()
    c.stack_size = int(ob["stack_size"])
    c.command_dim = int(ob["command_dim"])
    if c.command_dim < 0:
        raise AssertionError("command_dim must be equal or greater than 0.")
    if c.command_dim > CS_MAXCMD:
        raise ValueError(f"Invalid 'command_dim': expected 0> or <7; but got {c.command_dim}.")
    stacked, non_stacked = list(ob["stacked_obs_order"]), list(ob["non_stacked_obs_order"])
    if len(stacked) > CS_MAXFIELD or len(non_stacked) > CS_MAXFIELD:
        raise ValueError("too many observation fields")
    c.n_stacked, c.n_non_stacked = len(stacked), len(non_stacked)
    for i, n in enumerate(stacked):
        c.stacked_field[i] = OBS_IDS[n] if n in OBS_IDS else _unknown(n, obs_to_dim)
    for i, n in enumerate(non_stacked):
        c.non_stacked_field[i] = OBS_IDS[n] if n in OBS_IDS else _unknown(n, obs_to_dim)
    level = config["random"]["sensor_noise"]
    noise = config["random_table"]["sensor_noise"][level]
    for name, fid in OBS_IDS.items():
        c.field_dim[fid] = int(obs_to_dim[name])
        c.field_interval[fid] = 1
        c.field_scale[fid] = 1.0
        c.noise_std[fid] = 1.0
        if name in stacked + non_stacked and name != "command":
            n_cfg = ob[name]
            freq, scale = float(n_cfg["freq"]), float(n_cfg["scale"])
            if freq <= 0:
                raise ValueError(f"Invalid observation update frequency for '{name}': {freq}. Must be > 0.")
            c.field_interval[fid] = max(1, int(round(control_freq / freq)))
            c.field_scale[fid] = scale
        if name in noise:
            c.noise_mean[fid], c.noise_std[fid] = float(noise[name]["mean"]), float(noise[name]["std"])
            c.noise_lower[fid], c.noise_upper[fid] = float(noise[name]["lower"]), float(noise[name]["upper"])
    # level "none" perturbs by <= 1e-8 in the reference (random_table.yaml:25-55): below fp32 resolution, skipped
    c.noise_enabled = int(level != "none")
    c.position_command = int(bool(config["env"]["position_command"]))
    for i in range(c.command_dim):
        c.command_scales[i] = float(ob["command_scales"][str(i)])
    c.max_sim_step = int(config["env"]["max_duration"] * control_freq)
    c.action_delay_prob = float(config["random"]["action_delay_prob"])
    c.init_noise = float(config["random"]["init_noise"])
    c.auto_reset = int(auto_reset)
    hm = ob.get("height_map")
    if hm is not None:
        if config["env"]["terrain"] == "flat" and ("height_map" in list(ob["stacked_obs_order"]) + list(ob["non_stacked_obs_order"])):
            # the reference samples the map with mj_rayHfield on the ground geom (utils/mujoco_utils.py:169), which is an error
            # on the plane that terrain "flat" turns the ground into (xml_manager.py:24-28)
            raise ValueError("the height_map observation needs a heightfield terrain (mj_rayHfield rejects the plane of terrain 'flat')")
        c.hm_res_x, c.hm_res_y = int(hm["res_x"]), int(hm["res_y"])
        c.hm_size_x, c.hm_size_y = float(hm["size_x"]), float(hm["size_y"])
    return c


def _unknown(name, obs_to_dim):
    return obs_to_dim[name]  # raises KeyError like wrappers.py:116 does for names no env provides


class Engine:
    """Owning handle of one ``cosim_engine_t``; thin, typed wrappers over the C entry points."""

    def __init__(self, compiled, obs_cfg: ObsConfig, num_envs: int, device: int = 0, seed: int = 0, env_id0: int = 0):
        self.L = load_library()
        self.compiled = compiled
        self._keep = (np.ascontiguousarray(compiled.hull_vert, dtype=np.float32),
                      np.ascontiguousarray(compiled.hull_adr, dtype=np.int32),
                      np.ascontiguousarray(compiled.hull_nbr, dtype=np.int32),
                      np.ascontiguousarray(compiled.hfield, dtype=np.float32))
        h = ctypes.c_void_p()
        rc = self.L.cosim_create(ctypes.addressof(compiled.blob), self._keep[0].ctypes.data, self._keep[1].ctypes.data,
                                 self._keep[2].ctypes.data, self._keep[3].ctypes.data, ctypes.addressof(obs_cfg),
                                 int(num_envs), int(device), ctypes.c_uint64(seed & (2 ** 64 - 1)), ctypes.c_int64(env_id0),
                                 ctypes.byref(h))
        self._check(rc)
        self.h = h
        self.num_envs = num_envs
        self.device = device

    def _check(self, rc: int):
        if rc < 0:
            msg = self.L.cosim_last_error().decode()
            if rc == -1:
                raise ValueError(msg)
            raise RuntimeError(msg)
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.L.cosim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query(self, name: str) -> int:
        return self._check(self.L.cosim_query(self.h, name.encode()))

    def set_param(self, name: str, values: np.ndarray):
        v = np.ascontiguousarray(values, dtype=np.float32)
        self._check(self.L.cosim_set_param(self.h, name.encode(), v.ctypes.data, int(v.size)))

    def reset(self, mask_ptr, commands_ptr, state_out_ptr, stream=None):
        self._check(self.L.cosim_reset(self.h, mask_ptr, commands_ptr, state_out_ptr, stream))

    def step(self, actions_ptr, commands_ptr, state_out_ptr, term_ptr, trunc_ptr, info_ptr, stream=None):
        self._check(self.L.cosim_step(self.h, actions_ptr, commands_ptr, state_out_ptr, term_ptr, trunc_ptr, info_ptr, stream))

    def step_range(self, first, count, actions_ptr, commands_ptr, state_out_ptr, term_ptr, trunc_ptr, info_ptr, stream=None):
        self._check(self.L.cosim_step_range(self.h, int(first), int(count), actions_ptr, commands_ptr, state_out_ptr, term_ptr, trunc_ptr,
                                            info_ptr, stream))

    def rollout(self, steps, actions_ptr, commands_ptr, state_out_ptr, term_ptr, trunc_ptr, info_ptr, stream=None):
        self._check(self.L.cosim_rollout(self.h, int(steps), actions_ptr, commands_ptr, state_out_ptr, term_ptr, trunc_ptr, info_ptr, stream))

    def join(self, stream=None):
        self._check(self.L.cosim_join(self.h, stream))

    def range(self, i: int):
        """(first env, env count, hipStream_t or None) of range ``i`` (``cosim_range``)."""
        first, count, st = ctypes.c_int(), ctypes.c_int(), ctypes.c_void_p()
        self._check(self.L.cosim_range(self.h, int(i), ctypes.byref(first), ctypes.byref(count), ctypes.byref(st)))
        return first.value, count.value, st.value

    def range_mark(self, i: int):
        self._check(self.L.cosim_range_mark(self.h, int(i)))

    def get(self, name: str, out_ptr, stream=None):
        self._check(self.L.cosim_get(self.h, name.encode(), out_ptr, stream))

    def set(self, name: str, in_ptr, stream=None):
        self._check(self.L.cosim_set(self.h, name.encode(), in_ptr, stream))

    def push(self, v_ptr, mask_ptr, stream=None):
        self._check(self.L.cosim_event_push(self.h, v_ptr, mask_ptr, stream))

    def debug_forward(self, env: int) -> np.ndarray:
        out = np.zeros(8192, dtype=np.float32)
        self._check(self.L.cosim_debug_forward(self.h, int(env), b"all", out.ctypes.data, out.size))
        return out

    def profile_step(self, actions_ptr, commands_ptr, state_out_ptr, term_ptr, trunc_ptr) -> np.ndarray:
        out = np.zeros(32, dtype=np.float64)
        self._check(self.L.cosim_profile_step(self.h, actions_ptr, commands_ptr, state_out_ptr, term_ptr, trunc_ptr, out.ctypes.data))
        return out

    def debug_counters(self, clear: bool = True) -> np.ndarray:
        out = np.zeros(32, dtype=np.uint64)
        self.L.cosim_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        self.L.cosim_debug_counters.restype = ctypes.c_int
        self._check(self.L.cosim_debug_counters(self.h, out.ctypes.data, int(clear)))
        return out

    def set_timing(self, enabled: bool):
        self._check(self.L.cosim_set_timing(self.h, int(enabled)))

    def kernel_time(self):
        ms, n = ctypes.c_float(), ctypes.c_int()
        self._check(self.L.cosim_kernel_time(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value
